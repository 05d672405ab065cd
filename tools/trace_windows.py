"""Per-window kernel table of a rocprofv3 --kernel-trace run of an iterated loop: the trace is cut into windows at every
`per`-th launch of a marker kernel (default: composite_bwd_kernel, one launch per camera) and each kernel's time per
window is printed -- how a training loop's kernel mix moves while the scene trains.
    python tools/trace_windows.py <dir with *_kernel_trace.csv> [launches per window = 80] [marker substring]"""
import collections
import csv
import glob
import os
import re
import sys

d = sys.argv[1]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 80
marker = sys.argv[3] if len(sys.argv) > 3 else "composite_bwd_kernel"
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))


def short(n):
  m = re.search(r"(\w+)(<[^>(]*>)?\(", n)
  return (m.group(1) + (m.group(2) or "")) if m else n[:40]


windows, cur, seen = [], collections.defaultdict(lambda: [0, 0.0]), 0
for r in rows:
  name = short(r["Kernel_Name"])
  e = cur[name]
  e[0] += 1
  e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
  if marker in name:
    seen += 1
    if seen % per == 0:
      windows.append(cur)
      cur = collections.defaultdict(lambda: [0, 0.0])
names = sorted({n for w in windows for n in w}, key=lambda n: -sum(w[n][1] for w in windows if n in w))
print(f"{len(windows)} windows of {per} '{marker}' launches; microseconds per marker launch (all kernels of the window / {per})")
print(f"{'kernel':44s}" + "".join(f"{i:>9d}" for i in range(len(windows))))
for n in names[:28]:
  print(f"{n[:44]:44s}" + "".join(f"{(w[n][1] / per if n in w else 0):9.1f}" for w in windows))
print(f"{'sum of all kernels':44s}" + "".join(f"{sum(v[1] for v in w.values()) / per:9.1f}" for w in windows))
