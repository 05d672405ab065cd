"""Per-step GPU timeline from a rocprofv3 --kernel-trace CSV: busy time, gaps, launches (tools/collect_profiles.sh writes
<dir>/<host>/<pid>_kernel_trace.csv).   python tools/trace_gaps.py <kernel_trace.csv> [marker-kernel-substring]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "composite_bwd_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if len(ends) < 12:
  sys.exit("too few steps")
a, b = ends[-11], ends[-1]                       # the last ten steps (marker kernel to marker kernel)
seg = rows[a + 1:b + 1]
t0, t1 = int(rows[a]["End_Timestamp"]), int(rows[b]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print(f"10 steps: {(t1 - t0) / 1e4:.1f} us/step wall, {busy / 1e4:.1f} us/step kernels, {len(seg) / 10:.1f} launches/step")
gaps = {}
prev = rows[a]
for r in seg:
  g = int(r["Start_Timestamp"]) - int(prev["End_Timestamp"])
  key = (prev["Kernel_Name"][:48], r["Kernel_Name"][:48])
  gaps.setdefault(key, []).append(g)
  prev = r
tot = sorted(((sum(v) / 10, len(v) / 10, k) for k, v in gaps.items()), reverse=True)
for s, n, k in tot[:25]:
  print(f"{s / 1e3:8.1f} us/step gap  x{n:4.1f}  {k[0]}  ->  {k[1]}")
