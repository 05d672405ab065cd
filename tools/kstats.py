"""Prints the per-kernel averages of a rocprofv3 --kernel-trace --stats run (newest *kernel_stats.csv under a directory):
    python tools/kstats.py gpurun_out/x_trace [steps] [filter ...]"""
import csv
import glob
import os
import re
import sys

d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
filters = sys.argv[3:]
hits = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(hits[-1])))
total = 0.0
for r in rows:
  n = r["Name"]
  m = re.search(r"(\w+)(<[^>(]*>)?\(", n)
  name = (m.group(1) + (m.group(2) or "")) if m else n[:48]
  per_step = float(r["TotalDurationNs"]) / steps / 1000
  total += per_step
  if filters and not any(f in name for f in filters):
    continue
  print(f"{name[:52]:52s} x{int(r['Calls']) / steps:5.1f} avg {float(r['AverageNs']) / 1000:8.1f} us  /step {per_step:8.1f}")
print(f"sum /step {total:.1f} us")
