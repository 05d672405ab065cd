"""Prints one steady-state step of a rocprofv3 --kernel-trace run of bench.py as a timeline: kernel durations and the idle
gaps in front of them (host round trips, launch latency).   python tools/step_timeline.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("(anonymous namespace)::", "") for r in rows]
starts = [i for i, n in enumerate(names) if n.startswith("cull_count")]
a, b = starts[-3], starts[-2]
prev_end, busy, idle = None, 0.0, 0.0
for r, n in zip(rows[a:b], names[a:b]):
  st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
  gap = (st - prev_end) / 1e3 if prev_end else 0.0
  busy += (en - st) / 1e3
  idle += max(gap, 0.0)
  print("%7.1f us  gap %5.1f  %s" % ((en - st) / 1e3, gap, n[:90]))
  prev_end = en
print("step %.1f us: kernels %.1f us, idle %.1f us, %d launches" %
      ((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3, busy, idle, b - a))
