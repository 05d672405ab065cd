"""Condenses the rocprofv3 output of tools/collect_profiles.sh into the files kept under profiles/:

    python tools/make_profiles.py r02 c2 c3

  profiles/<round>_bench_<w>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (per-kernel calls / avg / total)
  profiles/<round>_pmc_<w>.json                 per-launch counter means of the hot kernels + the static ISA mix
  profiles/<round>_bench_<w>.json               the bench line of the same build

HBM bytes per launch follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE come from separate passes, are
reported in KiB, and on gfx950 FETCH_SIZE tallies a wide coalesced read at half its bytes -- `hbm_bytes_per_launch` doubles
it (the guide's correction; an upper bound for kernels that mix scalar and 4-byte loads), `hbm_bytes_uncorrected` does not.
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"K7": "composite_bwd_kernel", "K6": "composite_fwd_kernel", "K6_segC": "seg_composite_kernel",
           "K6_combine": "seg_combine_kernel", "sh_bwd_dense": "sh_bwd_dense_kernel", "reduce_grad": "reduce_grad_kernel",
           "tile_sort_scatter": "rs_scatter", "tile_emit": "tile_emit_kernel", "tile_count": "tile_count_kernel"}


def newest(pattern):
  hits = sorted(glob.glob(pattern), key=os.path.getmtime)
  return hits[-1:] if hits else []


def counters(path):
  agg = collections.defaultdict(lambda: collections.defaultdict(list))
  for f in newest(os.path.join(path, "*", "*counter_collection.csv")):      # one run per pass: the most recent
    for r in csv.DictReader(open(f)):
      agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
  return agg


def main():
  rnd, workloads = sys.argv[1], sys.argv[2:]
  src = os.path.join(ROOT, "gpurun_out", f"prof_{rnd}")
  isa = json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_stats.py"), "--json"], check=True,
                                  capture_output=True, text=True).stdout)
  isa.pop("all_kernels", None)
  for w in workloads:
    stats = newest(os.path.join(src, f"{w}_trace", "*", "*kernel_stats.csv"))
    if stats:
      shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{rnd}_bench_{w}_kernel_stats.csv"))
    b = os.path.join(src, f"bench_{w}.json")
    if os.path.exists(b) and os.path.getsize(b):
      shutil.copy(b, os.path.join(ROOT, "profiles", f"{rnd}_bench_{w}.json"))
    merged = collections.defaultdict(dict)
    for d in glob.glob(os.path.join(src, f"{w}_pmc_*")):
      if not os.path.isdir(d):
        continue
      for kname, vals in counters(d).items():
        flat = kname.replace(" ", "")
        for key, needle in KERNELS.items():
          templated = key in ("K7", "K6", "K6_segC", "K6_combine")     # take the 3-channel instantiation the bench runs
          if (needle + "<3" in flat) if templated else (needle in flat):
            for c, v in vals.items():
              merged[key][c] = sum(v) / len(v)
            merged[key]["kernel"] = needle
    for key, m in merged.items():
      if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        m["hbm_bytes_uncorrected"] = 1024.0 * (m["FETCH_SIZE"] + m["WRITE_SIZE"])
        m["hbm_bytes_per_launch"] = 1024.0 * (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"])
      if "TCC_HIT_sum" in m:
        m["L2_hit_rate"] = m["TCC_HIT_sum"] / max(m["TCC_HIT_sum"] + m["TCC_MISS_sum"], 1.0)
    out = {"workload": w, "round": rnd,
           "how": "rocprofv3 --kernel-trace --pmc <counters>, one pass per counter group (tools/collect_profiles.sh) of "
                  "`python3 bench.py --no-cpu-baseline --workload %s --steps 3 --warmup 1`; per-launch means. FETCH_SIZE / "
                  "WRITE_SIZE in KiB; hbm_bytes_per_launch = 1024 * (2 * FETCH_SIZE + WRITE_SIZE) (gfx950 correction of "
                  "MI355X_MICROARCH.md; upper bound for mixed scalar / 4-byte loads), hbm_bytes_uncorrected without the 2x" % w,
           "kernels": merged, "isa": isa}
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_{w}.json"), "w"), indent=1)
    print("wrote", f"profiles/{rnd}_pmc_{w}.json", sorted(merged))


if __name__ == "__main__":
  main()
