#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel-trace summaries and PMC passes of bench.py for one round.
#   bash tools/collect_profiles.sh r02 "c2 c3"
# Outputs land in gpurun_out/prof_<round>/ ; tools/make_profiles.py turns them into the files kept under profiles/.
# PMC counters are collected in passes of their own (never together with sys/hip traces), as the guide prescribes.
set -e
ROUND=${1:-r02}
WORKLOADS=${2:-"c2 c3"}
OUT=gpurun_out/prof_${ROUND}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$OUT"; mkdir -p "$OUT"
for W in $WORKLOADS; do
  echo "== $W kernel trace"; 
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${W}_trace" -- python3 bench.py --no-cpu-baseline --no-scale-workload --workload "$W" > "$OUT/${W}_trace.log" 2>&1
  for PASS in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE"; do
    TAG=$(echo "$PASS" | cut -d' ' -f1)
    echo "== $W pmc $TAG"
    rocprofv3 --kernel-trace --pmc $PASS --output-format csv -d "$OUT/${W}_pmc_${TAG}" -- python3 bench.py --no-cpu-baseline --no-scale-workload --workload "$W" --steps 3 --warmup 1 > "$OUT/${W}_pmc_${TAG}.log" 2>&1 || echo "pass $TAG failed"
  done
  python3 bench.py --workload "$W" $( [ "$W" = c2 ] || echo --no-cpu-baseline ) | tail -1 > "$OUT/bench_${W}.json"
done
ls "$OUT"
