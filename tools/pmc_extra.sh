# Extra SQ counters of K6 / K7 on c2 (two --pmc passes, never together with sys/hip traces): scalar-memory latency,
# instruction fetch, branches, and where the wave cycles go (waiting on counters / waiting for issue / issuing).
#   gpurun -- bash tools/pmc_extra.sh   -> prints per-launch means; kept under profiles/r03_pmc_extra_c2.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmcx; mkdir -p gpurun_out/pmcx
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d gpurun_out/pmcx/a -- python3 bench.py --no-cpu-baseline --no-scale-workload --workload c2 --steps 3 --warmup 1 > gpurun_out/pmcx/a.log 2>&1 || echo "pass a failed"
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SMEM --output-format csv -d gpurun_out/pmcx/b -- python3 bench.py --no-cpu-baseline --no-scale-workload --workload c2 --steps 3 --warmup 1 > gpurun_out/pmcx/b.log 2>&1 || echo "pass b failed"
python3 - <<'PY'
import csv,glob,collections
for p in ("a","b"):
    for f in glob.glob(f"gpurun_out/pmcx/{p}/*/*counter_collection.csv"):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n=r["Kernel_Name"]
            key="K6" if "composite_fwd_kernel" in n else "K7" if "composite_bwd_kernel" in n else None
            if key: agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in agg.items():
            print(p,k,{c: round(sum(x)/len(x)) for c,x in v.items()})
PY
