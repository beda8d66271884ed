# instruction / wave-time counters of the block-major re-integration launch next to k_integrate (bench.py's re-integration leg)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/reint_sq -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress --no-extra-rates --mode device > $R/gpurun_out/reint_sq.log 2>&1; echo sq rc=$?
# (a second pass with FETCH_SIZE WRITE_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM in ONE counter set went silent for
# 7 minutes on this pool and was killed: TCC and SQ counters are collected in separate passes, as collect_profiles.sh does)
cd $R
python - <<'P'
import csv, glob, collections
for d in ("reint_sq",):
    f = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no counters"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0][-60:]
        if not ("integrate" in k or "batch" in k): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen: seen.add(key); n[k] += 1
    for k in acc:
        print(d, k, "dispatches", n[k], {c: round(v / n[k], 1) for c, v in acc[k].items()})
P
