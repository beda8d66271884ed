# Round 4, VERDICT item 5: HBM traffic of the block-major re-integration launch -- FETCH_SIZE and WRITE_SIZE in two SEPARATE
# single-counter passes (the known-good form: a pass that mixed TCC and SQ counters went silent in round 3), and the SQ set in
# a pass of its own.  The command is the one collect_profiles.sh uses for the re-integration leg.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-stress --no-extra-rates --mode device"
cd /tmp
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/reint_fetch -- $CMD > $R/gpurun_out/reint_fetch.log 2>&1 && echo fetch ok &&
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/reint_write -- $CMD > $R/gpurun_out/reint_write.log 2>&1 && echo write ok &&
timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/reint_sq -- $CMD > $R/gpurun_out/reint_sq.log 2>&1 && echo sq ok
cd $R
python - <<'P'
import csv, glob, collections, json
out = {}
for d in ("reint_fetch", "reint_write", "reint_sq"):
    f = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no counters"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0][-48:]
        if not ("reintegrate" in k or "batch" in k): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k in acc:
        out.setdefault(k, {}).update({c: v / len(disp[k]) for c, v in acc[k].items()})
print(json.dumps(out, indent=1))
P
