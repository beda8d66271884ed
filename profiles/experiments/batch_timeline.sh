# dispatch-by-dispatch timeline of one dslam_reintegrate_batch call (bench.py's re-integration leg under rocprofv3 --kernel-trace)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/batch_tl -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-stress --no-extra-rates --mode device > $R/gpurun_out/batch_tl.log 2>&1; echo rc=$?
cd $R
python - <<'P'
import csv, glob
f = glob.glob('gpurun_out/batch_tl/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-40:]) for r in csv.DictReader(open(f))]
m = glob.glob('gpurun_out/batch_tl/**/*memory_copy_trace.csv', recursive=True)
if m:
    for r in csv.DictReader(open(m[0])):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')))
rows.sort()
idx = max(i for i, r in enumerate(rows) if 'reintegrate_blocks' in r[2])
end = rows[idx][1]
# walk back to the start of the batch: the memsets in front of the first allocation pass; print from 75 dispatches before
lo = max(0, idx - 80)
t0 = rows[lo][0]
prev_end = rows[lo][0]
for s, e, n in rows[lo:idx + 4]:
    print(f"{(s - t0) / 1e3:10.1f} +{(s - prev_end) / 1e3:7.1f} gap  {(e - s) / 1e3:9.1f} us  {n}")
    prev_end = e
P
