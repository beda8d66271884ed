# where a frame's wall time goes besides its kernels: dispatch-by-dispatch timeline of a few steady-state frames
# usage: bash profiles/experiments/frame_gaps.sh <bench.py arguments>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/gaps -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-extra-rates --reint 0 "$@" > $R/gpurun_out/gaps.log 2>&1; echo rc=$?
cd $R
python - <<'P'
import csv, glob
f = glob.glob('gpurun_out/gaps/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-34:]) for r in csv.DictReader(open(f))]
m = glob.glob('gpurun_out/gaps/**/*memory_copy_trace.csv', recursive=True)
if m:
    for r in csv.DictReader(open(m[0])):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].endswith('k_mark')]
# frame period (k_mark to k_mark) over the run, in groups of 20 frames
per = [(rows[marks[k + 1]][0] - rows[marks[k]][0]) / 1e3 for k in range(len(marks) - 1)]
for lo in range(0, len(per), 20):
    g = per[lo:lo + 20]
    print(f"frames {lo:3d}-{lo + len(g) - 1:3d}: period mean {sum(g) / len(g):7.1f} max {max(g):8.1f} us")
i0 = marks[len(marks) * 3 // 4]
i1 = marks[len(marks) * 3 // 4 + 3]
prev = rows[i0][0]
for s, e, n in rows[i0:i1 + 1]:
    print(f"{(s - rows[i0][0]) / 1e3:9.1f} +{(s - prev) / 1e3:7.1f}  {(e - s) / 1e3:8.1f} us  {n}")
    prev = e
P
rm -rf gpurun_out/gaps
