export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/stress_trace -- python3 $R/profiles/experiments/stress_frame_trace.py > $R/gpurun_out/stress_trace.log 2>&1; echo rc=$?
cd $R
python - <<'P'
import csv, glob
f = glob.glob('gpurun_out/stress_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
for r in rows[-60:]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:12.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:9.1f}  {r['Kernel_Name'][:90]}")
P
