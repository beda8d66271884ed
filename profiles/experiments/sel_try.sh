DSLAM_DBG_SELECT=gpurun_out/select.bin python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-extra-rates --reint 0 --mode device > /dev/null 2>&1
python profiles/experiments/select_timeline.py gpurun_out/select.bin > gpurun_out/select_tl.txt; tail -4 gpurun_out/select_tl.txt
bash profiles/experiments/quick_profile.sh > gpurun_out/qp.txt 2>&1 && python -c "
import csv,glob
f=glob.glob('gpurun_out/r3_stats_device/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,2))
"
