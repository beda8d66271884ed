export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for m in 0 1 2 3 4; do
DSLAM_DBG_MARKMODE=$m rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mm_$m -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-stress --no-extra-rates --reint 0 --mode device > $R/gpurun_out/mm_$m.log 2>&1
grep -h "k_mark" $R/gpurun_out/mm_$m/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
done
