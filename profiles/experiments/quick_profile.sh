export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_stats_device -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-extra-rates --reint 0 --mode device > $R/gpurun_out/r3_stats_device.log 2>&1; echo stats_device rc=$?
cd $R
python bench.py --no-cpu-baseline --reint 0 > gpurun_out/r3_bench_quick.json 2> gpurun_out/r3_bench_quick.err; echo rc=$?
tail -c 1500 gpurun_out/r3_bench_quick.json
