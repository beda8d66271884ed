# per-kernel times of the three forms of the re-integration batch (bench.py --reint), rocprofv3 kernel trace
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_reint_stats -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-stress --no-extra-rates --mode device > $R/gpurun_out/r3_reint_stats.log 2>&1; echo rc=$?
