# per-kernel times of the maintenance path (BASELINE configs[2]): harness/maint_bench.py (S-stress map calls + the keyframe
# loops with decay / window / swapping) under rocprofv3 --kernel-trace --stats
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_maint_stats -- python3 $R/denseslam-global-consistency-h_amd/harness/maint_bench.py > $R/gpurun_out/r3_maintenance_profiled.json 2> $R/gpurun_out/r3_maint_stats.err; echo rc=$?
cd $R
python denseslam-global-consistency-h_amd/harness/maint_bench.py > gpurun_out/r3_maintenance.json 2> gpurun_out/r3_maintenance.err; echo rc=$?
