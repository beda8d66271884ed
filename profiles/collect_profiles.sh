# collects the judged artifacts: default bench line, kernel stats, PMC passes (all the same bench command)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo bench rc=$?
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_stats -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --reint 0 > $R/gpurun_out/final_stats.log 2>&1; echo stats rc=$?
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final_fetch -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --reint 0 > $R/gpurun_out/final_fetch.log 2>&1; echo fetch rc=$?
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final_write -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --reint 0 > $R/gpurun_out/final_write.log 2>&1; echo write rc=$?
cd $R
python denseslam-global-consistency-h_amd/harness/stress.py 64 > gpurun_out/final_stress.json; cat gpurun_out/final_stress.json
python bench.py --sync --steps 100 --warmup 10 --no-cpu-baseline --reint 0 > gpurun_out/final_bench_sync.json 2>/dev/null; grep -o '"value": [0-9.]*' gpurun_out/final_bench_sync.json
python denseslam-global-consistency-h_amd/harness/side_bench.py 50 > gpurun_out/final_side_bench.json 2>gpurun_out/final_side_bench.err; cat gpurun_out/final_side_bench.json
python bench.py --host-io --steps 100 --warmup 10 --no-cpu-baseline --reint 0 > gpurun_out/final_bench_hostio.json 2>/dev/null; grep -o "\"value\": [0-9.]*" gpurun_out/final_bench_hostio.json
python denseslam-global-consistency-h_amd/harness/quality.py 40 > gpurun_out/final_quality.json 2>gpurun_out/final_quality.err; cat gpurun_out/final_quality.json
python denseslam-global-consistency-h_amd/harness/maint_bench.py > gpurun_out/final_maintenance.json 2>gpurun_out/final_maintenance.err; cat gpurun_out/final_maintenance.json
