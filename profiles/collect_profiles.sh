# collects the judged artifacts of a round (run ON the GPU box via gpurun; outputs under gpurun_out/, condensed into
# profiles/<tag>_* by profiles/make_profiles.py):  bash profiles/collect_profiles.sh
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
# two parts (a gpurun call lasts 20 minutes at most):  bash profiles/collect_profiles.sh 1   then   bash profiles/collect_profiles.sh 2
PART=${1:-all}
if [ "$PART" = 1 ] || [ "$PART" = all ]; then
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo bench rc=$?
BENCH_FAST="--steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-extra-rates --reint 0"
cd /tmp
# per-kernel times of the default (PCIe-inclusive, pipelined) loop and of the device-resident loop
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_stats -- python3 $R/bench.py $BENCH_FAST > $R/gpurun_out/final_stats.log 2>&1; echo stats rc=$?
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_stats_device -- python3 $R/bench.py $BENCH_FAST --mode device > $R/gpurun_out/final_stats_device.log 2>&1; echo stats_device rc=$?
# HBM traffic of k_integrate: two separate counter passes (MI355X_MICROARCH.md, HBM / rocprofv3)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final_fetch -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-stress --no-extra-rates --reint 0 --mode device > $R/gpurun_out/final_fetch.log 2>&1; echo fetch rc=$?
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final_write -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-stress --no-extra-rates --reint 0 --mode device > $R/gpurun_out/final_write.log 2>&1; echo write rc=$?
# instruction and wave-time counters (VALU instructions per block-wave; parked / issue-stalled / issuing shares)
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/final_sq -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-stress --no-extra-rates --reint 0 --mode device > $R/gpurun_out/final_sq.log 2>&1; echo sq rc=$?
cd $R
# per-wave dump of the ray march, per-tile timeline of the allocation sweep, per-wave timeline of the fusion kernel (diagnostic instantiations)
DSLAM_DBG_WAVETIME=gpurun_out/final_wavetime.bin DSLAM_DBG_SWEEP=gpurun_out/final_sweep.bin DSLAM_DBG_INTEGRATE=gpurun_out/final_integrate_waves.bin DSLAM_DBG_SELECT=gpurun_out/final_select.bin DSLAM_DBG_MARK=gpurun_out/final_mark.bin python bench.py $BENCH_FAST --mode device > /dev/null 2>&1; ls -la gpurun_out/final_wavetime.bin gpurun_out/final_sweep.bin gpurun_out/final_integrate_waves.bin gpurun_out/final_select.bin gpurun_out/final_mark.bin
python bench.py --mode sync --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-extra-rates --reint 0 > gpurun_out/final_bench_sync.json 2>/dev/null
python denseslam-global-consistency-h_amd/harness/stress.py 64 > gpurun_out/final_stress.json; cat gpurun_out/final_stress.json
python profiles/experiments/pipeline_breakdown.py 100 > gpurun_out/final_pipeline.json 2>/dev/null; cat gpurun_out/final_pipeline.json
python denseslam-global-consistency-h_amd/harness/side_bench.py 50 > gpurun_out/final_side_bench.json 2>gpurun_out/final_side_bench.err; tail -c 300 gpurun_out/final_side_bench.json
python denseslam-global-consistency-h_amd/harness/quality.py 40 > gpurun_out/final_quality.json 2>gpurun_out/final_quality.err; tail -c 300 gpurun_out/final_quality.json
fi
if [ "$PART" = 2 ] || [ "$PART" = all ]; then
python denseslam-global-consistency-h_amd/harness/maint_bench.py > gpurun_out/final_maintenance.json 2>gpurun_out/final_maintenance.err; tail -c 400 gpurun_out/final_maintenance.json
# the same script under the kernel trace: per-kernel split of the maintenance path (S-stress calls + keyframe loops, BASELINE configs[2])
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_maint_stats -- python3 $R/denseslam-global-consistency-h_amd/harness/maint_bench.py > /dev/null 2> $R/gpurun_out/final_maint_stats.err); echo maint_stats rc=$?
# per-kernel times of the three forms of the re-integration batch
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_reint_stats -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-extra-rates --mode device > $R/gpurun_out/final_reint_stats.log 2>&1); echo reint_stats rc=$?
# instruction / wave-time counters of the block-major batch launch (one pass of SQ counters; TCC counters are NOT mixed into it:
# a pass with FETCH_SIZE + WRITE_SIZE + SQ_INSTS_VMEM_* went silent for 7 minutes on this pool)
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/final_reint_sq -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-extra-rates --mode device > $R/gpurun_out/final_reint_sq.log 2>&1); echo reint_sq rc=$?
# HBM traffic of the block-major batch launch: FETCH_SIZE and WRITE_SIZE in two SEPARATE single-counter passes (round 4; the mixed TCC + SQ
# set of round 3 went silent)
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final_reint_fetch -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-extra-rates --mode device > $R/gpurun_out/final_reint_fetch.log 2>&1); echo reint_fetch rc=$?
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final_reint_write -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-extra-rates --mode device > $R/gpurun_out/final_reint_write.log 2>&1); echo reint_write rc=$?
# the drop-in: keyframe loops through the C++ ITMLib mirror (deferred completion vs every call waiting)
python denseslam-global-consistency-h_amd/harness/mirror_bench.py > gpurun_out/final_mirror.json 2> gpurun_out/final_mirror.err; tail -c 300 gpurun_out/final_mirror.json
python denseslam-global-consistency-h_amd/harness/memory_sensitivity.py 1500 > gpurun_out/final_memory_sensitivity.json 2> gpurun_out/final_memory_sensitivity.err; echo sensitivity rc=$?
python denseslam-global-consistency-h_amd/harness/shard_emulation.py 120 32 > gpurun_out/final_shard_emulation.json 2>/dev/null; tail -c 300 gpurun_out/final_shard_emulation.json
fi
