# round-3 auxiliary measurements: shape sensitivity of the memory curves, shard emulation with the block-major batch
R=$GRAFT_REPO_ROOT
cd $R
python denseslam-global-consistency-h_amd/harness/memory_sensitivity.py 1500 > gpurun_out/r3_memory_shape_sensitivity.json 2> gpurun_out/r3_memory_shape_sensitivity.err; echo sens rc=$?
python denseslam-global-consistency-h_amd/harness/shard_emulation.py 120 32 > gpurun_out/r3_shard_emulation.json 2> gpurun_out/r3_shard_emulation.err; echo shard rc=$?
