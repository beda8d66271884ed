#!/bin/bash
# Round 4: what spatial (Morton) order of the block-major batch's class lists would buy the image gathers -- lists sorted on the
# HOST (DSLAM_BATCH_SORT=1; 2 = plain slot order as the control), the block launch timed by rocprofv3.   run on the GPU box.
# Needs the experimental build: git apply profiles/experiments/batch_sort.patch && make -C denseslam-global-consistency-h_amd/csrc
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for m in 0 1 2 0 1; do
  if [ $m = 0 ]; then unset DSLAM_BATCH_SORT; else export DSLAM_BATCH_SORT=$m; fi
  rm -rf $R/gpurun_out/bs_$m
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bs_$m -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress --no-extra-rates --mode device > $R/gpurun_out/bs_$m.log 2>&1 || exit 1
  f=$(ls $R/gpurun_out/bs_$m/*/*_kernel_stats.csv | head -1)
  echo "sort=$m $(grep k_reintegrate_blocks $f | cut -d, -f2-4)"
  python3 -c "
import json,sys
d=json.loads(open('$R/gpurun_out/bs_$m.log').read().strip().split('\n')[-1]); print('   batch total_ms', d['reintegration']['block_major']['total_ms'])"
done
