# k_fill_range_tiles with different thresholds for the cooperative path (rebuilds raycast.o on the box)
cd denseslam-global-consistency-h_amd/csrc
for T in 32 48 96; do
  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fPIC -std=c++17 -DDSLAM_RANGE_BIG=$T -c raycast.hip -o raycast.o 2>/dev/null && hipcc --offload-arch=gfx950 -shared -fPIC -o libdslam_fusion.so capi.o alloc.o integrate.o raycast.o maintain.o view.o track.o mesh.o shard.o
  (cd ../..; DSLAM_SKIP_BUILD=1 bash profiles/experiments/quick_profile.sh > gpurun_out/qp.txt 2>&1; python -c "
import csv,glob
f=sorted(glob.glob('gpurun_out/r3_stats_device/**/*kernel_stats.csv',recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if 'fill_range' in r['Name']: print('threshold $T', round(float(r['AverageNs'])/1e3,2))
"; rm -rf gpurun_out/r3_stats_device)
done
