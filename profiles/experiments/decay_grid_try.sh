# k_decay_blocks / k_release on the S-stress map with different grids (rebuilds maintain.o on the box)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R/denseslam-global-consistency-h_amd/csrc
for W in 1024 2048 4096; do
  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fPIC -std=c++17 -DDSLAM_DECAY_WGS=$W -c maintain.hip -o maintain.o 2>/dev/null && hipcc --offload-arch=gfx950 -shared -fPIC -o libdslam_fusion.so capi.o alloc.o integrate.o raycast.o maintain.o view.o track.o mesh.o shard.o
  (cd /tmp && DSLAM_SKIP_BUILD=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dg -- python3 $R/denseslam-global-consistency-h_amd/harness/maint_bench.py > /dev/null 2>&1)
  python3 -c "
import csv,glob
f=sorted(glob.glob('$R/gpurun_out/dg/**/*kernel_trace.csv',recursive=True))[-1]
d=sorted([(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(f)) if 'k_decay_blocks' in r['Kernel_Name']],reverse=True)
print('decay wgs $W: longest launches', [round(x,1) for x in d[:5]])
"
  rm -rf $R/gpurun_out/dg
done
