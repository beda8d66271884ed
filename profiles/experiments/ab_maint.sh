# A/B of two builds of the library on one box: the maintenance bench with the tree's library, then with profiles/experiments/ab_old.so
# (the other build: `git archive <commit> denseslam-global-consistency-h_amd/csrc include | tar -x -C scratch/old`, make there, copy its
# libdslam_fusion.so to profiles/experiments/ab_old.so -- git-ignored, removed after use)
L=denseslam-global-consistency-h_amd/csrc/libdslam_fusion.so
show() { python -c "
import json,sys;m=json.load(open(sys.argv[1]))
print(sys.argv[1], [(k[14:], m[k]['us_per_keyframe']) for k in m if k.startswith('keyframe_loop')], 'pf', m['process_frame_all_visible']['us'], 'slide', m['slide_window_release_all']['us'])" $1; }
export DSLAM_SKIP_BUILD=1
python denseslam-global-consistency-h_amd/harness/maint_bench.py > gpurun_out/ab_new1.json 2>/dev/null; show gpurun_out/ab_new1.json
cp $L gpurun_out/new.so; cp profiles/experiments/ab_old.so $L
python denseslam-global-consistency-h_amd/harness/maint_bench.py > gpurun_out/ab_old1.json 2>/dev/null; show gpurun_out/ab_old1.json
cp gpurun_out/new.so $L
python denseslam-global-consistency-h_amd/harness/maint_bench.py > gpurun_out/ab_new2.json 2>/dev/null; show gpurun_out/ab_new2.json
cp profiles/experiments/ab_old.so $L
python denseslam-global-consistency-h_amd/harness/maint_bench.py > gpurun_out/ab_old2.json 2>/dev/null; show gpurun_out/ab_old2.json
rm -f gpurun_out/new.so
