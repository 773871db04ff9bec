#!/bin/bash
# round 5: the A/B experiments of NOTEBOOK.md in one run (experiment libraries built with make EXTRA=... LIBNAME=...; see the header lines).
# The source changes of the experiments that were NOT kept are recorded in profiles/r05_experiment_patches.txt; their macros no longer exist in csrc/.
O=gpurun_out/r05d; mkdir -p $O
{
echo "# libmoptix_base.so = round 4's node step (EXTRA=-DPT_NO_SIGNED_FETCH at a4e2eff); libmoptix.so = shipped; _s8 = -DPT_PK_STACKN=8; _t40 = + -DPT_PK_TOPN=40 (top of the tree in LDS);"
echo "# _fuse = -DPT_PK_FUSE=1 (leaf pass issues its gathers, one node step, then the triangle tests); _g1 = -DPT_SHARE_GTR1; _w2 = -DPT_PK_KP=224 -DPT_WAVES_PER_SIMD=2 with blocks_per_cu=2"
for lib in libmoptix_base.so libmoptix.so libmoptix_s8.so libmoptix_t40.so libmoptix_fuse.so libmoptix_g1.so libmoptix_base.so; do
  [ -f minimaloptix_amd/lib/$lib ] && MOPTIX_DEVICE_LIB=$lib timeout 600 python3 tools/scene_times.py 2>&1 | grep -v "^\[moptix\]"
done
echo "# 128-byte nodes: round 4's min / max form against the sign-addressed fetch"
NODE_FORMAT=128 MOPTIX_DEVICE_LIB=libmoptix_base.so timeout 600 python3 tools/scene_times.py 2>&1 | grep -v "^\[moptix\]"
NODE_FORMAT=128 timeout 600 python3 tools/scene_times.py 2>&1 | grep -v "^\[moptix\]"
echo "# two workgroups per CU x 896 slots"
[ -f minimaloptix_amd/lib/libmoptix_w2.so ] && SPP=64 OPTS=blocks_per_cu=2 MOPTIX_DEVICE_LIB=libmoptix_w2.so timeout 300 python3 tools/gpu_quick.py | tail -1
echo "# borrowed slots for every path"
SWEEP="aux_depth=1,4;slots_in_use=224,352,448" timeout 600 python3 tools/gpu_sweep.py 2>&1 | tail -6
} > $O/ab.log 2>&1
cat $O/ab.log | cut -c1-130
