#!/bin/bash
# round 5, experiment 1: Node128 fetched by the ray's signs against the 64-byte nodes and against round 4's library
mkdir -p gpurun_out/r05a
{
echo "== base (round 4), auto format"; MOPTIX_DEVICE_LIB=libmoptix_base.so timeout 600 python3 tools/scene_times.py
echo "== base, 128";  NODE_FORMAT=128 MOPTIX_DEVICE_LIB=libmoptix_base.so timeout 600 python3 tools/scene_times.py
echo "== signed fetch, 128"; NODE_FORMAT=128 timeout 600 python3 tools/scene_times.py
echo "== signed lib, 64"; NODE_FORMAT=64 timeout 600 python3 tools/scene_times.py
echo "== hashes"; SPP=64 MOPTIX_DEVICE_LIB=libmoptix_base.so timeout 300 python3 tools/gpu_quick.py | tail -1
SPP=64 OPTS=node_format=128 timeout 300 python3 tools/gpu_quick.py | tail -1
for k in fma_mix v_cvt_f32_f16 v_ashrrev v_sub_u32 v_or3 v_xad v_mul_u32_u24 v_sad v_cvt_f32_i32 v_pk_max_f16 v_pk_fma_f16 v_mul_legacy s_and; do timeout 120 tools/micro/valu_issue 3000 $k | grep -v "^#"; done
} > gpurun_out/r05a/log.txt 2>&1
tail -60 gpurun_out/r05a/log.txt
