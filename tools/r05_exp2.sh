#!/bin/bash
mkdir -p gpurun_out/r05b
{
echo "== masked gather"; timeout 120 tools/micro/gather_mask
echo "== census base"; MOPTIX_DEBUG=1 SPP=64 MOPTIX_DEVICE_LIB=libmoptix_base.so timeout 300 python3 tools/gpu_quick.py 2>&1 | grep -v "lane census\|\[moptix\]   " | tail -25
echo "== w2 (2 WG/CU x 896 slots)"; SPP=64 OPTS=blocks_per_cu=2 MOPTIX_DEVICE_LIB=libmoptix_w2.so timeout 300 python3 tools/gpu_quick.py | tail -1
echo "== w2, 128"; SPP=64 OPTS=blocks_per_cu=2,node_format=128 MOPTIX_DEVICE_LIB=libmoptix_w2.so timeout 300 python3 tools/gpu_quick.py | tail -1
} > gpurun_out/r05b/log.txt 2>&1
cat gpurun_out/r05b/log.txt
