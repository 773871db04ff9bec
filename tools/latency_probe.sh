#!/bin/bash
# round 5: how exposed is the latency of a node step / a leaf pass?  The wave idles 64 x n clocks (s_sleep n: no issue slot used, as in a memory wait)
# after every node step (libmoptix_snN.so, -DPT_PK_SLEEP_NODE=N) or before every leaf pass's tests (libmoptix_slN.so, -DPT_PK_SLEEP_LEAF=N).
O=gpurun_out/r05e; mkdir -p $O
{
for lib in libmoptix.so libmoptix_sn4.so libmoptix_sn8.so libmoptix_sn16.so libmoptix_sl16.so libmoptix_sl32.so libmoptix.so; do
  [ -f minimaloptix_amd/lib/$lib ] && MOPTIX_DEVICE_LIB=$lib timeout 600 python3 tools/scene_times.py 2>&1 | grep -v "^\[moptix\]"
done
} > $O/latency_probe.log 2>&1
cut -c1-110 $O/latency_probe.log
