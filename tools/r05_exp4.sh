#!/bin/bash
# is there synergy between a latency-side change (top of tree in LDS / more slots) and a VALU-side change (fast_shading as a proxy)?
for lib in libmoptix_base.so libmoptix_t40.so libmoptix_s8.so; do
  for o in "" "fast_shading=1"; do
    echo "== $lib $o"; SPP=64 OPTS=$o MOPTIX_DEVICE_LIB=$lib timeout 300 python3 tools/gpu_quick.py | tail -1
  done
done
