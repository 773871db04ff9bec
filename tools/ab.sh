#!/bin/bash
# A/B of device libraries on the GPU box: tools/ab.sh libA.so libB.so ...  (coffee full HD at SPP, three scenes optional)
set -u
export SPP=${SPP:-64}
for lib in "$@"; do
  echo "== $lib"
  MOPTIX_DEVICE_LIB=$lib timeout 300 python3 tools/gpu_quick.py 2>&1 | tail -2
done
