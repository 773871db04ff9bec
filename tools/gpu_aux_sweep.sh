#!/bin/bash
for part in 0/4 0/8; do
echo -n "HEAD  PART $part variant 4: "; MOPTIX_DEVICE_LIB=libmoptix_head.so SPP=256 WARM=2 PART=$part OPTS=kernel_variant=4 timeout 300 python tools/gpu_quick.py 2>&1 | tail -1
echo -n "new   PART $part variant 4 ud0 ad0: "; SPP=256 WARM=2 PART=$part OPTS=kernel_variant=4,urgent_depth=0,aux_depth=0 timeout 300 python tools/gpu_quick.py 2>&1 | tail -1
done
