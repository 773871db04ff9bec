#!/bin/bash
# PC sampling of the trace kernel (rocprofv3 beta): where the waves' program counters are, with stall reasons
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04_pcsamp
mkdir -p $OUT
cd /tmp
for method in stochastic host_trap; do
  unit=cycles; interval=1048576
  if [ $method = host_trap ]; then unit=time; interval=100; fi
  ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout -k 5 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $unit --pc-sampling-method $method --pc-sampling-interval $interval \
     --kernel-trace --output-format csv -d $OUT/$method -- python3 $ROOT/tools/prof_run.py --spp ${SPP:-32} --reps 1 > $OUT/$method.log 2>&1
  echo "$method rc=$?"; tail -5 $OUT/$method.log
  find $OUT/$method -type f | head; du -sh $OUT/$method
done
