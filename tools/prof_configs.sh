#!/bin/bash
# rocprofv3 evidence for BASELINE.json configs 2, 4, 5: kernel stats + SQ / TA counters of the trace kernel,
# separate passes over tools/prof_run.py.   tools/prof_configs.sh <tag>
set -u
TAG=${1:-r04}
OUT=gpurun_out/prof_configs_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
run() { cfg=$1; name=$2; shift; shift; args=$1; shift; (cd /tmp && timeout -k 5 90 rocprofv3 "$@" --output-format csv -d $ROOT/$OUT/${cfg}_$name -- python3 $ROOT/tools/prof_run.py $args > $ROOT/$OUT/${cfg}_$name.log 2>&1; echo "pass $cfg $name rc=$?"); }
for cfg in c2 c4 c5; do
  case $cfg in
    c2) A="--scene random_spheres --iarg 497 --width 1280 --height 720 --spp 64 --reps 2";;
    c4) A="--scene dining_standin --iarg 6 --width 1920 --height 1080 --spp 16 --reps 2";;
    c5) A="--scene million_standin --iarg 1000000 --width 1920 --height 1080 --spp 16 --reps 2";;
  esac
  run $cfg kt "$A" --kernel-trace --stats
  run $cfg sq "$A" --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
  run $cfg ta "$A" --kernel-trace --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE
  run $cfg fetch "$A" --kernel-trace --pmc FETCH_SIZE
  run $cfg write "$A" --kernel-trace --pmc WRITE_SIZE
  run $cfg l2 "$A" --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum
done
python3 tools/prof_configs_summary.py $OUT
