#!/bin/bash
# BASELINE.json configs 2, 4, 5 on the kernels that run them today: rocprofv3 kernel stats + SQ / TA / FETCH / WRITE / TCC
# counters, each in its OWN pass over `python3 bench.py --scene ...`, then the un-profiled bench.py line with those counters replayed.
#   tools/prof_configs.sh <tag>     -> gpurun_out/prof_configs_<tag>/{c2,c4,c5}/..., <tag>_traffic_<cfg>.json, line_<cfg>.json, configs.json
set -u
TAG=${1:?round tag, e.g. r06}
OUT=gpurun_out/prof_configs_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
run() { cfg=$1; name=$2; shift; shift; args=$1; shift; (cd /tmp && timeout -k 5 150 rocprofv3 "$@" --output-format csv -d $ROOT/$OUT/$cfg/$name -- python3 $ROOT/bench.py $args > $ROOT/$OUT/${cfg}_$name.log 2>&1; echo "pass $cfg $name rc=$?"); }
for cfg in ${CONFIGS:-c2 c4 c5}; do
  case $cfg in
    c2) A="--scene random_spheres --iarg 497 --width 1280 --height 720 --spp 64";;
    c4) A="--scene dining_standin --iarg 6 --width 1920 --height 1080 --spp 16";;
    c5) A="--scene million_standin --iarg 1000000 --width 1920 --height 1080 --spp 16";;
  esac
  P="$A --steps 2 --warmup 1 --no-cpu-baseline --no-fast-leg"
  run $cfg kt "$P" --kernel-trace --stats
  run $cfg sq "$P" --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
  run $cfg ta "$P" --kernel-trace --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE
  run $cfg fetch "$P" --kernel-trace --pmc FETCH_SIZE
  run $cfg write "$P" --kernel-trace --pmc WRITE_SIZE
  run $cfg l2 "$P" --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum
  python3 tools/make_traffic_json.py $OUT/$cfg $OUT/${TAG}_traffic_$cfg.json "$A" > $OUT/${cfg}_traffic.log 2>&1
  cp $OUT/${TAG}_traffic_$cfg.json profiles/${TAG}_traffic_$cfg.json       # (on the box: the line below replays it)
  python3 bench.py $A --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/line_$cfg.json
  for f in $(find $OUT/$cfg/kt -name "*kernel_stats.csv"); do echo "== $cfg kernel stats"; head -6 $f | cut -c1-200; done
  echo "== $cfg line"; cut -c1-600 $OUT/line_$cfg.json
done
python3 - <<PY
import json
out = {}
for c in "${CONFIGS:-c2 c4 c5}".split():
    try: out[c] = json.loads(open("$OUT/line_%s.json" % c).read())
    except Exception as e: print("no line for", c, e)
json.dump(out, open("$OUT/configs.json", "w"), indent=1)
PY
