#!/bin/bash
# round 4: final VALU ceiling file, un-profiled bench line, rocprofv3 evidence for the bench command
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r04c
timeout 400 ./tools/micro/valu_issue 20000 > gpurun_out/r04c/valu_ceiling.txt 2>&1; echo "valu rc=$?"; tail -2 gpurun_out/r04c/valu_ceiling.txt
timeout 900 python3 bench.py --steps 3 --warmup 1 > gpurun_out/r04c/bench_line.json 2> gpurun_out/r04c/bench.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/r04c/bench_line.json
bash tools/prof_bench.sh r04 2>&1 | tail -60
