#!/bin/bash
# round 4, first GPU call: VALU issue ceiling, baseline bench line, two ranks on one GPU through real RCCL
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r04a
timeout 300 ./tools/micro/valu_issue > gpurun_out/r04a/valu_ceiling.txt 2>&1; echo "valu rc=$?"
cat gpurun_out/r04a/valu_ceiling.txt
timeout 600 python3 bench.py --steps 3 --warmup 1 > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err; echo "bench rc=$?"
cut -c1-1500 gpurun_out/r04a/bench.json
NCCL_DEBUG=WARN timeout 300 python3 tools/rccl_two_ranks.py 2 --same-device > gpurun_out/r04a/rccl2.txt 2>&1; echo "rccl2 rc=$?"
tail -30 gpurun_out/r04a/rccl2.txt
