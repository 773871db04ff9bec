#!/bin/bash
# two frames in flight: one GPU whole frame, and rank 0's share of an 8-way / 4-way tile split
for env in "" "MOPTIX_BENCH_PIPELINE=1"; do
  echo "== whole frame $env"; env $env python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fast-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['ranks']['kernel_ms_per_frame'])"
  for n in 8 4; do echo "== emulated rank 0 of $n $env"; env $env MOPTIX_BENCH_EMULATE_RANKS=$n python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-fast-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['ranks']['kernel_ms_per_frame'], d['config']['ranks']['counted_launch_tail_ms'])"; done
done
