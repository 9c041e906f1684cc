#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r3h_sim_rank.txt
for args in "8 1" "8 0" "4 1" "2 1"; do
  echo "== sim_rank $args" >> gpurun_out/r3h_sim_rank.txt
  timeout -k 10 300 python tools/sim_rank.py $args 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3h_sim_rank.txt; echo "sim_rank $args rc=$?"
done
timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1-GPU product on this box:', d['roofline']['launch_us'], 'us')" >> gpurun_out/r3h_sim_rank.txt
cat gpurun_out/r3h_sim_rank.txt
