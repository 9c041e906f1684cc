#!/bin/bash
# round 3, GPU pass A: the whole -m gpu suite, the default bench line, the self-launched 2-rank bench on one card
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3a_pytest.log
tail -5 gpurun_out/r3a_pytest.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3a_bench_c4.json 2> gpurun_out/r3a_bench_c4.err; echo "bench c4 rc=$?"
MI355_FORCE_DEVICE=0 MI355_BENCH_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 2 --workload c2 --steps 50 --warmup 5 > gpurun_out/r3a_bench_c2_n2.json 2> gpurun_out/r3a_bench_c2_n2.err; echo "bench c2 n2 rc=$?"
tail -c 600 gpurun_out/r3a_bench_c2_n2.err
