#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
MI355_SPMM_TILE=1 timeout -k 10 600 python -m pytest tests/test_spmm_gpu.py -m gpu -x -q > gpurun_out/r3d_pytest_tile.log 2>&1; echo "pytest tile=1 rc=$?"
tail -3 gpurun_out/r3d_pytest_tile.log
MI355_SPMM_TILE=1 MI355_SPMM_TILE_SHAPE=512x2 timeout -k 10 600 python -m pytest tests/test_spmm_gpu.py -m gpu -x -q > gpurun_out/r3d_pytest_tile2.log 2>&1; echo "pytest tile=1 512x2 rc=$?"
tail -3 gpurun_out/r3d_pytest_tile2.log
: > gpurun_out/r3d_bench.jsonl
export MI355_SPMM_TILE=1
for w in fe_spmm4 fe_spmm8; do
  for shp in 256x1 512x1 256x2 512x2; do
    for p in 2 3 4; do
      MI355_SPMM_TILE_SHAPE=$shp MI355_SPMM_TILE_P=$p timeout -k 10 300 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline --no-parity >> gpurun_out/r3d_bench.jsonl 2>> gpurun_out/r3d_bench.err; echo "bench $w $shp P=$p rc=$?"
    done
  done
done
python - <<'PY'
import json
for ln in open('gpurun_out/r3d_bench.jsonl'):
    d = json.loads(ln)
    print(d['config']['name'], 'us', d['roofline']['launch_us'], 'frac', d['roofline']['frac'], 'GF', d['value'], d.get('kernel_info',{}).get('kernel'), d.get('kernel_info',{}).get('longest_list'))
PY
