#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r3d_bench.jsonl
export MI355_SPMM_TILE=1
for w in fe_spmm4 fe_spmm8; do
  for shp in 512x1 512x2; do
    for dbg in 0 1 2 3; do
      MI355_SPMM_DBG=$dbg MI355_SPMM_TILE_SHAPE=$shp MI355_SPMM_TILE_P=3 timeout -k 10 300 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline --no-parity >> gpurun_out/r3d_bench.jsonl 2>> gpurun_out/r3d_bench.err; echo "bench $w $shp dbg=$dbg rc=$?"
    done
  done
done
MI355_SPMM_TILE=0 timeout -k 10 300 python bench.py --workload fe_bcsr --steps 40 --warmup 5 --no-cpu-baseline --no-parity --no-extras >> gpurun_out/r3d_bench.jsonl 2>> gpurun_out/r3d_bench.err
python - <<'PY'
import json
for ln in open('gpurun_out/r3d_bench.jsonl'):
    d = json.loads(ln)
    print(d['config']['name'], 'us', d['roofline']['launch_us'], 'frac', d['roofline']['frac'], d.get('kernel_info',{}).get('kernel'))
PY
