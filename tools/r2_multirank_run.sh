set -u
mkdir -p gpurun_out; export TMPDIR=/tmp MI355_FORCE_DEVICE=0 MI355_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
port=29700
for cfg in "2 c2" "2 c3" "3 c3" "2 fe"; do
  set -- $cfg; port=$((port+1))
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus $1 --workload $2 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/mr_$1_$2.log 2>&1
  rc=$?; echo "ranks $1 workload $2 rc=$rc"; grep "^{" gpurun_out/mr_$1_$2.log | tail -1 | cut -c1-900
  [ $rc -ge 124 ] && exit $rc
done
echo MR_DONE
