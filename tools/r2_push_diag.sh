set -u
export TMPDIR=/tmp OMP_NUM_THREADS=1 HSA_ENABLE_IPC_MODE_LEGACY=0 MI355_DIST_FORCE_SELFCHECK=1 MI355_TEST_EXCHANGE=push MI355_TEST_TRACE=1 MI355_PUSH_SPIN_LOG2=19
mkdir -p gpurun_out
for i in 1 2 3 4; do
  timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port $((29650+i)) tests/dist_gpu_worker.py svar 160000 2000 > gpurun_out/diag_$i.log 2>&1
  rc=$?; echo "run $i rc=$rc"; grep -h "^\[rank\|MiError\|DIST_GPU" gpurun_out/diag_$i.log | tail -14
  [ $rc -ge 124 ] && exit $rc
done
