set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for w in c2 c3; do
  rm -rf gpurun_out/r2k_$w
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2k_$w -- python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --no-parity --no-extras > gpurun_out/r2k_$w.log 2>&1; echo "trace $w rc=$?"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/r2p_${w}_$c
    MI355_SPMV_KERNEL=ring MI355_RING_NT=0 timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/r2p_${w}_$c -- python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-extras > gpurun_out/r2p_${w}_$c.log 2>&1; echo "pmc $w $c rc=$?"
  done
done
