set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_tests.log 2>&1
rc=$?; tail -n 8 gpurun_out/t_tests.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py > gpurun_out/t_bench_c4.log 2>&1; rc=$?; echo "bench rc=$rc"; tail -c 3000 gpurun_out/t_bench_c4.log
[ $rc -ge 124 ] && exit $rc
timeout -k 10 300 python bench.py --workload fe --no-cpu-baseline > gpurun_out/t_bench_fe.log 2>&1; echo "fe rc=$?"; tail -c 1800 gpurun_out/t_bench_fe.log
echo RUN5_DONE
