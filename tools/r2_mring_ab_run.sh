set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_reorder_gpu.py -x -q -m gpu -k "mring or seeded or degenerate or relabelled or scrambled or golden" > gpurun_out/t_tests.log 2>&1
rc=$?; tail -n 5 gpurun_out/t_tests.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tools/mring_ab.py > gpurun_out/mring_ab.log 2>&1; cat gpurun_out/mring_ab.log
