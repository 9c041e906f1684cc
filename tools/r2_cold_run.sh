#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/cold_probe.log
run() { timeout -k 10 300 env "$@" >> gpurun_out/cold_probe.log 2>&1; rc=$?; [ $rc -ge 124 ] && { echo "timeout"; tail -5 gpurun_out/cold_probe.log; exit $rc; }; }
run MI355_X=1 python tools/cold_probe.py c4
run MI355_RING_DEPTH=3 python tools/cold_probe.py c4
run MI355_RING_DEPTH=4 python tools/cold_probe.py c4
run MI355_X=1 python tools/cold_probe.py c4 stream
run MI355_X=1 python tools/cold_probe.py fe
run MI355_X=1 python tools/cold_probe.py c2
cat gpurun_out/cold_probe.log
echo COLD_DONE
