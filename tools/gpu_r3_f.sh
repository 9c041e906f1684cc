#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "graph_capture or two_streams or reorder or one_launch or sharing_one_card" > gpurun_out/r3f_pytest.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/r3f_pytest.log
