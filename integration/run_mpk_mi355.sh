#!/bin/bash
# The reference's run-over-ten-matrices drivers (mpk/SpMV.sh, mpk/SpM2V.sh: `for i in {1..10}; do ./spm2v mat/matrix${i}_aij.mtx >> log/...`)
# for the GPU twins built by integration/mpk_mi355.mk.  Run from the reference's mpk/ directory; logs go beside the reference's own.
mkdir -p log
rm -f log/log_2SPMV_mi355.txt log/log_SPM2V_mi355.txt
for i in {1..10}; do
    [ -f mat/matrix${i}_aij.mtx ] || { echo "mat/matrix${i}_aij.mtx is missing (the reference does not ship its matrices)"; continue; }
    ./2spmv_mi355 mat/matrix${i}_aij.mtx >> log/log_2SPMV_mi355.txt 2>&1
    ./spm2v_mi355 mat/matrix${i}_aij.mtx >> log/log_SPM2V_mi355.txt 2>&1
done
