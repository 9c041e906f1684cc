set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 200 ./tools/cold_stream > gpurun_out/cold_stream.log 2>&1; rc=$?; echo "cold_stream rc=$rc"; cat gpurun_out/cold_stream.log
[ $rc -ge 124 ] && exit $rc
for cfg in "fe perm" "fe natural" "100 perm" "100 natural" "170 perm"; do
  set -- $cfg; tag=pc_$1_$2
  rm -rf gpurun_out/$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -- python tools/perm_cost.py $1 $2 > gpurun_out/$tag.log 2>&1
  rc=$?; echo "$tag rc=$rc"; grep -v "^W2\|rocprof" gpurun_out/$tag.log | tail -2
  [ $rc -ge 124 ] && exit $rc
done
echo RUN3_DONE
