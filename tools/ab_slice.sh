set -u
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r06
for sl in 2048 4096 8192 16384; do
  echo "== FP8Q_MSE_SLICE=$sl" 
  FP8Q_MSE_SLICE=$sl python tools/mb_calib_shapes.py fixed pre 24,56 32,28 192,14 96,14 576,7 960,7 1280,7 32,112 2>&1 | grep -v amdgpu
done > $R/gpurun_out/r06/ab_slice.txt
cd /tmp
for sl in 4096 16384; do
    rm -rf /tmp/kt_sl
    FP8Q_MSE_SLICE=$sl rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_sl -o t -- python $R/tools/mb_calib_shapes.py fixed pre 24,56 192,14 > /dev/null 2>&1
    f=$(find /tmp/kt_sl -name "*kernel_trace.csv" | head -1)
    python $R/tools/calib_timeline.py "$f" $R/gpurun_out/r06/ab_slice_timeline_$sl.txt
done
