# per-shape timing of the one-call calibration step + a kernel timeline of a few shapes; usage: bash tools/run_shapes.sh <suffix>
set -u
R=$PWD
SUF=${1:-}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r06
( for mode in fixed search; do python tools/mb_calib_shapes.py $mode pre 2>&1 | grep -v amdgpu.ids; done ) > $R/gpurun_out/r06/calib_shapes$SUF.txt
cd /tmp
for mode in fixed search; do
    rm -rf /tmp/kt_sh_$mode
    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_sh_$mode -o t -- python $R/tools/mb_calib_shapes.py $mode pre 32,112 24,56 192,14 64,14 160,7 > /dev/null 2>&1
    f=$(find /tmp/kt_sh_$mode -name "*kernel_trace.csv" | head -1)
    python $R/tools/calib_timeline.py "$f" $R/gpurun_out/r06/calib_timeline_$mode$SUF.txt
done
tail -3 $R/gpurun_out/r06/calib_shapes$SUF.txt
