# rocprofv3 --kernel-trace --stats of N warm MobileNetV2 MSE calibration passes -> gpurun_out/r06/calib_{fixed,search}_kernels<suffix>.txt
# usage (from the repo root on the GPU box): bash tools/trace_calib.sh [suffix] [modes...]
set -u
R=$PWD
SUF=${1:-}
shift || true
MODES=${*:-fixed search}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r06
cd /tmp
for m in $MODES; do
    rm -rf /tmp/kt_$m
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$m -o $m -- python $R/tools/calib_passes.py $m 10 > $R/gpurun_out/r06/kt_$m$SUF.log 2>&1
    f=$(find /tmp/kt_$m -name "*kernel_stats.csv" | head -1)
    python $R/tools/calib_passes.py summarize "$f" 11 $R/gpurun_out/r06/calib_${m}_kernels$SUF.txt
    grep "per warm" $R/gpurun_out/r06/kt_$m$SUF.log
done
