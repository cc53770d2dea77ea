# per-launch timeline of the one-call calibration step on a few MobileNetV2 activation shapes under a list of environment settings
# usage (GPU box, repo root): bash tools/ab_env_shapes.sh out_dir "fixed|search" "C,hw C,hw ..." "FP8Q_A=1" "FP8Q_A=2 FP8Q_B=3" ...
set -u
R=$PWD
OUT=$R/gpurun_out/$1
MODE=$2
SHAPES=$3
shift 3
export TMPDIR=/tmp
mkdir -p $OUT
cd /tmp
i=0
for setting in "$@"; do
    i=$((i+1))
    rm -rf /tmp/kt_es_$i
    env $setting rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_es_$i -o t -- python $R/tools/mb_calib_shapes.py $MODE pre $SHAPES > $OUT/shapes_$i.log 2>&1
    ( echo "== $setting ($MODE)"; grep "per step" $OUT/shapes_$i.log; python $R/tools/calib_timeline.py "$(find /tmp/kt_es_$i -name '*kernel_trace.csv' | head -1)" ) >> $OUT/summary_$MODE.txt 2>&1
done
cat $OUT/summary_$MODE.txt
