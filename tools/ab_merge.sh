# A/B of an FP8Q_* switch of the K4 interval-histogram route: K4 tests, the route's per-kernel timeline on [64,32,112,112] and the
# one-call calibration step per MobileNetV2 activation shape, with the switch at 1 and at 0.
# usage (GPU box, repo root): bash tools/ab_merge.sh [ENV_NAME] [out_dir]
set -u
R=$PWD
VAR=${1:-FP8Q_MSE_MERGE}
OUT=$R/gpurun_out/${2:-ab_merge}
export TMPDIR=/tmp
mkdir -p $OUT
python -m pytest tests -m gpu -x -q -k "mse or hist or calib or fuzz" 2>&1 | tail -4 > $OUT/tests.log
cd /tmp
for v in 1 0; do
    export $VAR=$v
    rm -rf /tmp/kt_ab_$v
    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_ab_$v -o mse -- python $R/tools/mb_mse.py > $OUT/mb_mse_$v.log 2>&1
    ( grep "MSE grid" $OUT/mb_mse_$v.log; python $R/tools/mse_timeline.py $(find /tmp/kt_ab_$v -name "*kernel_trace.csv" | head -1) ) > $OUT/mse_timeline_$v.txt 2>&1
    ( cd $R; for mode in fixed search; do python tools/mb_calib_shapes.py $mode pre 2>&1 | grep -v amdgpu.ids; done ) > $OUT/calib_shapes_$v.txt
    for mode in fixed search; do
        rm -rf /tmp/kt_sh_$mode
        rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_sh_$mode -o t -- python $R/tools/mb_calib_shapes.py $mode pre 32,112 24,56 192,14 64,14 160,7 > /dev/null 2>&1
        python $R/tools/calib_timeline.py "$(find /tmp/kt_sh_$mode -name '*kernel_trace.csv' | head -1)" $OUT/calib_timeline_${mode}_$v.txt
    done
done
cat $OUT/tests.log; grep TOTAL $OUT/calib_shapes_1.txt $OUT/calib_shapes_0.txt; cat $OUT/mse_timeline_1.txt
