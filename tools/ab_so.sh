# old-vs-new library on the one-call calibration step per activation shape (tools/mb_calib_shapes.py), alternating on one box
# usage (GPU box, repo root): bash tools/ab_so.sh out_name path/to/old/libfp8q_hip.so "fixed pre" "search pre" "fixed plain"
set -u
R=$PWD
OUT=$R/gpurun_out/$1
OLD=$2
shift 2
mkdir -p $(dirname $OUT)
: > $OUT
for mode in "$@"; do
    for rep in 1 2; do
        for tag in new old; do
            echo "== $mode / $tag (run $rep)" >> $OUT
            if [ $tag = old ]; then
                FP8Q_SO=$OLD python tools/mb_calib_shapes.py $mode 2>&1 | grep -v amdgpu.ids >> $OUT
            else
                python tools/mb_calib_shapes.py $mode 2>&1 | grep -v amdgpu.ids >> $OUT
            fi
        done
    done
done
grep -E "^==|TOTAL" $OUT
