# the K4 route's per-kernel timeline on [64,32,112,112] (tools/mb_mse.py) under a list of environment settings
# usage (GPU box, repo root): bash tools/ab_env.sh out_dir "FP8Q_A=1 FP8Q_B=2" "FP8Q_A=3" ...
set -u
R=$PWD
OUT=$R/gpurun_out/$1
shift
export TMPDIR=/tmp
mkdir -p $OUT
cd /tmp
i=0
for setting in "$@"; do
    i=$((i+1))
    rm -rf /tmp/kt_env_$i
    env $setting rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_env_$i -o mse -- python $R/tools/mb_mse.py > $OUT/mb_mse_$i.log 2>&1
    ( echo "== $setting"; grep "MSE grid act" $OUT/mb_mse_$i.log; python $R/tools/mse_timeline.py $(find /tmp/kt_env_$i -name "*kernel_trace.csv" | head -1) | grep -E "calls|k_|SPAN" ) >> $OUT/summary.txt 2>&1
done
cat $OUT/summary.txt
