cd /tmp && export TMPDIR=/tmp
for g in 1024 2048 4096; do
FP8Q_MULTI_GRID=$g rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_mb_kt_$g -o mb -- python $GRAFT_REPO_ROOT/tools/_mb.py > $GRAFT_REPO_ROOT/gpurun_out/r03_mb_kt.log 2>&1
done
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_api.py -m gpu -x -q -k "multi or plan or Plan" 2>&1 | tail -2
