set -u
TAG=r05
R=$PWD
export TMPDIR=/tmp
python bench.py 2> $R/gpurun_out/${TAG}_bench.err | tail -1 > $R/gpurun_out/${TAG}_bench_n1.json
( echo "# python tools/mb_mse_weights.py: K4 on MobileNetV2's 53 per-channel weight tensors, 111 candidates (HIP events around the Python call)"; echo "# FP8Q_MSE_GRID_TILE=2048 (whole rows per workgroup, two launches per call only when a row is split):"; FP8Q_MSE_GRID_TILE=2048 python tools/mb_mse_weights.py 2>&1 | grep -v amdgpu.ids; echo "# shipped (rows cut down to 64 elements until the launch has ~4096 workgroups; single-split rows write the table themselves):"; python tools/mb_mse_weights.py 2>&1 | grep -v amdgpu.ids ) > $R/gpurun_out/${TAG}_mse_weights.txt
cd /tmp
for cfg in c4 c4_search; do
    rm -rf $R/gpurun_out/${TAG}_${cfg}_kt
    rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_${cfg}_kt -o $cfg -- python $R/bench.py --only-model-config $cfg > $R/gpurun_out/${TAG}_${cfg}.log 2> $R/gpurun_out/${TAG}_${cfg}.err
done
cd $R
for d in c4_kt c4_search_kt; do find gpurun_out/${TAG}_$d -mindepth 2 -name "*.csv" -exec cp {} gpurun_out/${TAG}_$d/ \; ; done
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/final3_gpu_tests.log 2>&1; tail -2 gpurun_out/final3_gpu_tests.log
timeout 260 python tests/soak.py --seconds 170 --seed 31337 > gpurun_out/final3_soak.log 2>&1; tail -2 gpurun_out/final3_soak.log
tail -c 300 gpurun_out/${TAG}_bench_n1.json
