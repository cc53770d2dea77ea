#!/bin/bash
# Regenerate the judged artefacts on a GPU box: bench line, rocprofv3 kernel stats of the same command, PMC traffic,
# and the per-kernel profiles of K3 / K4 / the fused short-row kernel.
# usage (from the repo root, on the GPU box): bash tools/refresh_profiles.sh r04     -> gpurun_out/<tag>_*
set -u
TAG=${1:-r05}
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
python bench.py 2> $R/gpurun_out/${TAG}_bench.err | tail -1 > $R/gpurun_out/${TAG}_bench_n1.json
HEAD="--no-extras --no-cpu-baseline --no-north-star-path --no-traffic"     # the headline kernel alone
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_kt -o bench -- python $R/bench.py $HEAD > $R/gpurun_out/${TAG}_kt.log 2>&1
export FP8Q_BENCH_PREWARM_S=0
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pf -o bench -- python $R/bench.py --steps 5 --warmup 1 $HEAD > $R/gpurun_out/${TAG}_pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pw -o bench -- python $R/bench.py --steps 5 --warmup 1 $HEAD > $R/gpurun_out/${TAG}_pw.log 2>&1
unset FP8Q_BENCH_PREWARM_S
# K4 (MSE grid search), K3 (single-launch min/max), fused short rows: kernel stats + VALU / traffic counters
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_mse_kt -o mse -- python $R/tools/mb_mse.py > $R/gpurun_out/${TAG}_mse_kt.log 2>&1
# the interval-histogram route call by call (median per kernel, span) and both routes over tensor sizes (routing model)
( grep "MSE grid" $R/gpurun_out/${TAG}_mse_kt.log; python $R/tools/mse_timeline.py $(find $R/gpurun_out/${TAG}_mse_kt -name "*kernel_trace.csv" | head -1) ) > $R/gpurun_out/${TAG}_mse_timeline.txt 2>&1
( echo "# python tools/mb_mse_sizes.py: K4 on one per-tensor row, 111 candidates x n_m widths; FP8Q_MSE_HIST=0: lane-per-element kernel only, =1: default routing"; FP8Q_MSE_HIST=0 python $R/tools/mb_mse_sizes.py 2>&1 | grep HIST; FP8Q_MSE_HIST=1 python $R/tools/mb_mse_sizes.py 2>&1 | grep HIST ) > $R/gpurun_out/${TAG}_mse_sizes.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${TAG}_mse_pmc -o mse -- python $R/tools/mb_mse_one.py 1 > $R/gpurun_out/${TAG}_mse_pmc.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/${TAG}_mse_pmc2 -o mse -- python $R/tools/mb_mse_one.py 1 > $R/gpurun_out/${TAG}_mse_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_mse_pmc3 -o mse -- python $R/tools/mb_mse_one.py 1 > $R/gpurun_out/${TAG}_mse_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_mse_pmc4 -o mse -- python $R/tools/mb_mse_one.py 1 > $R/gpurun_out/${TAG}_mse_pmc4.log 2>&1
for d in mse_pmc mse_pmc2 mse_pmc3 mse_pmc4; do python $R/tools/pmc_by_kernel.py $R/gpurun_out/${TAG}_$d; done > $R/gpurun_out/${TAG}_mse_pmc_by_kernel.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_k3_kt -o k3 -- python $R/tools/mb_k3.py > $R/gpurun_out/${TAG}_k3_kt.log 2>&1
CHECK=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_staged_kt -o staged -- python $R/tools/mb_staged.py > $R/gpurun_out/${TAG}_staged_kt.log 2>&1
# BASELINE configs 3 / 4 at full size: kernel trace of one calibration batch + fix_ranges + validation forwards
for cfg in c3 c4 c4_search; do
    rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_${cfg}_kt -o $cfg -- python $R/bench.py --only-model-config $cfg > $R/gpurun_out/${TAG}_${cfg}.log 2> $R/gpurun_out/${TAG}_${cfg}.err
done
# the multi-tensor plan, the fused epilogue shapes and the short-row encode: GPU-side durations per (kernel, grid)
for set in multi epi enc; do
    ITERS=30 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_ab_${set}_kt -o $set -- python $R/tools/ab.py $set > $R/gpurun_out/${TAG}_ab_${set}.log 2>&1
    python $R/tools/trace_by_grid.py $R/gpurun_out/${TAG}_ab_${set}_kt k_ > $R/gpurun_out/${TAG}_ab_${set}_by_grid.txt 2>&1
done
# BASELINE config 1 at full size (float64 lane): wall time of the whole script + its printout
cd $R/fp8-quantization_amd && ( time python compute_quant_error.py ) > $R/gpurun_out/${TAG}_config1_full_size.txt 2>&1
cd $R && timeout 400 python tests/soak.py --seconds 240 --seed 5 > $R/gpurun_out/${TAG}_soak.txt 2>&1
cd $R
for d in c3_kt c4_kt c4_search_kt; do find gpurun_out/${TAG}_$d -mindepth 2 -name "*.csv" -exec cp {} gpurun_out/${TAG}_$d/ \; ; done
for d in kt pf pw mse_kt k3_kt staged_kt; do   # rocprofv3 nests its files under <dir>/<host>/: flatten
    find gpurun_out/${TAG}_$d -mindepth 2 -name "*.csv" -exec cp {} gpurun_out/${TAG}_$d/ \;
done
# gpurun merges only gpurun_out/ back; afterwards, in the container:
#   python tools/summarize_profiles.py gpurun_out/${TAG}_kt gpurun_out/${TAG}_pf gpurun_out/${TAG}_pw $TAG "k_rows_flat<0"
#   python tools/summarize_kernels.py $TAG
#   python tools/summarize_kernels.py model $TAG c3 gpurun_out/${TAG}_c3_kt/c3_kernel_trace.csv gpurun_out/${TAG}_c3.log "<title>"   (also c4, c4_search)
tail -c 600 gpurun_out/${TAG}_bench_n1.json; echo; head -4 gpurun_out/${TAG}_kt/bench_kernel_stats.csv | cut -c1-200
