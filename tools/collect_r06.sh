#!/bin/bash
# Container side of tools/refresh_r06.sh: turn what the GPU call merged into gpurun_out/ into the tracked files under profiles/.
# usage (repo root, after `gpurun -- bash tools/refresh_r06.sh`): bash tools/collect_r06.sh "<one line saying which build this is>"
set -eu
NOTE=${1:-"final build of round 6"}
G=gpurun_out
P=profiles
T=r06
python tools/summarize_profiles.py $G/${T}_kt $G/${T}_pf $G/${T}_pw $T "k_rows_flat<0"
python tools/summarize_kernels.py $T
for cfg in c3 c4 c4_search; do
    python tools/summarize_kernels.py model $T $cfg $G/${T}_${cfg}_kt/${cfg}_kernel_trace.csv $G/${T}_${cfg}.log "round 6, $NOTE"
done
cp $G/${T}_bench_n1.json $P/${T}_bench_n1.json
cp $G/${T}_mse_timeline.txt $P/${T}_mse_timeline.txt
cp $G/${T}_mse_sizes.txt $P/${T}_mse_sizes.txt
cp $G/${T}_mse_pmc_by_kernel.txt $P/${T}_mse_pmc_by_kernel.txt
cp $G/${T}_config1_full_size.txt $P/${T}_config1_full_size.txt
for set in multi epi enc; do cp $G/${T}_ab_${set}_by_grid.txt $P/${T}_${set}_kernels_by_grid.txt; done
( echo "# $NOTE"; cat $G/${T}_soak.txt ) > $P/${T}_soak.txt
# the per-pass / per-shape / per-launch tables of the MobileNetV2 MSE calibration (tools/trace_calib.sh, tools/run_shapes.sh)
for m in fixed search; do
    ms=$(grep "per warm" $G/r06/kt_${m}_final.log | tail -1 | sed 's/.*: //')
    ( echo "# tools/trace_calib.sh: rocprofv3 --kernel-trace --stats over 1 cold + 10 warm MobileNetV2 MSE calibration passes ($m), totals / 11; $NOTE"
      echo "# weight quantizers on a side stream next to the activations' chains: kernel durations overlap (they do not add up to the pass); under the tracer: $ms"
      cat $G/r06/calib_${m}_kernels_final.txt ) > $P/${T}_c4_pass_kernels_$m.txt
    ( echo "# rocprofv3 --kernel-trace of tools/mb_calib_shapes.py $m pre 32,112 24,56 192,14 64,14 160,7 (tools/calib_timeline.py): every launch of one calibration step, median of 10 steps; $NOTE"
      echo "# a kernel's figure runs from the end of the launch before it to its own end (launch gaps included: the figures add up to the span)"
      grep -v "^#" $G/r06/calib_timeline_${m}_final.txt ) > $P/${T}_calib_timeline_$m.txt
done
( echo "# tools/mb_calib_shapes.py: ONE MSE calibration step (fp8q_mse_calibrate_f32: epilogue + abs-max + grid, search, selection, quantization) per MobileNetV2 activation shape at batch 64,"
  echo "# first batch of a fresh estimator, GPU time per step by HIP events around 8 steps; 'pre' = behind BN + ReLU6 (the epilogue in the same call), 'plain' = on the tensor itself; $NOTE"
  cat $G/r06/calib_shapes_final.txt $G/r06/calib_shapes_plain_final.txt ) > $P/${T}_mse_c4_shapes.txt
( echo "# tools/host_profile.py search / fixed / c3: host timeline of one calibration pass and of fix_ranges(); $NOTE"
  cat $G/r06/host_search_final.txt; echo; echo "######## fixed mantissa width"; cat $G/r06/host_fixed_final.txt; echo; echo "######## ResNet-18 (config 3)"; cat $G/r06/host_c3_final.txt ) > $P/${T}_host_profile_after.txt
ls -l $P/${T}_* | wc -l
