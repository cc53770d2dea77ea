#!/bin/bash
# BASELINE configs 3 and 4 end to end through the CLI (synthetic ImageNet-shaped data, random-init weights)
cd "$(dirname "$0")/../fp8-quantization_amd"
C3="validate-quantized --architecture resnet18_quantized --batch-size 64 --seed 10 --n-bits 8 --cuda --load-type fp32 --quant-setup all --qmethod fp_quantizer --per-channel --fp8-mantissa-bits=2 --fp8-set-maxval --no-fp8-mse-include-mantissa-bits --weight-quant-method=current_minmax --act-quant-method=allminmax --num-est-batches=1 --synthetic-batches 4"
C4="validate-quantized --architecture mobilenet_v2_quantized --batch-size 64 --seed 10 --n-bits 8 --cuda --load-type fp32 --quant-setup all --qmethod fp_quantizer --per-channel --fp8-mantissa-bits=3 --fp8-set-maxval --fp8-mse-include-mantissa-bits --weight-quant-method=MSE --act-quant-method=MSE --num-est-batches=1 --synthetic-batches 4"
for c in "$C3" "$C4"; do
  t0=$(date +%s.%N)
  python image_net.py $c 2>&1 | grep "top_1_accuracy" | tail -1
  t1=$(date +%s.%N)
  python -c "print(\"wall %.1f s\" % ($t1 - $t0))"
done
