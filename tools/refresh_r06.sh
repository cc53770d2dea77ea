#!/bin/bash
# Round 6: everything under profiles/r06_* in one GPU call (bash tools/refresh_r06.sh from the repo root on the GPU box).
set -u
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r06
bash tools/refresh_profiles.sh r06 > $R/gpurun_out/r06/refresh_profiles.log 2>&1
# per calibration pass of MobileNetV2 (config 4): kernels of the one-call step, 11 passes averaged
bash tools/trace_calib.sh _final > $R/gpurun_out/r06/trace_calib_final.log 2>&1
# the step per activation shape: events + kernel timelines
bash tools/run_shapes.sh _final > $R/gpurun_out/r06/run_shapes_final.log 2>&1
( for mode in fixed search; do python tools/mb_calib_shapes.py $mode plain 2>&1 | grep -v amdgpu.ids; done ) > $R/gpurun_out/r06/calib_shapes_plain_final.txt
# host side after the round's changes
python tools/host_profile.py search $R/gpurun_out/r06/host_search_final.txt > /dev/null 2>&1
python tools/host_profile.py fixed $R/gpurun_out/r06/host_fixed_final.txt > /dev/null 2>&1
python tools/host_profile.py c3 $R/gpurun_out/r06/host_c3_final.txt > /dev/null 2>&1
ls $R/gpurun_out/r06 | head -50
