#!/bin/bash
# Regenerate the judged artefacts on a GPU box: bench line, rocprofv3 kernel stats of the same command, PMC traffic.
# usage (from the repo root, on the GPU box): bash tools/refresh_profiles.sh r01     -> gpurun_out/<tag>_*
set -u
TAG=${1:-r01}
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
python bench.py 2> $R/gpurun_out/${TAG}_bench.err | tail -1 > $R/gpurun_out/${TAG}_bench_n1.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_kt -o bench -- python $R/bench.py --no-extras --no-cpu-baseline > $R/gpurun_out/${TAG}_kt.log 2>&1
export FP8Q_BENCH_PREWARM_S=0
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pf -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-extras --no-cpu-baseline > $R/gpurun_out/${TAG}_pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pw -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-extras --no-cpu-baseline > $R/gpurun_out/${TAG}_pw.log 2>&1
cd $R
for d in kt pf pw; do   # rocprofv3 nests its files under <dir>/<host>/: flatten
    find gpurun_out/${TAG}_$d -mindepth 2 -name "bench_*.csv" -exec cp {} gpurun_out/${TAG}_$d/ \;
done
# gpurun merges only gpurun_out/ back; afterwards, in the container:
#   python tools/summarize_profiles.py gpurun_out/${TAG}_kt gpurun_out/${TAG}_pf gpurun_out/${TAG}_pw $TAG "k_rows_flat<0"
#   cp gpurun_out/${TAG}_bench_n1.json profiles/${TAG}_bench_n1.json
tail -c 400 gpurun_out/${TAG}_bench_n1.json; echo; head -4 gpurun_out/${TAG}_kt/bench_kernel_stats.csv | cut -c1-200
