#!/bin/bash
# Produces the artifacts committed under profiles/ (run on the GPU box through gpurun):
#   gpurun_out/prof/bench_default.json          python bench.py (default flags)
#   gpurun_out/prof/bench_att.json              python bench.py --decoder att
#   gpurun_out/prof/stats/*_kernel_stats.csv    rocprofv3 --kernel-trace --stats of the bench
#   gpurun_out/prof/bench_under_rocprof.json    the bench line printed under the profiler
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd $ROOT
timeout -k 10 400 python bench.py 2> $OUT/bench_default.log | tail -1 > $OUT/bench_default.json || exit 1
timeout -k 10 300 python bench.py --decoder att --no-cpu-baseline 2> $OUT/bench_att.log | tail -1 > $OUT/bench_att.json || exit 1
# SURVEY 8(d) Config 4: the attention decoder at 12 images per GPU (global 96 on 8 GPUs) and at 96 per GPU
timeout -k 10 300 python bench.py --decoder att --batch 12 --no-cpu-baseline --no-lstm-roofline 2> $OUT/bench_att_b12.log | tail -1 > $OUT/bench_att_b12.json || exit 1
timeout -k 10 300 python bench.py --decoder att --batch 96 --steps 40 --no-cpu-baseline --no-lstm-roofline 2> $OUT/bench_att_b96.log | tail -1 > $OUT/bench_att_b96.json || exit 1
# configs[4]'s decoder shape on capnet.stacked (perf-only, parity unpinned)
timeout -k 10 300 python bench.py --layers 3 --factored 1024 --batch 96 --steps 60 --no-cpu-baseline --no-lstm-roofline 2> $OUT/bench_stacked3.log | tail -1 > $OUT/bench_stacked3.json || exit 1
timeout -k 10 300 python bench.py --decoder nic --no-cpu-baseline 2> $OUT/bench_nic.log | tail -1 > $OUT/bench_nic.json || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- \
  python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/rocprof.log 2>&1 || exit 1
grep '^{' $OUT/rocprof.log | tail -1 > $OUT/bench_under_rocprof.json
ls $OUT $OUT/stats
