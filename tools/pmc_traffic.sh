#!/bin/bash
# HBM traffic of the conv kernels from the TCC counters, as MI355X_MICROARCH.md prescribes: one
# rocprofv3 pass per counter (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only.
# Output: gpurun_out/pmc/{fetch,write}/..._counter_collection.csv -> tools/pmc_summarize.py
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- \
  python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-conv-events --no-lstm-roofline > $OUT/fetch.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- \
  python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-conv-events --no-lstm-roofline > $OUT/write.log 2>&1 || exit 1
ls $OUT/fetch $OUT/write
