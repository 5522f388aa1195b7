#!/bin/bash
# SQ counters of round 4's kernels ALONE (tools/probes/fused_block_bench.py: the fused boundary kernel, the Gram statistics kernels and
# the launches they replace), two passes of eight counters -> gpurun_out/sq_r4/{a,b}/..counter_collection.csv
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $ROOT/gpurun_out/sq_r4/a -o p --output-format csv -- python3 $ROOT/tools/probes/fused_block_bench.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $ROOT/gpurun_out/sq_r4/b -o p --output-format csv -- python3 $ROOT/tools/probes/fused_block_bench.py > /dev/null 2>&1
ls $ROOT/gpurun_out/sq_r4/a $ROOT/gpurun_out/sq_r4/b
