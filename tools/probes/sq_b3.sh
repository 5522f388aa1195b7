#!/bin/bash
# SQ counters of gemm_b3_kernel alone on the decoders' product shapes (tools/probes/b3_bench.py): two passes of eight counters
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $ROOT/gpurun_out/sq_b3/a -o p --output-format csv -- python3 $ROOT/tools/probes/b3_bench.py > /dev/null 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $ROOT/gpurun_out/sq_b3/b -o p --output-format csv -- python3 $ROOT/tools/probes/b3_bench.py > /dev/null 2>&1 || exit 1
cd $ROOT && python3 tools/sq_summarize.py gpurun_out/sq_b3 gemm_b3 > gpurun_out/sq_b3.csv
