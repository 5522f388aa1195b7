#!/bin/bash
# SQ counters of the stride-1 3x3 kernel ALONE on the trunk's shapes at batch 64 (tools/probes/patch_bench.py), in its three
# wave arrangements: narrow (4 x 2 waves of 32 x 64, <= 128 VGPRs: the one the pipelined step uses), ksplit (a lone workgroup
# per CU), wide (2 x 4 waves of 64 x 64 on a 128 x 256 tile: VERDICT r3 #4). Two passes of eight counters each.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
B="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
export P3_NARROW=1 P3_SHARED=1
rocprofv3 --kernel-trace --pmc $A -d $ROOT/gpurun_out/sq_p3/narrow/a -o p --output-format csv -- python3 $ROOT/tools/probes/patch_bench.py > $ROOT/gpurun_out/sq_p3_narrow.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc $B -d $ROOT/gpurun_out/sq_p3/narrow/b -o p --output-format csv -- python3 $ROOT/tools/probes/patch_bench.py > /dev/null 2>&1 || exit 1
export P3_NARROW=1 P3_SHARED=0
rocprofv3 --kernel-trace --pmc $A -d $ROOT/gpurun_out/sq_p3/ksplit/a -o p --output-format csv -- python3 $ROOT/tools/probes/patch_bench.py > $ROOT/gpurun_out/sq_p3_ksplit.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc $B -d $ROOT/gpurun_out/sq_p3/ksplit/b -o p --output-format csv -- python3 $ROOT/tools/probes/patch_bench.py > /dev/null 2>&1 || exit 1
unset P3_NARROW; export P3_SHARED=0
rocprofv3 --kernel-trace --pmc $A -d $ROOT/gpurun_out/sq_p3/wide/a -o p --output-format csv -- python3 $ROOT/tools/probes/patch_bench.py > $ROOT/gpurun_out/sq_p3_wide.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc $B -d $ROOT/gpurun_out/sq_p3/wide/b -o p --output-format csv -- python3 $ROOT/tools/probes/patch_bench.py > /dev/null 2>&1 || exit 1
cd $ROOT
for v in narrow ksplit wide; do python3 tools/sq_summarize.py gpurun_out/sq_p3/$v conv3x3_patch > gpurun_out/sq_p3_$v.csv; done
# the same three alone, timed without the profiler
for e in "P3_NARROW=1 P3_SHARED=1" "P3_NARROW=1 P3_SHARED=0" "P3_SHARED=0"; do echo "== $e"; env -u P3_NARROW $e python3 tools/probes/patch_bench.py; done > gpurun_out/sq_p3_times.log 2>&1
