# SQ counters of fb_fused_kernel alone (tools/probes/fused_block_bench.py), two passes of 8 counters
cd /tmp && export TMPDIR=/tmp
for rs in ${FB_RS_LIST:-2 1}; do
export CAPNET_FB_RS=$rs
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d /root/repo/gpurun_out/fb_pmc_a$rs -o p --output-format csv -- python3 /root/repo/tools/probes/fused_block_bench.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d /root/repo/gpurun_out/fb_pmc_b$rs -o p --output-format csv -- python3 /root/repo/tools/probes/fused_block_bench.py > /dev/null 2>&1
done
ls /root/repo/gpurun_out/fb_pmc_a2 /root/repo/gpurun_out/fb_pmc_b2
