timeout -k 10 120 python tools/probes/fused_block_bench.py 2>&1 | grep stage
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/fbb_prof -o fbb -- python3 /root/repo/tools/probes/fused_block_bench.py > /dev/null 2>&1
