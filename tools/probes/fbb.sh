for rs in 1; do for d in 0; do echo "== rs $rs dbg $d"; CAPNET_FB_RS=$rs CAPNET_FB_DBG=$d timeout -k 10 120 python tools/fused_block_bench.py 2>&1 | grep stage; done; done
cd /tmp && export TMPDIR=/tmp && CAPNET_FB_RS=1 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/fbb_prof -o fbb -- python3 /root/repo/tools/fused_block_bench.py > /dev/null 2>&1
