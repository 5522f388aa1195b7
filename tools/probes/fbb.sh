for v in "CAPNET_FB_WIDE=1" "CAPNET_FB_WIDE=0"; do echo "== $v"; env $v timeout -k 10 120 python tools/probes/fused_block_bench.py 2>&1 | grep stage; done
