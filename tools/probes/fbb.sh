for d in 0 1 2 3 4 7; do echo "== NW8 dbg $d"; CAPNET_FB_NW=8 CAPNET_FB_DBG=$d timeout -k 10 120 python tools/fused_block_bench.py 2>&1 | grep "stage 14"; done
