# kernel trace (timestamps) of the attention decoder at 12 images per GPU: where the decoder's chain idles
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /root/repo/gpurun_out/att_trace -o t --output-format csv -- python3 /root/repo/bench.py --decoder att --batch 12 --steps 12 --warmup 4 --no-cpu-baseline --no-lstm-roofline --no-conv-events ${ATT_EXTRA} > /root/repo/gpurun_out/att_trace_bench.json 2> /root/repo/gpurun_out/att_trace_bench.err
ls -la /root/repo/gpurun_out/att_trace
