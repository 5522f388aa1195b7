# kernel trace (timestamps) of a launch-bound secondary workload: where the decoder's chain spends its time
#   TRACE_ARGS="--layers 3 --factored 1024 --batch 96" bash tools/probes/att_trace.sh     (default: the attention decoder at 12 images)
TRACE_ARGS=${TRACE_ARGS:-"--decoder att --batch 12"}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /root/repo/gpurun_out/att_trace -o t --output-format csv -- python3 /root/repo/bench.py $TRACE_ARGS --steps 12 --warmup 4 --no-cpu-baseline --no-lstm-roofline --no-conv-events --no-graph-trunk > /root/repo/gpurun_out/att_trace_bench.json 2> /root/repo/gpurun_out/att_trace_bench.err
ls -la /root/repo/gpurun_out/att_trace
