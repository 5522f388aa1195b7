cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_r4 -o b --output-format csv -- python3 /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lstm-roofline --no-conv-events > /root/repo/gpurun_out/prof_r4_bench.json 2> /root/repo/gpurun_out/prof_r4_bench.err
ls /root/repo/gpurun_out/prof_r4
