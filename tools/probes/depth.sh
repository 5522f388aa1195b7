for round in 1 2; do for d in 3 4 5; do
  timeout -k 10 200 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-lstm-roofline --pipeline-depth $d > gpurun_out/depth_$d.$round.json 2> gpurun_out/depth_$d.$round.err
  python -c "
import json
d=json.load(open('gpurun_out/depth_$d.$round.json')); print('depth $d round $round: %.0f images/s %.3f ms/step' % (d['value'], d['ms_per_step']))"
done; done
