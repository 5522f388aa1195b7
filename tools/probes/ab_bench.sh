# A/B of the pipelined bench on ONE box: variants given as NAME:ENV pairs in $AB_VARIANTS, interleaved, two rounds
for round in 1 2; do
  for v in ${AB_VARIANTS:-NOFUSED:CAPNET_NO_FUSED_BLOCK=1 RS1:CAPNET_FB_RS=1 RS2:CAPNET_FB_RS=2}; do
    name=${v%%:*}; envs=$(echo ${v#*:} | tr ',' ' ')
    env $envs timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-lstm-roofline > gpurun_out/ab_$name.$round.json 2> gpurun_out/ab_$name.$round.err
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/ab_$name.$round.json"))
    print("$name round $round: %.0f images/s  %.3f ms/step  loss %.5f" % (d["value"], d["ms_per_step"], d["loss_last"]))
except Exception as e:
    print("$name round $round: FAILED", e)
PY
  done
done
