# A/B of the attention decoder at 12 images per GPU: AB="NAME:ENV=V,ENV=V ..." (alternating, REPS rounds)
AB=${AB:-"three:CAPNET_ATT_CHAIN=3 one:CAPNET_ATT_CHAIN=0"}
REPS=${REPS:-3}
ARGS=${ARGS:-"--decoder att --batch 12 --steps 150 --warmup 10 --no-cpu-baseline --no-lstm-roofline"}
for r in $(seq $REPS); do
  for v in $AB; do
    name=${v%%:*}; envs=${v#*:}
    line=$(env $(echo $envs | tr ',' ' ') timeout -k 10 200 python bench.py $ARGS 2>/dev/null | tail -1)
    echo "$name $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
