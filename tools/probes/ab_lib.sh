# A/B of library builds on ONE box: build/ab/lib_<name>.so copied over the package's library between runs, interleaved, two rounds
#   AB_LIBS="noprefetch prefetch" bash tools/probes/ab_lib.sh [bench args]
LIB=image-caption-emotion-indonesia_amd/libcapnet_hip.so
cp $LIB build/ab/lib__orig.so
for round in 1 2; do
  for name in $AB_LIBS; do
    cp build/ab/lib_$name.so $LIB
    timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-lstm-roofline "$@" > gpurun_out/abl_$name.$round.json 2> gpurun_out/abl_$name.$round.err
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/abl_$name.$round.json"))
    print("$name round $round: %.0f images/s  %.3f ms/step  loss %.5f" % (d["value"], d["ms_per_step"], d["loss_last"]))
except Exception as e:
    print("$name round $round: FAILED", e)
PY
  done
done
cp build/ab/lib__orig.so $LIB
