#!/bin/bash
# HBM traffic of the persistent LSTM kernel (24 steps at b = 64 per launch), one rocprofv3 pass per counter.
# -> gpurun_out/pmc_lstm.json  {"lstm_persist_bytes_per_step": ...}
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_lstm
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $ROOT/tools/probes/persist_24.py > $OUT/fetch.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $ROOT/tools/probes/persist_24.py > $OUT/write.log 2>&1 || exit 1
python3 - $OUT <<'P' > $ROOT/gpurun_out/pmc_lstm.json
import csv, glob, json, os, sys
def tot(d, c):
    n = 0; v = 0.0
    for r in csv.DictReader(open(glob.glob(os.path.join(sys.argv[1], d, "**", "*counter_collection.csv"), recursive=True)[0])):
        if r["Counter_Name"] == c and "lstm_persist_kernel" in r["Kernel_Name"]:
            n += 1; v += float(r["Counter_Value"])
    return n, v
nf, f = tot("fetch", "FETCH_SIZE"); nw, w = tot("write", "WRITE_SIZE")
per_launch = (2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0
print(json.dumps({"lstm_persist_pmc_launches": nf, "lstm_persist_bytes_per_launch_24_steps": round(per_launch),
                  "lstm_persist_bytes_per_step": round(per_launch / 24),
                  "lstm_persist_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 tools/probes/persist_24.py; FETCH_SIZE doubled"}, indent=1))
P
cat $ROOT/gpurun_out/pmc_lstm.json
