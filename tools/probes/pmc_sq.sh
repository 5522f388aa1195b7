#!/bin/bash
# SQ counters of one conv_bench run, one rocprofv3 pass per counter group (kernel-trace only).
#   bash tools/probes/pmc_sq.sh <tag> <conv_bench args...>   ->  gpurun_out/sq_<tag>.txt
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -o p -- \
    python3 $ROOT/tools/probes/conv_bench.py "$@" > $OUT/g$i.log 2>&1 || { tail -5 $OUT/g$i.log; exit 1; }
done
python3 - $OUT <<'P' > $ROOT/gpurun_out/sq_$TAG.txt
import csv, glob, sys, os
per = {}
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0][:60]
        if "conv" not in name: continue
        a = per.setdefault((name, r["Counter_Name"]), [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
for (n, c), (k, v) in sorted(per.items()):
    print("%-62s %-28s launches %3d  per launch %14.0f" % (n, c, k, v / k))
P
cat $ROOT/gpurun_out/sq_$TAG.txt
