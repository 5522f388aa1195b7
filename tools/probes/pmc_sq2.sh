#!/bin/bash
# SQ counters of any script, one rocprofv3 pass per counter group (kernel-trace only).
#   bash tools/probes/pmc_sq2.sh <tag> <kernel-name-substring> <script.py> [args...]  ->  gpurun_out/sq_<tag>.txt
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; FILT=$2; shift 2
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -o p -- \
    python3 $ROOT/"$@" > $OUT/g$i.log 2>&1 || { tail -5 $OUT/g$i.log; exit 1; }
done
python3 - $OUT "$FILT" <<'P' > $ROOT/gpurun_out/sq_$TAG.txt
import csv, glob, sys, os
per = {}
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:70]
        if not any(f in name for f in sys.argv[2].split(",")): continue
        a = per.setdefault((name, r["Counter_Name"]), [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
for (n, c), (k, v) in sorted(per.items()):
    print("%-72s %-28s launches %4d  per launch %16.0f" % (n, c, k, v / k))
P
cat $ROOT/gpurun_out/sq_$TAG.txt
