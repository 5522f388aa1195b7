"""Per-kernel means of the SQ counters two rocprofv3 --pmc passes left (tools/probes/sq_round4.sh, sq_patch.sh):
python tools/sq_summarize.py <dir with a/ and b/> [name filter] > profiles/roundN_sq_<what>.csv

SQ_WAVE_CYCLES / WAIT_* / ACTIVE_* are quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles in which a SIMD's
matrix pipe was busy, summed over SIMDs; SQ_BUSY_CYCLES is per SE. Derived columns: the share of a wave's time spent waiting
(any reason) / stalled at issue / issuing; VALU instructions per MFMA; mfma_busy_per_wave_time = MFMA_BUSY / (4 WAVE_CYCLES /
waves per SIMD) is left to the reader (waves per SIMD depends on the launch: see DESIGN 4m)."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
meta = {}
for sub in ("a", "b"):
    for f in glob.glob("%s/%s/*counter_collection.csv" % (root, sub)):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("capnet::", "")
            if flt and flt not in name:
                continue
            key = (name, r["Grid_Size"], r["VGPR_Count"], r["LDS_Block_Size"])
            e = acc[key][r["Counter_Name"].replace("SQ_", "")]
            e[0] += float(r["Counter_Value"])
            e[1] += 1
cols = ["WAVE_CYCLES", "BUSY_CYCLES", "WAIT_ANY", "WAIT_INST_ANY", "ACTIVE_INST_ANY", "VALU_MFMA_BUSY_CYCLES", "WAIT_INST_LDS",
        "LDS_BANK_CONFLICT", "INSTS_VALU", "INSTS_MFMA", "INSTS_SALU", "INSTS_LDS", "ACTIVE_INST_LDS", "LDS_IDX_ACTIVE",
        "ACTIVE_INST_VALU", "ACTIVE_INST_SCA"]
print("kernel,grid_threads,vgprs,lds_bytes," + ",".join(cols) + ",wait_any_frac,wait_inst_frac,active_frac,valu_per_mfma")
for key in sorted(acc):
    m = {c: (acc[key][c][0] / acc[key][c][1] if acc[key][c][1] else float("nan")) for c in cols}
    wc = m["WAVE_CYCLES"]
    d = [m["WAIT_ANY"] / wc, m["WAIT_INST_ANY"] / wc, m["ACTIVE_INST_ANY"] / wc,
         m["INSTS_VALU"] / m["INSTS_MFMA"] if m["INSTS_MFMA"] else float("nan")]
    print(",".join([key[0].replace(",", ";"), key[1], key[2], key[3]] + ["%.4g" % m[c] for c in cols] + ["%.3f" % x for x in d]))
