"""Per-launch HBM traffic of the conv kernels from the two rocprofv3 --pmc passes of
tools/pmc_traffic.sh. Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE is reported in KB and counts 64 B per 128-B request on gfx950 for wide coalesced
reads -> doubled; WRITE_SIZE (KB) is exact for 16-B-per-lane stores.

usage: python tools/pmc_summarize.py gpurun_out/pmc > profiles/roundN_pmc_traffic.json
"""
import csv
import glob
import json
import os
import sys


def collect(d, counter):
    path = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not path:
        raise SystemExit("no counter_collection.csv under " + d)
    per = {}
    for r in csv.DictReader(open(path[0])):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        a = per.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return per


def main():
    root = sys.argv[1]
    fetch = collect(os.path.join(root, "fetch"), "FETCH_SIZE")
    write = collect(os.path.join(root, "write"), "WRITE_SIZE")
    if len(sys.argv) > 2 and sys.argv[2] == "--per-kernel":
        # raw per-kernel sums (profiles/roundN_pmc_per_kernel.csv)
        print("# rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / WRITE_SIZE (pass 2) -- python3 bench.py "
              "--steps 2 --warmup 1 --no-cpu-baseline --no-conv-events --no-lstm-roofline")
        print("# raw counter sums in KB over the run (pipelined: 3 + 3 trunk passes, 3 decoder steps); FETCH_SIZE "
              "must be doubled on gfx950 (MI355X_MICROARCH.md, HBM section)")
        print("kernel,dispatches,fetch_size_kb_raw,write_size_kb")
        for k, v in sorted(fetch.items(), key=lambda kv: -(2 * kv[1][1] + write.get(kv[0], [0, 0])[1]))[:14]:
            print("%s,%d,%.0f,%.0f" % (k.replace(",", ";"), v[0], v[1], write.get(k, [0, 0])[1]))
        return
    # round 4: a block boundary on fused_block.hip is two "launches" of the bench's roofline as well -- the statistics group
    # (fb_gram_kernel counted, its reduce / quadratic-form launches only add their bytes) in conv3's place and the fused
    # launch in the next conv1's
    convk = ("conv_f32", "conv_f16x3_kernel", "conv3x3_patch_kernel", "conv_stem_f16x3_kernel", "conv1x1_tail_kernel", "conv1x1_areg_kernel",
             "fb_fused_kernel", "fb_gram_kernel")
    conv = lambda n: any(k in n for k in convk) or "conv_tail_fixup" in n or "fb_gram_reduce_kernel" in n or "fb_quad_kernel" in n
    launches = sum(v[0] for k, v in fetch.items() if any(c in k for c in convk))   # fix-ups belong to a conv
    fetch_kb = sum(v[1] for k, v in fetch.items() if conv(k))
    write_kb = sum(v[1] for k, v in write.items() if conv(k))
    h3 = lambda n: "conv_f16x3_kernel" in n
    h3_n = sum(v[0] for k, v in fetch.items() if h3(k))
    h3_bytes = (2.0 * sum(v[1] for k, v in fetch.items() if h3(k)) + sum(v[1] for k, v in write.items() if h3(k))) * 1024.0
    p3 = lambda n: "conv3x3_patch_kernel" in n
    p3_n = sum(v[0] for k, v in fetch.items() if p3(k))
    p3_bytes = (2.0 * sum(v[1] for k, v in fetch.items() if p3(k)) + sum(v[1] for k, v in write.items() if p3(k))) * 1024.0
    tl = lambda n: "conv1x1_tail_kernel" in n
    tl_n = sum(v[0] for k, v in fetch.items() if tl(k))
    tl_bytes = (2.0 * sum(v[1] for k, v in fetch.items() if tl(k)) + sum(v[1] for k, v in write.items() if tl(k))) * 1024.0
    st = lambda n: "conv_stem_f16x3_kernel" in n
    st_n = sum(v[0] for k, v in fetch.items() if st(k))
    st_bytes = (2.0 * sum(v[1] for k, v in fetch.items() if st(k)) + sum(v[1] for k, v in write.items() if st(k))) * 1024.0
    fu = lambda n: "fb_fused_kernel" in n
    fu_n = sum(v[0] for k, v in fetch.items() if fu(k))
    fu_bytes = (2.0 * sum(v[1] for k, v in fetch.items() if fu(k)) + sum(v[1] for k, v in write.items() if fu(k))) * 1024.0
    gs = lambda n: "fb_gram_kernel" in n or "fb_gram_reduce_kernel" in n or "fb_quad_kernel" in n
    gs_n = sum(v[0] for k, v in fetch.items() if "fb_gram_kernel" in k)
    gs_bytes = (2.0 * sum(v[1] for k, v in fetch.items() if gs(k)) + sum(v[1] for k, v in write.items() if gs(k))) * 1024.0
    lp = lambda n: "lstm_persist_kernel" in n
    lp_n = sum(v[0] for k, v in fetch.items() if lp(k))
    lp_bytes = (2.0 * sum(v[1] for k, v in fetch.items() if lp(k)) + sum(v[1] for k, v in write.items() if lp(k))) * 1024.0
    ls = lambda n: "lstm_step_fused_kernel" in n
    ls_n = sum(v[0] for k, v in fetch.items() if ls(k))
    ls_bytes = (2.0 * sum(v[1] for k, v in fetch.items() if ls(k)) +
                sum(v[1] for k, v in write.items() if ls(k))) * 1024.0
    out = {
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 2 "
                   "--warmup 1 --no-cpu-baseline --no-conv-events --no-lstm-roofline",
        "conv_launches": launches,
        "fetch_size_kb_raw": fetch_kb,
        "write_size_kb": write_kb,
        "conv_bytes_per_launch": round((2.0 * fetch_kb + write_kb) * 1024.0 / max(launches, 1)),
        "algorithmic_bytes_per_launch": round((232e6 + 2 * 90e6 * 64) / 155),
        # with the 49 fused block tails: + 3 passes over each block output (3 365 MB per pass of the trunk at B = 64:
        # read y3, read the identity, write the block output) - the conv1 read of that output the fusion removes
        "algorithmic_bytes_per_launch_with_tails": round((232e6 + 2 * 90e6 * 64 + 3 * 3288e6 - 3288e6) / 155),
        "f16x3_launches": h3_n,
        "f16x3_bytes_per_launch": round(h3_bytes / max(h3_n, 1)),
        "patch3x3_launches": p3_n,
        "patch3x3_bytes_per_launch": round(p3_bytes / max(p3_n, 1)),
        "tail_conv1_launches": tl_n,
        "tail_conv1_bytes_per_launch": round(tl_bytes / max(tl_n, 1)),
        "fused_boundary_launches": fu_n,
        "fused_boundary_bytes_per_launch": round(fu_bytes / max(fu_n, 1)),
        "boundary_statistics_groups": gs_n,
        "boundary_statistics_bytes_per_group": round(gs_bytes / max(gs_n, 1)),
        # the 44 inner boundaries of stages 1-3 no longer write and re-read y3 (2 x 2 434 MB per pass at B = 64) nor re-read
        # the block output for conv1: what remains per boundary is y2 twice, identity, out, y1
        "algorithmic_bytes_per_launch_with_fused_boundaries": round((232e6 + 2 * 90e6 * 64 + 2 * 3288e6 - 2 * 2434e6 + 44 * 0) / 155),
        "stem_launches": st_n,
        "stem_bytes_per_launch": round(st_bytes / max(st_n, 1)),
        "lstm_persist_launches": lp_n,
        "lstm_persist_bytes_per_launch": round(lp_bytes / max(lp_n, 1)),
        "lstm_step_launches": ls_n,
        "lstm_step_bytes_per_launch": round(ls_bytes / max(ls_n, 1)),
        "note": "FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B); per conv launch incl. "
                "its tail fix-up; algorithmic = (232 MB weights + 2 x 90 MB x 64 activations) / 155 convs; the 49 "
                "conv1 launches that absorb a block's tail also carry its traffic (identity read, block output written)",
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
