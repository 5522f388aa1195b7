"""Condense a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats file into the
short table committed under profiles/ (kernel, calls, total_ms, avg_us, pct).

usage: python tools/summarize_rocprof.py <..._kernel_stats.csv> "<command line profiled>" > profiles/<name>.csv
"""
import csv
import re
import sys


def main():
    path, cmd = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
    rows = list(csv.DictReader(open(path)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# {cmd}")
    print(f"# MI355X (gfx950); all kernels of the run, total kernel time {total / 1e6:.3f} ms")
    print("# columns: kernel, calls, total_ms, avg_us, pct")
    for r in rows:
        name = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "")).replace(",", ";").strip()
        print(f"{name},{int(r['Calls'])},{float(r['TotalDurationNs']) / 1e6:.3f},"
              f"{float(r['AverageNs']) / 1e3:.2f},{float(r['Percentage']):.2f}")


if __name__ == "__main__":
    main()
