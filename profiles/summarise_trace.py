"""Per-kernel totals from a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv): calls, total/avg/min/max us, share.

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace -- python3 bench.py ...
    python profiles/summarise_trace.py gpurun_out/trace "<header line>" > profiles/rNN_bench_kernel_stats.txt
"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name if len(name) <= 92 else name[:92]


def main():
    rows = collections.defaultdict(list)
    for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                rows[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    total = sum(sum(v) for v in rows.values())
    if len(sys.argv) > 2:
        print("# " + sys.argv[2])
    print("# total kernel time %.1f ms over %d kernels" % (total / 1e3, sum(len(v) for v in rows.values())))
    print("%-92s %6s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:70]:
        print("%-92s %6d %12.1f %10.1f %10.1f %10.1f %6.2f" % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v),
                                                                100.0 * sum(v) / total))


if __name__ == "__main__":
    main()
