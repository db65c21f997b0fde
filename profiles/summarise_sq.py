"""Per-kernel SQ counter table from ONE rocprofv3 PMC pass:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS \
        SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d <dir> -- python3 bench.py ...
    python profiles/summarise_sq.py <dir> "<header>" > profiles/rNN_sq_counters.txt

mfma/32: SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES (the MFMA counter sums over 32 units: 32.0 = pipe always busy; the
column is already divided by 32).  valu/mfma, lds/mfma: instruction ratios.  wait_lds%: SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES.
Averages per launch; the PMC pass serialises kernels: use it for ratios, never for times.
"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name if len(name) <= 84 else name[:84]


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    seen = set()
    for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                key = (k, r.get("Dispatch_Id"))
                if key not in seen:
                    seen.add(key)
                    n[k] += 1
    if len(sys.argv) > 2:
        print("# " + sys.argv[2])
    print("%-84s %5s %8s %9s %9s %10s %9s" % ("kernel", "n", "mfma/32", "valu/mfma", "lds/mfma", "bankconf", "wait_lds%"))
    rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0.0))
    for k, c in rows[:60]:
        mf, bz = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("SQ_BUSY_CYCLES", 0.0)
        im = c.get("SQ_INSTS_MFMA", 0.0)
        print("%-84s %5d %8.3f %9.2f %9.2f %10.3g %9.2f" % (
            k, n[k], mf / bz / 32.0 if bz else 0.0, c.get("SQ_INSTS_VALU", 0.0) / im if im else float("nan"),
            c.get("SQ_INSTS_LDS", 0.0) / im if im else float("nan"), c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(n[k], 1),
            100.0 * c.get("SQ_WAIT_INST_LDS", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else 0.0))


if __name__ == "__main__":
    main()
