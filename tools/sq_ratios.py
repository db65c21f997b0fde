import collections, csv, glob, os, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen=set()
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:8]:
    wc = c.get("SQ_WAVE_CYCLES", 1)
    print(k)
    print("   " + "  ".join("%s=%.3f" % (name.replace("SQ_", ""), val / wc) for name, val in sorted(c.items()) if name != "SQ_WAVE_CYCLES"))
