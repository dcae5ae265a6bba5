"""Sum rocprofv3 --pmc counters per kernel from *_counter_collection.csv files under a directory.
usage: python tools/pmc_summary.py DIR [kernel-name-substring]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if flt not in k: continue
        k = k.split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
for k in acc:
    print(k, "dispatches", len(nd[k]))
    for c, v in sorted(acc[k].items()): print("   %-28s %.6g" % (c, v))
