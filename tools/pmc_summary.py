"""summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean counter value per dispatch"""
import csv
import glob
import sys
from collections import defaultdict

for path in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", path)
    for k, d in acc.items():
        print("  ", k)
        for c, v in sorted(d.items()):
            print("      %-28s n=%-3d mean=%.6g" % (c, len(v), sum(v) / len(v)))
