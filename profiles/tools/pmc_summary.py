"""Average the counters of a rocprofv3 --pmc run per kernel name.

    python profiles/tools/pmc_summary.py <dir with *_counter_collection.csv> [substring of kernel names to keep]
"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
keep = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if keep and keep not in name:
                continue
            short = name.split("(")[0].replace("void amof::", "")[:70]
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for kern in sorted(acc):
    print(kern)
    for c in sorted(acc[kern]):
        v = acc[kern][c]
        print("    %-28s mean %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
