"""Per-dispatch listing of this library's kernels from a rocprofv3 --kernel-trace --output-format csv directory.

    python profiles/tools/dispatches.py DIR [substring]
"""
import csv
import glob
import sys

d = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "amof"
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if not files:
    sys.exit("no *kernel_trace.csv under %s" % d)
rows = []
for fn in files:
    with open(fn) as fh:
        rows += list(csv.DictReader(fh))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    name = r["Kernel_Name"]
    if want not in name:
        continue
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    short = name.split("(")[0].replace("void amof::", "")[:60]
    print("%-60s %10.1f us  grid %s x %s  wg %s  lds %s  vgpr %s" % (
        short, us, r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Workgroup_Size_X", "?"),
        r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?")))
