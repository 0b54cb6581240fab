"""Which kernels ran BESIDE the RDF tile launch?  Reads a rocprofv3 --kernel-trace CSV (*_kernel_trace.csv) and prints,
per kernel name, how many of its dispatches lie inside the time span of a `rdf_tile_kernel_fast` dispatch, how long they
took there, and how long the same kernel takes when nothing else runs (dispatches outside any tile span).

    python profiles/tools/overlap.py gpurun_out/r05/trace/**/*_kernel_trace.csv
"""
import csv
import sys
from collections import defaultdict

rows = []
for path in sys.argv[1:]:
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "?"))))
tiles = sorted((s, e) for n, s, e, _ in rows if n.startswith("void amof::rdf_tile_kernel_fast") or "rdf_tile_kernel_fast" in n)
print("%d dispatches, %d rdf_tile_kernel_fast launches (mean %.3f ms)" %
      (len(rows), len(tiles), sum(e - s for s, e in tiles) / max(1, len(tiles)) / 1e6))
inside, outside = defaultdict(list), defaultdict(list)
for n, s, e, q in rows:
    if "rdf_tile_kernel_fast" in n:
        continue
    short = n.split("(")[0].replace("void ", "").replace("amof::", "")[:60]
    hit = any(ts <= s and e <= te for ts, te in tiles)
    part = any(ts < e and s < te for ts, te in tiles)
    (inside if hit else outside)[short].append(((e - s) / 1e3, q, part and not hit))
print("%-62s %8s %12s | %8s %12s" % ("kernel", "inside", "mean us", "outside", "mean us"))
for k in sorted(set(inside) | set(outside)):
    a, b = inside.get(k, []), outside.get(k, [])
    print("%-62s %8d %12.1f | %8d %12.1f   queues %s" %
          (k, len(a), sum(x[0] for x in a) / max(1, len(a)), len(b), sum(x[0] for x in b) / max(1, len(b)),
           sorted(set(x[1] for x in a + b))))
