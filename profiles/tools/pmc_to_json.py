"""Condense the raw rocprofv3 --pmc CSVs of profiles/tools/collect_pmc.sh into the JSON files bench.py reads.

    python profiles/tools/pmc_to_json.py gpurun_out/r03/pmc profiles/r03

Writes, next to a per-kernel table of every counter (``pmc_<workload>.json``, mean per launch):

* ``profiles/traffic.json`` -- HBM-side bytes per launch of the headline kernels.  gfx950 corrections as prescribed by
  MI355X_MICROARCH.md (HBM section): FETCH_SIZE is reported in KB and tallies 128-B requests at 64 B -> x 1024 x 2;
  WRITE_SIZE (KB) is taken as is -> x 1024.  Infinity-Cache hits are included in both.
* ``profiles/valu_model.json`` -- for the all-pairs RDF tile kernel: VALU instructions per launch by class, the issue
  cycles they need at the per-instruction costs measured on this box, and the share of the kernel's SIMD cycles that is
  (issue-slot utilisation).  Costs: ``profiles/r03/ubench2_valu_issue.txt`` (profiles/tools/ubench2.hip: >= 15 ms
  kernels, cycles from s_memtime per SIMD, 8 waves resident on every SIMD -- 2.19 cycles for the full-rate classes,
  4.1 for conversions / fract / compares / min / shift-add, 8.1 for sqrt; round 2 priced with a table that carried a
  fixed launch offset and came out above 1).  Instructions outside the counted classes (min, fract, compares,
  selects, moves, lane reads/writes) are priced twice: all at the full rate and all at the half rate -- a low and a
  high figure.  Kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs); the kernel's duration in
  the same counter run (dispatch timestamps of the CSV) gives the effective clock, with which bench.py turns its
  LIVE kernel time into cycles.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

import hashlib

N_SIMD = 1024
N_XCC = 8           # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs: cycles of the launch = value / 8
src, dst = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def source_stamp():
    """sha256 over the kernel sources these counters belong to: bench.py drops roofline_valu / traffic when the built sources
    no longer match (the numbers would be another kernel's)"""
    h = hashlib.sha256()
    for fn in ("rdf.hip", "msd.hip", "quant.hip", "amof_internal.h", "guard_math.h"):
        with open(os.path.join(root, "amof_amd", "csrc", fn), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def short(name):
    name = name.replace("void amof::", "").replace("amof::", "")
    return re.sub(r"\(.*$", "", name).strip()


def load(workload):
    acc = defaultdict(lambda: defaultdict(list))
    for gdir in sorted(glob.glob(os.path.join(src, workload + "_g*"))):
        if not os.path.isdir(gdir):
            continue
        # (a directory merged from several collections holds one CSV per run: the newest one counts)
        csvs = glob.glob(os.path.join(gdir, "**", "*counter_collection.csv"), recursive=True)
        if not csvs:
            continue
        path = max(csvs, key=os.path.getmtime)
        with open(path) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if "amof" not in name:
                    continue
                acc[short(name)][row["Counter_Name"]].append(float(row["Counter_Value"]))
                if row.get("Start_Timestamp") and row.get("End_Timestamp") and row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    acc[short(name)]["_duration_ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    # the first launch of a run includes cold caches; every run does RUN_ONCE_REPS identical calls: keep the mean
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"launches_seen": max(len(v) for v in cs.values())}
            for k, cs in acc.items()}


# issue cycles per wave instruction and SIMD, measured in-kernel (profiles/r03/ubench2_valu_issue.txt; first table
# only: the occupancy rows further down repeat two names)
UB = {}
with open(os.path.join(root, "profiles", "r03", "ubench2_valu_issue.txt")) as fh:
    for line in fh:
        m = re.match(r"(\w+)\s+1024 SIMDs, 8\.\.8 waves each.*cycles/wave-instr/SIMD\s+([\d.]+)", line)
        if m and m.group(1) not in UB:
            UB[m.group(1)] = float(m.group(2))

os.makedirs(dst, exist_ok=True)
tables = {}
for wl in ("rdf", "msd", "bad", "cn", "cfg4"):
    t = load(wl)
    if t:
        tables[wl] = t
        with open(os.path.join(dst, "pmc_%s.json" % wl), "w") as fh:
            json.dump(t, fh, indent=1, sort_keys=True)


def traffic(table, pick):
    tot = 0.0
    for k, c in table.items():
        if pick(k) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            tot += c["FETCH_SIZE"] * 1024.0 * 2.0 + c["WRITE_SIZE"] * 1024.0
    return tot or None


out = {"_comment": "HBM-side bytes per launch on the headline workload (9792 atoms x 5000 frames), derived by "
                   "profiles/tools/pmc_to_json.py from the separate FETCH_SIZE / WRITE_SIZE passes of "
                   "profiles/tools/collect_pmc.sh (raw per-kernel means: profiles/r02/pmc_*.json). gfx950 correction per "
                   "MI355X_MICROARCH.md: FETCH_SIZE (KB) tallies 128-B requests at 64 B -> doubled; WRITE_SIZE (KB) as "
                   "is. Infinity-Cache hits are included.", "cfg3": {}}
if "rdf" in tables:
    out["cfg3"]["rdf_tile_kernel_fast"] = traffic(tables["rdf"], lambda k: k.startswith("rdf_tile_kernel_fast"))
    out["cfg3"]["quantize_kernel"] = traffic(tables["rdf"], lambda k: k.startswith("quantize") or k.startswith("species_bytes"))
if "msd" in tables:
    out["cfg3"]["msd_pipeline"] = traffic(tables["msd"], lambda k: True)
    out["cfg3"]["msd_kernels"] = {k: traffic(tables["msd"], lambda q, k=k: q == k) for k in tables["msd"]}
if "bad" in tables:
    out["cfg3"]["bad_pipeline"] = traffic(tables["bad"], lambda k: True)
if "cn" in tables:
    out["cfg3"]["cn_pipeline"] = traffic(tables["cn"], lambda k: True)
if "cfg4" in tables:
    out["cfg4_64_frames"] = {k: traffic(tables["cfg4"], lambda q, k=k: q == k) for k in tables["cfg4"]}
out["_sources_sha256"] = source_stamp()
with open(os.path.join(root, "profiles", "traffic.json"), "w") as fh:
    json.dump(out, fh, indent=1)

model = {}
for wl, prefix in (("rdf", "rdf_tile_kernel_fast"), ("cfg4", "rdf_cell")):      # (rdf_cellwave_kernel | rdf_cell_kernel)
    if wl not in tables:
        continue
    for k, c in tables[wl].items():
        if not k.startswith(prefix) or "SQ_INSTS_VALU" not in c:
            continue
        cls = {"cvt": c.get("SQ_INSTS_VALU_CVT", 0.0), "trans_f32": c.get("SQ_INSTS_VALU_TRANS_F32", 0.0),
               "add_f32": c.get("SQ_INSTS_VALU_ADD_F32", 0.0), "mul_f32": c.get("SQ_INSTS_VALU_MUL_F32", 0.0),
               "fma_f32": c.get("SQ_INSTS_VALU_FMA_F32", 0.0), "int32": c.get("SQ_INSTS_VALU_INT32", 0.0),
               "f64": c.get("SQ_INSTS_VALU_ADD_F64", 0.0) + c.get("SQ_INSTS_VALU_MUL_F64", 0.0) +
                      c.get("SQ_INSTS_VALU_FMA_F64", 0.0)}
        other = c["SQ_INSTS_VALU"] - sum(cls.values())
        # int32: the chain's own mix is two full-rate v_sub_u32 per half-rate v_lshl_add_u32; priced low / high like "other"
        cost = {"cvt": UB["k_cvt_f32_i32"], "trans_f32": UB["k_sqrt32"], "add_f32": UB["k_add32"],
                "mul_f32": UB["k_mul32"], "fma_f32": UB["k_fma32"], "int32": UB["k_subu32"], "f64": UB["k_add64"]}
        known = sum(cls[x] * cost[x] for x in cls)
        issue_lo = known + max(other, 0.0) * UB["k_mul32"]       # the remainder (and every int32) at the full rate ...
        issue_hi = known + max(other, 0.0) * UB["k_rwlane"] + cls["int32"] * (UB["k_lshladd"] - UB["k_subu32"]) / 3.0
        # ... or all of it at the half rate (min / fract / compare / select / lane moves), a third of int32 as shift-adds
        gui = c.get("GRBM_GUI_ACTIVE")
        cyc = gui / N_XCC if gui else None
        dur = c.get("_duration_ns")
        model[prefix] = {
            "kernel": k, "valu_instructions_per_launch": c["SQ_INSTS_VALU"], "by_class": cls, "other_valu": other,
            "cycles_per_instruction_used": dict(cost, other_low=UB["k_mul32"], other_high=UB["k_rwlane"],
                                                int32_shift_add=UB["k_lshladd"]),
            "cost_table": "profiles/r03/ubench2_valu_issue.txt",
            "valu_issue_cycles_per_simd": [issue_lo / N_SIMD, issue_hi / N_SIMD],
            "kernel_cycles": cyc,
            "kernel_seconds_in_counter_run": dur * 1e-9 if dur else None,
            "effective_clock_ghz": (cyc / dur) if (cyc and dur) else None,
            "issue_slot_utilisation": [issue_lo / N_SIMD / cyc, issue_hi / N_SIMD / cyc] if cyc else None,
            "lane_utilisation": (c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0))
                                if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU") else None,
            "wave_cycle_split": {x: c.get(x) for x in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                                                       "SQ_ACTIVE_INST_ANY")},
            "lds_instructions": c.get("SQ_INSTS_LDS"), "lds_atomics": c.get("SQ_INSTS_LDS_ATOMIC"),
            "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT"), "lds_active_cycles": c.get("SQ_LDS_IDX_ACTIVE"),
        }
model["_sources_sha256"] = source_stamp()
with open(os.path.join(root, "profiles", "valu_model.json"), "w") as fh:
    json.dump(model, fh, indent=1)
print("wrote", sorted(tables), "->", dst, "; traffic.json, valu_model.json")
