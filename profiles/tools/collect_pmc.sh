#!/bin/bash
# Hardware-counter passes for the kernels of every single-GPU config (run on the GPU box, from the repo root):
#     bash profiles/tools/collect_pmc.sh [out_dir] ["rdf msd ..."]
# One rocprofv3 --pmc run per (workload, counter group); counters only -- no tracing domains in the same run.
# The program follows `--` directly (python3 script): no env/bash wrapper between rocprofv3 and the process
# that initialises the GPU.  Raw CSVs land under out_dir; profiles/tools/pmc_to_json.py condenses them.
set -u
OUT=${1:-gpurun_out/r03/pmc}
WLS=${2:-rdf msd bad cn cfg4}
mkdir -p "$OUT"
export TMPDIR=/tmp RUN_ONCE_REPS=2
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
G2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_LDS_ATOMIC"
G3="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
G4="FETCH_SIZE"
G5="WRITE_SIZE"
for wl in $WLS; do
  n=0
  for grp in "$G1" "$G2" "$G3" "$G4" "$G5"; do
    n=$((n+1))
    d="$OUT/${wl}_g${n}"
    rm -rf "$d"
    echo "== $wl group $n: $grp"
    # shellcheck disable=SC2086
    rocprofv3 --pmc $grp -d "$d" --output-format csv -- python3 profiles/tools/run_once.py $wl > "$OUT/${wl}_g${n}.log" 2>&1 || { echo "FAILED ($wl g$n)"; tail -5 "$OUT/${wl}_g${n}.log"; exit 1; }
    tail -1 "$OUT/${wl}_g${n}.log"
  done
done
echo "all passes done"
