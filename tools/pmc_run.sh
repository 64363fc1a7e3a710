#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_run.sh OUTDIR
# Separate rocprofv3 passes (the counter blocks have few slots; --pmc only with --kernel-trace), over the forward
# headline (OUTDIR/fwd) and over the training step at the reference's resolution (OUTDIR/train).
set -u
OUT=${1:-gpurun_out/pmc}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$OUT"; mkdir -p "$OUT/fwd" "$OUT/train"
PASSES=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"
        "FETCH_SIZE" "WRITE_SIZE"
        "SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE")
run_passes() {  # subdir cmd...
  local sub=$1; shift
  local i=0
  for C in "${PASSES[@]}"; do
    i=$((i+1))
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$sub/pass$i" -- "$@" > "$OUT/$sub/pass$i.log" 2>&1 || echo "$sub pass $i failed"
  done
  python3 tools/summarize_pmc.py "$OUT/$sub" "$OUT/$sub/summary.md" > /dev/null
}
run_passes fwd python3 bench.py --steps 8 --warmup 2 --headline-only --frames-in-flight 1
run_passes train python3 tools/exp_train.py 1000000 1600 1066 10
python3 tools/make_traffic_json.py "$OUT/fwd" "$OUT/pmc_traffic.json" n1000000 10
python3 tools/make_traffic_json.py "$OUT/train" "$OUT/pmc_traffic.json" train_n1000000_1600x1066 13 train
