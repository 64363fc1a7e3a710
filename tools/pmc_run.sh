#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_run.sh OUTDIR
# Separate rocprofv3 passes (the counter blocks have few slots; --pmc only with --kernel-trace).
set -u
OUT=${1:-gpurun_out/pmc}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python3 bench.py --steps 8 --warmup 2 --headline-only --frames-in-flight 1"
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
         "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pass$i" -- $CMD > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
python3 tools/summarize_pmc.py "$OUT" "$OUT/summary.md"
python3 tools/make_traffic_json.py "$OUT" "$OUT/pmc_traffic.json" n1000000 10
