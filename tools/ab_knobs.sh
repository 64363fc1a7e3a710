#!/bin/bash
# usage (on the GPU box): tools/ab_knobs.sh KERNEL_SUBSTRING "knobs A" "knobs B" ...   (each arg: "k=v k=v" or "-")
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K=$1; shift
i=0
for cfg in "$@"; do
  i=$((i+1))
  d=gpurun_out/ab_$i
  if [ "$cfg" = "-" ]; then args=""; else args="$cfg"; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/exp_knob.py $args > gpurun_out/ab_$i.log 2>&1 || exit 1
  python tools/summarize_rocprof.py $d gpurun_out/ab_$i.md 8 > /dev/null 2>&1
  echo "== [$cfg]"; grep -E "$K" gpurun_out/ab_$i.md | cut -c1-150
  rm -rf $d
done
