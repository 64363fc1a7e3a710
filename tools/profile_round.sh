#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh TAG      e.g. TAG = r02
# Produces under gpurun_out/: the bench line, rocprofv3 kernel stats of the headline command (two frames in
# flight, the default) and of the same frames one at a time, of the street scenes and of the train step, and
# the PMC passes.  Copy the *.md / *.json you want judged into profiles/.
set -u
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python3 bench.py --steps 50 --warmup 5 --cpu-python-frames 3 > $O/${TAG}_bench_1gpu_S1M.jsonl 2> $O/${TAG}_bench.err; echo "bench rc=$?"
prof() {  # name frames cmd...
  local name=$1 frames=$2; shift 2
  rm -rf $O/prof_tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tmp -- "$@" > $O/${TAG}_${name}.log 2>&1; echo "$name rc=$?"
  python3 tools/summarize_rocprof.py $O/prof_tmp $O/${TAG}_kernel_stats_${name}.md $frames > /dev/null
  rm -rf $O/prof_tmp
}
prof 1gpu_S1M 55 python3 bench.py --headline-only --steps 50 --warmup 5
prof 1gpu_S1M_one_in_flight 55 python3 bench.py --headline-only --steps 50 --warmup 5 --frames-in-flight 1
prof street1m 20 python3 tools/prof_scene.py street1m 20
prof street3m 12 python3 tools/prof_scene.py street3m 12
prof sky 20 python3 tools/prof_scene.py sky 20
prof train_1gpu_S1M 13 python3 tools/exp_train.py 1000000 1600 1066 10
bash tools/pmc_run.sh $O/pmc_${TAG} > $O/${TAG}_pmc.log 2>&1; echo "pmc rc=$?"
cp $O/pmc_${TAG}/fwd/summary.md $O/${TAG}_pmc_counters_S1M.md
cp $O/pmc_${TAG}/train/summary.md $O/${TAG}_pmc_counters_train_S1M.md
cp $O/pmc_${TAG}/pmc_traffic.json $O/${TAG}_pmc_traffic.json
rm -rf $O/pmc_${TAG}/fwd/pass* $O/pmc_${TAG}/train/pass*
