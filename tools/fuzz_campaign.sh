#!/bin/bash
# usage (on the GPU box, from the repo root): tools/fuzz_campaign.sh [BASE] [N]
# N seeds (default 8) from BASE (default 100) of every randomised cross-check: tools/fuzz_paths.py (fast HIP paths vs the
# reference-shaped ones), tests/fuzz/fuzz_oracle.py (vs the C oracle; pixels over the flat bar judged in float64),
# fuzz_knn.py, fuzz_grad.py, fuzz_two_pass.py.  Logs under gpurun_out/camp_*; prints a tally.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BASE=${1:-100}; N=${2:-8}
fail=0
for i in $(seq 1 $N); do s=$((BASE + i))
  timeout -k 10 400 python tools/fuzz_paths.py $s 6 > gpurun_out/camp_paths_$s.log 2>&1 || { echo "fuzz_paths $s FAILED"; fail=1; }
  timeout -k 10 400 python tests/fuzz/fuzz_oracle.py $s 8 > gpurun_out/camp_oracle_$s.log 2>&1 || { echo "fuzz_oracle $s FAILED"; fail=1; }
  if [ $((i % 2)) -eq 0 ]; then
    timeout -k 10 300 python tests/fuzz/fuzz_knn.py $s 10 > gpurun_out/camp_knn_$s.log 2>&1 || { echo "fuzz_knn $s FAILED"; fail=1; }
    timeout -k 10 400 python tests/fuzz/fuzz_grad.py $s 6 > gpurun_out/camp_grad_$s.log 2>&1 || { echo "fuzz_grad $s FAILED"; fail=1; }
  fi
  if [ $((i % 4)) -eq 0 ]; then
    timeout -k 10 400 python tests/fuzz/fuzz_two_pass.py $s 4 > gpurun_out/camp_two_$s.log 2>&1 || { echo "fuzz_two_pass $s FAILED"; fail=1; }
  fi
  echo "seed $s done (fail=$fail)"
done
echo "campaign fail=$fail"
grep -h "stable pixels over the flat\|all paths agree\|agrees with the oracle\|distCUDA2 agrees\|agree with\|FAILED" gpurun_out/camp_*.log | sed 's/: [0-9]*; judged against float64: [0-9]*;/: N; judged against float64: N;/' | sort | uniq -c | sort -rn | head -20
