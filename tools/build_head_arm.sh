#!/bin/bash
# Builds the kernels of the last commit (HEAD) as the diagnostic library lib/libstreet_crafter_hip_diag_<tag>.so, so that
# tools/ab_lib.py can time the working tree (shipped build) against it:  tools/build_head_arm.sh base && python tools/ab_lib.py s1m 4 10 base
set -e
tag=${1:-base}
cd "$(dirname "$0")/.."
git stash -q
trap 'git stash pop -q' EXIT
SC_DIAG_TAG=$tag python -m street_crafter_amd.build --diag 2>&1 | grep "^built.*diag"
git stash pop -q
trap - EXIT
python -m street_crafter_amd.build 2>&1 | grep "^built"      # the shipped build again, from the working tree
