#!/bin/bash
# Builds the kernels of the last commit (HEAD) as the diagnostic library lib/libstreet_crafter_hip_diag_<tag>.so, so that
# tools/ab_lib.py can time the working tree (shipped build) against it:  tools/build_head_arm.sh base && python tools/ab_lib.py s1m 4 10 base
# HEAD's sources are checked out into a temporary worktree (nothing is stashed: the working tree, its untracked files and any
# stash entries of yours stay as they are); only the built library is copied back.
set -e
tag=${1:-base}
root="$(cd "$(dirname "$0")/.." && pwd)"
tmp="$(mktemp -d /tmp/sc_head_arm.XXXXXX)"
trap 'git -C "$root" worktree remove --force "$tmp" >/dev/null 2>&1 || rm -rf "$tmp"' EXIT
git -C "$root" worktree add --detach -q "$tmp" HEAD
(cd "$tmp" && SC_DIAG_TAG=$tag python -m street_crafter_amd.build --diag 2>&1 | grep "^built.*diag")
cp "$tmp/street_crafter_amd/lib/libstreet_crafter_hip_diag_${tag}.so" "$root/street_crafter_amd/lib/"
echo "copied lib/libstreet_crafter_hip_diag_${tag}.so (HEAD $(git -C "$root" rev-parse --short HEAD))"
