"""A short run of each randomised cross-check tool (tools/fuzz_paths.py; tests/fuzz/*.py for the ones that use the
oracle) as part of the GPU suite: the seeds that found
something in round 2 plus one more each.  The tools compare the fast HIP paths with the reference-shaped ones, with
the C / numpy / float64-autograd oracles and with scipy; see their docstrings."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, *args, timeout=420):
    path = os.path.join(ROOT, "tools", tool) if tool == "fuzz_paths.py" else os.path.join(ROOT, "tests", "fuzz", tool)
    p = subprocess.run([sys.executable, path, *map(str, args)], cwd=ROOT, capture_output=True,
                       text=True, timeout=timeout)
    tail = "\n".join((p.stdout + p.stderr).splitlines()[-25:])
    assert p.returncode == 0, f"{tool} {args} failed:\n{tail}"
    return p.stdout


@pytest.mark.parametrize("seed,rounds", [(9, 2), (3, 4)])
def test_fast_paths_agree_with_the_reference_shaped_ones(seed, rounds):
    # seed 9, round 1: big sky splats clipped to one column of super-tiles (the record-scatter bug of round 2)
    assert "all paths agree" in _run("fuzz_paths.py", seed, rounds)


def test_caller_sequence_agrees_with_the_c_oracle():
    assert "agrees with the oracle" in _run("fuzz_oracle.py", 3, 5)


def test_backward_kernels_agree_with_the_float64_oracle():
    assert "agree with the float64 oracle" in _run("fuzz_grad.py", 8, 5)


def test_two_pass_uint8_frame_agrees_with_the_oracle():
    assert "agree with the oracle" in _run("fuzz_two_pass.py", 2, 3)


def test_knn_agrees_on_degenerate_clouds():
    assert "distCUDA2 agrees" in _run("fuzz_knn.py", 4, 8)
