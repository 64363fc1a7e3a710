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


@pytest.mark.parametrize("seed,rnd", [(28, 5), (42, 7), (57, 3)])
def test_ill_conditioned_pixels_are_judged_by_float64(seed, rnd):
    """VERDICT r2 item 1.  The configurations on which round 2's fuzz campaign found "stable" pixels beyond the flat
    1e-4 bar (gpurun_out/fuzz_oracle_cond2.log: 50 giant splats seen from half a metre, 900x634, 1.26e-4; two more of
    that kind): every pixel that is over the flat bar against the fp32 oracle under the FIXED 2e-5 window is blended in
    float64 from the same fp32 inputs (oracle/blend_f64.py) and must be within 1e-4 of that truth, or at least as close
    to it as the fp32 oracle's literal formula is.  No widened window takes part in the verdict."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))
    import fuzz_oracle as F
    rng = np.random.default_rng(seed)
    for it in range(rnd + 1):
        sc, cam, cfg = F.draw_config(rng)
    ok, rep = F.check_config(sc, cam, cfg, tag=f"[seed {seed} round {rnd}] ")
    assert rep["ints_ok"] and rep["floats_ok"] and rep["inputs_same"]
    assert rep["n_over_flat_bar"] > 0, "this configuration is here BECAUSE it has pixels over the flat bar"
    assert rep["f64_judged"] == rep["n_over_flat_bar"] and rep["f64_failed"] == 0, rep["rows"][:3]
    assert ok


def test_backward_kernels_agree_with_the_float64_oracle():
    assert "agree with the float64 oracle" in _run("fuzz_grad.py", 8, 5)


def test_two_pass_uint8_frame_agrees_with_the_oracle():
    assert "agree with the oracle" in _run("fuzz_two_pass.py", 2, 3)


def test_knn_agrees_on_degenerate_clouds():
    assert "distCUDA2 agrees" in _run("fuzz_knn.py", 4, 8)
