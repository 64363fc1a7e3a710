"""A/B of a KERNEL EXPERIMENT: the shipped library against the diagnostic library built with extra -D flags
(SC_EXP_DEFS="-DSC_EXP_..." python -m street_crafter_amd.build --diag), per-operator HIP-event times on the same
box in one gpurun call.  The two libraries cannot live in one process (same symbols), so each arm is a child process;
arms alternate (A B A B ...) and the medians over all rounds are printed.

    python tools/ab_lib.py [s1m|s100k|street1m|sky|train] [rounds] [frames] [arm,arm,...]
    (arms: "diag" or the SC_DIAG_TAG names of experiment builds: lib/libstreet_crafter_hip_diag_<tag>.so)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(which, lib, frames):
    import torch
    from street_crafter_amd import _lib
    if lib != "shipped":
        _lib.use_diagnostic_build("" if lib == "diag" else lib)
    _lib.set_fast_binding(False)       # both arms through the ctypes table (the binding layer is linked to the shipped
                                       # library only; an operator's event bracket contains its host-side work)
    from harness.caller import render_gaussians
    from street_crafter_amd.scenes import make_camera, make_scene, make_street_scene
    import bench
    dev = "cuda"
    train = which == "train"
    W, H = (1600, 1066) if train else (1920, 1280)
    sc = {"s1m": lambda: make_scene(1_000_000), "train": lambda: make_scene(1_000_000), "s100k": lambda: make_scene(100_000),
          "street1m": lambda: make_street_scene(1_000_000)[0], "sky": lambda: make_street_scene(1_000_000)[1]}[which]().to(dev)
    ev = {}
    if train:
        from street_crafter_amd import rendering
        cam = make_camera(W, H, 2050.0 * W / 1920.0, 2050.0 * W / 1920.0).to(dev)
        ps = (sc.means, sc.quats, sc.scales, sc.opacities, sc.sh)
        for t in ps:
            t.requires_grad_(True)
        target = torch.rand(3, H, W, device=dev)
        for i in range(frames + 3):
            for t in ps:
                t.grad = None
            rec = i >= 3
            prev = rendering.set_backward_probe(ev if rec else None)      # (before the forward: probed steps run the Python Functions)
            o = render_gaussians(sc, cam, mode="train", stage_events=ev if rec else None)
            ((o["rgb"] - target).abs().mean() + 0.01 * o["acc"].mean()).backward()
            rendering.set_backward_probe(prev)
    else:
        cams = [bench.frame_camera(s, W, H).to(dev) for s in range(frames + 4)]
        with torch.no_grad():
            for s in range(4):
                render_gaussians(sc, cams[s])
            for s in range(4, frames + 4):
                render_gaussians(sc, cams[s], stage_events=ev)
    torch.cuda.synchronize()
    med = {k: sorted(a.elapsed_time(b) for a, b in v)[len(v) // 2] * 1e3 for k, v in ev.items()}
    print("AB_RESULT " + json.dumps(med), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3], int(sys.argv[4]))
        sys.exit(0)
    which = sys.argv[1] if len(sys.argv) > 1 else "s1m"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    arms = ["shipped"] + (sys.argv[4].split(",") if len(sys.argv) > 4 else ["diag"])     # diag, or SC_DIAG_TAG names
    res = {a: [] for a in arms}
    for r in range(rounds):
        for lib in arms:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", which, lib, str(frames)],
                               capture_output=True, text=True, timeout=600)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("AB_RESULT ")]
            if p.returncode != 0 or not line:
                print(f"{lib} round {r} failed:\n" + "\n".join((p.stdout + p.stderr).splitlines()[-15:]))
                sys.exit(1)
            res[lib].append(json.loads(line[0][10:]))
    keys = list(res["shipped"][0])
    print(f"# {which}, {rounds} alternating rounds x {frames} frames, median per-operator HIP-event time (us); "
          f"diag = SC_EXP_DEFS build")
    for k in keys:
        a = sorted(x[k] for x in res["shipped"])
        line = f"{k:28s} shipped {a[len(a) // 2]:8.1f} (min {a[0]:7.1f})"
        for arm in arms[1:]:
            b = sorted(x[k] for x in res[arm])
            line += f"   {arm} {b[len(b) // 2]:8.1f} (min {b[0]:7.1f}) {b[len(b) // 2] - a[len(a) // 2]:+6.1f}"
        print(line)
