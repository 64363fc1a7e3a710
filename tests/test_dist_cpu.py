"""world_size-2 (and 3) gloo tests of the frame sharding + gather used for the multi-GPU path
(street_crafter_amd/dist.py).  The renderer is replaced by a deterministic CPU stand-in: what is
under test is the frame -> rank map, the async gather and the global frame order on rank 0."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from street_crafter_amd.dist import FrameGatherer, frames_for_rank, render_sharded, to_uint8_frame


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_frame(f, h=12, w=20):
    g = torch.Generator().manual_seed(1000 + f)
    return to_uint8_frame(torch.rand(3, h, w, generator=g))


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rendered = []

        def render(f):
            rendered.append(f)
            return _fake_frame(f)

        frames = render_sharded(n_frames, render)
        assert rendered == frames_for_rank(n_frames, rank, world)
        if rank == 0:
            ok = len(frames) == n_frames and all(torch.equal(fr, _fake_frame(i)) for i, fr in enumerate(frames))
            q.put(("ok" if ok else "mismatch", len(frames)))
        else:
            assert frames is None
        # a second, explicitly asynchronous use: submit everything, then drain once
        # batches of 3 rounds through a ring of 2 staging buffers: full batches, a partial tail, ring reuse,
        # frames written straight into the staging slot
        g = FrameGatherer((12, 20, 3), "cpu", dst=0, batch=3, ring=2)
        for r, f in enumerate(frames_for_rank(n_frames, rank, world)):
            if r % 2:
                g.slot(r).copy_(_fake_frame(f))
                g.submit(r)
            else:
                g.submit(r, _fake_frame(f))
        out = g.drain()
        if rank == 0:
            q.put(("ok" if all(torch.equal(fr, _fake_frame(i)) for i, fr in enumerate(out)) else "mismatch2", len(out)))
        else:
            assert out == []
        assert world == 1 or g.stats["gathers"] == -(-len(frames_for_rank(n_frames, rank, world)) // 3)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 8), (3, 6), (2, 26)])
def test_sharded_render_and_gather_gloo(world, n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    assert q.get(timeout=5) == ("ok", n_frames)
    assert q.get(timeout=5) == ("ok", n_frames)


def test_frame_to_rank_map():
    assert frames_for_rank(8, 0, 2) == [0, 2, 4, 6]
    assert frames_for_rank(8, 1, 2) == [1, 3, 5, 7]
    assert sorted(sum((frames_for_rank(24, r, 8) for r in range(8)), [])) == list(range(24))


def _worker_bad_count(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        try:
            render_sharded(3, _fake_frame)          # 3 frames over 2 ranks: not a whole number of rounds
            q.put("no error")
        except ValueError as e:
            q.put("ValueError" if "multiple of world size 2" in str(e) else f"wrong message: {e}")
    finally:
        dist.destroy_process_group()


def test_render_sharded_rejects_a_frame_count_that_is_not_a_multiple_of_the_world_size():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bad_count, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) == "ValueError" and q.get(timeout=5) == "ValueError"


def test_single_process_path():
    frames = render_sharded(4, _fake_frame)
    assert len(frames) == 4 and all(torch.equal(fr, _fake_frame(i)) for i, fr in enumerate(frames))
    u8 = to_uint8_frame(torch.tensor([[[0.0, 1.0]], [[0.5, 2.0]], [[-1.0, 0.25]]]), rounding="save_image")
    assert u8.shape == (1, 2, 3) and u8.dtype == torch.uint8
    assert u8[0, 0].tolist() == [0, 128, 0] and u8[0, 1].tolist() == [255, 255, 64]


def test_uint8_rounding_modes_are_the_reference_formulas():
    """'video' = `(rgb * 255).astype(np.uint8)` (street_gaussian_visualizer.py:97, base_visualizer.py:37: what
    mode=novel_view keeps), 'save_image' = torchvision.utils.save_image's mul(255).add_(0.5).clamp_(0, 255)
    .to(uint8) (visualizer :92); with and without the sky composite of renderer.py:152,159."""
    import numpy as np
    g = torch.Generator().manual_seed(5)
    rgb = torch.rand(3, 33, 47, generator=g) * 1.4 - 0.2          # some values outside [0, 1]
    sky = torch.rand(3, 33, 47, generator=g) * 1.2 - 0.1
    acc = torch.rand(1, 33, 47, generator=g)
    for comp in (False, True):
        if comp:
            ref = torch.clamp(rgb.clamp(0, 1) + sky.clamp(0, 1) * (1 - acc), 0.0, 1.0)     # renderer.py:152,159
            kw = dict(acc=acc[0], sky_rgb_chw=sky)
        else:
            ref, kw = rgb.clamp(0, 1), {}
        video = (ref.numpy().transpose(1, 2, 0) * 255).astype(np.uint8)
        png = ref.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
        assert np.array_equal(to_uint8_frame(rgb, rounding="video", **kw).numpy(), video)
        assert np.array_equal(to_uint8_frame(rgb, rounding="save_image", **kw).numpy(), png)
        assert np.array_equal(to_uint8_frame(rgb, **kw).numpy(), video)               # default: the video frames
        assert (video != png).mean() > 0.2                                              # the modes really differ
    with pytest.raises(ValueError):
        to_uint8_frame(rgb, rounding="nearest")
    with pytest.raises(ValueError):
        to_uint8_frame(rgb, acc=acc[0])


def test_bench_launches_its_own_ranks_selftest():
    """`python bench.py --gpus 2` without a launcher: the parent starts two ranks (gloo, CPU stand-in frames),
    frames are sharded, gathered in batches, checked on rank 0, and ONE JSON line comes out.  Same launcher /
    sharding / gather / timing code as the GPU run; only the frame source and the backend differ."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None), env.pop("MASTER_PORT", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--selftest-cpu", "--gpus", "2", "--steps", "21",
                        "--warmup", "3", "--gather-batch", "4"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 21 and d["warmup"] == 3 and d["scaling"] == "weak" and d["selftest"]
    assert len(d["per_rank_frames_per_s"]) == 2 and d["value"] > 0
    assert d["gather"]["frames_per_collective"] == 4 and d["gather"]["collectives"] == 6      # ceil(21 / 4)
    # asking for more GPUs than the machine has is a clear message and exit code 2, not an assert
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 2 and "exposes" in r.stderr
    # a rank that dies takes the job down with its exit code
    from street_crafter_amd.dist import launch_ranks
    code = "import os,sys,time; sys.exit(7) if os.environ['RANK']=='1' else time.sleep(60)"
    assert launch_ranks([sys.executable, "-c", code], 2) == 7


def _worker_world1_forced(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        g = FrameGatherer((12, 20, 3), "cpu", batch=3, ring=2, force_collective=True)
        for r in range(10):                       # 3 full batches + a tail of 1; the third batch reuses buffer 0
            if r % 2:
                g.slot(r).copy_(_fake_frame(r))
                g.submit(r)
            else:
                g.submit(r, _fake_frame(r))
        out = g.drain()
        same = len(out) == 10 and all(torch.equal(fr, _fake_frame(i)) for i, fr in enumerate(out))
        frames = render_sharded(7, _fake_frame, batch=2, ring=2, force_collective=True)
        same2 = len(frames) == 7 and all(torch.equal(fr, _fake_frame(i)) for i, fr in enumerate(frames))
        q.put((same, g.stats["gathers"], same2, render_sharded.last_stats["gathers"]))
        plain = FrameGatherer((12, 20, 3), "cpu", batch=3)          # the default at world 1: no collective at all
        plain.submit(0, _fake_frame(0))
        plain.drain()
        q.put(plain.stats["gathers"])
    finally:
        dist.destroy_process_group()


def test_forced_collective_sends_a_world_of_one_through_the_real_ring():
    """FrameGatherer(force_collective=True): the staging ring + async gather run even when there is one rank (what
    tests/rccl_world1.py does with backend nccl on the GPU box); without the flag a world of one issues none.
    Without a process group the flag is refused."""
    with pytest.raises(RuntimeError, match="init_process_group"):
        FrameGatherer((4, 4, 3), "cpu", force_collective=True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_world1_forced, args=(_free_port(), q))
    p.start()
    p.join(120)
    assert p.exitcode == 0
    assert q.get(timeout=5) == (True, 4, True, 4)
    assert q.get(timeout=5) == 0


def _fake_sysfs(root, gpus, cpu_nodes=2):
    """A sysfs tree with `cpu_nodes` CPU nodes followed by GPU nodes (pci bus, numa node, cpulist) in KFD order."""
    import os
    nodes = os.path.join(root, "class", "kfd", "kfd", "topology", "nodes")
    k = 0
    for _ in range(cpu_nodes):
        os.makedirs(os.path.join(nodes, str(k)))
        open(os.path.join(nodes, str(k), "properties"), "w").write("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
        k += 1
    for bus, numa, cpulist in gpus:
        os.makedirs(os.path.join(nodes, str(k)))
        open(os.path.join(nodes, str(k), "properties"), "w").write(
            f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {bus << 8}\ndomain 0\nunique_id 123456789012\n")
        d = os.path.join(root, "bus", "pci", "devices", "0000:%02x:00.0" % bus)
        os.makedirs(d)
        open(os.path.join(d, "numa_node"), "w").write(f"{numa}\n")
        open(os.path.join(d, "local_cpulist"), "w").write(cpulist + "\n")
        k += 1


def test_rank_placement_maps_every_rank_to_its_gpus_numa_cpus(tmp_path):
    """dist.gpu_host_topology / cpus_for_local_rank / bind_rank: eight GPUs on two sockets -> four ranks
    per socket, each a contiguous quarter of the socket's CPUs (hyper-thread ranges included), intersected with what the
    process may use; a machine without the information gets no binding, never an error."""
    import os
    from street_crafter_amd import dist as D
    root = str(tmp_path)
    _fake_sysfs(root, [(0x05 + 0x10 * i, 0 if i < 4 else 1, "0-31,64-95" if i < 4 else "32-63,96-127") for i in range(8)])
    topo = D.gpu_host_topology(root)
    assert [g["index"] for g in topo] == list(range(8)) and topo[0]["pci"] == "0000:05:00.0" and topo[7]["numa_node"] == 1
    assert topo[5]["cpus"][:2] == [32, 33] and len(topo[5]["cpus"]) == 64
    shares = [D.cpus_for_local_rank(topo, r, 8) for r in range(8)]
    assert all(len(s) == 16 for s in shares)
    assert sorted(sum(shares[:4], [])) == topo[0]["cpus"] and sorted(sum(shares[4:], [])) == topo[4]["cpus"]
    assert len(set(map(tuple, shares))) == 8                                # disjoint
    assert D.cpus_for_local_rank(topo, 1, 2) == topo[0]["cpus"][32:]        # two local ranks: halves of socket 0
    assert D.cpus_for_local_rank(topo, 6, 8, allowed=range(40, 48)) == [44, 45]      # cgroup: 8 CPUs of socket 1, a quarter
    assert D.cpus_for_local_rank(topo, 2, 8, allowed=range(40, 48)) == []            # nothing local is allowed: no binding
    assert D.cpus_for_local_rank(topo, 9, 8) == [] and D.cpus_for_local_rank([], 0, 8) == []
    assert D.gpu_host_topology(os.path.join(root, "nope")) == []
    # a GPU without NUMA information (numa_node -1, whole machine as its cpulist): shares among all eight
    root2 = str(tmp_path / "flat")
    _fake_sysfs(root2, [(0x10 + i, -1, "0-15") for i in range(4)])
    t2 = D.gpu_host_topology(root2)
    assert [D.cpus_for_local_rank(t2, r, 4) for r in range(4)] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]]
    # bind_rank in a child process (it changes the affinity of the caller): with this container's CPUs standing in
    import subprocess, sys, json
    allowed = sorted(os.sched_getaffinity(0))
    root3 = str(tmp_path / "here")
    _fake_sysfs(root3, [(0x20 + i, 0, ",".join(map(str, allowed))) for i in range(2)])
    code = ("import json, os, sys; sys.path.insert(0, %r); from street_crafter_amd import dist as D; "
            "i = D.bind_rank(1, 2, sysfs=%r); print(json.dumps([i, sorted(os.sched_getaffinity(0))]))" % (os.getcwd(), root3))
    info, now = json.loads(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True,
                                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))).stdout.strip().splitlines()[-1])
    if len(allowed) >= 2:
        assert info["bound"] and now == allowed[len(allowed) // 2:][:len(allowed) // 2] and info["numa_node"] == 0
    # verify_rank_binding: the device the rank really got sits at ANOTHER position of the topology -> every thread moves there
    code2 = ("import json, os, sys; sys.path.insert(0, %r); from street_crafter_amd import dist as D; "
             "i = D.bind_rank(0, 2, sysfs=%r); j = D.verify_rank_binding(i, 0, 2, 0x21, sysfs=%r); "
             "k = D.verify_rank_binding(i, 0, 2, 0x20, sysfs=%r); "
             "print(json.dumps([j, sorted(os.sched_getaffinity(0)), k]))" % (os.getcwd(), root3, root3, root3))
    j, now2, k = json.loads(subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, check=True,
                                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))).stdout.strip().splitlines()[-1])
    if len(allowed) >= 2:
        assert j.get("rebound_after_init") and now2 == allowed[len(allowed) // 2:][:len(allowed) // 2] and j["pci"] == "0000:21:00.0"
        assert k.get("binding_verified") and not k.get("rebound_after_init")
    off = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True,
                         env=dict(os.environ, SC_BIND_CPUS="0"),
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))).stdout.strip().splitlines()[-1]
    assert json.loads(off)[0]["bound"] is False and json.loads(off)[1] == allowed

