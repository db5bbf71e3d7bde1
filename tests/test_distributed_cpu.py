"""world_size-2 gloo tests (CPU) of the N>1 plumbing: frame partition, all-gather order, ragged T.
The local encoder is a stand-in (a deterministic function of the pixels); the GPU path plugs TeoEngine.vit_features in."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from teochat_amd.parallel import frame_partition, shard_conversations, sharded_frame_features


def _fake_encode(px):
    # [c,3,H,W] -> [c,4,6]: depends on every frame's content and keeps frame identity
    c = px.shape[0]
    base = px.reshape(c, -1)[:, :24].reshape(c, 4, 6)
    return base * 2.0 + 1.0


def _worker(rank, ws, port, T, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        g = torch.Generator().manual_seed(0)
        px = torch.randn(T, 3, 4, 4, generator=g)
        out = sharded_frame_features(_fake_encode, px)
        ok = torch.equal(out, _fake_encode(px))
        q.put((rank, bool(ok), tuple(out.shape)))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("T", [16, 5, 1])
def test_sharded_frames_world2(T):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, T, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, shape in res:
        assert ok and shape == (T, 4, 6), (rank, ok, shape)


def test_partition_properties():
    for T in range(0, 40):
        for ws in (1, 2, 3, 8):
            parts = frame_partition(T, ws)
            assert len(parts) == ws and sum(c for _, c in parts) == T
            assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(ws - 1))     # contiguous, ordered
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    assert frame_partition(16, 8) == [(2 * r, 2) for r in range(8)]          # config C4: 2 frames per GPU
    assert sorted(sum((shard_conversations(8, r, 4) for r in range(4)), [])) == list(range(8))


def test_single_process_passthrough():
    px = torch.randn(3, 3, 4, 4)
    assert torch.equal(sharded_frame_features(_fake_encode, px), _fake_encode(px))
