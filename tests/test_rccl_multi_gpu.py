"""The RCCL data path at world size > 1 (teo_ctx_create over N ranks + teo_allgather_visual), on a node with >= 2 GPUs.
Skipped on the one-GPU test boxes; the driver's multi-GPU node runs it.  (the world > 1 path had no test.)"""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, q):
    import torch.distributed as dist
    from teochat_amd.parallel import TeoComm, sharded_frame_features
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=ws, device_id=torch.device(f"cuda:{rank}"))
    try:
        ok = True
        comms = [TeoComm(rank, ws, rank), TeoComm(rank, ws, rank)]          # two communicators in one process group: ids must not collide
        for comm in comms:
            r, w, cu, hbm = comm.info()
            ok = ok and (r, w) == (rank, ws) and cu > 0
            for T in (16, 5, 1):                                             # equal, ragged, T < world
                g = torch.Generator().manual_seed(T)
                px = torch.randn(T, 3, 8, 8, generator=g).to(f"cuda:{rank}", torch.bfloat16)
                enc = lambda p: (p.reshape(p.shape[0], -1)[:, :48].reshape(p.shape[0], 4, 12) * 2 + 1)     # noqa: E731
                out = sharded_frame_features(enc, px, comm=comm)
                torch.cuda.synchronize()
                ok = ok and torch.equal(out, enc(px)) and tuple(out.shape) == (T, 4, 12)
        for comm in comms:
            comm.close()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL refuses two ranks on one device)")
def test_allgather_visual_two_ranks_over_rccl():
    import torch.multiprocessing as mp
    ws = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, ws, port, q)) for r in range(ws)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res
