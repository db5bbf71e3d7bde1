"""Config C4's frame-sharded tower with the REAL `TeoEngine.vit_features` (CLIP-ViT-L/14 shapes, 23 layers): two ranks share
cuda:0, each encodes its contiguous block of the T frames with the HIP tower, the visual tokens are all-gathered (gloo here;
RCCL through teo_allgather_visual on a multi-GPU node) and must equal the unsharded encode BIT FOR BIT on every rank --
frames are independent through the tower (T is the batch dim, modeling_image.py:641-643) and the kernels' reduction order
does not depend on how many frames share a launch."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _engine():
    """Full-size tower + projector in front of a two-layer LLaMA (the LLM is not under test here)."""
    from teochat_amd.config import LlavaConfig, VisionConfig
    from teochat_amd.engine import TeoEngine
    from teochat_amd.synthetic import synthetic_state_dict
    cfg = LlavaConfig(hidden_size=4096, num_attention_heads=32, num_key_value_heads=32, intermediate_size=11008,
                      num_hidden_layers=2, vocab_size=32000, mm_hidden_size=1024, max_position_embeddings=4096,
                      vision_config=VisionConfig(hidden_act="gelu"))
    sd = synthetic_state_dict(cfg, seed=2, device="cuda:0")
    return TeoEngine(sd, cfg, dtype=torch.bfloat16, device="cuda:0", max_seq=256)


def _worker(rank, ws, port, T, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        from oracle import teo_oracle as O
        from teochat_amd.parallel import frame_partition, sharded_frame_features
        eng = _engine()
        px = torch.stack(O.synthetic_frames(T, 224, seed=0)).to("cuda:0", dtype=torch.bfloat16)
        whole = eng.vit_features(px)
        calls = []

        def encode(p):
            calls.append(p.shape[0])
            return eng.vit_features(p)

        got = sharded_frame_features(encode, px)
        torch.cuda.synchronize()
        proj_equal = torch.equal(eng.project(got), eng.project(whole))
        q.put((rank, bool(torch.equal(got, whole)), tuple(got.shape), calls, frame_partition(T, ws)[rank][1], bool(proj_equal),
               float(whole.float().abs().sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("T", [16, 5])
def test_sharded_real_tower_equals_unsharded_bitwise(T):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, T, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sums = set()
    for rank, equal, shape, calls, mine, proj_equal, s in res:
        assert shape == (T, 256, 1024), (rank, shape)
        assert calls == [mine], (rank, calls)          # the rank encoded only its own block of frames
        assert equal, f"rank {rank}: gathered features differ from the unsharded encode"
        assert proj_equal
        sums.add(s)
    assert len(sums) == 1                               # both ranks hold the same features


def test_rccl_allgather_behind_the_c_abi_single_rank():
    """teo_ctx_create / teo_allgather_visual run RCCL itself (ncclCommInitRank + ncclAllGather) on this GPU: a one-rank
    communicator is all a single-GPU box can host (two ranks on one device are refused by RCCL), the N-rank path is the same
    call with a shared unique id.  Also the sharded tower routed through it (world 1: rank 0 owns every frame)."""
    import ctypes as C
    from teochat_amd import _lib as L
    from teochat_amd.parallel import TeoComm, sharded_frame_features
    comm = TeoComm(0, 1, 0)
    r, w, cu, hbm = C.c_int(-1), C.c_int(-1), C.c_int(0), C.c_size_t(0)
    L.check(comm.lib.teo_ctx_info(comm.handle, C.byref(r), C.byref(w), C.byref(cu), C.byref(hbm)), "teo_ctx_info")
    assert (r.value, w.value) == (0, 1) and cu.value == 256 and hbm.value > 200 * 2 ** 30      # MI355X: 256 CUs, 288 GB
    for dtype in (torch.bfloat16, torch.float32):
        send = torch.randn(512, 1024, device="cuda:0").to(dtype)
        recv = torch.zeros_like(send)
        comm.all_gather_rows(send, recv)
        torch.cuda.synchronize()
        assert torch.equal(send, recv)
    # a second unique id can be drawn and is 128 opaque bytes
    uid = (C.c_char * L.COMM_ID_BYTES)()
    L.check(comm.lib.teo_comm_unique_id(uid), "teo_comm_unique_id")
    assert any(b != 0 for b in uid.raw)
    # the sharded tower through the communicator
    from oracle import teo_oracle as O
    eng = _engine()
    px = torch.stack(O.synthetic_frames(3, 224, seed=1)).to("cuda:0", dtype=torch.bfloat16)
    assert torch.equal(sharded_frame_features(eng.vit_features, px, comm=comm), eng.vit_features(px))
    # argument errors come back as status codes, not crashes
    assert comm.lib.teo_allgather_visual(comm.handle, None, None, 4, 8, L.TEO_BF16, None) == -1
    assert comm.lib.teo_allgather_visual(None, None, None, 4, 8, L.TEO_BF16, None) == -1
    comm.close()
