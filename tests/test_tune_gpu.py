"""Performance knobs are state of teo_tune blocks, not of the process ("thread-compatible, no globals except an opaque
teo_ctx*"; VERDICT r04 'What's weak' #7): two engines in one process keep their own choices, two host threads do not race on them, a
descriptor's block wins over the thread's, and nothing a test sets outlives the block it set it in."""
import ctypes as C
import threading

import pytest
import torch

from teochat_amd import _lib as L
from tests import _gpu as G
from tests import _tiny as TY

pytestmark = pytest.mark.gpu


def _engine(name="tinyB", dtype=torch.bfloat16):
    from teochat_amd.config import LlavaConfig, VisionConfig
    from teochat_amd.engine import TeoEngine
    t = TY.TINY[name]
    cfg = LlavaConfig(**t["llm"], mm_hidden_size=t["vit"]["hidden_size"], max_position_embeddings=1024, vision_config=VisionConfig(**t["vit"]))
    return TeoEngine(TY.state_dict(name), cfg, dtype=dtype, device="cuda:0", max_seq=1024)


def test_two_engines_keep_their_own_knobs():
    """Engine A forces the 128-row plain tile, engine B keeps the dispatch's choice (the software-pipelined 64 x 64 tile for this small projector GEMM).  Calls
    are interleaved; each engine's projector runs the family ITS block names every time (teo_last_kernel), the thread's bound block is a
    third, different choice that neither engine sees, and the values are bit-identical across families."""
    lib = G.lib()
    a, b = _engine(), _engine()
    a.tune_set("gemm_bm", 128)
    L.tune_set(b"gemm_bm", 64)                           # the calling thread's block: descriptor blocks win over it
    L.tune_set(b"gemm_wide", 0)
    feats = torch.randn(2, 256, a.cfg.mm_hidden_size, generator=torch.Generator().manual_seed(3)).to(torch.bfloat16).cuda()
    ya0 = a.project(feats)
    assert lib.teo_last_kernel().decode() == "gemm_mfma_128"
    for _ in range(3):
        yb = b.project(feats)
        assert lib.teo_last_kernel().decode() == "gemm_pipe_64x64"
        ya = a.project(feats)
        assert lib.teo_last_kernel().decode() == "gemm_mfma_128"
        assert torch.equal(ya, ya0) and torch.equal(ya, yb)
    # a primitive call from this thread sees the THREAD's block (64-row tiles forced), whatever the engines hold
    A = feats.view(-1, feats.shape[-1])
    W = torch.randn(256, A.shape[1], generator=torch.Generator().manual_seed(4)).to(torch.bfloat16).cuda()
    G.gemm(A, W)
    assert lib.teo_last_kernel().decode() == "gemm_mfma_64"
    assert a.tune.get("gemm_bm") == 128 and b.tune.get("gemm_bm") == 0 and L.thread_tune().get("gemm_bm") == 64
    # an engine's knob change drops its captured decode graph (the capture keeps kernel choices) and only its own
    a.tune_reset()
    a.project(feats)
    assert lib.teo_last_kernel().decode() == "gemm_pipe_64x64"


def test_two_threads_do_not_share_knobs():
    """Two host threads, each with its own bound block (128-row vs 64-row tiles), each on its own stream, hammer the same GEMM: every
    call of a thread dispatches the family of that thread's block, and both get the same bits."""
    lib = G.lib()
    g = torch.Generator().manual_seed(11)
    A = torch.randn(512, 256, generator=g).to(torch.bfloat16).cuda()
    W = (torch.randn(384, 256, generator=g) * 0.05).to(torch.bfloat16).cuda()
    want = G.gemm(A, W)
    torch.cuda.synchronize()
    seen, outs, errs = {}, {}, []
    barrier = threading.Barrier(2)

    def worker(name, bm, expect):
        try:
            t = L.Tune()
            assert t.set("gemm_bm", bm) == 0 and t.set("gemm_wide", 0) == 0
            t.bind()
            st = torch.cuda.Stream()
            kinds = set()
            Cc = torch.empty(512, 384, dtype=torch.bfloat16, device="cuda")
            barrier.wait()
            with torch.cuda.stream(st):
                for _ in range(200):
                    L.check(lib.teo_gemm(G.p(A), G.p(W), None, None, G.p(Cc), 512, 384, 256, 256, 384, 0, 0, L.TEO_BF16, L.TEO_BF16,
                                         C.c_void_p(st.cuda_stream)), "gemm")
                    kinds.add(lib.teo_last_kernel().decode())
            st.synchronize()
            seen[name], outs[name] = kinds, Cc.clone()
            lib.teo_tune_bind(None)
            t.close()
            assert kinds == {expect}, (name, kinds)
        except Exception as e:  # noqa: BLE001
            errs.append((name, repr(e)))
            try:
                barrier.abort()
            except Exception:  # noqa: BLE001
                pass

    th = [threading.Thread(target=worker, args=("t128", 128, "gemm_mfma_128")), threading.Thread(target=worker, args=("t64", 64, "gemm_mfma_64"))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert seen == {"t128": {"gemm_mfma_128"}, "t64": {"gemm_mfma_64"}}
    assert torch.equal(outs["t128"], want) and torch.equal(outs["t64"], want)


def test_nothing_outlives_a_released_block():
    """What the conftest fixture does after every GPU test: the thread's block is unbound and destroyed; the next call runs on the shipped
    defaults without anybody having 'reset' a global."""
    lib = G.lib()
    g = torch.Generator().manual_seed(5)
    A = torch.randn(512, 256, generator=g).to(torch.bfloat16).cuda()
    W = torch.randn(384, 256, generator=g).to(torch.bfloat16).cuda()
    L.tune_set(b"gemm_bm", 128)
    L.tune_set(b"gemm_wide", 0)
    G.gemm(A, W)
    assert lib.teo_last_kernel().decode() == "gemm_mfma_128"
    L.tune_release()
    G.gemm(A, W)
    assert lib.teo_last_kernel().decode() == "gemm_pipe_64x64"       # 48 tiles of 64 x 64: the dispatch's own choice is the software-pipelined small tile


@pytest.mark.parametrize("M,N,K,extra,want", [
    (257, 1024, 4096, "bias_res", "gemm_pipe_64x64"),        # tower fc2 at T = 1: 5 x 16 = 80 tiles of 64 x 64
    (514, 3072, 1024, "bias", "gemm_pipe_64x64"),            # tower qkv at T = 2: 432 tiles of 64 x 64, short K loop and wide N (limit 512)
    (514, 4096, 1024, "bias_gelu", "gemm_narrow_64"),        # tower fc1 + GELU at T = 2: 576 / 288 tiles, short K with an activation -> the LDS-DMA tile stays
    (1285, 1024, 4096, "bias_res", "gemm_pipe_64_r4"),       # tower fc2 at T = 5: 21 x 8 = 168 tiles of 64 x 128, long K -> ring of 4
    (2056, 1024, 4096, "bias_res", "gemm_pipe_128x96"),      # tower fc2 at T = 8: 17 x 11 = 187 tiles of 128 x 96
    (2056, 1024, 1024, "bias_res", "gemm_pipe_128x96"),      # tower out_proj at T = 8
    (200, 4096, 4096, "res", "gemm_pipe_64x64"),             # LLaMA o at M <= 256
    (638, 4096, 11008, "res", "gemm_pipe_128x96"),           # LLaMA down at C2's row count: 5 x 43 = 215 tiles
    (900, 4096, 4096, "res", "gemm_pipe_128"),               # LLaMA o at M = 641 .. 1024
    (100, 22016, 4096, "swiglu", "gemm_pipe_128"),           # gate/up + SwiGLU on a text-only prompt
    (2168, 4096, 4096, "res", "gemm_quad_160"),              # C3: unchanged
])
def test_production_dispatch_takes_the_small_pipelined_tiles_where_round_six_measured_them_ahead(M, N, K, extra, want):
    """The few-tile rule of gemm.hip (round 6, late; profiles/r06_pipe_candidates.txt): the tile is chosen for about one workgroup per CU.  Pins the
    rule at the shapes the model produces (CLIP tower at T = 1 / 2 / 5 / 8: modeling_image.py:136-151; LLaMA o / down / gate-up: llava_llama.py:88-99)
    and checks the result against the register-staged 128 x 128 kernel, bitwise."""
    lib = G.lib()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16).cuda()
    swiglu = extra == "swiglu"
    bias = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16).cuda() if "bias" in extra else None
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda() if "res" in extra else None
    kw = dict(bias=bias, res=res, act=L.ACT_GELU_ERF if "gelu" in extra else L.ACT_NONE, flags=L.GEMM_SWIGLU16 if swiglu else 0)
    L.tune_reset()
    got = G.gemm(A, W, **kw)
    ran = lib.teo_last_kernel().decode()
    for k, v in (("gemm_wide", 0), ("gemm_big", 0), ("gemm_sk", 0), ("gemm_narrow", 0), ("gemm_quad", 0), ("gemm_bm", 128)):
        assert L.tune_set(k.encode(), v) == 0
    ref = G.gemm(A, W, **kw)
    assert lib.teo_last_kernel().decode() == "gemm_mfma_128"
    L.tune_reset()
    assert torch.equal(got, ref), (ran, float((got.float() - ref.float()).abs().max()))
    assert ran == want, (ran, want, M, N, K)
