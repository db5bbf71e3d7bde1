"""GPU parity tests of the HIP primitives, called through the C ABI, against the CPU oracle (oracle/teo_oracle.py)
and plain torch-CPU fp32 references on the same seeded inputs.

Tolerances (stated per test):
  fp32 kernels : absolute 1e-5 on O(1) values (accumulation order is the only difference)
  bf16 kernels : the oracle is evaluated in fp32 on the SAME bf16-rounded inputs and rounded to bf16 at the same
                 boundary; kernels must agree within 1 bf16 ulp (up to 2^-7 relative) plus an absolute 1e-3 floor.
  integer / data-movement kernels : bit-exact.
"""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import teo_oracle as O
from teochat_amd import _lib as L
from teochat_amd.engine import rope_tables
from tests import _gpu as G

pytestmark = pytest.mark.gpu

FP32_ATOL = 1e-5


def close_bf16(got, ref, ulps=1.0, floor=1e-3):
    got, ref = got.float().cpu(), ref.float().cpu()
    tol = ulps * (2.0 ** -7) * ref.abs() + floor
    bad = (got - ref).abs() > tol
    assert not bad.any(), f"{int(bad.sum())} / {bad.numel()} outside tolerance; max abs diff {float((got - ref).abs().max()):.3e}"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ---------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("rows,dim", [(1, 64), (5, 1024), (3, 4096), (2, 100)])
def test_layernorm_rmsnorm_fp32(rows, dim):
    x, w, b = rnd(rows, dim, seed=1), 1 + 0.1 * rnd(dim, seed=2), rnd(dim, seed=3, scale=0.1)
    y = G.layernorm(G.dev(x), G.dev(w), G.dev(b), 1e-5).cpu()
    torch.testing.assert_close(y, F.layer_norm(x, (dim,), w, b, 1e-5), atol=FP32_ATOL, rtol=1e-5)
    y = G.rmsnorm(G.dev(x), G.dev(w), 1e-5).cpu()
    torch.testing.assert_close(y, O.rmsnorm(x, w, 1e-5), atol=FP32_ATOL, rtol=1e-5)


@pytest.mark.parametrize("rows,dim", [(4, 1024), (3, 4096)])
def test_layernorm_rmsnorm_bf16(rows, dim):
    bf = torch.bfloat16
    x, w, b = G.bf16_round(rnd(rows, dim, seed=1)), G.bf16_round(1 + 0.1 * rnd(dim, seed=2)), G.bf16_round(rnd(dim, seed=3, scale=0.1))
    y = G.layernorm(G.dev(x, bf), G.dev(w, bf), G.dev(b, bf), 1e-5)
    close_bf16(y, G.bf16_round(F.layer_norm(x, (dim,), w, b, 1e-5)))
    y = G.rmsnorm(G.dev(x, bf), G.dev(w, bf), 1e-5)
    close_bf16(y, G.bf16_round(O.rmsnorm(x, w, 1e-5)))


# ---------------------------------------------------------------------------------------------- GEMM
def _gemm_ref(A, W, bias, res, act):
    y = A @ W.t()
    if bias is not None:
        y = y + bias
    if act == L.ACT_GELU_ERF:
        y = F.gelu(y)
    elif act == L.ACT_QUICK_GELU:
        y = y * torch.sigmoid(1.702 * y)
    if res is not None:
        y = y + res
    return y


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (7, 13, 5), (65, 130, 33), (257, 64, 588), (64, 300, 64)])
@pytest.mark.parametrize("act", [L.ACT_NONE, L.ACT_GELU_ERF, L.ACT_QUICK_GELU])
def test_gemm_generic_fp32(M, N, K, act):
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    y = G.gemm(G.dev(A), G.dev(W), G.dev(bias), G.dev(res), act=act).cpu()
    torch.testing.assert_close(y, _gemm_ref(A, W, bias, res, act), atol=2e-5, rtol=1e-5)
    y = G.gemm(G.dev(A), G.dev(W)).cpu()
    torch.testing.assert_close(y, A @ W.t(), atol=2e-5, rtol=1e-5)


def _swiglu_ref(A, gate, up):
    return F.silu(A @ gate.t()) * (A @ up.t())


def test_gemm_swiglu_fp32_and_layout():
    from teochat_amd.engine import interleave_gate_up
    M, K, Fd = 37, 48, 96
    A, gate, up = rnd(M, K, seed=1), rnd(Fd, K, seed=2, scale=0.3), rnd(Fd, K, seed=3, scale=0.3)
    gu = interleave_gate_up(gate, up)
    y = G.gemm(G.dev(A), G.dev(gu), flags=L.GEMM_SWIGLU16).cpu()
    torch.testing.assert_close(y, _swiglu_ref(A, gate, up), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 384, 128), (514, 256, 640), (1, 128, 64), (129, 132, 192)])
@pytest.mark.parametrize("act", [L.ACT_NONE, L.ACT_GELU_ERF])
def test_gemm_mfma_bf16(M, N, K, act):
    bf = torch.bfloat16
    assert G.lib().teo_gemm_uses_mfma(M, N, K, L.TEO_BF16, 0) == 1
    A, W = G.bf16_round(rnd(M, K, seed=1)), G.bf16_round(rnd(N, K, seed=2, scale=0.2))
    bias, res = G.bf16_round(rnd(N, seed=3)), G.bf16_round(rnd(M, N, seed=4))
    ref = _gemm_ref(A, W, bias, res, act)
    y = G.gemm(G.dev(A, bf), G.dev(W, bf), G.dev(bias, bf), G.dev(res, bf), act=act)
    close_bf16(y, G.bf16_round(ref))
    # fp32 output (logits form) must match the exact product tightly: no output rounding
    y32 = G.gemm(G.dev(A, bf), G.dev(W, bf), out_dtype=torch.float32).cpu()
    torch.testing.assert_close(y32, A @ W.t(), atol=1e-3, rtol=1e-4)
    # and the generic kernel on the same inputs (on-GPU cross-check used at full size below)
    ys = G.gemm(G.dev(A, bf), G.dev(W, bf), out_dtype=torch.float32, flags=L.GEMM_FORCE_SIMPLE).cpu()
    torch.testing.assert_close(y32, ys, atol=1e-3, rtol=1e-4)


def test_gemm_mfma_swiglu_bf16():
    from teochat_amd.engine import interleave_gate_up
    bf = torch.bfloat16
    M, K, Fd = 150, 128, 256
    A = G.bf16_round(rnd(M, K, seed=1))
    gate, up = G.bf16_round(rnd(Fd, K, seed=2, scale=0.2)), G.bf16_round(rnd(Fd, K, seed=3, scale=0.2))
    gu = interleave_gate_up(gate, up)
    assert G.lib().teo_gemm_uses_mfma(M, 2 * Fd, K, L.TEO_BF16, L.GEMM_SWIGLU16) == 1
    y = G.gemm(G.dev(A, bf), G.dev(gu, bf), flags=L.GEMM_SWIGLU16)
    close_bf16(y, G.bf16_round(_swiglu_ref(A, gate, up)))


def test_gemm_mfma_identity_asymmetric():
    """A = I against an asymmetric W catches transposed fragment/epilogue mappings exactly."""
    bf = torch.bfloat16
    n = 256
    A = torch.eye(n)
    W = (torch.arange(n * n, dtype=torch.float32).view(n, n) % 251) - 125.0     # exactly representable in bf16
    y = G.gemm(G.dev(A, bf), G.dev(W, bf), out_dtype=torch.float32).cpu()
    assert torch.equal(y, W.t())


def test_gemm_full_size_7b_shapes_mfma_vs_generic():
    """BASELINE shapes (prefill rows of config C3, one LLaMA layer's FFN): MFMA kernel vs the generic kernel on the
    GPU for all outputs, and vs the CPU fp32 product on a sample of rows/cols."""
    bf = torch.bfloat16
    M, K, N = 2168, 4096, 11008
    g = torch.Generator(device="cuda").manual_seed(0)
    A = torch.randn(M, K, generator=g, device="cuda").to(bf)
    W = (torch.randn(N, K, generator=g, device="cuda") * 0.02).to(bf)
    y = G.gemm(A, W, out_dtype=torch.float32)
    ys = G.gemm(A, W, out_dtype=torch.float32, flags=L.GEMM_FORCE_SIMPLE)
    torch.testing.assert_close(y, ys, atol=2e-3, rtol=1e-3)
    rows = torch.tensor([0, 1, 127, 128, 1000, 2047, 2048, 2167])
    cols = torch.tensor([0, 5, 127, 128, 5000, 11007])
    ref = A[rows].float().cpu() @ W[cols].float().cpu().t()
    torch.testing.assert_close(y[rows][:, cols].cpu(), ref, atol=2e-3, rtol=1e-3)
    # linearity (size-independent property): f(2A) == 2 f(A) exactly in fp32 output (powers of two are exact)
    y2 = G.gemm((A.float() * 2).to(bf), W, out_dtype=torch.float32)
    assert torch.equal(y2, 2 * y)


# ---------------------------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, causal, scale, round_p=False):
    B, H, Sq, d = q.shape
    Hk, Sk = k.shape[1], k.shape[2]
    if Hk != H:
        k = k.repeat_interleave(H // Hk, dim=1)
        v = v.repeat_interleave(H // Hk, dim=1)
    s = (q @ k.transpose(-1, -2)) * scale
    if causal:
        qpos = torch.arange(Sq).view(Sq, 1) + (Sk - Sq)
        s = s.masked_fill(torch.arange(Sk).view(1, Sk) > qpos, float("-inf"))
    m = s.max(-1, keepdim=True).values
    p = torch.exp(s - m)
    pn = G.bf16_round(p) if round_p else p
    o = (pn @ v) / p.sum(-1, keepdim=True)
    return o.transpose(1, 2).reshape(B, Sq, H * d)


@pytest.mark.parametrize("B,H,Hk,Sq,Sk,d,causal", [(2, 4, 4, 257, 257, 16, False), (1, 4, 2, 70, 70, 16, True),
                                                   (1, 2, 2, 5, 133, 128, True), (1, 3, 1, 1, 300, 64, True)])
def test_attention_generic_fp32(B, H, Hk, Sq, Sk, d, causal):
    q, k, v = rnd(B, H, Sq, d, seed=1), rnd(B, Hk, Sk, d, seed=2), rnd(B, Hk, Sk, d, seed=3)
    sc = d ** -0.5
    o = G.attention(G.dev(q), G.dev(k), G.dev(v), causal, sc).cpu()
    torch.testing.assert_close(o, _attn_ref(q, k, v, causal, sc), atol=FP32_ATOL, rtol=1e-5)


@pytest.mark.parametrize("B,H,Hk,Sq,Sk,d,causal", [(2, 2, 2, 257, 257, 64, False),     # ViT shape (N = 257)
                                                   (1, 2, 2, 300, 300, 128, True),      # LLaMA prefill
                                                   (1, 4, 2, 100, 420, 128, True),      # continuation with a past, GQA
                                                   (1, 2, 2, 64, 64, 64, True),
                                                   (1, 1, 1, 1, 65, 128, True)])
def test_attention_mfma_bf16(B, H, Hk, Sq, Sk, d, causal):
    bf = torch.bfloat16
    q, k, v = (G.bf16_round(rnd(B, H, Sq, d, seed=1)), G.bf16_round(rnd(B, Hk, Sk, d, seed=2)),
               G.bf16_round(rnd(B, Hk, Sk, d, seed=3)))
    sc = d ** -0.5
    qd, kd, vd = G.dev(q, bf), G.dev(k, bf), G.dev(v, bf)
    vt = G.make_vt(vd)
    vt[..., Sk:] = float("nan")          # padding beyond kv_len must never reach the output
    o = G.attention(qd, kd, vd, causal, sc, vt=vt)
    assert torch.isfinite(o.float()).all()
    # the oracle's "flash64" mode rounds P exactly where the kernel does (64-key tiles, running max)
    kk, vv = (k, v) if Hk == H else (k.repeat_interleave(H // Hk, dim=1), v.repeat_interleave(H // Hk, dim=1))
    vis = None
    if causal:
        vis = (torch.arange(Sk).view(1, Sk) <= (torch.arange(Sq).view(Sq, 1) + (Sk - Sq))).view(1, 1, Sq, Sk)
    ref = O.attention_core(q, kk, vv, vis, sc, G.bf16_round, mode="flash64").transpose(1, 2).reshape(B, Sq, H * d)
    close_bf16(o, G.bf16_round(ref), ulps=2.0, floor=2e-3)
    os_ = G.attention(qd, kd, vd, causal, sc, force_simple=True)
    torch.testing.assert_close(o.float(), os_.float(), atol=1.5e-2, rtol=1.5e-2)


@pytest.mark.parametrize("B,H,Hk,Sq,Sk,d,causal", [(1, 8, 8, 700, 700, 128, True),      # several query blocks, tail tile (700 % 64 = 60)
                                                   (1, 8, 2, 333, 901, 128, True),      # chunked prefill (past 568), GQA, ragged both ways
                                                   (2, 16, 16, 257, 257, 64, False),    # the tower's shape: four tile pairs, one-key tail
                                                   (1, 8, 8, 130, 130, 64, True)])
def test_attention_flash_pipeline_and_workgroup_order_do_not_change_a_bit(B, H, Hk, Sq, Sk, d, causal):
    """The in-wave software pipeline (score MFMAs of tile t+1 issued with the softmax of tile t; "flash_pipe") and the mirrored order of
    the causal workgroups ("flash_order") change WHEN things happen, never what is computed: outputs are bit-identical to the
    one-tile-at-a-time loop in plain heavy-first order, also with NaN in the V^T padding beyond kv_len."""
    bf = torch.bfloat16
    lib = G.lib()
    q, k, v = (G.bf16_round(rnd(B, H, Sq, d, seed=4)), G.bf16_round(rnd(B, Hk, Sk, d, seed=5)), G.bf16_round(rnd(B, Hk, Sk, d, seed=6)))
    qd, kd, vd = G.dev(q, bf), G.dev(k, bf), G.dev(v, bf)
    vt = G.make_vt(vd)
    vt[..., Sk:] = float("nan")
    outs = {}
    for pipe, order in ((0, 0), (1, 1), (1, 0), (0, 1), (-1, 1)):
        assert L.tune_set(b"flash_pipe", pipe) == 0 and L.tune_set(b"flash_order", order) == 0
        outs[(pipe, order)] = G.attention(qd, kd, vd, causal, d ** -0.5, vt=vt)
        assert lib.teo_last_kernel() == b"attn_flash32"
    L.tune_reset()
    base = outs[(0, 0)]
    assert torch.isfinite(base.float()).all()
    for key, o in outs.items():
        assert torch.equal(o, base), key


def test_attention_mfma_online_softmax_rescale_branch():
    """Force the running max to jump late in the key sequence (spike one key against one query)."""
    bf = torch.bfloat16
    B, H, S, d = 1, 1, 256, 128
    q, k, v = rnd(B, H, S, d, seed=1, scale=0.3), rnd(B, H, S, d, seed=2, scale=0.3), rnd(B, H, S, d, seed=3)
    k[0, 0, 200] = q[0, 0, 230] * 40.0       # key 200 dominates query 230 (third KV tile)
    q, k, v = G.bf16_round(q), G.bf16_round(k), G.bf16_round(v)
    sc = d ** -0.5
    vd = G.dev(v, bf)
    o = G.attention(G.dev(q, bf), G.dev(k, bf), vd, True, sc, vt=G.make_vt(vd)).float().cpu()
    ref = _attn_ref(q.double(), k.double(), v.double(), True, sc).float()
    torch.testing.assert_close(o, ref, atol=1.5e-2, rtol=1.5e-2)


def test_attention_mfma_full_size_properties():
    """Config C3 prefill shape (L = 2168, 32 heads, d = 128): V = ones gives ones (softmax rows sum to 1);
    a sample of rows matches the generic kernel; two runs are bit-identical."""
    bf = torch.bfloat16
    H, S, d = 32, 2168, 128
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(1, H, S, d, generator=g, device="cuda").to(bf)
    k = torch.randn(1, H, S, d, generator=g, device="cuda").to(bf)
    v = torch.randn(1, H, S, d, generator=g, device="cuda").to(bf)
    sc = d ** -0.5
    ones = torch.ones_like(v)
    o1 = G.attention(q, k, ones, True, sc, vt=G.make_vt(ones)).float()
    torch.testing.assert_close(o1, torch.ones_like(o1), atol=8e-3, rtol=0)
    vt = G.make_vt(v)
    o = G.attention(q, k, v, True, sc, vt=vt)
    o2 = G.attention(q, k, v, True, sc, vt=vt)
    assert torch.equal(o, o2)
    rows = [0, 1, 63, 64, 1000, 2167]
    for r in rows:
        osr = G.attention(q[:, :, r:r + 1].contiguous(), k[:, :, :r + 1].contiguous(), v[:, :, :r + 1].contiguous(), True, sc,
                          force_simple=True)
        torch.testing.assert_close(o[:, r:r + 1].float(), osr.float(), atol=1.5e-2, rtol=1.5e-2)


# ---------------------------------------------------------------------------------------------- RoPE + KV append
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("S,past", [(37, 0), (1, 12), (70, 5)])
def test_rope_kv_append(dtype, S, past):
    from teochat_amd.engine import rope_tables
    H, Hk, hd, S_max = 4, 2, 32, 128
    ld = (H + 2 * Hk) * hd
    qkv = rnd(S, ld, seed=1)
    if dtype == torch.bfloat16:
        qkv = G.bf16_round(qkv)
    pos = torch.arange(past, past + S)
    cs, sn = rope_tables(hd, 10000.0, 256)
    d_qkv = G.dev(qkv, dtype)
    kc = torch.zeros(Hk, S_max, hd, dtype=dtype, device="cuda")
    vc, vtc = torch.zeros_like(kc), torch.zeros(Hk, hd, S_max, dtype=dtype, device="cuda")
    d_pos, d_cs, d_sn = pos.to(torch.int32).cuda(), cs.cuda(), sn.cuda()      # keep device buffers alive across the call
    L.check(G.lib().teo_rope_kv_append(G.p(d_qkv), ld, G.p(d_pos), G.p(d_cs), G.p(d_sn),
                                       G.p(kc), G.p(vc), G.p(vtc), S, past, S_max, H, Hk, hd, G.DT[dtype], G.stream()), "rope")
    c, s = O.rope_cos_sin(pos, hd, 10000.0, torch.float32)
    q = qkv[:, :H * hd].view(S, H, hd)
    k = qkv[:, H * hd:(H + Hk) * hd].view(S, Hk, hd)
    v = qkv[:, (H + Hk) * hd:].view(S, Hk, hd)
    qr = q * c[:, None] + O.rotate_half(q) * s[:, None]
    kr = k * c[:, None] + O.rotate_half(k) * s[:, None]
    got_q = d_qkv[:, :H * hd].float().cpu().view(S, H, hd)
    got_k = kc[:, past:past + S].float().cpu().transpose(0, 1)
    if dtype == torch.float32:
        torch.testing.assert_close(got_q, qr, atol=1e-6, rtol=1e-6)
        torch.testing.assert_close(got_k, kr, atol=1e-6, rtol=1e-6)
    else:
        close_bf16(got_q, G.bf16_round(qr))
        close_bf16(got_k, G.bf16_round(kr))
    assert torch.equal(vc[:, past:past + S].float().cpu().transpose(0, 1), v)            # V is a bit-exact copy
    assert torch.equal(vtc[:, :, past:past + S].float().cpu().permute(2, 0, 1), v)       # and so is V^T
    assert float(kc[:, :past].abs().sum()) == 0 and float(kc[:, past + S:].abs().sum()) == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("S,past", [(300, 0), (129, 64), (16, 8)])
def test_rope_kv_vt_one_launch_equals_the_two_launches(dtype, S, past):
    """Round 6: RoPE + K append and V / V^T append run as ONE launch (`rope_vt_fused`, default on) where the 16-byte paths apply; same kernels' bodies,
    so q (in place), the K / V caches and the V^T cache must equal the two-launch form bit for bit -- at LLaMA's head geometry, ragged row counts
    and a non-zero past (apply_rotary_pos_emb + cache append, tf LlamaAttention.forward reached from llava_llama.py:88-99)."""
    from teochat_amd.engine import rope_tables
    H, Hk, hd, S_max = 32, 32, 128, 512
    ld = (H + 2 * Hk) * hd
    qkv = rnd(S, ld, seed=S + past).to(dtype)
    cs, sn = rope_tables(hd, 10000.0, S_max)
    d_pos, d_cs, d_sn = torch.arange(past, past + S).to(torch.int32).cuda(), cs.cuda(), sn.cuda()
    outs = []
    try:
        for knob in (1, 0):
            assert L.tune_set(b"rope_vt_fused", knob) == 0
            d_qkv = qkv.clone().cuda()
            kc = torch.zeros(Hk, S_max, hd, dtype=dtype, device="cuda")
            vc, vtc = torch.zeros_like(kc), torch.zeros(Hk, hd, S_max, dtype=dtype, device="cuda")
            L.check(G.lib().teo_rope_kv_append(G.p(d_qkv), ld, G.p(d_pos), G.p(d_cs), G.p(d_sn), G.p(kc), G.p(vc), G.p(vtc), S, past, S_max, H, Hk, hd,
                                               G.DT[dtype], G.stream()), "rope")
            torch.cuda.synchronize()
            outs.append((d_qkv, kc, vc, vtc))
    finally:
        L.tune_reset()
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)
    v = qkv[:, (H + Hk) * hd:].view(S, Hk, hd)
    assert torch.equal(outs[0][2][:, past:past + S].cpu().transpose(0, 1), v) and torch.equal(outs[0][3][:, :, past:past + S].cpu().permute(2, 0, 1), v)
    assert float(outs[0][1][:, :past].float().abs().sum()) == 0 and bool((outs[0][0][:, :H * hd] != qkv.cuda()[:, :H * hd]).any())


# ---------------------------------------------------------------------------------------------- data movement
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embed_splice_bit_exact(dtype):
    V, D, NV = 50, 72, 6
    emb, vis = rnd(V, D, seed=1).to(dtype), rnd(2 * NV, D, seed=2).to(dtype)
    plan = torch.tensor([1, 7, -1, -2, -12, L.INT32_MIN, 49, 0, -6], dtype=torch.int32)
    out = torch.empty(plan.numel(), D, dtype=dtype, device="cuda")
    d_plan, d_emb, d_vis = plan.cuda(), emb.cuda(), vis.cuda()
    L.check(G.lib().teo_embed_splice(G.p(d_plan), G.p(d_emb), G.p(d_vis), G.p(out), plan.numel(), D,
                                     G.DT[dtype], G.stream()), "splice")
    for r, pl in enumerate(plan.tolist()):
        exp = torch.zeros(D, dtype=dtype) if pl == L.INT32_MIN else (emb[pl] if pl >= 0 else vis[-pl - 1])
        assert torch.equal(out[r].cpu(), exp), r


def test_im2col_vt_dropcls_bit_exact():
    T, Cc, img, P = 2, 3, 28, 14
    px = rnd(T, Cc, img, img, seed=1)
    ld = 640
    cols = torch.empty(T * 4, ld, device="cuda")
    d_px = px.cuda()
    L.check(G.lib().teo_im2col_patches(G.p(d_px), G.p(cols), T, Cc, img, P, ld, L.TEO_F32, G.stream()), "im2col")
    ref = F.unfold(px, kernel_size=P, stride=P).transpose(1, 2).reshape(T * 4, Cc * P * P)
    assert torch.equal(cols[:, :Cc * P * P].cpu(), ref) and float(cols[:, Cc * P * P:].abs().sum()) == 0
    # ViT value transpose
    N, H, hd = 37, 2, 64
    D = H * hd
    qkv = rnd(T * N, 3 * D, seed=2)
    ldv = 64
    vt = torch.full((T, H, hd, ldv), 7.0, device="cuda")
    d_qkv = qkv.cuda()
    L.check(G.lib().teo_vit_value_transpose(G.p(d_qkv), G.p(vt), T, N, H, hd, ldv, L.TEO_F32, G.stream()), "vt")
    v = qkv.view(T, N, 3, H, hd)[:, :, 2]                       # [T,N,H,hd]
    assert torch.equal(vt[..., :N].cpu(), v.permute(0, 2, 3, 1)) and float(vt[..., N:].abs().sum()) == 0
    # the 16-bit form (16-byte accesses through an LDS transpose) at the tower's shape: N = 257, ldv = 320, and a ragged one
    for (Tv, Nv, Hv, hdv, ldvv, dt, code) in ((3, 257, 16, 64, 320, torch.bfloat16, L.TEO_BF16), (2, 70, 2, 128, 128, torch.float16, L.TEO_F16),
                                              (1, 9, 1, 32, 16, torch.bfloat16, L.TEO_BF16)):
        Dv = Hv * hdv
        q16 = rnd(Tv * Nv, 3 * Dv, seed=5).to(dt)
        vt16 = torch.full((Tv, Hv, hdv, ldvv), 7.0, dtype=dt, device="cuda")
        d_q16 = q16.cuda()
        L.check(G.lib().teo_vit_value_transpose(G.p(d_q16), G.p(vt16), Tv, Nv, Hv, hdv, ldvv, code, G.stream()), "vt16")
        v16 = q16.view(Tv, Nv, 3, Hv, hdv)[:, :, 2]
        assert torch.equal(vt16[..., :Nv].cpu(), v16.permute(0, 2, 3, 1)) and float(vt16[..., Nv:].float().abs().sum()) == 0
    # drop CLS
    h = rnd(T, N, D, seed=3)
    out = torch.empty(T, N - 1, D, device="cuda")
    d_h = h.cuda()
    L.check(G.lib().teo_drop_cls(G.p(d_h), G.p(out), T, N, D, L.TEO_F32, G.stream()), "drop_cls")
    assert torch.equal(out.cpu(), h[:, 1:])


def test_vit_embed_ln_fp32():
    T, NP, D = 2, 16, 64
    patch, cls, pos = rnd(T * NP, D, seed=1), rnd(D, seed=2), rnd(NP + 1, D, seed=3)
    w, b = 1 + 0.1 * rnd(D, seed=4), rnd(D, seed=5, scale=0.1)
    out = torch.empty(T, NP + 1, D, device="cuda")
    dv = [t.cuda() for t in (patch, cls, pos, w, b)]
    L.check(G.lib().teo_vit_embed_ln(G.p(dv[0]), G.p(dv[1]), G.p(dv[2]), G.p(dv[3]), G.p(dv[4]),
                                     G.p(out), T, NP, D, 1e-5, L.TEO_F32, G.stream()), "embed_ln")
    emb = torch.cat([cls.view(1, 1, D).expand(T, 1, D), patch.view(T, NP, D)], dim=1) + pos
    torch.testing.assert_close(out.cpu(), F.layer_norm(emb, (D,), w, b, 1e-5), atol=FP32_ATOL, rtol=1e-5)


def test_argmax_first_index_on_ties():
    x = torch.zeros(3, 32000)
    x[0, 31999] = 5.0
    x[1, 77] = 2.0; x[1, 20000] = 2.0
    x[2] = -1.0
    tok = torch.empty(3, dtype=torch.int64, device="cuda")
    d_x = x.cuda()
    L.check(G.lib().teo_argmax(G.p(d_x), G.p(tok), 3, 32000, G.stream()), "argmax")
    assert tok.tolist() == [31999, 77, 0]
    y = rnd(1, 32000, seed=9)
    d_y = y.cuda()
    L.check(G.lib().teo_argmax(G.p(d_y), G.p(tok), 1, 32000, G.stream()), "argmax")
    assert int(tok[0]) == int(y.argmax())


# ---------------------------------------------------------------------------------------------- GEMV
@pytest.mark.parametrize("N,K", [(64, 64), (300, 1376), (4096, 4096), (130, 520)])
def test_gemv_fp32(N, K):
    x, W, res, nw = rnd(K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3), 1 + 0.1 * rnd(K, seed=4)
    y = G.gemv(G.dev(x), G.dev(W)).cpu()
    torch.testing.assert_close(y, W @ x, atol=3e-5, rtol=1e-5)
    y = G.gemv(G.dev(x), G.dev(W), norm_w=G.dev(nw), res=G.dev(res)).cpu()
    torch.testing.assert_close(y, W @ O.rmsnorm(x, nw, 1e-5) + res, atol=3e-5, rtol=1e-5)


@pytest.mark.parametrize("N,K", [(256, 256), (12288, 4096), (4096, 11008), (32000, 4096)])
def test_gemv_bf16_real_shapes(N, K):
    bf = torch.bfloat16
    x, W, res = G.bf16_round(rnd(K, seed=1)), G.bf16_round(rnd(N, K, seed=2, scale=0.02)), G.bf16_round(rnd(N, seed=3))
    nw = G.bf16_round(1 + 0.1 * rnd(K, seed=4))
    y = G.gemv(G.dev(x, bf), G.dev(W, bf), res=G.dev(res, bf))
    close_bf16(y, G.bf16_round(W @ x + res))
    y32 = G.gemv(G.dev(x, bf), G.dev(W, bf), norm_w=G.dev(nw, bf), out_dtype=torch.float32).cpu()
    xn = G.bf16_round(O.rmsnorm(x, nw, 1e-5))
    torch.testing.assert_close(y32, W @ xn, atol=2e-4, rtol=1e-4)


def test_gemv_swiglu_bf16_and_matches_gemm_row():
    from teochat_amd.engine import interleave_gate_up
    bf = torch.bfloat16
    K, Fd = 4096, 11008
    x = G.bf16_round(rnd(K, seed=1))
    gate, up = G.bf16_round(rnd(Fd, K, seed=2, scale=0.02)), G.bf16_round(rnd(Fd, K, seed=3, scale=0.02))
    nw = G.bf16_round(1 + 0.1 * rnd(K, seed=4))
    gu = G.dev(interleave_gate_up(gate, up), bf)
    y = G.gemv(G.dev(x, bf), gu, norm_w=G.dev(nw, bf), flags=L.GEMM_SWIGLU16)
    xn = G.bf16_round(O.rmsnorm(x, nw, 1e-5))
    close_bf16(y, G.bf16_round(F.silu(gate @ xn) * (up @ xn)))
    # the prefill GEMM on a 1-row batch of the same (already normalised) activations agrees
    yg = G.gemm(G.dev(xn.view(1, K), bf), gu, flags=L.GEMM_SWIGLU16)
    close_bf16(y, yg[0].float(), ulps=2.0)


def test_abi_error_convention():
    lib = G.lib()
    rc = lib.teo_gemm(None, None, None, None, None, 4, 4, 4, 4, 4, 0, 0, 7, 0, None)
    assert rc == -1 and b"dtype" in lib.teo_last_error()
    x = torch.zeros(8, device="cuda")
    rc = lib.teo_gemv(G.p(x), G.p(x), None, None, G.p(x), 2, 3, 1e-5, 0, L.TEO_F32, L.TEO_F32, G.stream())
    assert rc == -1 and b"multiple" in lib.teo_last_error()
    assert lib.teo_layernorm(None, None, None, None, 0, 8, 1e-5, L.TEO_F32, None) == 0      # empty input is fine


# ---------------------------------------------------------------------------------------------- sampler (N1)
def _sample(lg, temperature, top_k, seed, draw, top_p=1.0):
    tok = torch.empty(1, dtype=torch.int64, device="cuda")
    L.check(G.lib().teo_sample_topk(G.p(lg), G.p(tok), lg.numel(), temperature, top_k, top_p, seed, draw, G.stream()), "sample")
    return int(tok.item())


def test_sampler_limits_and_determinism():
    lg = rnd(32000, seed=5, scale=3.0).cuda()
    am = int(lg.argmax())
    assert _sample(lg, 1e-4, 50, 1, 0) == am                 # temperature -> 0 is greedy
    assert all(_sample(lg, 1.0, 1, s, d) == am for s in (1, 2) for d in (0, 7))     # top-1 is greedy
    a = [_sample(lg, 0.8, 50, 1234, d) for d in range(64)]
    b = [_sample(lg, 0.8, 50, 1234, d) for d in range(64)]
    c = [_sample(lg, 0.8, 50, 999, d) for d in range(64)]
    assert a == b and a != c                                   # counter-based: (seed, draw) fixes the token
    top50 = set(lg.topk(50).indices.tolist())
    assert set(a) <= top50 and set(c) <= top50


def test_sampler_distribution_matches_softmax_of_topk():
    g = torch.Generator().manual_seed(11)
    lg = torch.randn(1000, generator=g) * 2.0
    k, temp, n = 8, 0.7, 4000
    d_lg = lg.cuda()
    draws = torch.tensor([_sample(d_lg, temp, k, 42, d) for d in range(n)])
    top = lg.topk(k)
    p = torch.softmax(top.values / temp, dim=0)
    counts = torch.stack([(draws == i).sum() for i in top.indices]).float()
    assert int(counts.sum()) == n                              # never outside the top-k
    # each frequency within 5 sigma of the binomial expectation
    sigma = torch.sqrt(n * p * (1 - p)).clamp_min(1.0)
    assert bool(((counts - n * p).abs() < 5 * sigma).all()), (counts, n * p)


@pytest.mark.parametrize("vocab,top_k,top_p", [(32000, 50, 1.0), (32000, 1000, 0.9), (32000, 7, 1.0), (5000, 50, 0.8), (32768, 50, 1.0)])
def test_sampler_register_form_draws_what_the_radix_form_draws(vocab, top_k, top_p):
    """vocab <= 32768: the row is held in registers and the k-th largest key found by bisection (round 4); larger rows -- and rows that
    are not 16-byte aligned -- take the 4-pass radix select.  Same threshold, same survivors in the same order, same draw: the two forms
    return the same token for every (seed, draw), duplicates of the k-th value included."""
    g = torch.Generator().manual_seed(vocab + top_k)
    lg = torch.randn(vocab, generator=g) * 3.0
    lg[torch.randint(0, vocab, (40,), generator=g)] = float(lg.topk(top_k).values[-1])       # ties AT the threshold
    aligned = lg.cuda()
    pad = torch.empty(vocab + 1, device="cuda")
    pad[1:] = aligned
    shifted = pad[1:]                                                                          # 4 bytes off: the radix form
    assert aligned.data_ptr() % 16 == 0 and shifted.data_ptr() % 16 != 0
    for seed in (1, 77):
        a = [_sample(aligned, 3.0, top_k, seed, d, top_p) for d in range(48)]
        b = [_sample(shifted, 3.0, top_k, seed, d, top_p) for d in range(48)]
        assert a == b, (seed, a, b)


def test_sampler_ties_and_small_vocab():
    lg = torch.zeros(300).cuda()                               # all tied: every index may be drawn, never out of range
    s = {_sample(lg, 1.0, 50, 3, d) for d in range(200)}
    assert min(s) >= 0 and max(s) < 300 and len(s) > 20
    lg2 = torch.tensor([0.0, 5.0, -1.0]).cuda()
    assert _sample(lg2, 0.05, 0, 1, 0) == 1                    # top_k = 0 disables the filter


# ---------------------------------------------------------------------------------------------- fp8 weights (config C5)
@pytest.mark.parametrize("N,K,norm,swiglu", [(12288, 4096, True, False), (4096, 11008, False, False), (22016, 4096, True, True),
                                             (4096, 4096, False, False), (32000, 4096, True, False), (64, 48, False, False)])
def test_gemv_fp8_weights(N, K, norm, swiglu):
    """fp8-e4m3 weights with power-of-two row scales: the GEMV must equal the bf16 GEMV on the dequantised weights
    (which are exactly representable in bf16) up to fp32 summation order."""
    from teochat_amd.engine import quantize_fp8_rows
    bf = torch.bfloat16
    W = G.bf16_round(rnd(N, K, seed=2, scale=0.02))
    q, s, dq = quantize_fp8_rows(W.to(bf))
    assert torch.equal(dq.float(), q.view(torch.float8_e4m3fn).float() * s[:, None])          # exact in bf16
    assert float((dq.float() - W).abs().max()) <= 0.07 * float(W.abs().max())                # e4m3: 3 mantissa bits
    x = G.bf16_round(rnd(K, seed=1))
    nw = G.bf16_round(1 + 0.1 * rnd(K, seed=4)) if norm else None
    res = None if (swiglu or norm) else G.bf16_round(rnd(N, seed=3))
    flags = L.GEMM_SWIGLU16 if swiglu else 0
    dx, dq_d, q_d, s_d = G.dev(x, bf), dq.cuda(), q.cuda(), s.cuda()
    dn = G.dev(nw, bf) if norm else None
    dr = G.dev(res, bf) if res is not None else None
    Ny = N // 2 if swiglu else N
    y8 = torch.empty(Ny, dtype=torch.float32, device="cuda")
    L.check(G.lib().teo_gemv_w8(G.p(dx), G.p(q_d), G.p(s_d), G.p(dn), G.p(dr), G.p(y8), N, K, 1e-5, flags, L.TEO_F32, G.stream()), "gemv_w8")
    y16 = G.gemv(dx, dq_d, norm_w=dn, res=dr, flags=flags, out_dtype=torch.float32)
    torch.testing.assert_close(y8, y16, atol=2e-4, rtol=1e-4)
    xn = G.bf16_round(O.rmsnorm(x, nw, 1e-5)) if norm else x
    ref = dq.float() @ xn
    if swiglu:
        idx = torch.arange(N // 2)
        g_rows = (idx // 16) * 32 + idx % 16
        ref = F.silu(ref[g_rows]) * ref[g_rows + 16]
    if res is not None:
        ref = ref + res
    torch.testing.assert_close(y8.cpu(), ref, atol=3e-4, rtol=2e-4)


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("N,K", [(4096, 4096), (4096, 11008), (130, 1168), (6, 64)])
def test_gemv_splitk_rows_and_chunks_per_step_do_not_change_a_bit(N, K, fp8):
    """The o / down projections of the decode step run a split-K GEMV (no fused norm, N <= 8192); `gemv_splitk_r` / `gemv_splitk_u`
    pick rows per workgroup and 16-byte chunks per thread and step.  A chunk's (wave, lane) and every lane's order of accumulation
    do not depend on either, so every form must give the bits of every other -- including rows shorter than a step (masked lanes),
    ragged row counts and K that is not a multiple of the step."""
    from teochat_amd.engine import quantize_fp8_rows
    bf = torch.bfloat16
    lib = G.lib()
    W = G.bf16_round(rnd(N, K, seed=2, scale=0.02 if K > 256 else 0.1))
    x, res = G.dev(G.bf16_round(rnd(K, seed=1)), bf), G.dev(G.bf16_round(rnd(N, seed=3)), bf)
    q, s, dq = quantize_fp8_rows(W.to(bf))
    q_d, s_d, dW = q.cuda(), s.cuda(), G.dev(W, bf)
    outs = {}
    try:
        for r in (0, 2, 4):
            for u in (0, 1, 2, 3, 4, 6):
                assert L.tune_set(b"gemv_splitk_r", r) == 0 and L.tune_set(b"gemv_splitk_u", u) == 0
                y = torch.empty(N, dtype=torch.float32, device="cuda")
                if fp8:
                    L.check(lib.teo_gemv_w8(G.p(x), G.p(q_d), G.p(s_d), None, G.p(res), G.p(y), N, K, 1e-5, 0, L.TEO_F32, G.stream()), "gemv_w8")
                else:
                    y = G.gemv(x, dW, res=res, out_dtype=torch.float32)
                outs[(r, u)] = y.cpu()
    finally:
        L.tune_set(b"gemv_splitk_r", 0)
        L.tune_set(b"gemv_splitk_u", 0)
    base = outs[(2, 2)]
    for k_, v in outs.items():
        assert torch.equal(v, base), k_
    ref = ((dq.float() if fp8 else W).double() @ x.cpu().double() + res.cpu().double()).float()
    torch.testing.assert_close(base, ref, atol=3e-4, rtol=2e-4)


# ---------------------------------------------------------------------------------------------- batched-decode GEMM
@pytest.mark.parametrize("MB", [1, 3, 8, 16])
@pytest.mark.parametrize("N,K", [(64, 64), (300, 128), (4096, 4096), (12288, 4096), (4096, 11008), (32000, 4096)])
def test_gemm_skinny_bf16(MB, N, K):
    bf = torch.bfloat16
    x = G.bf16_round(rnd(MB, K, seed=1))
    W = G.bf16_round(rnd(N, K, seed=2, scale=0.02 if K > 256 else 0.1))
    res = G.bf16_round(rnd(MB, N, seed=3))
    dx, dW, dr = G.dev(x, bf), G.dev(W, bf), G.dev(res, bf)
    from teochat_amd.engine import tile_weights
    y32 = G.gemm_skinny(dx, dW, out_dtype=torch.float32).cpu()
    torch.testing.assert_close(y32, x @ W.T, atol=2e-4, rtol=1e-4)
    y = G.gemm_skinny(dx, dW, res=dr)
    close_bf16(y, G.bf16_round(x @ W.T + res))
    # the operand-tiled weight layout gives bit-identical results (same products, same summation order)
    dWt = tile_weights(dW)
    assert torch.equal(G.gemm_skinny(dx, dWt, out_dtype=torch.float32, flags=L.GEMM_WTILED, N=N).cpu(), y32)
    assert torch.equal(G.gemm_skinny(dx, dWt, res=dr, flags=L.GEMM_WTILED, N=N), y)
    # every conversation's row equals the single-conversation GEMV on that row (fp32 accumulation order aside)
    yv = G.gemv(dx[MB - 1].contiguous(), dW, out_dtype=torch.float32).cpu()
    torch.testing.assert_close(y32[MB - 1], yv, atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("tiles", [0, 2, 4, 8])
@pytest.mark.parametrize("MB", [2, 8, 16])
def test_gemm_skinny_swiglu_and_tilings(MB, tiles):
    from teochat_amd.engine import interleave_gate_up
    bf = torch.bfloat16
    K, Fd = 4096, 11008
    x = G.bf16_round(rnd(MB, K, seed=1))
    gate, up = G.bf16_round(rnd(Fd, K, seed=2, scale=0.02)), G.bf16_round(rnd(Fd, K, seed=3, scale=0.02))
    gu = G.dev(interleave_gate_up(gate, up), bf)
    dx = G.dev(x, bf)
    assert L.tune_set(b"skinny_tiles", tiles) == 0
    try:
        from teochat_amd.engine import tile_weights
        y = G.gemm_skinny(dx, gu, flags=L.GEMM_SWIGLU16)
        assert torch.equal(G.gemm_skinny(dx, tile_weights(gu), flags=L.GEMM_SWIGLU16 | L.GEMM_WTILED), y)
        yp = G.gemm_skinny(dx, gu, out_dtype=torch.float32).cpu()           # plain product on the interleaved rows
    finally:
        L.tune_set(b"skinny_tiles", 0)
    close_bf16(y, G.bf16_round(F.silu(x @ gate.T) * (x @ up.T)))
    torch.testing.assert_close(yp, x @ interleave_gate_up(gate, up).T, atol=2e-4, rtol=1e-4)
    yg = G.gemm(dx, gu, flags=L.GEMM_SWIGLU16)                              # the prefill GEMM on the same rows
    close_bf16(y, yg.float(), ulps=2.0)


@pytest.mark.parametrize("MB", [1, 8, 16])
@pytest.mark.parametrize("N,K,swiglu", [(4096, 4096, False), (22016, 4096, True), (4096, 11008, False), (320, 128, False)])
def test_gemm_skinny_fp8_weights(MB, N, K, swiglu):
    from teochat_amd.engine import quantize_fp8_rows
    bf = torch.bfloat16
    W = G.bf16_round(rnd(N, K, seed=2, scale=0.02))
    q, s, dq = quantize_fp8_rows(W.to(bf))
    x = G.bf16_round(rnd(MB, K, seed=1))
    flags = L.GEMM_SWIGLU16 if swiglu else 0
    dx, q_d, s_d, dq_d = G.dev(x, bf), q.cuda(), s.cuda(), dq.cuda()
    from teochat_amd.engine import tile_weights
    y8 = G.gemm_skinny(dx, q_d, scale=s_d, flags=flags, out_dtype=torch.float32)
    y16 = G.gemm_skinny(dx, dq_d, flags=flags, out_dtype=torch.float32)
    torch.testing.assert_close(y8, y16, atol=2e-4, rtol=1e-4)
    y8t = G.gemm_skinny(dx, tile_weights(q_d), scale=s_d, flags=flags | L.GEMM_WTILED, out_dtype=torch.float32, N=N)
    assert torch.equal(y8t, y8)
    ref = x @ dq.float().T
    if swiglu:
        idx = torch.arange(N // 2)
        g_rows = (idx // 16) * 32 + idx % 16
        ref = F.silu(ref[:, g_rows]) * ref[:, g_rows + 16]
    torch.testing.assert_close(y8.cpu(), ref, atol=3e-4, rtol=2e-4)


@pytest.mark.parametrize("MB", [1, 8, 16])
@pytest.mark.parametrize("N,K,fp8", [(4096, 4096, True), (4096, 11008, True), (4096, 4096, False), (4096, 11008, False), (320, 128, False), (64, 2048, True)])
def test_gemm_skinny_sixteen_waves(MB, N, K, fp8):
    """Round 6: the tile kernel with 16 waves per workgroup (K split 16 ways, `skinny_waves` = 16) computes the same products with
    another fp32 partition of K: fp32 output within the accumulation-order tolerance of the 8-wave kernel and of the fp64 product,
    residual epilogue and the norm hand-off (xg_out / ssq partials are produced by the same code) unchanged in form."""
    from teochat_amd.engine import quantize_fp8_rows, tile_weights
    bf = torch.bfloat16
    W = G.bf16_round(rnd(N, K, seed=2, scale=0.02))
    x = G.bf16_round(rnd(MB, K, seed=1))
    res = G.bf16_round(rnd(MB, N, seed=3))
    dx, dr = G.dev(x, bf), G.dev(res, bf)
    scale = None
    if fp8:
        q, s_, dq = quantize_fp8_rows(W.to(bf))
        W, dW, scale = dq.float(), tile_weights(q.cuda()), s_.cuda()
    else:
        dW = tile_weights(G.dev(W, bf))
    y8 = G.gemm_skinny(dx, dW, scale=scale, flags=L.GEMM_WTILED, out_dtype=torch.float32, N=N)
    assert G.lib().teo_last_kernel().decode() in ("skinny_gemm", "skinny_gemm_u8")
    assert L.tune_set(b"skinny_waves", 16) == 0
    try:
        y16 = G.gemm_skinny(dx, dW, scale=scale, flags=L.GEMM_WTILED, out_dtype=torch.float32, N=N)
        took16 = G.lib().teo_last_kernel().decode() == "skinny_gemm_w16"
        yr = G.gemm_skinny(dx, dW, scale=scale, res=dr, flags=L.GEMM_WTILED, N=N)
    finally:
        L.tune_set(b"skinny_waves", 0)
    assert took16 == (K // (64 if fp8 else 32) // 16 >= 2)
    ref = (x.double() @ W.double().T).float()
    torch.testing.assert_close(y16.cpu(), ref, atol=3e-4, rtol=2e-4)
    torch.testing.assert_close(y16, y8, atol=3e-4, rtol=2e-4)
    close_bf16(yr, G.bf16_round(ref + res))


def test_gemm_skinny_rejects_unsupported():
    x = torch.zeros(2, 48, dtype=torch.bfloat16, device="cuda")
    W = torch.zeros(16, 48, dtype=torch.bfloat16, device="cuda")
    out = torch.zeros(2, 16, dtype=torch.bfloat16, device="cuda")
    rc = G.lib().teo_gemm_skinny(G.p(x), G.p(W), None, 0, None, 0.0, None, G.p(out), 2, 16, 48, 48, 16, 0, L.TEO_BF16, G.stream())
    assert rc == -2 and b"skinny" in G.lib().teo_last_error()
    rc = G.lib().teo_gemm_skinny(G.p(x), G.p(W), None, 0, None, 0.0, None, G.p(out), 17, 16, 64, 64, 16, 0, L.TEO_BF16, G.stream())
    assert rc == -2


@pytest.mark.parametrize("MB", [1, 8, 16])
@pytest.mark.parametrize("N,K,swiglu,fp8", [(12288, 4096, False, False), (22016, 4096, True, False), (22016, 4096, True, True),
                                            (32000, 4096, False, True), (320, 128, False, False)])
def test_gemm_skinny_fused_rmsnorm(MB, N, K, swiglu, fp8):
    """norm_w: rsqrt(mean x^2 + eps) * (W . bf16(x*g)) against the fp64-accumulated definition, and against the
    unfused composition rmsnorm kernel -> skinny GEMM (differs only by where the bf16 rounding of x_normed sits)."""
    from teochat_amd.engine import quantize_fp8_rows, tile_weights
    bf = torch.bfloat16
    x = G.bf16_round(rnd(MB, K, seed=1) * 3.0)
    g = G.bf16_round(1 + 0.1 * rnd(K, seed=4))
    W = G.bf16_round(rnd(N, K, seed=2, scale=0.02))
    flags = L.GEMM_SWIGLU16 if swiglu else 0
    dx, dg = G.dev(x, bf), G.dev(g, bf)
    scale = None
    if fp8:
        q, s_, dq = quantize_fp8_rows(W.to(bf))
        W = dq.float()
        dW, scale = tile_weights(q.cuda()), s_.cuda()
    else:
        dW = tile_weights(G.dev(W, bf))
    y = G.gemm_skinny(dx, dW, scale=scale, flags=flags | L.GEMM_WTILED, out_dtype=torch.float32, N=N, norm_w=dg, eps=1e-5).cpu()
    inv = torch.rsqrt((x.double() ** 2).mean(-1, keepdim=True) + 1e-5)
    ref = ((G.bf16_round(x * g).double() @ W.double().T) * inv).float()
    if swiglu:
        idx = torch.arange(N // 2)
        g_rows = (idx // 16) * 32 + idx % 16
        ref = F.silu(ref[:, g_rows]) * ref[:, g_rows + 16]
    torch.testing.assert_close(y, ref, atol=3e-4, rtol=2e-4)
    # unfused composition: one more bf16 rounding of the normalised activations -> bf16-level agreement
    xn = torch.empty_like(dx)
    L.check(G.lib().teo_rmsnorm(G.p(dx), G.p(dg), G.p(xn), MB, K, 1e-5, L.TEO_BF16, G.stream()), "rmsnorm")
    y2 = G.gemm_skinny(xn, dW, scale=scale, flags=flags | L.GEMM_WTILED, out_dtype=torch.float32, N=N).cpu()
    assert float((y - y2).abs().max()) <= 2e-2 * float(y2.abs().max())


# ---------------------------------------------------------------------------------------------- preprocessing (N3)
@pytest.mark.parametrize("H,W", [(224, 224), (256, 320), (320, 256), (1024, 1024), (100, 150), (231, 517), (2000, 1500)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_preprocess_frames_matches_torchvision_semantics(H, W, dtype):
    """teo_preprocess_frames == ToTensor -> Resize(224, bicubic, antialias) -> CenterCrop(224) -> Normalize on the CPU
    (oracle.preprocess_image restates processing_image.py:15-25; it is pinned in tests/test_oracle_golden.py by analytic
    cases and a PIL cross-check)."""
    import ctypes as C
    from oracle import teo_oracle as O
    from teochat_amd.processor import OPENAI_DATASET_MEAN, OPENAI_DATASET_STD
    g = torch.Generator().manual_seed(H * 7 + W)
    T = 3
    raw = torch.randint(0, 256, (T, H, W, 3), generator=g, dtype=torch.uint8)
    # smooth content on top of the noise so that interpolation errors would show
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    raw[1] = ((torch.sin(yy / 9.0) * torch.cos(xx / 13.0) * 0.5 + 0.5) * 255).to(torch.uint8)[..., None].expand(H, W, 3)
    want = torch.stack([O.preprocess_image(raw[t]) for t in range(T)])
    src = raw.cuda()
    out = torch.empty(T, 3, 224, 224, dtype=dtype, device="cuda")
    mean, std = (C.c_float * 3)(*OPENAI_DATASET_MEAN), (C.c_float * 3)(*OPENAI_DATASET_STD)
    L.check(G.lib().teo_preprocess_frames(G.p(src), G.p(out), T, H, W, 224, mean, std, G.DT[dtype], G.stream()), "preprocess")
    if dtype == torch.float32:
        torch.testing.assert_close(out.cpu(), want, atol=3e-5, rtol=0)
        if (H, W) == (224, 224):
            exact = (raw.permute(0, 3, 1, 2).float() / 255.0 - torch.tensor(OPENAI_DATASET_MEAN).view(1, 3, 1, 1)) \
                / torch.tensor(OPENAI_DATASET_STD).view(1, 3, 1, 1)
            torch.testing.assert_close(out.cpu(), exact, atol=2e-6, rtol=0)
    else:
        close_bf16(out, G.bf16_round(want), ulps=1.0)


@pytest.mark.parametrize("H,W", [(224, 224), (300, 448), (448, 300), (101, 224), (1000, 37)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_preprocess_frames_pad_mode_matches_expand2square(H, W, dtype):
    """teo_preprocess_frames_pad == preprocess(expand2square(img, fill)) (mm_utils.py:14-36, image_aspect_ratio == 'pad'),
    the padded canvas being virtual on the device."""
    import ctypes as C
    from oracle import teo_oracle as O
    g = torch.Generator().manual_seed(H + 3 * W)
    T = 2
    raw = torch.randint(0, 256, (T, H, W, 3), generator=g, dtype=torch.uint8)
    fill = O.pad_fill_from_mean(O.OPENAI_DATASET_MEAN)
    want = torch.stack([O.preprocess_image(O.expand2square_u8(raw[t], fill)) for t in range(T)])
    out = torch.empty(T, 3, 224, 224, dtype=dtype, device="cuda")
    mean, std = (C.c_float * 3)(*O.OPENAI_DATASET_MEAN), (C.c_float * 3)(*O.OPENAI_DATASET_STD)
    d_raw = raw.cuda()
    L.check(G.lib().teo_preprocess_frames_pad(G.p(d_raw), G.p(out), T, H, W, 224, mean, std, (C.c_ubyte * 3)(*fill), G.DT[dtype],
                                              G.stream()), "preprocess_pad")
    if dtype == torch.float32:
        torch.testing.assert_close(out.cpu(), want, atol=3e-5, rtol=0)
    else:
        close_bf16(out, G.bf16_round(want), ulps=1.0)


def test_preprocess_frames_rejects_huge_downscale():
    import ctypes as C
    src = torch.zeros(1, 5000, 5000, 3, dtype=torch.uint8, device="cuda")
    out = torch.empty(1, 3, 224, 224, device="cuda")
    m = (C.c_float * 3)(0, 0, 0)
    s_ = (C.c_float * 3)(1, 1, 1)
    rc = G.lib().teo_preprocess_frames(G.p(src), G.p(out), 1, 5000, 5000, 224, m, s_, L.TEO_F32, G.stream())
    assert rc == -2 and b"taps" in G.lib().teo_last_error()


# ---------------------------------------------------------------------------------------------- loss (N4)
@pytest.mark.parametrize("rows,vocab", [(1, 300), (37, 512), (64, 32000)])
def test_cross_entropy_matches_torch(rows, vocab):
    lg = rnd(rows, vocab, seed=3, scale=4.0)
    g = torch.Generator().manual_seed(9)
    lab = torch.randint(0, vocab, (rows,), generator=g)
    lab[::5] = -100
    dl, dlab = lg.cuda(), lab.cuda()
    per = torch.empty(rows, device="cuda")
    out = torch.empty(3, device="cuda")
    L.check(G.lib().teo_cross_entropy(G.p(dl), vocab, G.p(dlab), G.p(per), G.p(out), rows, vocab, -100, G.stream()), "ce")
    want_per = F.cross_entropy(lg.double(), lab, ignore_index=-100, reduction="none")
    torch.testing.assert_close(per.cpu().double(), want_per, atol=2e-5, rtol=1e-6)
    n = int((lab != -100).sum())
    if n:
        assert abs(float(out[0]) - float(want_per.sum() / n)) < 2e-5 and int(out[2]) == n
    else:
        assert bool(torch.isnan(out[0]))


@pytest.mark.parametrize("MB", [3, 8, 16])
@pytest.mark.parametrize("fp8", [False, True])
def test_gemm_skinny_swiglu_block8_layout(MB, fp8):
    """TEO_GEMM_SWIGLU8: gate/up pairs interleaved in blocks of 8 rows (one 16-row tile holds 8 pairs) == SWIGLU16."""
    from teochat_amd.engine import interleave_gate_up, quantize_fp8_rows, reinterleave_gate_up, tile_weights
    bf = torch.bfloat16
    K, Fd = 4096, 11008
    x = G.bf16_round(rnd(MB, K, seed=1))
    gate, up = G.bf16_round(rnd(Fd, K, seed=2, scale=0.02)), G.bf16_round(rnd(Fd, K, seed=3, scale=0.02))
    gu16 = interleave_gate_up(gate, up).to(bf).cuda()
    dx = G.dev(x, bf)
    if fp8:
        q, s_, dq = quantize_fp8_rows(gu16)
        y16 = G.gemm_skinny(dx, tile_weights(q), scale=s_, flags=L.GEMM_SWIGLU16 | L.GEMM_WTILED, out_dtype=torch.float32, N=2 * Fd)
        q8 = reinterleave_gate_up(q, 8)
        s8 = reinterleave_gate_up(s_.view(-1, 1), 8).view(-1).contiguous()
        y8 = G.gemm_skinny(dx, tile_weights(q8), scale=s8, flags=L.GEMM_SWIGLU8 | L.GEMM_WTILED, out_dtype=torch.float32, N=2 * Fd)
    else:
        y16 = G.gemm_skinny(dx, tile_weights(gu16), flags=L.GEMM_SWIGLU16 | L.GEMM_WTILED, out_dtype=torch.float32, N=2 * Fd)
        gu8 = reinterleave_gate_up(gu16, 8)
        assert torch.equal(gu8.view(Fd // 8, 2, 8, K)[:, 0].reshape(Fd, K).cpu().float(), gate)
        y8 = G.gemm_skinny(dx, tile_weights(gu8), flags=L.GEMM_SWIGLU8 | L.GEMM_WTILED, out_dtype=torch.float32, N=2 * Fd)
        torch.testing.assert_close(y8.cpu(), F.silu(x @ gate.T) * (x @ up.T), atol=3e-4, rtol=2e-4)
    torch.testing.assert_close(y8, y16, atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("MB", [1, 5, 8, 16])
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("N,K", [(64, 64), (300, 128), (1040, 2048), (12288, 4096), (32000, 4096)])
def test_gemm_skinny_stream_form(MB, fp8, N, K):
    """The persistent streaming form (activations in registers, weight ring across tiles, epilogue by all waves) against the
    one-tile-per-workgroup kernel: bit-identical where the K partition is the same (K / KS a multiple of 8 slices of the template's
    size: K = 4096), fp32-summation-order close elsewhere; plain + residual, f32 and bf16 outputs, row-major and tiled.  A ragged
    last tile (N % 16 != 0) is not eligible: the tile kernel runs."""
    from teochat_amd.engine import quantize_fp8_rows, tile_weights
    bf = torch.bfloat16
    x = G.bf16_round(rnd(MB, K, seed=1))
    W = G.bf16_round(rnd(N, K, seed=2, scale=0.02 if K > 256 else 0.1))
    dx, dr = G.dev(x, bf), G.dev(G.bf16_round(rnd(MB, N, seed=3)), bf)
    dW, scale = G.dev(W, bf), None
    if fp8:
        dW, scale, _ = quantize_fp8_rows(dW)
    same = K == 4096
    outs = {}
    try:
        for mode in (0, 2):
            assert L.tune_set(b"skinny_stream", mode) == 0
            outs[mode] = (G.gemm_skinny(dx, dW, scale=scale, out_dtype=torch.float32).cpu(),
                          G.gemm_skinny(dx, dW, scale=scale, res=dr).cpu(),
                          G.gemm_skinny(dx, tile_weights(dW), scale=scale, res=dr, flags=L.GEMM_WTILED, N=N).cpu())
            assert G.lib().teo_last_kernel().startswith(b"skinny_stream" if (mode and N % 16 == 0) else b"skinny_gemm")
    finally:
        L.tune_set(b"skinny_stream", 1)
    a, b = outs[0], outs[2]
    assert torch.equal(b[1], b[2])                        # layouts agree within the streaming form
    if same:
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    else:
        torch.testing.assert_close(b[0], a[0], atol=2e-4, rtol=1e-4)
        close_bf16(b[1], a[1].float())


@pytest.mark.parametrize("MB", [3, 8, 16])
@pytest.mark.parametrize("fp8", [False, True])
def test_gemm_skinny_stream_swiglu8(MB, fp8):
    """TEO_GEMM_SWIGLU8 through the streaming form == through the tile kernel (7B gate/up shape), bf16 and f32 outputs."""
    from teochat_amd.engine import interleave_gate_up, quantize_fp8_rows, reinterleave_gate_up, tile_weights
    bf = torch.bfloat16
    K, Fd = 4096, 11008
    x = G.bf16_round(rnd(MB, K, seed=1))
    gate, up = G.bf16_round(rnd(Fd, K, seed=2, scale=0.02)), G.bf16_round(rnd(Fd, K, seed=3, scale=0.02))
    gu8 = reinterleave_gate_up(interleave_gate_up(gate, up).to(bf).cuda(), 8)
    dx, scale = G.dev(x, bf), None
    if fp8:
        gu8, scale, _ = quantize_fp8_rows(gu8)
    wt = tile_weights(gu8)
    outs = {}
    try:
        for mode in (0, 2):
            assert L.tune_set(b"skinny_stream", mode) == 0
            outs[mode] = (G.gemm_skinny(dx, wt, scale=scale, flags=L.GEMM_SWIGLU8 | L.GEMM_WTILED, out_dtype=torch.float32, N=2 * Fd).cpu(),
                          G.gemm_skinny(dx, wt, scale=scale, flags=L.GEMM_SWIGLU8 | L.GEMM_WTILED, N=2 * Fd).cpu())
    finally:
        L.tune_set(b"skinny_stream", 1)
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])
    if not fp8:
        torch.testing.assert_close(outs[2][0], F.silu(x @ gate.T) * (x @ up.T), atol=3e-4, rtol=2e-4)


# ---------------------------------------------------------------------------------------------- decode attention
def _decode_attn_ref(q, K, V, n):
    """q [H, d]; K, V [Hk, S, d]; n keys -> [H*d] (fp64 softmax(q K^T / sqrt(d)) V, GQA by head // (H/Hk))."""
    H, d = q.shape
    rep = H // K.shape[0]
    out = []
    for h in range(H):
        k, v = K[h // rep, :n].double(), V[h // rep, :n].double()
        p = torch.softmax((k @ q[h].double()) / d ** 0.5, dim=0)
        out.append(p @ v)
    return torch.cat(out).float()


@pytest.mark.parametrize("dtype,H,Hk,d,S,ctx", [(torch.bfloat16, 32, 32, 128, 2560, [2299, 64, 1, 2559]),
                                                (torch.bfloat16, 8, 2, 64, 512, [300, 511]),
                                                (torch.float32, 4, 2, 16, 256, [17, 200, 255])])
@pytest.mark.parametrize("chunk", [0, 32, 128, 256])
def test_attn_decode_batched_vs_reference(dtype, H, Hk, d, S, ctx, chunk):
    """teo_attn_decode (pre-rotated q, caches already hold every key): every conversation of the batch against an fp64
    softmax reference, for every KV-split size."""
    B = len(ctx)
    g = torch.Generator().manual_seed(11)
    q = torch.randn(B, H, d, generator=g)
    K = torch.randn(B, Hk, S, d, generator=g)
    V = torch.randn(B, Hk, S, d, generator=g)
    if dtype == torch.bfloat16:
        q, K, V = G.bf16_round(q), G.bf16_round(K), G.bf16_round(V)
    dq, dK, dV = q.to("cuda", dtype).contiguous(), K.to("cuda", dtype).contiguous(), V.to("cuda", dtype).contiguous()
    pos = torch.tensor([n - 1 for n in ctx], dtype=torch.int32, device="cuda")
    out = torch.empty(B, H * d, dtype=dtype, device="cuda")
    lib = G.lib()
    part = torch.empty(lib.teo_attn_decode_workspace_bytes(H, d, S, B), dtype=torch.uint8, device="cuda")
    assert L.tune_set(b"attn_chunk", chunk) == 0
    try:
        L.check(lib.teo_attn_decode(G.p(dq), G.p(dK), G.p(dV), None, None, None, G.p(out), G.p(part), G.p(pos), S, H, Hk, d,
                                    1.0 / d ** 0.5, G.DT[dtype], B, H * d, Hk * S * d, H * d, G.stream()), "attn_decode")
    finally:
        L.tune_set(b"attn_chunk", 0)
    for b, n in enumerate(ctx):
        ref = _decode_attn_ref(q[b], K[b], V[b], n)
        if dtype == torch.float32:
            torch.testing.assert_close(out[b].cpu(), ref, atol=2e-5, rtol=1e-5)
        else:
            close_bf16(out[b], G.bf16_round(ref), ulps=2.0, floor=4e-3)    # P is rounded to bf16 before the PV product


@pytest.mark.parametrize("dtype,H,Hk,d,S,ctx", [(torch.bfloat16, 32, 32, 128, 2560, [2299, 64, 1, 2560, 129, 2305, 640, 2048]),
                                                (torch.bfloat16, 32, 32, 128, 4608, [4255, 4608, 100]),
                                                (torch.bfloat16, 8, 2, 64, 512, [300, 512, 5]),
                                                (torch.float32, 4, 2, 32, 512, [17, 200, 512])])
@pytest.mark.parametrize("chunk", [0, 32, 64, 128])
@pytest.mark.parametrize("rope", [False, True])
def test_attn_decode_whole_context_is_bitwise_the_split_pair(dtype, H, Hk, d, S, ctx, chunk, rope):
    """attn_decode_whole_kernel (one workgroup per (conversation, head) walks the whole context; the split records stay in LDS)
    against attn_decode_partial + attn_decode_combine at the SAME chunk size: same chunk records, same merge code -> BIT-identical
    outputs, with and without the in-kernel RoPE + KV append (appended rows compared too), ragged / full / one-key contexts;
    and against the fp64 softmax reference."""
    from teochat_amd.engine import rope_tables
    B = len(ctx)
    lib = G.lib()
    g = torch.Generator().manual_seed(31 + chunk + int(rope))
    K = torch.randn(B, Hk, S, d, generator=g)
    V = torch.randn(B, Hk, S, d, generator=g)
    qkv = torch.randn(B, H + 2 * Hk, d, generator=g)                           # raw q | k | v rows of the new tokens
    if dtype == torch.bfloat16:
        K, V, qkv = G.bf16_round(K), G.bf16_round(V), G.bf16_round(qkv)
    cs, sn = rope_tables(d, 10000.0, S)
    d_cs, d_sn = cs.cuda(), sn.cuda()
    pos = torch.tensor([n - 1 for n in ctx], dtype=torch.int32, device="cuda")
    cw = chunk if chunk else 64                                                 # the whole-context kernel's default chunk
    rpi = 64 // (d * (4 if dtype == torch.float32 else 2) // 16)
    if cw // 4 < rpi:
        cw = 4 * rpi
    if cw // 4 // rpi > 8:
        pytest.skip("quarter-chunk above 8 load instructions: the whole-context form does not take this shape")
    outs, caches = {}, {}
    part = torch.empty(lib.teo_attn_decode_workspace_bytes(H, d, S, B), dtype=torch.uint8, device="cuda")
    for whole in (2, 0):
        dK, dV = K.to("cuda", dtype).contiguous(), V.to("cuda", dtype).contiguous()
        dVT = torch.zeros(B, Hk, d, S, dtype=dtype, device="cuda")
        out = torch.zeros(B, H * d, dtype=dtype, device="cuda")
        if rope:
            dq = qkv.reshape(B, -1).to("cuda", dtype).contiguous()
            qs = (H + 2 * Hk) * d
        else:
            dq = qkv[:, :H].reshape(B, -1).to("cuda", dtype).contiguous()
            qs = H * d
        assert L.tune_set(b"attn_whole", whole) == 0 and L.tune_set(b"attn_chunk", cw) == 0
        L.check(lib.teo_attn_decode(G.p(dq), G.p(dK), G.p(dV), G.p(dVT) if rope else None, G.p(d_cs) if rope else None,
                                    G.p(d_sn) if rope else None, G.p(out), G.p(part), G.p(pos), S, H, Hk, d, 1.0 / d ** 0.5,
                                    G.DT[dtype], B, qs, Hk * S * d, H * d, G.stream()), "attn_decode")
        torch.cuda.synchronize()
        if whole == 2:
            assert lib.teo_last_kernel() == b"attn_decode_whole"
        outs[whole] = out.clone()
        caches[whole] = (dK, dV, dVT)
    assert torch.equal(outs[2], outs[0])
    for a, b_ in zip(caches[2], caches[0]):
        assert torch.equal(a, b_)
    for b, n in enumerate(ctx):
        if rope:
            c, s_ = cs[n - 1], sn[n - 1]
            c, s_ = torch.cat([c, c]), torch.cat([s_, s_])
            rnd_ = G.bf16_round if dtype == torch.bfloat16 else (lambda t: t)
            q_rot = rnd_(qkv[b, :H] * c + O.rotate_half(qkv[b, :H]) * s_)
            Kb, Vb = K[b].clone(), V[b].clone()
            Kb[:, n - 1] = rnd_(qkv[b, H:H + Hk] * c + O.rotate_half(qkv[b, H:H + Hk]) * s_)
            Vb[:, n - 1] = qkv[b, H + Hk:]
        else:
            q_rot, Kb, Vb = qkv[b, :H], K[b], V[b]
        ref = _decode_attn_ref(q_rot, Kb, Vb, n)
        if dtype == torch.float32:
            torch.testing.assert_close(outs[2][b].cpu(), ref, atol=3e-5, rtol=2e-5)
        else:
            close_bf16(outs[2][b], G.bf16_round(ref), ulps=2.0, floor=4e-3)


def _hf_top_p_keep(logits, temperature, top_k, top_p):
    """HF order: temperature -> TopKLogitsWarper -> TopPLogitsWarper (ascending sort, cumulative <= 1 - top_p removed,
    at least one token kept).  Returns (kept index set, renormalised probabilities)."""
    x = logits.double() / temperature
    if top_k and top_k < x.numel():
        kth = torch.topk(x, top_k).values[-1]
        x = torch.where(x < kth, torch.full_like(x, -float("inf")), x)
    srt, idx = torch.sort(x, descending=False)
    cum = torch.softmax(srt, dim=-1).cumsum(-1)
    remove = cum <= (1 - top_p)
    remove[-1:] = False
    x[idx[remove]] = -float("inf")
    p = torch.softmax(x, dim=-1)
    return set(torch.nonzero(p > 0).flatten().tolist()), p


@pytest.mark.parametrize("top_k,top_p", [(50, 0.9), (8, 0.5), (50, 0.999), (50, 0.05)])
def test_sampler_top_p_matches_hf_warper(top_k, top_p):
    g = torch.Generator().manual_seed(21)
    lg = torch.randn(1000, generator=g) * 2.5
    keep, p = _hf_top_p_keep(lg, 0.8, top_k, top_p)
    dl = lg.cuda()
    draws = [_sample(dl, 0.8, top_k, 77, d, top_p) for d in range(1500)]
    assert set(draws) <= keep, sorted(set(draws) - keep)
    counts = torch.bincount(torch.tensor(draws), minlength=1000).double() / len(draws)
    assert float((counts - p).abs().max()) < 0.05
    if len(keep) == 1:
        assert set(draws) == keep


def test_sampler_top_p_without_top_k_small_vocab():
    g = torch.Generator().manual_seed(22)
    lg = torch.randn(1000, generator=g) * 3.0
    keep, p = _hf_top_p_keep(lg, 1.0, 0, 0.6)
    dl = lg.cuda()
    draws = [_sample(dl, 1.0, 0, 5, d, 0.6) for d in range(1500)]
    assert set(draws) <= keep
    counts = torch.bincount(torch.tensor(draws), minlength=1000).double() / len(draws)
    assert float((counts - p).abs().max()) < 0.05


def test_sampler_full_vocabulary_when_top_k_is_disabled():
    """HF: top_k = 0 (or >= vocab) means TopKLogitsWarper is not applied -> multinomial over the WHOLE vocabulary, not over
    the device sampler's 1024-candidate cap.  Mass outside the top 1024 must be reachable and the frequencies
    must follow softmax(logits / T)."""
    V, n, temp = 5000, 3000, 1.0
    lg = torch.zeros(V)
    lg[:40] = 2.0                      # 40 likely tokens ...
    p = torch.softmax(lg.double() / temp, dim=0)
    tail_mass = float(p[40:].sum())    # ... and 4960 tokens that together hold ~94 % of the mass
    assert tail_mass > 0.9
    dl = lg.cuda()
    for top_k in (0, V, V + 5):
        draws = torch.tensor([_sample(dl, temp, top_k, 9, d) for d in range(n)])
        assert int(draws.min()) >= 0 and int(draws.max()) < V
        frac_tail = float((draws >= 40).double().mean())
        assert abs(frac_tail - tail_mass) < 0.03, (top_k, frac_tail, tail_mass)
        assert int((draws >= 1024).sum()) > 0.5 * n           # indices far beyond any 1024-candidate window
        assert len(set(draws.tolist())) > 1500
    # index-ordered inverse CDF: a one-hot distribution returns its index for every uniform
    lg2 = torch.full((40000,), -50.0)
    lg2[31999] = 50.0
    assert all(_sample(lg2.cuda(), 1.0, 0, s, 0) == 31999 for s in range(5))
    # determinism: (seed, draw) fixes the token
    assert [_sample(dl, temp, 0, 4, d) for d in range(32)] == [_sample(dl, temp, 0, 4, d) for d in range(32)]


def test_sampler_rejects_what_it_cannot_do_exactly():
    lg = torch.randn(32000).cuda()
    tok = torch.empty(1, dtype=torch.int64, device="cuda")
    # top_k above the candidate cap but not "off"
    assert G.lib().teo_sample_topk(G.p(lg), G.p(tok), 32000, 1.0, 2000, 1.0, 1, 0, G.stream()) == -2
    assert b"top_k" in G.lib().teo_last_error()
    # nucleus filter over the whole vocabulary
    assert G.lib().teo_sample_topk(G.p(lg), G.p(tok), 32000, 1.0, 0, 0.9, 1, 0, G.stream()) == -2
    assert b"top_p" in G.lib().teo_last_error()
    assert G.lib().teo_sample_topk(G.p(lg), G.p(tok), 32000, 1.0, 50, 0.9, 1, 0, G.stream()) == 0


# ---------------------------------------------------------------------------------------------- stream-K GEMM
def _gemm_ws(A, W, ws, bias=None, res=None, act=L.ACT_NONE, flags=0, out_dtype=None):
    M, K = A.shape
    N = W.shape[0]
    out_dtype = out_dtype or A.dtype
    Nc = N // 2 if flags & L.GEMM_SWIGLU16 else N
    Cc = torch.empty(M, Nc, dtype=out_dtype, device=A.device)
    L.check(G.lib().teo_gemm_ws(G.p(A), G.p(W), G.p(bias), G.p(res), G.p(Cc), M, N, K, A.stride(0), Nc, act, flags, G.DT[A.dtype],
                                G.DT[out_dtype], G.p(ws), G.stream()), "gemm_ws")
    return Cc


@pytest.mark.parametrize("wide", [0, 1])
@pytest.mark.parametrize("M,N,K,flags,extra", [(2168, 4096, 4096, 0, "res"), (2168, 4096, 11008, 0, "res"), (2168, 12288, 4096, 0, ""),
                                               (2168, 22016, 4096, L.GEMM_SWIGLU16, ""), (2056, 4096, 1024, 0, "bias_gelu"),
                                               (4208, 4096, 4096, 0, "f32out"), (1300, 8192, 512, 0, ""), (17000, 512, 64, 0, "bias")])
def test_gemm_stream_k_is_bitwise_the_plain_kernel(M, N, K, flags, extra, wide):
    """teo_gemm_ws (persistent stream-K grid: equal k-tile ranges per workgroup, partial tiles handed to the neighbour through
    fp32 slabs and continued in the same k-order) against teo_gemm (one workgroup per tile): BIT-identical outputs, launch after
    launch on the same workspace (slabs and flags are re-used: a stale or torn hand-off would show), also with a concurrent HBM
    stream on a second HIP stream."""
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g)).to(bf).cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).to(bf).cuda()
    Nc = N // 2 if flags else N
    bias = (torch.randn(N, generator=g) * 0.1).to(bf).cuda() if "bias" in extra else None
    res = (torch.randn(M, Nc, generator=g)).to(bf).cuda() if "res" in extra else None
    act = L.ACT_GELU_ERF if "gelu" in extra else L.ACT_NONE
    od = torch.float32 if "f32out" in extra else bf
    lib = G.lib()
    ws = torch.empty(lib.teo_gemm_workspace_bytes(), dtype=torch.uint8, device="cuda")
    L.check(lib.teo_gemm_workspace_init(G.p(ws), G.stream()), "ws init")
    assert L.tune_set(b"gemm_wide", 0) == 0
    want = G.gemm(A, W, bias=bias, res=res, act=act, flags=flags, out_dtype=od)      # the 128 x 128 kernel, one workgroup per tile
    side = torch.cuda.Stream()
    big = torch.empty(64 * 2 ** 20, dtype=torch.float32, device="cuda")
    big2 = torch.empty_like(big)
    bad = 0
    # wide = 0: stream-K grid of the 128 x 128 kernel; wide = 1: stream-K grid of the wide-tile LDS-DMA kernel (non-SwiGLU shapes
    # with more than 256 wide tiles; the others fall through to the 128 x 128 form).  Forced: the heuristic only picks ~1 round.
    assert L.tune_set(b"gemm_wide", wide) == 0
    assert L.tune_set(b"gemm_sk", 2) == 0
    assert L.tune_set(b"gemm_big", 0) == 0            # the 256 x 256 hybrid has its own test below
    try:
        for it in range(8):
            if it % 2:
                with torch.cuda.stream(side):
                    big2.copy_(big)
            got = _gemm_ws(A, W, ws, bias=bias, res=res, act=act, flags=flags, out_dtype=od)
            torch.cuda.synchronize()
            bad += int(not torch.equal(got, want))
    finally:
        L.tune_set(b"gemm_sk", 1)
        L.tune_set(b"gemm_wide", 1)
        L.tune_set(b"gemm_big", 1)
    assert bad == 0, f"{bad}/8 launches differ from the plain kernel"
    # without a workspace teo_gemm_ws is teo_gemm; tuning the stream-K path off gives the same bits too
    assert torch.equal(_gemm_ws(A, W, None, bias=bias, res=res, act=act, flags=flags, out_dtype=od), want)
    assert L.tune_set(b"gemm_sk", 0) == 0
    try:
        assert torch.equal(_gemm_ws(A, W, ws, bias=bias, res=res, act=act, flags=flags, out_dtype=od), want)
    finally:
        L.tune_set(b"gemm_sk", 1)


# ---------------------------------------------------------------------------------------------- w8a8 prefill GEMM (fp8 MFMA, X1)
def _quant_ref(x):
    """per-row e4m3 quantisation of a float tensor the way teo_quant_rows_fp8 defines it (scale = amax / 448, RNE)."""
    amax = x.abs().amax(dim=1)
    s = torch.where(amax > 0, amax * (1.0 / 448.0), torch.ones_like(amax))
    q = (x * (1.0 / s)[:, None]).to(torch.float8_e4m3fn)
    return q, s


def _gemm_fp8(A8, sa, W8, sw, res=None, flags=0, out_dtype=torch.bfloat16):
    M, K = A8.shape
    N = W8.shape[0]
    Nc = N // 2 if flags & L.GEMM_SWIGLU16 else N
    Cc = torch.empty(M, Nc, dtype=out_dtype, device="cuda")
    L.check(G.lib().teo_gemm_fp8(G.p(A8), G.p(sa), G.p(W8), G.p(sw), G.p(res), G.p(Cc), M, N, K, A8.stride(0), Nc, flags, G.DT[out_dtype],
                                 G.stream()), "gemm_fp8")
    return Cc


@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (200, 260, 256), (2168, 4096, 4096), (333, 4096, 11008), (2168, 512, 1152)])
def test_gemm_fp8_mfma_is_exact_against_dequantised_operands(M, N, K):
    """teo_gemm_fp8 (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales, row scales in the epilogue) against an fp64 product
    of the DEQUANTISED operands.  The products are exact, but the instruction adds its 128 terms in a reduced-precision
    internal tree (measured on MI355X: up to 1.1e-5 of the row's sum |a||w|, vs 1e-7 for an fp32 FMA chain): the bound is
    3e-5 of sum |a||w|.  Asymmetric operands (a transposed or permuted k-layout is off by O(1) and cannot pass)."""
    g = torch.Generator().manual_seed(M * 3 + N + K)
    A = torch.randn(M, K, generator=g) * torch.linspace(0.5, 2.0, K)[None, :]           # column structure: k order matters
    W = torch.randn(N, K, generator=g) * 0.05 * torch.linspace(2.0, 0.25, K)[None, :]
    qa, sa = _quant_ref(A)
    qw, sw = _quant_ref(W)
    A8, W8 = qa.view(torch.uint8).cuda(), qw.view(torch.uint8).cuda()
    dsa, dsw = sa.float().cuda(), sw.float().cuda()
    got = _gemm_fp8(A8, dsa, W8, dsw, out_dtype=torch.float32).cpu().double()
    Ad, Wd = qa.double() * sa.double()[:, None], qw.double() * sw.double()[:, None]
    want = Ad @ Wd.T
    bound = (Ad.abs() @ Wd.abs().T) * 3e-5 + 1e-12
    assert bool(((got - want).abs() <= bound).all()), float(((got - want).abs() / bound).max())
    # bf16 output + residual: one rounding of (product + residual)
    res = torch.randn(M, N, generator=g).to(torch.bfloat16)
    got16 = _gemm_fp8(A8, dsa, W8, dsw, res=res.cuda(), out_dtype=torch.bfloat16)
    close_bf16(got16, G.bf16_round((want + res.double()).float()), ulps=1.0)
    # the wide-tile (LDS-DMA) and the 128 x 128 kernels add the k tiles in the same order: bit-identical
    lib = G.lib()
    try:
        outs = []
        for mode in (0, 2):
            assert L.tune_set(b"gemm_fp8_wide", mode) == 0
            outs.append(_gemm_fp8(A8, dsa, W8, dsw, res=res.cuda(), out_dtype=torch.bfloat16))
        assert K < 256 or torch.equal(outs[0], outs[1])
    finally:
        L.tune_set(b"gemm_fp8_wide", 1)


@pytest.mark.parametrize("M,N,K", [(2168, 4096, 4096), (2168, 4096, 11008), (1100, 2304, 512), (333, 768, 1024)])
def test_gemm_fp8_stream_k_is_bit_identical(M, N, K):
    """teo_gemm_fp8_ws (persistent stream-K grid of the wide fp8 kernel, forced with gemm_fp8_wide = 3 where the shape would not
    pick it) against teo_gemm_fp8: same k order per output element => same bits; repeated launches share the slabs and flags."""
    g = torch.Generator().manual_seed(M + N + K)
    A8 = _quant_ref(torch.randn(M, K, generator=g))[0].view(torch.uint8).cuda()
    W8 = _quant_ref(torch.randn(N, K, generator=g) * 0.02)[0].view(torch.uint8).cuda()
    sa = (torch.rand(M, generator=g) + 0.5).cuda()
    sw = (torch.rand(N, generator=g) * 0.01 + 0.001).cuda()
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
    lib = G.lib()
    ws = torch.empty(lib.teo_gemm_workspace_bytes(), dtype=torch.uint8, device="cuda")
    L.check(lib.teo_gemm_workspace_init(G.p(ws), G.stream()), "ws init")
    try:
        for od in (torch.bfloat16, torch.float32):
            L.tune_set(b"gemm_fp8_wide", 0)
            want = _gemm_fp8(A8, sa, W8, sw, res=res, out_dtype=od)
            L.tune_set(b"gemm_fp8_wide", 3)
            for _ in range(3):
                got = torch.full((M, N), float("nan"), dtype=od, device="cuda")
                L.check(lib.teo_gemm_fp8_ws(G.p(A8), G.p(sa), G.p(W8), G.p(sw), G.p(res), G.p(got), M, N, K, K, N, 0, G.DT[od], G.p(ws),
                                            G.stream()), "gemm_fp8_ws")
                assert torch.equal(got, want), (od, (got.float() - want.float()).abs().max().item())
    finally:
        L.tune_set(b"gemm_fp8_wide", 1)


def test_gemm_fp8_swiglu_pairs_gate_and_up_rows():
    from teochat_amd.engine import interleave_gate_up
    M, Fd, K = 300, 704, 512
    g = torch.Generator().manual_seed(4)
    A = torch.randn(M, K, generator=g)
    gate, up = torch.randn(Fd, K, generator=g) * 0.06, torch.randn(Fd, K, generator=g) * 0.06
    qa, sa = _quant_ref(A)
    qg, sg = _quant_ref(gate)
    qu, su = _quant_ref(up)
    w8 = interleave_gate_up(qg.view(torch.uint8), qu.view(torch.uint8)).cuda()
    s8 = interleave_gate_up(sg.view(-1, 1), su.view(-1, 1)).view(-1).float().cuda()
    got = _gemm_fp8(qa.view(torch.uint8).cuda(), sa.float().cuda(), w8, s8, flags=L.GEMM_SWIGLU16, out_dtype=torch.float32).cpu()
    Ad = qa.double() * sa.double()[:, None]
    gd, ud = Ad @ (qg.double() * sg.double()[:, None]).T, Ad @ (qu.double() * su.double()[:, None]).T
    want = (F.silu(gd) * ud).float()
    # the MFMA's internal sum is good to ~1e-5 of sum |a||w| (previous test): here ~3e-4 absolute on g and u of size ~1.4
    torch.testing.assert_close(got, want, atol=2e-3, rtol=2e-3)


@pytest.mark.parametrize("M,K,norm", [(7, 4096, False), (300, 11008, False), (300, 4096, True), (5, 1152, True)])
def test_quant_rows_fp8_matches_the_definition(M, K, norm):
    g = torch.Generator().manual_seed(K + M)
    x = (torch.randn(M, K, generator=g) * torch.logspace(-2, 1, M)[:, None]).to(torch.bfloat16)
    x[M // 2] = 0                                                     # an all-zero row: scale 1, codes 0
    w = (1.0 + 0.1 * torch.randn(K, generator=g)).to(torch.bfloat16)
    q = torch.empty(M, K, dtype=torch.uint8, device="cuda")
    s = torch.empty(M, dtype=torch.float32, device="cuda")
    dx, dw = x.cuda(), w.cuda()                                       # keep the device tensors alive across the launch
    L.check(G.lib().teo_quant_rows_fp8(G.p(dx), G.p(dw) if norm else None, G.p(q), G.p(s), M, K, K, 1e-5, G.stream()), "quant")
    xf = x.float()
    if norm:
        xf = G.bf16_round(O.rmsnorm(xf, w.float(), 1e-5))
    qr, sr = _quant_ref(xf)
    torch.testing.assert_close(s.cpu(), sr, rtol=2e-6, atol=0)
    got = q.cpu().view(torch.float8_e4m3fn).float()
    # identical codes except where x / s sits on a rounding boundary and the two divisions differ in the last fp32 bit
    diff = (got - qr.float()).abs()
    step = qr.float().abs().clamp_min(2 ** -6) * 0.125
    assert bool((diff <= step + 1e-9).all())
    assert float((diff > 0).float().mean()) < 2e-3
    assert float(got.abs().max()) <= 448.0 and bool((got[M // 2] == 0).all()) and float(s[M // 2]) == 1.0
    # rows that are not all zero use the full range: their largest code is exactly +-448
    nz = [i for i in range(M) if i != M // 2]
    assert bool((got[nz].abs().amax(dim=1) == 448.0).all())


@pytest.mark.parametrize("M,N,K,flags,extra", [(2168, 22016, 4096, L.GEMM_SWIGLU16, ""), (300, 700, 192, 0, "bias_gelu"), (129, 260, 128, 0, "res"),
                                               (1000, 1024, 4096, 0, "f32out"), (257, 512, 64, 0, ""), (2056, 4096, 1024, 0, "bias_res")])
def test_gemm_wide_tile_kernel_is_bitwise_the_plain_kernel(M, N, K, flags, extra):
    """gemm_wide.hip (128 x 256 tiles, LDS-DMA operand staging into a three-stage ring, counted vmcnt + raw barrier) against the
    128 x 128 register-staged kernel: same k-order per output element -> BIT-identical, incl. ragged M / N edges, K of one or
    three tiles (ring warm-up / drain), every epilogue."""
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(M + 7 * N + K)
    A = torch.randn(M, K, generator=g).to(bf).cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).to(bf).cuda()
    Nc = N // 2 if flags else N
    bias = (torch.randn(N, generator=g) * 0.1).to(bf).cuda() if "bias" in extra else None
    res = torch.randn(M, Nc, generator=g).to(bf).cuda() if "res" in extra else None
    act = L.ACT_GELU_ERF if "gelu" in extra else L.ACT_NONE
    od = torch.float32 if "f32out" in extra else bf
    lib = G.lib()
    try:
        assert L.tune_set(b"gemm_wide", 0) == 0
        want = G.gemm(A, W, bias=bias, res=res, act=act, flags=flags, out_dtype=od)
        assert L.tune_set(b"gemm_wide", 2) == 0
        for _ in range(3):
            got = G.gemm(A, W, bias=bias, res=res, act=act, flags=flags, out_dtype=od)
            assert torch.equal(got, want)
    finally:
        L.tune_set(b"gemm_wide", 1)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,extra", [(2056, 1024, 4096, "bias_res"), (2056, 1024, 1024, "bias_res"), (2056, 4096, 1024, "bias_gelu"),
                                         (300, 700, 192, "bias_quick"), (129, 260, 128, "res"), (1000, 1024, 2048, "f32out"), (65, 132, 64, ""),
                                         (514, 3072, 1024, "bias")])
def test_gemm_narrow_tiles_are_bitwise_the_plain_kernel(M, N, K, extra, dt):
    """gemm_narrow.hip (round 5: 64 x 128 and 128 x 128 tiles, 4 waves, LDS-DMA ring of 3 / 2 stages, two workgroups per CU) against the
    register-staged kernel: same LDS image, same fragment reads, same k-ascending MFMA chain -> BIT-identical in both 16-bit formats, incl.
    the tower's real shapes (fc2 / out_proj / fc1 at M = 2056), ragged M / N edges, K of one / two / three tiles and every epilogue."""
    g = torch.Generator().manual_seed(M + 7 * N + K)
    A = torch.randn(M, K, generator=g).to(dt).cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).to(dt).cuda()
    bias = (torch.randn(N, generator=g) * 0.1).to(dt).cuda() if "bias" in extra else None
    res = torch.randn(M, N, generator=g).to(dt).cuda() if "res" in extra else None
    act = L.ACT_GELU_ERF if "gelu" in extra else (L.ACT_QUICK_GELU if "quick" in extra else L.ACT_NONE)
    od = torch.float32 if "f32out" in extra else dt
    lib = G.lib()
    assert L.tune_set(b"gemm_narrow", 0) == 0 and L.tune_set(b"gemm_big", 0) == 0 and L.tune_set(b"gemm_wide", 0) == 0
    want = G.gemm(A, W, bias=bias, res=res, act=act, out_dtype=od)
    assert lib.teo_last_kernel().decode() in ("gemm_mfma_64", "gemm_mfma_128")
    assert L.tune_set(b"gemm_narrow", 2) == 0
    for bm in (64, 128):
        assert L.tune_set(b"gemm_narrow_bm", bm) == 0
        for _ in range(3):
            got = G.gemm(A, W, bias=bias, res=res, act=act, out_dtype=od)
            assert lib.teo_last_kernel().decode() == f"gemm_narrow_{bm}"
            assert torch.equal(got, want), (bm, float((got.float() - want.float()).abs().max()))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,extra", [(2168, 4096, 4096, "res"), (2168, 4096, 11008, "res"), (2056, 4096, 1024, "bias_gelu"), (2056, 4096, 1024, "bias_quick"),
                                         (300, 700, 192, "bias_quick"), (129, 260, 128, "res"), (1000, 1024, 2048, "f32out"), (65, 132, 64, ""),
                                         (257, 164, 256, "bias_res"), (514, 3072, 320, "bias")])
def test_gemm_quad_tiles_are_bitwise_the_plain_kernel(M, N, K, extra, dt):
    """gemm_quad.hip (round 5: 256 x 160 tiles, accumulators pinned to AGPRs by inline-asm MFMAs, hand-written issue order, three-stage
    LDS-DMA ring, one barrier per K tile; eight waves of 64 x 80 -- the default -- or four of 128 x 80, one per SIMD) against the
    register-staged kernel: same LDS image, same fragment reads, same k-ascending MFMA chain -> BIT-identical in both 16-bit formats and
    both wave layouts, at the shapes it is dispatched for (LLaMA o / down at M = 2168, the tower's fc1), at ragged M / N edges (N not a
    multiple of 160, of 16; M = 1 row over a tile), K of one .. five tiles (ring shorter than / equal to / longer than the loop) and with
    every epilogue.  The production dispatch picks it for the first four shapes by itself."""
    g = torch.Generator().manual_seed(M + 7 * N + K)
    A = torch.randn(M, K, generator=g).to(dt).cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).to(dt).cuda()
    bias = (torch.randn(N, generator=g) * 0.1).to(dt).cuda() if "bias" in extra else None
    res = torch.randn(M, N, generator=g).to(dt).cuda() if "res" in extra else None
    act = L.ACT_GELU_ERF if "gelu" in extra else (L.ACT_QUICK_GELU if "quick" in extra else L.ACT_NONE)
    od = torch.float32 if "f32out" in extra else dt
    lib = G.lib()
    if M >= 2056 and N == 4096:
        G.gemm(A, W, bias=bias, res=res, act=act, out_dtype=od)
        assert lib.teo_last_kernel().decode() == "gemm_quad_160"           # one round of 256 x 160 tiles, more than one of 128 x 256
    assert L.tune_set(b"gemm_narrow", 0) == 0 and L.tune_set(b"gemm_big", 0) == 0 and L.tune_set(b"gemm_wide", 0) == 0
    want = G.gemm(A, W, bias=bias, res=res, act=act, out_dtype=od)
    assert lib.teo_last_kernel().decode() in ("gemm_mfma_64", "gemm_mfma_128")
    assert L.tune_set(b"gemm_quad", 2) == 0
    for waves, name in ((8, "gemm_quad_160"), (4, "gemm_quad_160_w4")):
        assert L.tune_set(b"gemm_quad_waves", waves) == 0
        for _ in range(3):
            got = G.gemm(A, W, bias=bias, res=res, act=act, out_dtype=od)
            assert lib.teo_last_kernel().decode() == name
            assert torch.equal(got, want), (waves, float((got.float() - want.float()).abs().max()))


@pytest.mark.parametrize("M,N,K,flags,extra", [(2168, 12288, 4096, 0, ""), (4208, 2048, 512, L.GEMM_SWIGLU16, ""), (300, 700, 192, 0, "bias_gelu"),
                                               (129, 260, 128, 0, "res"), (1000, 1024, 2048, 0, "f32out"), (257, 512, 128, 0, ""),
                                               (2056, 4096, 1024, 0, "bias_res"), (4096, 1024, 1024, 0, "group")])
def test_gemm_256x256_kernel_is_bitwise_the_plain_kernel(M, N, K, flags, extra):
    """gemm_big.hip (256 x 256 tiles, 8 waves of 128 x 64, two-stage LDS-DMA ring, carried second half) against the 128 x 128
    register-staged kernel: same k-order per output element -> BIT-identical, incl. ragged M / N edges, K of two tiles, every
    epilogue, and the grouped tile walk (>= 16 row tiles)."""
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(M + 7 * N + K)
    A = torch.randn(M, K, generator=g).to(bf).cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).to(bf).cuda()
    Nc = N // 2 if flags else N
    bias = (torch.randn(N, generator=g) * 0.1).to(bf).cuda() if "bias" in extra else None
    res = torch.randn(M, Nc, generator=g).to(bf).cuda() if "res" in extra else None
    act = L.ACT_GELU_ERF if "gelu" in extra else L.ACT_NONE
    od = torch.float32 if "f32out" in extra else bf
    lib = G.lib()
    try:
        assert L.tune_set(b"gemm_big", 0) == 0 and L.tune_set(b"gemm_wide", 0) == 0
        want = G.gemm(A, W, bias=bias, res=res, act=act, flags=flags, out_dtype=od)
        assert L.tune_set(b"gemm_big", 2) == 0
        for _ in range(3):
            got = G.gemm(A, W, bias=bias, res=res, act=act, flags=flags, out_dtype=od)
            assert torch.equal(got, want)
    finally:
        L.tune_set(b"gemm_big", 1)
        L.tune_set(b"gemm_wide", 1)


@pytest.mark.parametrize("M,N,K,flags,extra", [(2168, 12288, 4096, 0, ""), (2168, 22016, 4096, L.GEMM_SWIGLU16, ""), (4208, 12288, 512, 0, "res"),
                                               (2100, 8192, 256, 0, "bias_gelu"), (4096, 4352, 1024, 0, "f32out"), (2168, 12288 + 256, 128, 0, "")])
def test_gemm_256x256_hybrid_is_bitwise_the_plain_kernel(M, N, K, flags, extra):
    """The persistent form of gemm_big.hip behind teo_gemm_ws: whole tiles for the first rounds (one per workgroup per round), the
    remaining 256..511 tiles cut into 256 equal (tile, k) ranges with the sequential slab hand-off -- BIT-identical to the
    128 x 128 kernel, launch after launch on the same workspace, also next to a concurrent HBM stream.  Shapes: pure stream-K
    (432 tiles), two whole rounds + 262 (774), K of two and four tiles (per == nk), grouped tile walk, every epilogue."""
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g)).to(bf).cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).to(bf).cuda()
    Nc = N // 2 if flags else N
    bias = (torch.randn(N, generator=g) * 0.1).to(bf).cuda() if "bias" in extra else None
    res = (torch.randn(M, Nc, generator=g)).to(bf).cuda() if "res" in extra else None
    act = L.ACT_GELU_ERF if "gelu" in extra else L.ACT_NONE
    od = torch.float32 if "f32out" in extra else bf
    lib = G.lib()
    T = -(-M // 256) * -(-N // 256)
    assert T > 256 and T % 256 != 0
    ws = torch.empty(lib.teo_gemm_workspace_bytes(), dtype=torch.uint8, device="cuda")
    L.check(lib.teo_gemm_workspace_init(G.p(ws), G.stream()), "ws init")
    side = torch.cuda.Stream()
    big = torch.empty(64 * 2 ** 20, dtype=torch.float32, device="cuda")
    big2 = torch.empty_like(big)
    try:
        assert L.tune_set(b"gemm_big", 0) == 0 and L.tune_set(b"gemm_wide", 0) == 0
        want = G.gemm(A, W, bias=bias, res=res, act=act, flags=flags, out_dtype=od)
        assert L.tune_set(b"gemm_big", 2) == 0 and L.tune_set(b"gemm_big_hybrid", 2) == 0
        # round 5: the stream-K part as XCD-local cohorts (columns of 8 / 16 / 32 tiles walked in step, hand-off to the same slot of the
        # next chain link) next to the linear ranges (0) and the shipped choice (-1): the same bits from every arrangement
        # round 6: the same with the ragged last row block covered by 128 x 512 tiles (gemm_big_ragged = 2: whenever the shape allows)
        for cohort, ragged in ((-1, 0), (0, 0), (8, 0), (16, 0), (32, 0), (-1, 2), (0, 2), (16, 2)):
            assert L.tune_set(b"gemm_big_cohort", cohort) == 0 and L.tune_set(b"gemm_big_ragged", ragged) == 0
            bad = 0
            for it in range(4):
                if it % 2:
                    with torch.cuda.stream(side):
                        big2.copy_(big)
                got = _gemm_ws(A, W, ws, bias=bias, res=res, act=act, flags=flags, out_dtype=od)
                torch.cuda.synchronize()
                bad += int(not torch.equal(got, want))
            assert bad == 0, f"cohort {cohort}, ragged {ragged}: {bad}/4 launches differ from the plain kernel"
            assert lib.teo_last_kernel().decode() in ("gemm_big_hybrid", "gemm_big_hybrid_cohort")
        st = C.c_int(0)
        L.check(lib.teo_gemm_workspace_status(G.p(ws), C.byref(st), G.stream()), "ws status")
        assert st.value == 0
    finally:
        L.tune_set(b"gemm_big_cohort", -1)
        L.tune_set(b"gemm_big_ragged", 1)
        L.tune_set(b"gemm_big_hybrid", 1)
        L.tune_set(b"gemm_big", 1)
        L.tune_set(b"gemm_wide", 1)


@pytest.mark.parametrize("M,N,K,swiglu", [(2168, 12288, 4096, False), (4208, 2048, 512, True), (300, 704, 256, False), (257, 512, 128, False),
                                          (4096, 1024, 1024, False)])
def test_gemm_fp8_256x256_kernel_is_bitwise_the_plain_fp8_kernel(M, N, K, swiglu):
    """gemm_mfma_fp8_big_kernel (256 x 256 tiles, two-stage LDS-DMA ring) against the 128 x 128 fp8 kernel: same k order per output
    element -> same bits; ragged edges, K of one / two tiles, SwiGLU and residual epilogues, fp32 and bf16 outputs, grouped walk."""
    g = torch.Generator().manual_seed(M + N + K)
    A8 = _quant_ref(torch.randn(M, K, generator=g))[0].view(torch.uint8).cuda()
    W8 = _quant_ref(torch.randn(N, K, generator=g) * 0.02)[0].view(torch.uint8).cuda()
    sa = (torch.rand(M, generator=g) + 0.5).cuda()
    sw = (torch.rand(N, generator=g) * 0.01 + 0.001).cuda()
    res = None if swiglu else torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
    flags = L.GEMM_SWIGLU16 if swiglu else 0
    lib = G.lib()
    try:
        for od in (torch.bfloat16, torch.float32):
            L.tune_set(b"gemm_fp8_big", 0); L.tune_set(b"gemm_fp8_wide", 0)
            want = _gemm_fp8(A8, sa, W8, sw, res=res, flags=flags, out_dtype=od)
            L.tune_set(b"gemm_fp8_big", 2)
            for _ in range(2):
                got = _gemm_fp8(A8, sa, W8, sw, res=res, flags=flags, out_dtype=od)
                assert torch.equal(got, want), (od, (got.float() - want.float()).abs().max().item())
    finally:
        L.tune_set(b"gemm_fp8_big", 1)
        L.tune_set(b"gemm_fp8_wide", 1)


@pytest.mark.parametrize("T,img,P,D", [(2, 224, 14, 1024), (1, 56, 14, 192), (3, 64, 16, 260)])
def test_patch_embed_fused_equals_im2col_plus_gemm_through_the_abi(T, img, P, D):
    """teo_patch_embed against teo_im2col_patches + teo_gemm on the same padded weight: bit-identical (same k order and MFMA chain);
    P = 14 (rows of 14 pixels never line up with the 8-element chunks), P = 16, ragged output width, several tiles of patches."""
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(T + img + D)
    Cc = 3
    KV = Cc * P * P
    ld = (KV + 63) // 64 * 64
    px = torch.randn(T, Cc, img, img, generator=g).to(bf).cuda()
    W = torch.zeros(D, ld, dtype=bf)
    W[:, :KV] = (torch.randn(D, KV, generator=g) * 0.05).to(bf)
    W = W.cuda()
    npatch = (img // P) ** 2
    cols = torch.empty(T * npatch, ld, dtype=bf, device="cuda")
    lib = G.lib()
    L.check(lib.teo_im2col_patches(G.p(px), G.p(cols), T, Cc, img, P, ld, L.TEO_BF16, G.stream()), "im2col")
    want = G.gemm(cols, W)
    got = torch.full((T * npatch, D), float("nan"), dtype=bf, device="cuda")
    L.check(lib.teo_patch_embed(G.p(px), G.p(W), G.p(got), T, Cc, img, P, ld, D, L.TEO_BF16, G.stream()), "patch_embed")
    assert torch.equal(got, want)
    # and the definition: the convolution with kernel = stride = P
    ref = torch.nn.functional.conv2d(px.float().cpu(), W[:, :KV].float().cpu().view(D, Cc, P, P), stride=P).flatten(2).transpose(1, 2).reshape(T * npatch, D)
    close_bf16(got, G.bf16_round(ref), ulps=1.0)


def test_gemm_dispatch_fuzz_is_bitwise_the_plain_kernel():
    """Whatever kernel the dispatch picks (128 x 128, 128 x 256, 256 x 256, their stream-K / hybrid forms behind teo_gemm_ws) for a
    shape, the bits are those of the 128 x 128 one-workgroup-per-tile kernel: 40 random shapes incl. ragged edges, SwiGLU, residual in
    place, fp32 output, with and without a workspace."""
    import random
    rnd_ = random.Random(1234)
    bf = torch.bfloat16
    lib = G.lib()
    ws = torch.empty(lib.teo_gemm_workspace_bytes(), dtype=torch.uint8, device="cuda")
    L.check(lib.teo_gemm_workspace_init(G.p(ws), G.stream()), "ws init")
    shapes = [(2168, 12288, 4096), (4208, 22016, 1024), (638, 12288, 512), (2168, 4096, 2048), (17344, 4096, 256), (300, 256, 128)]
    while len(shapes) < 40:
        M = rnd_.choice([1, 7, 129, 257, 638, 1000, 2056, 2168, 3000, 4208, 6000])
        N = rnd_.choice([4, 36, 256, 260, 1024, 3072, 4096, 8192, 12288, 22016]) + rnd_.choice([0, 0, 4, 32])
        K = 64 * rnd_.choice([1, 2, 3, 8, 16, 33, 64])
        if M * N * K > 3.2e11:
            continue
        shapes.append((M, N, K))
    for i, (M, N, K) in enumerate(shapes):
        g = torch.Generator().manual_seed(i)
        A = torch.randn(M, K, generator=g).to(bf).cuda()
        W = (torch.randn(N, K, generator=g) * 0.05).to(bf).cuda()
        swiglu = N % 32 == 0 and i % 3 == 0
        flags = L.GEMM_SWIGLU16 if swiglu else 0
        Nc = N // 2 if swiglu else N
        res = None if swiglu or i % 2 else torch.randn(M, Nc, generator=g).to(bf).cuda()
        od = torch.float32 if i % 5 == 4 else bf
        try:
            for k_, v_ in ((b"gemm_big", 0), (b"gemm_wide", 0), (b"gemm_sk", 0)):
                L.tune_set(k_, v_)
            want = G.gemm(A, W, res=res, flags=flags, out_dtype=od)
        finally:
            for k_ in (b"gemm_big", b"gemm_wide", b"gemm_sk"):
                L.tune_set(k_, 1)
        got = G.gemm(A, W, res=res, flags=flags, out_dtype=od)
        assert torch.equal(got, want), ("teo_gemm", M, N, K, swiglu)
        got = _gemm_ws(A, W, ws, res=res, flags=flags, out_dtype=od)
        assert torch.equal(got, want), ("teo_gemm_ws", M, N, K, swiglu)
        if res is not None and od == bf:                    # residual in place (C == residual), as the prefill loop calls it
            buf = res.clone()
            L.check(lib.teo_gemm_ws(G.p(A), G.p(W), None, G.p(buf), G.p(buf), M, N, K, K, Nc, L.ACT_NONE, flags, L.TEO_BF16, L.TEO_BF16,
                                    G.p(ws), G.stream()), "gemm_ws in place")
            assert torch.equal(buf, want), ("in place", M, N, K)


def test_gemm_fp8_dispatch_fuzz_is_bitwise_the_plain_fp8_kernel():
    """Same property for the fp8 family (128 x 128, 128 x 256, 256 x 256, stream-K form): 24 random shapes, with and without a
    workspace, against the 128 x 128 fp8 kernel."""
    import random
    rnd_ = random.Random(99)
    lib = G.lib()
    ws = torch.empty(lib.teo_gemm_workspace_bytes(), dtype=torch.uint8, device="cuda")
    L.check(lib.teo_gemm_workspace_init(G.p(ws), G.stream()), "ws init")
    shapes = [(2168, 12288, 4096), (2168, 4096, 4096), (4208, 22016, 512), (17344, 4096, 256)]
    while len(shapes) < 24:
        M = rnd_.choice([1, 33, 257, 638, 2056, 2168, 4208])
        N = rnd_.choice([32, 256, 1024, 4096, 12288, 22016]) + rnd_.choice([0, 0, 32])
        K = 128 * rnd_.choice([1, 2, 3, 8, 16, 32])
        if M * N * K > 2.5e11:
            continue
        shapes.append((M, N, K))
    for i, (M, N, K) in enumerate(shapes):
        g = torch.Generator().manual_seed(1000 + i)
        A8 = _quant_ref(torch.randn(M, K, generator=g))[0].view(torch.uint8).cuda()
        W8 = _quant_ref(torch.randn(N, K, generator=g) * 0.02)[0].view(torch.uint8).cuda()
        sa = (torch.rand(M, generator=g) + 0.5).cuda()
        sw = (torch.rand(N, generator=g) * 0.01 + 0.001).cuda()
        swiglu = i % 3 == 0
        flags = L.GEMM_SWIGLU16 if swiglu else 0
        Nc = N // 2 if swiglu else N
        res = None if swiglu or i % 2 else torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
        od = torch.float32 if i % 5 == 4 else torch.bfloat16
        try:
            L.tune_set(b"gemm_fp8_big", 0); L.tune_set(b"gemm_fp8_wide", 0)
            want = _gemm_fp8(A8, sa, W8, sw, res=res, flags=flags, out_dtype=od)
        finally:
            L.tune_set(b"gemm_fp8_big", 1); L.tune_set(b"gemm_fp8_wide", 1)
        assert torch.equal(_gemm_fp8(A8, sa, W8, sw, res=res, flags=flags, out_dtype=od), want), ("teo_gemm_fp8", M, N, K, swiglu)
        got = torch.full((M, Nc), float("nan"), dtype=od, device="cuda")
        L.check(lib.teo_gemm_fp8_ws(G.p(A8), G.p(sa), G.p(W8), G.p(sw), G.p(res), G.p(got), M, N, K, K, Nc, flags, G.DT[od], G.p(ws),
                                    G.stream()), "gemm_fp8_ws")
        assert torch.equal(got, want), ("teo_gemm_fp8_ws", M, N, K, swiglu)
