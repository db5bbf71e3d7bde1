"""The oracle ON THE BASELINE SHAPES (VERDICT r02 "What's missing" #2, "What's weak" #2 / #4).

bench.py::cpu_baseline shows the oracle runs one LLaMA-2-7B layer at L = 2168 in ~1.2 s and a full-depth C3 prefill in
~15-40 s on the GPU box's host, so parity at size is checked directly, not through bit-identity chains:

  (a) teacher-forced WALK at the true shapes -- LLaMA D=4096, F=11008, H=32, hd=128, L=2168; ViT-L/14 D=1024, N=257, T=8;
      projector 1024 -> 4096 -> 4096.  Every kernel gets the oracle's bf16-rounded input and must land within ONE bf16 ulp
      of the oracle's output on EVERY element.  The GEMMs are run through teo_gemm_ws with the production workspace (the
      dispatch the runtime uses) AND with every tile family forced in turn (128x128, 128x256, 256x256, their stream-K /
      hybrid forms), and the test states which kernel each call dispatched to (teo_last_kernel).
  (b) C2 (T=2, L=638, 128 out) and C3 (T=8, L=2168, 256 out) end to end against the oracle: ViT-L/14 (23 layers), projector,
      splice, LLaMA at full width but N_LAYERS deep, prefill logits at every position + 8 teacher-forced decode steps;
      bf16 engine vs the rounding="bf16" oracle (max / p99 / median reported), fp32 engine vs the oracle in fp64
      (<= 1e-5 of max|logit|: north_star's fp32 bar).
  (c) ONE full-depth (32-layer) C3 prefill against the oracle (slow: ~40 s of CPU).

Reference being matched: videollava/model/language_model/llava_llama.py:56-99 via llava_arch.py:148-346 (H13-H16 of SURVEY 8a).
CPU budget of the file: ~2-3 min on the box's 16 CPUs."""
import ctypes as C
import math
import os
import time

import pytest
import torch
import torch.nn.functional as F

from oracle import teo_oracle as O
from teochat_amd import _lib as L
from teochat_amd.engine import interleave_gate_up, rope_tables
from tests import _gpu as G

pytestmark = pytest.mark.gpu
bf = torch.bfloat16          # the 16-bit format under test: the `fmt` fixture below switches it (and FMT) to IEEE half for the fp16 legs
DEV = "cuda:0"
N_LAYERS_DEEP = 3            # (b): LLaMA layers at full width
# round 5 (VERDICT r04 "Next round" #2): every test of this file runs in BOTH 16-bit formats -- bfloat16 and the reference's own
# inference type, IEEE binary16 (model/builder.py:104-105, eval/inference.py:53).  name = the oracle's rounding mode, mant = mantissa
# bits incl. the hidden one (what one ulp is), dt = the C ABI's dtype code
_FORMATS = {"bf16": dict(name="bf16", dtype=torch.bfloat16, mant=8, dt=L.TEO_BF16), "fp16": dict(name="fp16", dtype=torch.float16, mant=11, dt=L.TEO_F16)}
FMT = dict(_FORMATS["bf16"])


@pytest.fixture(params=["bf16", "fp16"])
def fmt(request):
    global bf
    old = dict(FMT), bf
    FMT.clear(); FMT.update(_FORMATS[request.param])
    bf = FMT["dtype"]
    yield request.param
    FMT.clear(); FMT.update(old[0])
    bf = old[1]


def R(t):
    return t.to(bf).to(t.dtype)


def _threads():
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        n = int(float(q) / float(per)) if q != "max" else (os.cpu_count() or 8)
    except (OSError, ValueError):
        n = os.cpu_count() or 8
    torch.set_num_threads(max(1, min(n, 32)))


FP32_SUM_ABS = 1e-5      # absolute allowance for the fp32 summation-order difference of a sum whose terms cancel (|ref| << its operands):
                         # eps_fp32 * sqrt(K) * |terms| ~ 6e-8 * 100 * 1 -- 100x below the round-3 floor of 1e-3, i.e. 0.1 ulp at |ref| = 0.03


def ulp_check(got, ref, tag, report, kernel=True, abs_tol=FP32_SUM_ABS):
    """got: device tensor (bf16 or f32); ref: oracle output rounded to bf16 (fp32 tensor).  EVERY element within ONE EXACT bf16 ulp
    of the oracle (G.ulp16: from torch.frexp, not 2^-7 |ref| which is up to 2 ulps) plus `abs_tol`; the worst element is printed in
    ulps."""
    got = got.float().cpu().reshape(ref.shape)
    d = (got - ref).abs()
    ulp = G.ulp16(ref, FMT['mant'])
    if torch.is_tensor(abs_tol):
        abs_tol = abs_tol.float().reshape(ref.shape)
    bad = int((d > ulp + abs_tol).sum())
    exact = float((d == 0).float().mean())
    worst = float((d / ulp).max())
    kn = L.load().teo_last_kernel().decode() if kernel else "-"
    report.append(f"  {tag:<58s} {tuple(ref.shape)!s:<16s} kernel={kn:<18s} bit-equal {exact * 100:6.2f} %  worst {worst:5.2f} ulp  beyond 1 ulp: {bad}")
    assert bad == 0, f"{tag}: {bad} / {d.numel()} elements beyond 1 bf16 ulp + allowance (worst {worst:.2f} ulp, max diff {float(d.max()):.3e})"


def attn_p_noise(q, k, v, visible, scale):
    """Per-output allowance for the bf16 rounding of P in softmax(QK^T) V: the kernels and the oracle both round the exponentiated
    scores to bf16 before the PV product, each at its own 1-ulp-different value, so an output moves by up to
    2^-8 * sum_j p_j |v_j| whatever its own magnitude (outputs of late causal rows are sums of ~2000 terms that largely cancel).
    Returned in the layout attention_core returns ([B, H, S, d])."""
    return (2.0 ** -FMT['mant']) * O.attention_core(q, k, v.abs(), visible, scale, lambda t: t, "exact")


def _rand(shape, gen, std=1.0):
    return R(torch.randn(*shape, generator=gen) * std)


class GemmWs:
    def __init__(self):
        lib = G.lib()
        self.ws = torch.empty(lib.teo_gemm_workspace_bytes(), dtype=torch.uint8, device=DEV)
        L.check(lib.teo_gemm_workspace_init(G.p(self.ws), G.stream()), "ws init")

    def __call__(self, A, W, bias=None, res=None, act=L.ACT_NONE, flags=0, out_dtype=None, ws=True):
        M, K = A.shape
        N = W.shape[0]
        out_dtype = out_dtype or A.dtype
        Nc = N // 2 if flags & L.GEMM_SWIGLU16 else N
        Cc = torch.empty(M, Nc, dtype=out_dtype, device=A.device)
        L.check(G.lib().teo_gemm_ws(G.p(A), G.p(W), G.p(bias), G.p(res), G.p(Cc), M, N, K, A.stride(0), Nc, act, flags, G.DT[A.dtype],
                                    G.DT[out_dtype], G.p(self.ws if ws else None), G.stream()), "gemm_ws")
        return Cc


# forced tile families (tune knobs are perf-only; every family must agree with the oracle, not merely with each other)
FAMILIES = (("production dispatch", {}),
            ("128x128 tile", {"gemm_wide": 0, "gemm_big": 0, "gemm_sk": 0, "gemm_narrow": 0}),
            ("128x128 stream-K", {"gemm_wide": 0, "gemm_big": 0, "gemm_sk": 2, "gemm_narrow": 0}),
            ("64x128 LDS-DMA", {"gemm_narrow": 2, "gemm_narrow_bm": 64}),           # round 5 (SwiGLU GEMMs have no such form: they fall
            ("128x128 LDS-DMA", {"gemm_narrow": 2, "gemm_narrow_bm": 128}),         # through to the production dispatch under these knobs)
            ("64x64 software-pipelined", {"gemm_narrow": 2, "gemm_narrow_bm": 64, "gemm_narrow_pipe": 2, "gemm_pipe_bn": 64}),      # round 6
            ("64x128 software-pipelined, ring of 4", {"gemm_narrow": 2, "gemm_narrow_bm": 64, "gemm_narrow_pipe": 2, "gemm_pipe_stages": 4}),
            ("128x128 software-pipelined", {"gemm_narrow": 2, "gemm_narrow_bm": 128, "gemm_narrow_pipe": 2}),
            ("128x96 software-pipelined", {"gemm_narrow": 2, "gemm_narrow_bm": 128, "gemm_narrow_pipe": 2, "gemm_pipe_bn": 96}),
            ("256x160 eight waves", {"gemm_quad": 2}),
            ("256x160 four waves", {"gemm_quad": 2, "gemm_quad_waves": 4}),
            ("128x256 tile", {"gemm_wide": 2, "gemm_big": 0, "gemm_sk": 0}),
            ("128x256 stream-K", {"gemm_wide": 2, "gemm_big": 0, "gemm_sk": 2}),
            ("256x256 tile", {"gemm_big": 2, "gemm_big_hybrid": 0}),
            ("256x256 hybrid", {"gemm_big": 2, "gemm_big_hybrid": 2}))
DEFAULTS = {"gemm_wide": 1, "gemm_big": 1, "gemm_sk": 1, "gemm_big_hybrid": 1, "gemm_narrow": 1, "gemm_narrow_bm": 0, "gemm_quad": 1, "gemm_quad_waves": 8,
            "gemm_narrow_pipe": 1, "gemm_pipe_bn": 0, "gemm_pipe_stages": 0}


def gemm_all_families(gw, A, W, ref, tag, report, families=FAMILIES, **kw):
    lib = G.lib()
    seen = set()
    for name, knobs in families:
        try:
            for k, v in knobs.items():
                assert L.tune_set(k.encode(), v) == 0, k
            got = gw(A, W, **kw)
            seen.add(lib.teo_last_kernel().decode())
            ulp_check(got, ref, f"{tag} [{name}]", report)
        finally:
            for k, v in DEFAULTS.items():
                L.tune_set(k.encode(), v)
    return seen


# ------------------------------------------------------------------------------------------------------------ (a) LLaMA
def test_llama_layer_walk_at_7b_shapes(fmt):
    _threads()
    t0 = time.perf_counter()
    report = []
    S, D, H, hd, Fi = 2168, 4096, 32, 128, 11008
    gen = torch.Generator().manual_seed(0)
    w = {k: _rand(s, gen, 0.02) for k, s in (("q", (D, D)), ("k", (D, D)), ("v", (D, D)), ("o", (D, D)), ("gate", (Fi, D)),
                                              ("up", (Fi, D)), ("down", (D, Fi)))}
    g_in = R(1.0 + 0.1 * torch.randn(D, generator=gen))
    g_post = R(1.0 + 0.1 * torch.randn(D, generator=gen))
    h = _rand((S, D), gen)
    gw = GemmWs()
    # rmsnorm
    n1 = R(O.rmsnorm(h, g_in, 1e-5))
    ulp_check(G.rmsnorm(G.dev(h, bf), G.dev(g_in, bf), 1e-5), n1, "rmsnorm", report, kernel=False)
    # fused q/k/v projection: every tile family
    Wqkv = torch.cat([w["q"], w["k"], w["v"]], 0)
    qkv = R(n1 @ Wqkv.t())
    d_n1, d_Wqkv = G.dev(n1, bf), G.dev(Wqkv, bf)
    seen = gemm_all_families(gw, d_n1, d_Wqkv, qkv, "qkv GEMM M=2168 N=12288 K=4096", report)
    assert {"gemm_mfma_128", "gemm_wide", "gemm_big", "gemm_narrow_64", "gemm_narrow_128", "gemm_pipe_64x64", "gemm_pipe_64_r4", "gemm_pipe_128", "gemm_pipe_128x96"} <= seen, seen
    # RoPE + KV append
    pos = torch.arange(S)
    c, s_ = O.rope_cos_sin(pos, hd, 10000.0, torch.float32)
    q, k, v = (qkv[:, i * D:(i + 1) * D].view(S, H, hd) for i in range(3))
    qr = R(q * c[:, None] + O.rotate_half(q) * s_[:, None])
    kr = R(k * c[:, None] + O.rotate_half(k) * s_[:, None])
    S_max = 2432
    cs, sn = rope_tables(hd, 10000.0, 4096)
    d_qkv = G.dev(qkv, bf)
    kc = torch.zeros(H, S_max, hd, dtype=bf, device=DEV)
    vc, vtc = torch.zeros_like(kc), torch.zeros(H, hd, S_max, dtype=bf, device=DEV)
    d_pos, d_cs, d_sn = pos.int().to(DEV), cs.to(DEV), sn.to(DEV)
    L.check(G.lib().teo_rope_kv_append(G.p(d_qkv), 3 * D, G.p(d_pos), G.p(d_cs), G.p(d_sn), G.p(kc), G.p(vc), G.p(vtc), S, 0, S_max,
                                       H, H, hd, FMT['dt'], G.stream()), "rope")
    ulp_check(d_qkv[:, :D].reshape(S, H, hd), qr, "RoPE q", report, kernel=False)
    ulp_check(kc[:, :S].transpose(0, 1), kr, "RoPE k -> K cache", report, kernel=False)
    assert torch.equal(vc[:, :S].transpose(0, 1).cpu().float(), v), "V cache rows"
    assert torch.equal(vtc[:, :, :S].permute(2, 0, 1).cpu().float(), v), "V^T cache"
    # causal flash attention at L = 2168, 32 heads (the production prefill kernel)
    vis = (torch.arange(S).view(1, S) <= torch.arange(S).view(S, 1)).view(1, 1, S, S)
    qq, kk, vv = qr.transpose(0, 1)[None], kr.transpose(0, 1)[None], v.transpose(0, 1)[None]
    o_ref = R(O.attention_core(qq, kk, vv, vis, 1 / math.sqrt(hd), R, "flash64").transpose(1, 2).reshape(S, D))
    a = L.AttnArgs()
    o_k = torch.empty(S, D, dtype=bf, device=DEV)
    dq = G.dev(qr.reshape(S, D), bf)
    a.q, a.k, a.v, a.vt, a.o = dq.data_ptr(), kc.data_ptr(), vc.data_ptr(), vtc.data_ptr(), o_k.data_ptr()
    a.q_hs, a.q_rs, a.k_hs, a.k_rs, a.v_hs, a.v_rs = hd, D, S_max * hd, hd, S_max * hd, hd
    a.vt_hs, a.vt_rs, a.o_rs = hd * S_max, S_max, D
    a.batch, a.heads, a.kv_heads, a.head_dim, a.q_len, a.kv_len, a.causal, a.scale = 1, H, H, hd, S, S, 1, 1 / math.sqrt(hd)
    L.check(G.lib().teo_attention(C.byref(a), FMT['dt'], G.stream()), "attn")
    assert G.lib().teo_last_kernel() == b"attn_flash32"
    ulp_check(o_k, o_ref, "causal flash attention L=2168 H=32 d=128", report,
              abs_tol=attn_p_noise(qq, kk, vv, vis, 1 / math.sqrt(hd)).transpose(1, 2).reshape(S, D) + FP32_SUM_ABS)
    # o projection + residual (stream-K shapes: 272 wide tiles / 544 narrow tiles)
    h1 = R(h + o_ref @ w["o"].t())
    d_o, d_Wo, d_h = G.dev(o_ref, bf), G.dev(w["o"], bf), G.dev(h, bf)
    seen = gemm_all_families(gw, d_o, d_Wo, h1, "o GEMM + residual N=4096 K=4096", report, res=d_h)
    assert {"gemm_mfma_128_sk", "gemm_wide_sk", "gemm_quad_160"} <= seen, seen     # (production dispatch: the hand-scheduled 256 x 160 tile)
    # post norm, gate/up with SwiGLU epilogue, down + residual
    n2 = R(O.rmsnorm(h1, g_post, 1e-5))
    act = R(F.silu(n2 @ w["gate"].t()) * (n2 @ w["up"].t()))
    d_n2, d_gu = G.dev(n2, bf), G.dev(interleave_gate_up(w["gate"], w["up"]), bf)
    gemm_all_families(gw, d_n2, d_gu, act, "gate/up GEMM + SwiGLU N=22016 K=4096", report, flags=L.GEMM_SWIGLU16)
    h2 = R(h1 + act @ w["down"].t())
    d_act, d_Wd, d_h1 = G.dev(act, bf), G.dev(w["down"], bf), G.dev(h1, bf)
    gemm_all_families(gw, d_act, d_Wd, h2, "down GEMM + residual N=4096 K=11008", report, res=d_h1)
    # ---- the decode-step kernels on the same layer (one activation row, ctx = L)
    x = h2[-1]
    xn = R(O.rmsnorm(x, g_in, 1e-5))
    y = G.gemv(G.dev(x, bf), d_Wqkv, norm_w=G.dev(g_in, bf))
    ulp_check(y, R(Wqkv @ xn), "decode GEMV rmsnorm + qkv N=12288", report, kernel=False)
    xo = o_ref[-1]
    ulp_check(G.gemv(G.dev(xo, bf), d_Wo, res=G.dev(x, bf)), R(x + w["o"] @ xo), "decode GEMV o + residual (split-K)", report, kernel=False)
    xp = R(O.rmsnorm(x, g_post, 1e-5))
    ulp_check(G.gemv(G.dev(x, bf), d_gu, norm_w=G.dev(g_post, bf), flags=L.GEMM_SWIGLU16), R(F.silu(w["gate"] @ xp) * (w["up"] @ xp)),
              "decode GEMV rmsnorm + gate/up + SwiGLU N=22016", report, kernel=False)
    ulp_check(G.gemv(G.dev(act[-1], bf), d_Wd, res=G.dev(x, bf)), R(x + w["down"] @ act[-1]), "decode GEMV down + residual K=11008", report, kernel=False)
    # decode attention over the L cached keys + the new one (chunk 64 = production default for one conversation)
    qd = qr[-1]                                                     # a rotated query [H, hd]
    n_keys = S
    lib = G.lib()
    part = torch.empty(lib.teo_attn_decode_workspace_bytes(H, hd, S_max, 1), dtype=torch.uint8, device=DEV)
    out = torch.empty(D, dtype=bf, device=DEV)
    posd = torch.tensor([n_keys - 1], dtype=torch.int32, device=DEV)
    d_qd = G.dev(qd.reshape(D), bf)
    L.check(lib.teo_attn_decode(G.p(d_qd), G.p(kc), G.p(vc), None, None, None, G.p(out), G.p(part), G.p(posd), S_max, H,
                                H, hd, 1.0 / math.sqrt(hd), FMT['dt'], 1, D, H * S_max * hd, D, G.stream()), "attn_decode")
    o_dec = R(O.attention_core(qd[None, :, None, :], kk, vv, None, 1 / math.sqrt(hd), R, "split64")[0, :, 0].reshape(D))
    ulp_check(out, o_dec, "decode attention ctx=2168 (split-KV + combine)", report, kernel=False,
              abs_tol=attn_p_noise(qd[None, :, None, :], kk, vv, None, 1 / math.sqrt(hd))[0, :, 0].reshape(D) + FP32_SUM_ABS)
    # lm_head GEMV with the final norm, fp32 logits
    Wlm = _rand((32000, D), gen, 0.02)
    g_f = R(1.0 + 0.1 * torch.randn(D, generator=gen))
    xf = R(O.rmsnorm(x, g_f, 1e-5))
    lg = G.gemv(G.dev(x, bf), G.dev(Wlm, bf), norm_w=G.dev(g_f, bf), out_dtype=torch.float32)
    want = Wlm.double() @ xf.double()
    err = float((lg.cpu().double() - want).abs().max()) / float(want.abs().max())
    report.append(f"  {'decode GEMV rmsnorm + lm_head (fp32 logits) N=32000':<58s} max|d|/max|logit| {err:.2e}")
    assert err < 1e-5
    print(f"\n[7B-shape LLaMA layer walk in {fmt}, every kernel fed the oracle's input]\n" + "\n".join(report)
          + f"\n  wall {time.perf_counter() - t0:.1f} s")


# ------------------------------------------------------------------------------------------------------------ (a) ViT + projector
def test_vit_layer_and_projector_walk_at_vit_l14_shapes(fmt):
    _threads()
    t0 = time.perf_counter()
    report = []
    T, N, D, H, Fi = 8, 257, 1024, 16, 4096
    hd = D // H
    gen = torch.Generator().manual_seed(1)
    gw = GemmWs()
    lib = G.lib()
    # patch embedding: conv(k = stride = 14, no bias) as one kernel, pixels gathered into the MFMA tile's LDS image
    px = _rand((T, 3, 224, 224), gen)
    pw = _rand((D, 3, 14, 14), gen, 0.02)
    cols = px.view(T, 3, 16, 14, 16, 14).permute(0, 2, 4, 1, 3, 5).reshape(T * 256, 588)
    patches = R(cols @ pw.reshape(D, 588).t())
    pw_pad = torch.zeros(D, 640)
    pw_pad[:, :588] = pw.reshape(D, 588)
    d_out = torch.empty(T * 256, D, dtype=bf, device=DEV)
    d_px, d_pw = G.dev(px, bf), G.dev(pw_pad, bf)            # (device operands are kept in variables: a temporary would be freed,
    L.check(lib.teo_patch_embed(G.p(d_px), G.p(d_pw), G.p(d_out), T, 3, 224, 14, 640, D, FMT['dt'], G.stream()),   # and its block reused, before the launch)
            "patch_embed")
    assert lib.teo_last_kernel() == b"patch_embed_mfma"
    ulp_check(d_out, patches, "patch embedding (fused gather + MFMA) K=588", report)
    # + CLS + position embedding, pre-LayerNorm
    cls, posw = _rand((D,), gen, 0.02), _rand((N, D), gen, 0.02)
    g0, b0 = R(1.0 + 0.1 * torch.randn(D, generator=gen)), _rand((D,), gen, 0.02)
    emb = R(torch.cat([cls.view(1, 1, D).expand(T, 1, D), patches.view(T, 256, D)], 1) + posw[None])
    h0 = R(F.layer_norm(emb, (D,), g0, b0, 1e-5)).reshape(T * N, D)
    d_h0 = torch.empty(T * N, D, dtype=bf, device=DEV)
    dv = [G.dev(t, bf) for t in (patches, cls, posw, g0, b0)]
    L.check(lib.teo_vit_embed_ln(G.p(dv[0]), G.p(dv[1]), G.p(dv[2]), G.p(dv[3]), G.p(dv[4]),
                                 G.p(d_h0), T, 256, D, 1e-5, FMT['dt'], G.stream()), "vit_embed_ln")
    ulp_check(d_h0, h0, "CLS + position embedding + pre-LayerNorm", report, kernel=False)
    # one encoder layer at M = T * 257 = 2056
    h = _rand((T * N, D), gen)
    w = {k: _rand(s, gen, 0.02) for k, s in (("q", (D, D)), ("k", (D, D)), ("v", (D, D)), ("o", (D, D)), ("fc1", (Fi, D)), ("fc2", (D, Fi)))}
    b = {k: _rand((n,), gen, 0.02) for k, n in (("qkv", 3 * D), ("o", D), ("fc1", Fi), ("fc2", D))}
    ln = {k: (R(1.0 + 0.1 * torch.randn(D, generator=gen)), _rand((D,), gen, 0.02)) for k in ("1", "2")}
    ln1 = R(F.layer_norm(h, (D,), ln["1"][0], ln["1"][1], 1e-5))
    ulp_check(G.layernorm(G.dev(h, bf), G.dev(ln["1"][0], bf), G.dev(ln["1"][1], bf), 1e-5), ln1, "LayerNorm", report, kernel=False)
    Wqkv = torch.cat([w["q"], w["k"], w["v"]], 0)
    qkv = R(ln1 @ Wqkv.t() + b["qkv"])
    fam = [f for f in FAMILIES if f[0] in ("production dispatch", "128x128 tile", "128x256 tile", "256x256 tile", "64x128 LDS-DMA", "128x128 LDS-DMA")]
    fam.sort(key=lambda f: ("production dispatch", "128x128 tile", "64x128 LDS-DMA", "128x128 LDS-DMA", "128x256 tile", "256x256 tile").index(f[0]))
    gemm_all_families(gw, G.dev(ln1, bf), G.dev(Wqkv, bf), qkv, "qkv GEMM + bias M=2056 N=3072 K=1024", report, families=fam, bias=G.dev(b["qkv"], bf))
    q, k, v = (qkv[:, i * D:(i + 1) * D].view(T, N, H, hd).transpose(1, 2) for i in range(3))
    o_ref = R(O.attention_core(q, k, v, None, hd ** -0.5, R, "flash64").transpose(1, 2).reshape(T * N, D))
    qd, kd, vd = (G.dev(t.contiguous(), bf) for t in (q, k, v))
    o_k = G.attention(qd, kd, vd, False, hd ** -0.5, vt=G.make_vt(vd))
    assert lib.teo_last_kernel() == b"attn_flash32"
    ulp_check(o_k.reshape(T * N, D), o_ref, "flash attention N=257 d=64 16 heads x 8 frames", report,
              abs_tol=attn_p_noise(q, k, v, None, hd ** -0.5).transpose(1, 2).reshape(T * N, D) + FP32_SUM_ABS)
    h1 = R(h + o_ref @ w["o"].t() + b["o"])
    gemm_all_families(gw, G.dev(o_ref, bf), G.dev(w["o"], bf), h1, "out_proj + bias + residual N=1024 K=1024", report,
                      families=fam[:4], bias=G.dev(b["o"], bf), res=G.dev(h, bf))
    ln2 = R(F.layer_norm(h1, (D,), ln["2"][0], ln["2"][1], 1e-5))
    m = R(F.gelu(ln2 @ w["fc1"].t() + b["fc1"]))
    gemm_all_families(gw, G.dev(ln2, bf), G.dev(w["fc1"], bf), m, "fc1 + bias + GELU(erf) N=4096", report, families=fam,
                      bias=G.dev(b["fc1"], bf), act=L.ACT_GELU_ERF)
    zq = ln2 @ w["fc1"].t() + b["fc1"]                     # one rounding, after the activation: what oracle.vit_layer and the kernel do
    mq = R(zq * torch.sigmoid(1.702 * zq))
    gemm_all_families(gw, G.dev(ln2, bf), G.dev(w["fc1"], bf), mq, "fc1 + bias + quick_gelu N=4096", report, families=fam[:1],
                      bias=G.dev(b["fc1"], bf), act=L.ACT_QUICK_GELU)
    h2 = R(h1 + m @ w["fc2"].t() + b["fc2"])
    gemm_all_families(gw, G.dev(m, bf), G.dev(w["fc2"], bf), h2, "fc2 + bias + residual K=4096", report, families=fam[:4],
                      bias=G.dev(b["fc2"], bf), res=G.dev(h1, bf))
    # projector mlp2x_gelu on the T * 256 visual tokens
    feats = _rand((T * 256, D), gen)
    p0, p2 = _rand((4096, D), gen, 0.02), _rand((4096, 4096), gen, 0.02)
    pb0, pb2 = _rand((4096,), gen, 0.02), _rand((4096,), gen, 0.02)
    mid = R(F.gelu(feats @ p0.t() + pb0))
    gemm_all_families(gw, G.dev(feats, bf), G.dev(p0, bf), mid, "projector.0 + bias + GELU M=2048 1024->4096", report, families=fam,
                      bias=G.dev(pb0, bf), act=L.ACT_GELU_ERF)
    outp = R(mid @ p2.t() + pb2)
    gemm_all_families(gw, G.dev(mid, bf), G.dev(p2, bf), outp, "projector.2 + bias 4096->4096", report, families=fam, bias=G.dev(pb2, bf))
    print(f"\n[ViT-L/14 layer + projector walk at T=8 in {fmt}, every kernel fed the oracle's input]\n" + "\n".join(report)
          + f"\n  wall {time.perf_counter() - t0:.1f} s")


# ------------------------------------------------------------------------------------------------------------ (b), (c)
def _full_width_model(n_layers, dtype, max_seq, sd=None):
    from teochat_amd.config import teochat_7b_config
    from teochat_amd.engine import TeoEngine
    from teochat_amd.model import LlavaLlamaForCausalLM
    from teochat_amd.synthetic import synthetic_state_dict
    cfg = teochat_7b_config()
    cfg.num_hidden_layers = n_layers
    if sd is None:
        sd = synthetic_state_dict(cfg, seed=2, std=0.02, dtype=bf, device=DEV)          # bf16-valued weights for every leg
    eng = TeoEngine(sd, cfg, dtype=dtype, device=DEV, max_seq=max_seq)
    return LlavaLlamaForCausalLM(cfg, eng), sd, cfg


def _oracle_cfgs(n_layers):
    return O.VitCfg(hidden_act="gelu", num_hidden_layers=24), O.LlamaCfg(num_hidden_layers=n_layers), O.MMCfg()


def _stats(got, ref):
    d = (got.double() - ref.double()).abs().flatten()
    scale = float(ref.abs().max())
    k99 = max(1, int(0.99 * d.numel()))
    return float(d.max()) / scale, float(d.kthvalue(k99).values) / scale, float(d.median()) / scale, scale


def test_tile_family_knobs_do_not_change_one_bit_of_the_c3_forward_at_full_width():
    """The kernel-level claim (every GEMM tile family is bit-identical to every other: tests/test_kernels_gpu.py, test_gemm_fuzz_gpu.py)
    at the level a caller sees: the ViT-L/14 tower (23 layers, T = 8) + projector and a two-layer LLaMA prefill at 7B width, L = 2168 --
    the shapes at which the dispatch picks the 256 x 160 hand-scheduled tile, the 256 x 256 hybrid, the narrow tiles and the 128 x 256
    tile by itself -- give the SAME BITS (visual features and the logits of every position) whichever families an engine's tune block
    allows.  A performance knob is a performance knob."""
    _threads()
    m16, sd_dev, cfg = _full_width_model(2, bf, 2304)
    eng = m16.engine
    lib = G.lib()
    g = torch.Generator().manual_seed(17)
    pixels = torch.randn(8, 3, 224, 224, generator=g).to(bf).to(DEV)
    embeds = (torch.randn(2168, cfg.hidden_size, generator=g) * 0.5).to(bf).to(DEV)

    def run():
        eng.reset_cache()
        feats = eng.encode_images(pixels)
        return feats.clone(), eng.prefill(embeds).clone()

    f0, l0 = run()
    assert torch.isfinite(l0).all() and l0.shape == (2168, cfg.vocab_size)
    variants = (("no 256x160 tile", {"gemm_quad": 0}), ("256x160 on four waves", {"gemm_quad_waves": 4}),
                ("no 256x256 tile", {"gemm_big": 0}), ("no stream-K / hybrid forms", {"gemm_sk": 0, "gemm_big_hybrid": 0}),
                ("no LDS-DMA narrow tiles", {"gemm_narrow": 0}),
                ("register-staged 128x128 tile only", {"gemm_wide": 0, "gemm_big": 0, "gemm_sk": 0, "gemm_narrow": 0, "gemm_quad": 0, "gemm_bm": 128}))
    for name, knobs in variants:
        eng.tune_reset()
        for k, v in knobs.items():
            eng.tune_set(k, v)
        f1, l1 = run()
        assert torch.equal(f1, f0), (name, float((f1.float() - f0.float()).abs().max()))
        assert torch.equal(l1, l0), (name, float((l1 - l0).abs().max()))
    eng.tune_reset()
    # and the default dispatch did use the new tile at these shapes
    A = torch.randn(2168, 4096, generator=g).to(bf).to(DEV)
    W = (torch.randn(4096, 4096, generator=g) * 0.02).to(bf).to(DEV)
    G.gemm(A, W, res=A)
    assert lib.teo_last_kernel().decode() == "gemm_quad_160"


@pytest.mark.parametrize("T,n_out,tag", [(2, 128, "C2"), (8, 256, "C3")])
def test_c2_c3_prefill_and_decode_against_the_oracle_at_full_width(T, n_out, tag, fmt):
    """ViT-L/14 (23 layers) + projector + splice + LLaMA at 7B width, N_LAYERS_DEEP layers: prefill logits of EVERY position and 8
    teacher-forced decode steps against the oracle, bf16 and fp32."""
    _threads()
    t0 = time.perf_counter()
    n_text = 128
    Lseq = n_text - T + 256 * T
    vcfg, lcfg, mm = _oracle_cfgs(N_LAYERS_DEEP)
    m16, sd_dev, cfg = _full_width_model(N_LAYERS_DEEP, bf, Lseq + n_out + 8)
    sd = {k: v.cpu() for k, v in sd_dev.items()}                     # bf16 storage; the oracle upcasts at every use
    frames = O.synthetic_frames(T, 224, seed=0)
    ids = O.synthetic_prompt_ids(n_text, T, 32000, seed=1).unsqueeze(0)
    forced = torch.randint(3, 32000, (8,), generator=torch.Generator().manual_seed(5)).tolist()
    emb_w = sd["model.embed_tokens.weight"].float()
    mask = torch.ones(1, n_text, dtype=torch.long, device=DEV)

    def gpu_run(model, dt):
        imgs = [f.to(DEV, dtype=dt) for f in frames]
        out = model(input_ids=ids.to(DEV), images=imgs, use_cache=True)
        logits = out.logits[0].float().cpu()
        pkv, steps = out.past_key_values, []
        for t in forced:
            _in = model.prepare_inputs_for_generation(torch.tensor([[t]], device=DEV), past_key_values=pkv, images=imgs,
                                                      attention_mask=mask, use_cache=True)
            out = model(**_in)
            pkv = out.past_key_values
            steps.append(out.logits[0, -1].float().cpu())
        return logits, torch.stack(steps)

    def oracle_run(rounding, dtype):
        lg, cache, _ = O.mm_forward(ids, frames, sd, vcfg, lcfg, mm, None, rounding, dtype)
        steps = []
        for t in forced:
            e = emb_w[torch.tensor([[t]])].to(dtype)
            sl, cache = O.llama_forward(e, None, None, cache, sd, lcfg, rounding, decode_kernel=True)
            steps.append(sl[0, -1])
        return lg[0], torch.stack(steps)

    # ---- bf16 engine vs the boundary-rounded oracle
    g_pre, g_dec = gpu_run(m16, bf)
    assert g_pre.shape == (Lseq, 32000)
    o_pre, o_dec = oracle_run(FMT["name"], torch.float32)
    mx, p99, med, sc = _stats(g_pre, o_pre)
    dmx, dp99, dmed, _ = _stats(g_dec, o_dec)
    agree = float((g_pre.argmax(-1) == o_pre.argmax(-1)).float().mean())
    print(f"\n[{tag} {fmt}, {N_LAYERS_DEEP} LLaMA layers at 7B width, L={Lseq}] prefill logits vs oracle({fmt} boundaries), |d|/max|logit| "
          f"(max|logit| {sc:.2f}): max {mx:.2e}  p99 {p99:.2e}  median {med:.2e};  argmax agreement {agree * 100:.1f} % of {Lseq} rows;  "
          f"8 teacher-forced decode steps: max {dmx:.2e}  p99 {dp99:.2e}  median {dmed:.2e}")
    # bars = measured + ~35 % (round 3, MI355X: C2 max 2.26e-2 / p99 8.9e-3 / median 2.1e-3, decode steps 1.05e-2; C3 2.21e-2 / 8.1e-3 /
    # 2.0e-3, decode 8.5e-3).  This is rounding noise, not error: every kernel is within 1 ulp of the oracle on every element
    # (the walks above), ~0.05-0.3 % of a kernel's outputs land on the other side of a bf16 rounding boundary, and 26 layers x ~8
    # kernels of such flips random-walk to a per-logit sigma of ~3e-3 of max|logit| -- the max over 7e7 logits is a 5.7-sigma event.
    # fp16 (round 5, three more mantissa bits; measured on MI355X: see BASELINE.md section 4): the same statement an order of magnitude
    # lower -- and the format in which north_star's 1e-3 is met by the p99 and the median, though not by the maximum over 2e7-7e7 logits
    bars = {"bf16": (3e-2, 1.2e-2, 3e-3, 1.5e-2), "fp16": (4e-3, 1.6e-3, 4e-4, 2.5e-3)}[fmt]
    assert mx < bars[0] and p99 < bars[1] and med < bars[2] and dmx < bars[3], (mx, p99, med, dmx)
    # ---- the control (tests/test_noise_floor.py): the oracle against ITSELF with only the fp32 summation order of its Linear layers
    # changed (K in 8 chunks, descending, vs the single matmul above) -- the HIP path may differ from the oracle by at most 1.5x what
    # the oracle differs from itself by, statistic by statistic.  That is a parity statement, not a regression guard.
    try:
        O.K_ORDER = (8, True)
        s_pre, _, _ = O.mm_forward(ids, frames, sd, vcfg, lcfg, mm, None, FMT["name"], torch.float32)
    finally:
        O.K_ORDER = None
    smx, sp99, smed, _ = _stats(s_pre[0], o_pre)
    print(f"[{tag} {fmt}] oracle vs itself (K order changed, nothing else): max {smx:.2e}  p99 {sp99:.2e}  median {smed:.2e};  "
          f"HIP / self: max {mx / smx:.2f}x  p99 {p99 / sp99:.2f}x  median {med / smed:.2f}x")
    assert mx <= 1.5 * smx and p99 <= 1.5 * sp99 and med <= 1.5 * smed, (mx, smx, p99, sp99, med, smed)
    del m16
    torch.cuda.empty_cache()
    if fmt != "bf16":
        return                                                       # the fp32 leg below does not depend on the 16-bit format: run once
    # ---- fp32 engine (same bf16-valued weights) vs the oracle in fp64: north_star's 1e-5
    m32, _, _ = _full_width_model(N_LAYERS_DEEP, torch.float32, Lseq + n_out + 8, sd=sd_dev)
    g_pre, g_dec = gpu_run(m32, torch.float32)
    o_pre, o_dec = oracle_run(None, torch.float64)
    mx, p99, med, sc = _stats(g_pre, o_pre)
    dmx, _, _, _ = _stats(g_dec, o_dec)
    print(f"[{tag} fp32] prefill logits vs oracle(fp64): max {mx:.2e}  p99 {p99:.2e}  median {med:.2e} of max|logit| {sc:.2f};  "
          f"decode steps max {dmx:.2e};  wall {time.perf_counter() - t0:.1f} s")
    assert mx < 1e-5 and dmx < 1e-5
    assert float((g_pre.argmax(-1) == o_pre.float().argmax(-1)).float().mean()) > 0.995


def test_c3_full_depth_prefill_against_the_oracle():
    """(c) all 32 layers: C3 prefill (ViT-L/14 -> projector -> splice -> LLaMA-2-7B shapes, L = 2168) on the bf16 engine against
    the oracle with bf16 rounding at the kernel boundaries.  ~40 s of CPU."""
    _threads()
    t0 = time.perf_counter()
    T, n_text = 8, 128
    vcfg, lcfg, mm = _oracle_cfgs(32)
    m, sd_dev, cfg = _full_width_model(32, bf, 2304)
    frames = O.synthetic_frames(T, 224, seed=0)
    ids = O.synthetic_prompt_ids(n_text, T, 32000, seed=1).unsqueeze(0)
    imgs = [f.to(DEV, dtype=bf) for f in frames]
    got = m(input_ids=ids.to(DEV), images=imgs).logits[0].float().cpu()
    sd = {k: v.cpu() for k, v in sd_dev.items()}
    del m, sd_dev
    torch.cuda.empty_cache()
    want, _, _ = O.mm_forward(ids, frames, sd, vcfg, lcfg, mm, None, "bf16", torch.float32)
    want = want[0]
    mx, p99, med, sc = _stats(got, want)
    last = float((got[-1] - want[-1]).abs().max()) / float(want[-1].abs().max())
    top2 = torch.topk(want, 2, dim=-1).values
    decisive = (top2[:, 0] - top2[:, 1]) > 2 * mx * sc
    agree = got.argmax(-1) == want.argmax(-1)
    print(f"\n[C3 full depth, 32 layers, L=2168, bf16] prefill logits vs oracle(bf16 boundaries): max {mx:.2e}  p99 {p99:.2e}  median {med:.2e} "
          f"of max|logit| {sc:.2f};  last row {last:.2e};  argmax agreement {float(agree.float().mean()) * 100:.1f} % "
          f"({int(decisive.sum())} rows decisive at 2 x max|d|, all agree: {bool(agree[decisive].all())});  wall {time.perf_counter() - t0:.1f} s")
    # 32-layer random walk of 1-ulp flips on random weights; measured (round 3) max 6.0e-2 / p99 2.0e-2 / median 4.9e-3 + ~35 %
    assert mx < 8e-2 and p99 < 2.7e-2 and med < 7e-3
    assert bool(agree[decisive].all())


# ------------------------------------------------------------------------------------------------------------ X1: w8a8 prefill
def _q_rows(x):
    """(e4m3 bytes [M, K] uint8, scales [M] fp32) of the oracle's per-token quantiser, for feeding the GPU GEMM the oracle's operand."""
    xf = x.float()
    amax = xf.abs().amax(dim=-1, keepdim=True)
    s = torch.where(amax > 0, amax * torch.tensor(1.0 / 448.0), torch.ones_like(amax))
    q = (xf * (1.0 / s)).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).contiguous(), s.reshape(-1).contiguous()


def test_w8a8_prefill_layer_walk_at_7b_shapes():
    """Config C5's "fp8 weight path on CDNA4 MFMA" for the prefill: every Linear layer as a w8a8 GEMM on
    v_mfma_scale_f32_16x16x128_f8f6f4 (per-token e4m3 activations, per-row e4m3 weights with power-of-two scales), at the true
    LLaMA-2-7B shapes, against the oracle's quantised-activation mode (O.quant_rows_e4m3 + the dequantised weights):
      * the quantiser kernel reproduces the oracle's bytes and scales exactly (no norm), and within the 1-ulp norm flips otherwise;
      * each GEMM, fed the ORACLE's quantised operand, lands within one bf16 ulp (+ the instruction's 128-term partial sums:
        measured <= 1.1e-5 of sum|a||w|) on every element, through the production dispatch (128x128 / wide / 256x256 / stream-K).
    This replaces the round-2 end-to-end bound `rms < 0.35`."""
    from teochat_amd.engine import quantize_fp8_rows
    _threads()
    t0 = time.perf_counter()
    report = []
    lib = G.lib()
    S, D, Fi = 2168, 4096, 11008
    gen = torch.Generator().manual_seed(3)
    gw = GemmWs()

    def qgemm(x, W, tag, res=None, flags=0, norm_w=None):
        """x [S, K] bf16-valued (pre-norm if norm_w), W [N, K] fp32 -> GPU w8a8 vs oracle on the same quantised operand."""
        q8, sw, Wdq = quantize_fp8_rows(W.to(bf))
        Wdq = Wdq.float()
        xin = R(O.rmsnorm(x, norm_w, 1e-5)) if norm_w is not None else x
        x_dq = O.quant_rows_e4m3(xin)
        # (1) the device quantiser
        d_x = G.dev(x, bf)
        d_q = torch.empty(S, x.shape[1], dtype=torch.uint8, device=DEV)
        d_s = torch.empty(S, dtype=torch.float32, device=DEV)
        d_nw = G.dev(norm_w, bf) if norm_w is not None else None
        L.check(lib.teo_quant_rows_fp8(G.p(d_x), G.p(d_nw), G.p(d_q), G.p(d_s), S, x.shape[1], x.shape[1], 1e-5, G.stream()), "quant")
        oq, os_ = _q_rows(xin)
        same = float((d_q.cpu() == oq).float().mean())
        srel = float(((d_s.cpu() - os_).abs() / os_).max())
        if norm_w is None:
            assert same == 1.0 and srel == 0.0, (tag, same, srel)
        else:
            assert same > 0.995 and srel < 2.0 ** -7, (tag, same, srel)        # 1-ulp flips of the fused norm move a few bytes
        # (2) the GEMM on the oracle's operand
        A8, sa = oq.to(DEV), os_.to(DEV)
        N = W.shape[0]
        Nc = N // 2 if flags & L.GEMM_SWIGLU16 else N
        out = torch.empty(S, Nc, dtype=bf, device=DEV)
        d_res = G.dev(res, bf) if res is not None else None
        L.check(lib.teo_gemm_fp8_ws(G.p(A8), G.p(sa), G.p(q8.to(DEV)), G.p(sw.to(DEV)), G.p(d_res), G.p(out), S, N, x.shape[1], x.shape[1], Nc,
                                    flags, FMT['dt'], G.p(gw.ws), G.stream()), "gemm_fp8_ws")
        kern = lib.teo_last_kernel().decode()
        return x_dq, Wdq, out, same, kern

    def check(out, ref, tag, same, kern, sum_abs):
        """one exact bf16 ulp + the scaled fp8 MFMA's internal 128-term sums: <= 3e-5 of sum|a||w| (the kernel's stated parity bound,
        DESIGN.md section 5; an fp32 FMA chain would need 1e-7) -- `sum_abs` is that bound per element, computed from the operands"""
        got = out.float().cpu()
        d = (got - ref).abs()
        ulp = G.ulp16(ref, FMT['mant'])
        bad = int((d > ulp + sum_abs + FP32_SUM_ABS).sum())
        worst = float((d / ulp).max())
        report.append(f"  {tag:<44s} kernel={kern:<18s} quantiser bytes equal {same * 100:7.3f} %  worst {worst:5.2f} ulp  beyond 1 ulp + 3e-5 sum|a||w|: {bad} of {d.numel()}")
        assert bad == 0, f"{tag}: {bad} elements beyond tolerance (worst {worst:.2f} ulp, max {float(d.max()):.3e})"

    def sabs(x_dq, Wdq):
        return 3e-5 * (x_dq.abs() @ Wdq.abs().t())

    h = _rand((S, D), gen)
    g_in = R(1.0 + 0.1 * torch.randn(D, generator=gen))
    Wqkv = _rand((3 * D, D), gen, 0.02)
    x_dq, Wdq, out, same, kern = qgemm(h, Wqkv, "qkv", norm_w=g_in)
    check(out, R(x_dq @ Wdq.t()), "rmsnorm + quantise + qkv GEMM N=12288", same, kern, sabs(x_dq, Wdq))
    a = _rand((S, D), gen)
    Wo = _rand((D, D), gen, 0.02)
    x_dq, Wdq, out, same, kern = qgemm(a, Wo, "o", res=h)
    check(out, R(h + x_dq @ Wdq.t()), "quantise + o GEMM + residual", same, kern, sabs(x_dq, Wdq))
    gate, up = _rand((Fi, D), gen, 0.02), _rand((Fi, D), gen, 0.02)
    Wgu = interleave_gate_up(gate, up)
    q8, sw, Wgu_dq = quantize_fp8_rows(Wgu.to(bf))
    g_post = R(1.0 + 0.1 * torch.randn(D, generator=gen))
    x_dq, _, out, same, kern = qgemm(h, Wgu, "gate/up", flags=L.GEMM_SWIGLU16, norm_w=g_post)
    Wf = Wgu_dq.float().view(Fi // 16, 2, 16, D)
    gate_dq, up_dq = Wf[:, 0].reshape(Fi, D), Wf[:, 1].reshape(Fi, D)
    gg, uu = x_dq @ gate_dq.t(), x_dq @ up_dq.t()
    # first-order bound through silu(g) * u (|silu'| <= 1.1)
    check(out, R(F.silu(gg) * uu), "rmsnorm + quantise + gate/up + SwiGLU N=22016", same, kern,
          1.1 * sabs(x_dq, gate_dq) * uu.abs() + sabs(x_dq, up_dq) * F.silu(gg).abs())
    act = _rand((S, Fi), gen, 0.5)
    Wd = _rand((D, Fi), gen, 0.02)
    x_dq, Wdq, out, same, kern = qgemm(act, Wd, "down", res=h)
    check(out, R(h + x_dq @ Wdq.t()), "quantise + down GEMM + residual K=11008", same, kern, sabs(x_dq, Wdq))
    print("\n[w8a8 prefill walk at 7B shapes: device quantiser + fp8 MFMA GEMM vs the oracle's quantised-activation mode]\n" + "\n".join(report)
          + f"\n  wall {time.perf_counter() - t0:.1f} s")


def test_c5_w8a8_prefill_against_the_oracle_at_full_width():
    """w8a8 prefill end to end (engine option prefill_fp8), N_LAYERS_DEEP layers at 7B width, L = 2168, against the oracle with
    act_quant="e4m3" on the engine's dequantised weights: logits statistics reported and bounded like the bf16 leg of (b)."""
    from teochat_amd.config import teochat_7b_config
    from teochat_amd.engine import TeoEngine, quantize_fp8_rows
    from teochat_amd.model import LlavaLlamaForCausalLM
    from teochat_amd.synthetic import synthetic_state_dict
    _threads()
    t0 = time.perf_counter()
    T, n_text = 8, 128
    cfg = teochat_7b_config()
    cfg.num_hidden_layers = N_LAYERS_DEEP
    sd_dev = synthetic_state_dict(cfg, seed=2, std=0.02, dtype=bf, device=DEV)
    eng = TeoEngine(sd_dev, cfg, dtype=bf, device=DEV, max_seq=2304, weight_format="fp8")
    m = LlavaLlamaForCausalLM(cfg, eng)
    frames = O.synthetic_frames(T, 224, seed=0)
    ids = O.synthetic_prompt_ids(n_text, T, 32000, seed=1).unsqueeze(0)
    imgs = [f.to(DEV, dtype=bf) for f in frames]
    lib = eng.lib
    eng.set_options(prefill_fp8=True)
    try:
        got = m(input_ids=ids.to(DEV), images=imgs).logits[0].float().cpu()
    finally:
        eng.set_options(prefill_fp8=False)
    exact = m(input_ids=ids.to(DEV), images=imgs).logits[0].float().cpu()         # bf16 MFMA on the same dequantised weights
    # the oracle sees the engine's weights: Linear layers of the LLM dequantised from their e4m3 rows, everything else as drawn
    sd = {k: v.cpu() for k, v in sd_dev.items()}
    for k in list(sd):
        if k.startswith("model.layers.") and k.endswith("proj.weight") or k == "lm_head.weight":
            sd[k] = quantize_fp8_rows(sd[k])[2]
    vcfg, lcfg, mm = _oracle_cfgs(N_LAYERS_DEEP)
    pix = torch.stack(frames)
    feats = O.encode_images(pix, sd, vcfg, mm, "bf16")
    emb_w = sd["model.embed_tokens.weight"].float()
    _, pos, mask, _, embeds, _ = O.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [feats[i] for i in range(T)], emb_w, mm)
    want, _ = O.llama_forward(embeds, pos, mask, None, sd, lcfg, "bf16", act_quant="e4m3")
    want = want[0]
    mx, p99, med, sc = _stats(got, want)
    emx, ep99, emed, _ = _stats(got, exact)
    print(f"\n[C5 w8a8 prefill, {N_LAYERS_DEEP} layers at 7B width, L=2168] vs oracle(act_quant=e4m3, bf16 boundaries): max {mx:.2e}  p99 {p99:.2e}  "
          f"median {med:.2e} of max|logit| {sc:.2f};  quantisation effect itself (w8a8 vs exact bf16 prefill on the same weights): "
          f"max {emx:.2e}  p99 {ep99:.2e}  median {emed:.2e};  wall {time.perf_counter() - t0:.1f} s")
    # Measured (round 3): max 8.95e-2 / p99 3.55e-2 / median 9.1e-3 -- ~4.5x the bf16 leg of (b).  Expected: every kernel is within
    # 1 ulp of the oracle (walk above, quantiser bytes 100 % equal on identical inputs), but a 1-ulp bf16 flip in front of a quantiser
    # crosses an e4m3 code boundary ~7 % of the time and then moves that activation by a whole e4m3 step (6-12 %): the rounding noise
    # of the bf16 pipeline re-enters ~5x amplified at each of the 4 quantisers per layer.  Bound = measured + ~35 %.
    assert mx < 1.2e-1 and p99 < 5e-2 and med < 1.3e-2
