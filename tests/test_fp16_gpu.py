"""The IEEE binary16 (fp16) path -- the reference's actual inference dtype (videollava/model/builder.py:105 torch_dtype=torch.float16,
eval/inference.py:53 casts the frames to match) -- against the oracle's fp16 rounding mode.

  (a) every arithmetic kernel family at small shapes, fed identical fp16-valued inputs: within ONE exact fp16 ulp (11 significand
      bits; tests/_gpu.py ulp16) of an fp64 reference rounded to fp16 -- MFMA GEMM families (v_mfma_f32_16x16x32_f16), GEMV
      (v_dot2_f32_f16), norms, RoPE, flash attention (v_mfma_f32_32x32x16_f16), decode attention (split pair and whole-context form,
      bit-identical to each other), the batched-decode GEMM in both forms;
  (b) the tiny models end to end: logits against oracle(rounding="fp16") on the fp16-rounded weights -- 3 more mantissa bits than
      bf16: the bound is ~8x tighter than tests/test_model_gpu.py's BF16_REL -- greedy tokens through generate(), batched generate.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import teo_oracle as O
from teochat_amd import _lib as L
from teochat_amd.engine import interleave_gate_up, rope_tables, tile_weights
from tests import _gpu as G
from tests import _tiny as TY

pytestmark = pytest.mark.gpu
h16 = torch.float16
FP16_REL = 2.5e-3       # end to end, tiny models: bf16 measures 0.9e-2 / 1.3e-2 (bar 1.4e-2); 8 more ulps per binade -> ~1.5e-3 expected


def R(t):
    return t.to(h16).to(torch.float32)


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def one_ulp(got, ref, tag, abs_tol=1e-6):
    got, ref = got.float().cpu().reshape(ref.shape), ref.float()
    d = (got - ref).abs()
    ulp = G.ulp16(ref, mant_bits=11)
    bad = int((d > ulp + abs_tol).sum())
    assert bad == 0, f"{tag}: {bad} / {d.numel()} elements beyond 1 fp16 ulp (worst {float((d / ulp).max()):.2f} ulp, max diff {float(d.max()):.3e})"


def test_fp16_norms_and_rope():
    x, w, b = R(rnd(5, 1024, seed=1)), R(1 + 0.1 * rnd(1024, seed=2)), R(rnd(1024, seed=3, scale=0.1))
    one_ulp(G.layernorm(G.dev(x, h16), G.dev(w, h16), G.dev(b, h16), 1e-5), R(F.layer_norm(x, (1024,), w, b, 1e-5)), "layernorm")
    one_ulp(G.rmsnorm(G.dev(x, h16), G.dev(w, h16), 1e-5), R(O.rmsnorm(x, w, 1e-5)), "rmsnorm")
    S, H, hd, S_max = 40, 4, 128, 64
    qkv = R(rnd(S, 3 * H * hd, seed=4))
    cs, sn = rope_tables(hd, 10000.0, S_max)
    pos = torch.arange(S)
    c, s_ = O.rope_cos_sin(pos, hd, 10000.0, torch.float32)
    q, k, v = (qkv[:, i * H * hd:(i + 1) * H * hd].view(S, H, hd) for i in range(3))
    qr = R(q * c[:, None] + O.rotate_half(q) * s_[:, None])
    kr = R(k * c[:, None] + O.rotate_half(k) * s_[:, None])
    d_qkv = G.dev(qkv, h16)
    kc = torch.zeros(H, S_max, hd, dtype=h16, device="cuda")
    vc, vtc = torch.zeros_like(kc), torch.zeros(H, hd, S_max, dtype=h16, device="cuda")
    d_pos, d_cs, d_sn = pos.int().cuda(), cs.cuda(), sn.cuda()        # named: a temporary would be freed before the launch
    L.check(G.lib().teo_rope_kv_append(G.p(d_qkv), 3 * H * hd, G.p(d_pos), G.p(d_cs), G.p(d_sn), G.p(kc), G.p(vc), G.p(vtc),
                                       S, 0, S_max, H, H, hd, L.TEO_F16, G.stream()), "rope")
    one_ulp(d_qkv[:, :H * hd].view(S, H, hd), qr, "rope q")
    one_ulp(kc[:, :S].transpose(0, 1), kr, "rope k")
    assert torch.equal(vc[:, :S].transpose(0, 1).cpu().float(), v) and torch.equal(vtc[:, :, :S].permute(2, 0, 1).cpu().float(), v)


@pytest.mark.parametrize("M,N,K", [(300, 512, 256), (2056, 1024, 1024), (640, 4096, 1024)])
def test_fp16_gemm_families_and_epilogues(M, N, K):
    """every MFMA tile family on fp16 operands (bit-identical to each other: same k-order), bias + GELU / residual / SwiGLU / f32 out."""
    lib = G.lib()
    A, W = R(rnd(M, K, seed=1)), R(rnd(N, K, seed=2, scale=0.05))
    b, r = R(rnd(N, seed=3, scale=0.1)), R(rnd(M, N, seed=4))
    dA, dW, db, dr = (G.dev(t, h16) for t in (A, W, b, r))
    z = A.double() @ W.double().t()
    ws = torch.empty(lib.teo_gemm_workspace_bytes(), dtype=torch.uint8, device="cuda")
    L.check(lib.teo_gemm_workspace_init(G.p(ws), G.stream()), "ws")
    outs = []
    for knobs in ({}, {"gemm_wide": 0, "gemm_big": 0, "gemm_sk": 0}, {"gemm_wide": 2, "gemm_big": 0}, {"gemm_big": 2}, {"gemm_wide": 0, "gemm_big": 0, "gemm_sk": 2}):
        for k_, v_ in knobs.items():
            assert L.tune_set(k_.encode(), v_) == 0
        C = torch.empty(M, N, dtype=h16, device="cuda")
        L.check(lib.teo_gemm_ws(G.p(dA), G.p(dW), G.p(db), G.p(dr), G.p(C), M, N, K, K, N, 0, 0, L.TEO_F16, L.TEO_F16, G.p(ws), G.stream()), "gemm")
        outs.append((lib.teo_last_kernel().decode(), C.clone()))
        L.tune_reset()
    assert all(k.startswith("gemm_") and k != "gemm_simple" for k, _ in outs), [k for k, _ in outs]
    for k, C in outs:
        one_ulp(C, R((z + b.double() + r.double()).float()), f"gemm + bias + residual [{k}]", abs_tol=2e-5)
        assert torch.equal(C, outs[0][1]), k
    one_ulp(G.gemm(dA, dW, bias=db, act=L.ACT_GELU_ERF), R(F.gelu((z + b.double()).float())), "gemm + bias + gelu", abs_tol=2e-5)
    torch.testing.assert_close(G.gemm(dA, dW, out_dtype=torch.float32).cpu(), z.float(), atol=3e-4, rtol=1e-5)
    if N % 32 == 0:
        gate, up = W[:N // 2], W[N // 2:]
        got = G.gemm(dA, G.dev(interleave_gate_up(gate, up), h16), flags=L.GEMM_SWIGLU16)
        one_ulp(got, R(F.silu((A.double() @ gate.double().t()).float()) * (A.double() @ up.double().t()).float()), "gate/up SwiGLU", abs_tol=2e-5)
    # the generic kernel on the same operands (fp16 element type through Elem<f16_t>)
    Cs = torch.empty(M, N, dtype=h16, device="cuda")
    L.check(lib.teo_gemm(G.p(dA), G.p(dW), G.p(db), G.p(dr), G.p(Cs), M, N, K, K, N, 0, L.GEMM_FORCE_SIMPLE, L.TEO_F16, L.TEO_F16, G.stream()), "gemm simple")
    one_ulp(Cs, R((z + b.double() + r.double()).float()), "gemm_simple", abs_tol=2e-5)


def test_fp16_gemv_family():
    D, Fi = 1024, 2816
    x, nw = R(rnd(D, seed=1)), R(1 + 0.1 * rnd(D, seed=2))
    Wq = R(rnd(3 * D, D, seed=3, scale=0.03))
    xn = R(O.rmsnorm(x, nw, 1e-5))
    one_ulp(G.gemv(G.dev(x, h16), G.dev(Wq, h16), norm_w=G.dev(nw, h16)), R((Wq.double() @ xn.double()).float()), "gemv rmsnorm + qkv", abs_tol=2e-5)
    gate, up = R(rnd(Fi, D, seed=4, scale=0.03)), R(rnd(Fi, D, seed=5, scale=0.03))
    y = G.gemv(G.dev(x, h16), G.dev(interleave_gate_up(gate, up), h16), norm_w=G.dev(nw, h16), flags=L.GEMM_SWIGLU16)
    one_ulp(y, R(F.silu((gate.double() @ xn.double()).float()) * (up.double() @ xn.double()).float()), "gemv gate/up SwiGLU", abs_tol=2e-5)
    a, Wd, res = R(rnd(Fi, seed=6, scale=0.5)), R(rnd(D, Fi, seed=7, scale=0.03)), R(rnd(D, seed=8))
    one_ulp(G.gemv(G.dev(a, h16), G.dev(Wd, h16), res=G.dev(res, h16)), R((res.double() + Wd.double() @ a.double()).float()), "gemv down + residual (split-K)",
            abs_tol=2e-5)
    lg = G.gemv(G.dev(x, h16), G.dev(Wq, h16), norm_w=G.dev(nw, h16), out_dtype=torch.float32)
    torch.testing.assert_close(lg.cpu(), (Wq.double() @ xn.double()).float(), atol=2e-4, rtol=1e-5)


def _p_noise(q, k, v, visible, scale):
    return (2.0 ** -11) * O.attention_core(q, k, v.abs(), visible, scale, lambda t: t, "exact")      # P rounded to 11 significand bits


@pytest.mark.parametrize("H,d,S", [(4, 128, 300), (8, 64, 257)])
def test_fp16_flash_attention(H, d, S):
    causal = d == 128
    q, k, v = (R(rnd(1, H, S, d, seed=i)) for i in (1, 2, 3))
    vis = (torch.arange(S).view(1, S) <= torch.arange(S).view(S, 1)).view(1, 1, S, S) if causal else None
    ref = R(O.attention_core(q, k, v, vis, d ** -0.5, R, "flash64"))
    qd, kd, vd = (G.dev(t.contiguous(), h16) for t in (q, k, v))
    got = G.attention(qd, kd, vd, causal, d ** -0.5, vt=G.make_vt(vd))
    assert G.lib().teo_last_kernel() == b"attn_flash32"
    one_ulp(got.view(1, S, H, d).transpose(1, 2), ref, "flash attention", abs_tol=_p_noise(q, k, v, vis, d ** -0.5) + 1e-6)


@pytest.mark.parametrize("rope", [False, True])
def test_fp16_decode_attention_split_and_whole(rope):
    H, d, S = 32, 128, 1024
    ctx = [700, 1024, 65, 1, 300, 999, 512, 640]
    B = len(ctx)
    lib = G.lib()
    g = torch.Generator().manual_seed(5 + int(rope))
    K, V = R(torch.randn(B, H, S, d, generator=g)), R(torch.randn(B, H, S, d, generator=g))
    qkv = R(torch.randn(B, 3 * H, d, generator=g))
    cs, sn = rope_tables(d, 10000.0, S)
    d_cs, d_sn = cs.cuda(), sn.cuda()
    pos = torch.tensor([n - 1 for n in ctx], dtype=torch.int32, device="cuda")
    part = torch.empty(lib.teo_attn_decode_workspace_bytes(H, d, S, B), dtype=torch.uint8, device="cuda")
    outs = {}
    for whole in (2, 0):
        dK, dV = G.dev(K, h16), G.dev(V, h16)
        dVT = torch.zeros(B, H, d, S, dtype=h16, device="cuda")
        out = torch.zeros(B, H * d, dtype=h16, device="cuda")
        dq = (qkv if rope else qkv[:, :H]).reshape(B, -1).to("cuda", h16).contiguous()
        assert L.tune_set(b"attn_whole", whole) == 0 and L.tune_set(b"attn_chunk", 64) == 0
        L.check(lib.teo_attn_decode(G.p(dq), G.p(dK), G.p(dV), G.p(dVT) if rope else None, G.p(d_cs) if rope else None,
                                    G.p(d_sn) if rope else None, G.p(out), G.p(part), G.p(pos), S, H, H, d, d ** -0.5, L.TEO_F16, B,
                                    dq.shape[1], H * S * d, H * d, G.stream()), "attn_decode")
        torch.cuda.synchronize()
        outs[whole] = out.clone()
    L.tune_reset()
    assert torch.equal(outs[2], outs[0])
    for b, n in enumerate(ctx):
        if rope:
            c, s_ = torch.cat([cs[n - 1]] * 2), torch.cat([sn[n - 1]] * 2)
            q_rot = R(qkv[b, :H] * c + O.rotate_half(qkv[b, :H]) * s_)
            Kb, Vb = K[b].clone(), V[b].clone()
            Kb[:, n - 1] = R(qkv[b, H:2 * H] * c + O.rotate_half(qkv[b, H:2 * H]) * s_)
            Vb[:, n - 1] = qkv[b, 2 * H:]
        else:
            q_rot, Kb, Vb = qkv[b, :H], K[b], V[b]
        qq, kk, vv = q_rot[None, :, None, :], Kb[None, :, :n], Vb[None, :, :n]
        ref = R(O.attention_core(qq, kk, vv, None, d ** -0.5, R, "split64"))[0, :, 0].reshape(-1)
        one_ulp(outs[2][b], ref, f"decode attention ctx {n}", abs_tol=_p_noise(qq, kk, vv, None, d ** -0.5)[0, :, 0].reshape(-1) + 1e-6)


@pytest.mark.parametrize("MB", [3, 8, 16])
def test_fp16_skinny_gemm_both_forms(MB):
    """the batched-decode GEMM with fp16 activations / weights: tile kernel == streaming form (bit-identical at K = 4096), both within
    one fp16 ulp of the fp64 product; SwiGLU8, residual, f32 output; fp8 weights are refused with fp16 activations."""
    from teochat_amd.engine import reinterleave_gate_up
    lib = G.lib()
    K, N = 4096, 2048
    x, W, r = R(rnd(MB, K, seed=1)), R(rnd(N, K, seed=2, scale=0.02)), R(rnd(MB, N, seed=3))
    dx, dW, dr = G.dev(x, h16), G.dev(W, h16), G.dev(r, h16)
    z = x.double() @ W.double().t()
    outs = {}
    for mode in (0, 2):
        assert L.tune_set(b"skinny_stream", mode) == 0
        outs[mode] = (G.gemm_skinny(dx, tile_weights(dW), res=dr, flags=L.GEMM_WTILED, N=N), G.gemm_skinny(dx, dW, flags=L.GEMM_F16, out_dtype=torch.float32))
        assert lib.teo_last_kernel().startswith(b"skinny_stream" if mode else b"skinny_gemm")
    L.tune_reset()
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])
    one_ulp(outs[2][0], R((z + r.double()).float()), "skinny + residual", abs_tol=2e-5)
    torch.testing.assert_close(outs[2][1].cpu(), z.float(), atol=3e-4, rtol=1e-5)
    gate, up = W[:N // 2], W[N // 2:]
    gu8 = reinterleave_gate_up(interleave_gate_up(gate, up).to(h16).cuda(), 8)
    got = G.gemm_skinny(dx, tile_weights(gu8), flags=L.GEMM_SWIGLU8 | L.GEMM_WTILED, N=N)
    one_ulp(got, R(F.silu((x.double() @ gate.double().t()).float()) * (x.double() @ up.double().t()).float()), "skinny SwiGLU8", abs_tol=2e-5)
    q8 = torch.zeros(N, K, dtype=torch.uint8, device="cuda")
    out = torch.empty(MB, N, dtype=h16, device="cuda")
    rc = lib.teo_gemm_skinny(G.p(dx), G.p(q8), G.p(torch.ones(N, device="cuda")), 1, None, 1e-5, None, G.p(out), MB, N, K, K, N, 0, L.TEO_F16, G.stream())
    assert rc != 0 and b"fp8" in lib.teo_last_error()


def _build(name, dtype):
    from tests.test_model_gpu import build
    return build(name, dtype)


@pytest.mark.parametrize("name", ["tinyA", "tinyB"])
def test_fp16_tiny_models_against_the_oracle_fp16_mode(name):
    """prefill logits of every position and 4 greedy tokens: fp16 engine vs oracle(rounding="fp16") on the fp16-rounded weights;
    the same run in bf16 for scale (3 fewer mantissa bits)."""
    g = TY.load_npz(name)
    vcfg, lcfg, mm = TY.cfgs(name)
    sd = TY.state_dict(name)
    frames = O.synthetic_frames(int(g["T"]), vcfg.image_size, seed=0)
    ids = torch.from_numpy(g["input_ids"])
    errs = {}
    for dt, rounding in ((h16, "fp16"), (torch.bfloat16, "bf16")):
        sdr = {k: v.to(dt).float() for k, v in sd.items()}
        want, _, _ = O.mm_forward(ids, frames, sdr, vcfg, lcfg, mm, None, rounding, torch.float32)
        model, _ = _build(name, dt)
        imgs = [f.to("cuda:0", dtype=dt) for f in frames]
        out = model(input_ids=ids.cuda(), images=imgs)
        errs[rounding] = float((out.logits[0].cpu().float() - want[0]).abs().max()) / float(want[0].abs().max())
        # the bar: the oracle's own self-difference at these rounding points, computed here (tests/test_model_gpu.py SELF_DIFF_FACTOR)
        floor = O.self_difference(ids, frames, sdr, vcfg, lcfg, mm, rounding, base=want[0])
        mine = O.logit_stats(out.logits[0].cpu().float(), want[0])
        print(f"\n[{name}] {rounding} HIP vs oracle max / p99 / median {mine[0]:.2e} / {mine[1]:.2e} / {mine[2]:.2e}; oracle vs itself "
              f"{floor[0]:.2e} / {floor[1]:.2e} / {floor[2]:.2e}")
        assert all(m <= 1.25 * f for m, f in zip(mine, floor)), (rounding, mine, floor)
        if dt == h16:
            ref_tokens, _, _ = O.greedy_generate(ids, frames, sdr, vcfg, lcfg, mm, max_new_tokens=4, rounding="fp16")
            gen = model.generate(input_ids=ids.cuda(), images=imgs, do_sample=False, max_new_tokens=4, eos_token_id=None)
            toks = gen[0, ids.shape[1]:].tolist()
            top2 = torch.topk(want[0, -1], 2).values
            if float(top2[0] - top2[1]) > 4 * errs["fp16"] * float(want[0].abs().max()):
                assert toks[0] == ref_tokens[0]
            both = model.generate_batch([ids[0].cuda(), ids[0].cuda()], [imgs, imgs], do_sample=False, max_new_tokens=4, eos_token_id=None)
            assert both[0].tolist() == both[1].tolist() and len(both[0]) >= 4
    print(f"\n[{name}] end-to-end logits vs the oracle at the same rounding points, max|d| / max|logit|: fp16 {errs['fp16']:.2e}   bf16 {errs['bf16']:.2e}")
    assert errs["fp16"] < 0.5 * errs["bf16"]            # 3 more mantissa bits: well below the bf16 drift (expected ~1/8)
