"""bf16 parity, stated rigorously: a TEACHER-FORCED walk through one LLaMA layer and one ViT layer at MFMA-friendly
dims.  Every kernel receives the oracle's (bf16-rounded) input for that stage, so single-ulp flips cannot compound, and
its output must equal the oracle's output rounded to bf16 within ONE bf16 ulp on EVERY element.

(End to end, two correct bf16 pipelines that differ only in fp32 summation order drift apart by ~1e-2 of max|logit|
on these deliberately sensitive random tiny models -- the same size as the reference's own bf16-vs-fp32 deviation;
tests/test_model_gpu.py bounds that loosely.  This file is the tight statement.)"""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import teo_oracle as O
from teochat_amd import _lib as L
from teochat_amd.engine import interleave_gate_up, rope_tables
from tests import _gpu as G
from tests import _tiny as TY

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


def R(t):
    return t.to(bf).float()


def p_noise(q, k, v, visible, scale):
    """allowance for the bf16 rounding of P before the PV product (both sides round P, each at its own 1-ulp-different value):
    2^-8 * sum_j p_j |v_j| per output, whatever the output's own magnitude -- see tests/test_true_shapes_gpu.py attn_p_noise"""
    return (2.0 ** -8) * O.attention_core(q, k, v.abs(), visible, scale, lambda t: t, "exact")


def within_one_ulp(got, ref, tag, abs_tol=0.0):
    got = got.float().cpu()
    d = (got - ref).abs()
    ulp = G.ulp16(ref)                                       # exact ulp of the oracle's value (frexp), + 1e-5 for fp32 summation order
    bad = int((d > ulp + 1e-5 + abs_tol).sum())
    assert bad == 0, f"{tag}: {bad} / {d.numel()} elements beyond 1 bf16 ulp (worst {float((d / ulp).max()):.2f} ulp, max diff {float(d.max()):.3e})"


def test_llama_layer_walk_bf16_mfma():
    vcfg, lcfg, mm = TY.cfgs("tinyB")
    sd = {k: R(v) for k, v in TY.state_dict("tinyB").items()}
    S, D, H, hd = 534, lcfg.hidden_size, lcfg.num_attention_heads, lcfg.head_dim
    g = torch.Generator().manual_seed(0)
    h = R(torch.randn(S, D, generator=g))
    p = "model.layers.0."
    n1 = R(O.rmsnorm(h, sd[p + "input_layernorm.weight"], 1e-5))
    within_one_ulp(G.rmsnorm(G.dev(h, bf), G.dev(sd[p + "input_layernorm.weight"], bf), 1e-5), n1, "rmsnorm")
    Wqkv = torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
    qkv = R(n1 @ Wqkv.t())
    assert G.lib().teo_gemm_uses_mfma(S, 3 * H * hd, D, L.TEO_BF16, 0) == 1
    within_one_ulp(G.gemm(G.dev(n1, bf), G.dev(Wqkv, bf)), qkv, "qkv gemm")
    pos = torch.arange(S)
    c, s_ = O.rope_cos_sin(pos, hd, 10000.0, torch.float32)
    q, k, v = (qkv[:, i * H * hd:(i + 1) * H * hd].view(S, H, hd) for i in range(3))
    qr = R(q * c[:, None] + O.rotate_half(q) * s_[:, None])
    kr = R(k * c[:, None] + O.rotate_half(k) * s_[:, None])
    cs, sn = rope_tables(hd, 10000.0, 1024)
    d_qkv, S_max = G.dev(qkv, bf), 576
    kc = torch.zeros(H, S_max, hd, dtype=bf, device="cuda")
    vc, vtc = torch.zeros_like(kc), torch.zeros(H, hd, S_max, dtype=bf, device="cuda")
    d_pos, d_cs, d_sn = pos.int().cuda(), cs.cuda(), sn.cuda()
    L.check(G.lib().teo_rope_kv_append(G.p(d_qkv), 3 * H * hd, G.p(d_pos), G.p(d_cs), G.p(d_sn), G.p(kc), G.p(vc), G.p(vtc), S, 0,
                                       S_max, H, H, hd, L.TEO_BF16, G.stream()), "rope")
    within_one_ulp(d_qkv[:, :H * hd].view(S, H, hd), qr, "rope q")
    within_one_ulp(kc[:, :S].transpose(0, 1), kr, "rope k")
    vis = (torch.arange(S).view(1, S) <= torch.arange(S).view(S, 1)).view(1, 1, S, S)
    qq, kk, vv = qr.transpose(0, 1)[None], kr.transpose(0, 1)[None], v.transpose(0, 1)[None]
    o_ref = R(O.attention_core(qq, kk, vv, vis, 1 / math.sqrt(hd), R, "flash64").transpose(1, 2).reshape(S, H * hd))
    a = L.AttnArgs()
    o_k = torch.empty(S, H * hd, dtype=bf, device="cuda")
    dq = G.dev(qr.reshape(S, H * hd), bf)
    a.q, a.k, a.v, a.vt, a.o = dq.data_ptr(), kc.data_ptr(), vc.data_ptr(), vtc.data_ptr(), o_k.data_ptr()
    a.q_hs, a.q_rs, a.k_hs, a.k_rs, a.v_hs, a.v_rs = hd, H * hd, S_max * hd, hd, S_max * hd, hd
    a.vt_hs, a.vt_rs, a.o_rs = hd * S_max, S_max, H * hd
    a.batch, a.heads, a.kv_heads, a.head_dim, a.q_len, a.kv_len, a.causal, a.scale = 1, H, H, hd, S, S, 1, 1 / math.sqrt(hd)
    L.check(G.lib().teo_attention(C.byref(a), L.TEO_BF16, G.stream()), "attn")
    within_one_ulp(o_k, o_ref, "flash attention", p_noise(qq, kk, vv, vis, 1 / math.sqrt(hd)).transpose(1, 2).reshape(S, H * hd))
    Wo = sd[p + "self_attn.o_proj.weight"]
    h1 = R(h + o_ref @ Wo.t())
    within_one_ulp(G.gemm(G.dev(o_ref, bf), G.dev(Wo, bf), res=G.dev(h, bf)), h1, "o_proj + residual")
    n2 = R(O.rmsnorm(h1, sd[p + "post_attention_layernorm.weight"], 1e-5))
    gate, up = sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]
    act = R(F.silu(n2 @ gate.t()) * (n2 @ up.t()))
    within_one_ulp(G.gemm(G.dev(n2, bf), G.dev(interleave_gate_up(gate, up), bf), flags=L.GEMM_SWIGLU16), act, "gate/up SwiGLU")
    Wd = sd[p + "mlp.down_proj.weight"]
    within_one_ulp(G.gemm(G.dev(act, bf), G.dev(Wd, bf), res=G.dev(h1, bf)), R(h1 + act @ Wd.t()), "down + residual")
    # decode-step kernels on the same layer: GEMV forms agree with the oracle too
    x = h1[-1]
    xn = R(O.rmsnorm(x, sd[p + "post_attention_layernorm.weight"], 1e-5))
    y = G.gemv(G.dev(x, bf), G.dev(interleave_gate_up(gate, up), bf), norm_w=G.dev(sd[p + "post_attention_layernorm.weight"], bf),
               flags=L.GEMM_SWIGLU16)
    within_one_ulp(y, R(F.silu(gate @ xn) * (up @ xn)), "gemv rmsnorm + gate/up SwiGLU")
    within_one_ulp(G.gemv(G.dev(act[-1], bf), G.dev(Wd, bf), res=G.dev(x, bf)), R(x + Wd @ act[-1]), "gemv down + residual")


def test_vit_layer_walk_bf16_mfma():
    vcfg, lcfg, mm = TY.cfgs("tinyB")
    sd = {k: R(v) for k, v in TY.state_dict("tinyB").items()}
    T, N, D, H = 2, 257, vcfg.hidden_size, vcfg.num_attention_heads
    hd = D // H
    g = torch.Generator().manual_seed(1)
    h = R(torch.randn(T * N, D, generator=g))
    p = O.VIT_PREFIX + "encoder.layers.0."
    ln1 = R(F.layer_norm(h, (D,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], 1e-5))
    within_one_ulp(G.layernorm(G.dev(h, bf), G.dev(sd[p + "layer_norm1.weight"], bf), G.dev(sd[p + "layer_norm1.bias"], bf), 1e-5),
                   ln1, "layernorm")
    Wqkv = torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
    bqkv = torch.cat([sd[p + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0)
    qkv = R(ln1 @ Wqkv.t() + bqkv)
    k_qkv = G.gemm(G.dev(ln1, bf), G.dev(Wqkv, bf), bias=G.dev(bqkv, bf))
    within_one_ulp(k_qkv, qkv, "qkv gemm + bias")
    q, k, v = (qkv[:, i * D:(i + 1) * D].view(T, N, H, hd).transpose(1, 2) for i in range(3))
    o_ref = R(O.attention_core(q, k, v, None, hd ** -0.5, R, "flash64").transpose(1, 2).reshape(T * N, D))
    qd, kd, vd = (G.dev(t.contiguous(), bf) for t in (q, k, v))
    o_k = G.attention(qd, kd, vd, False, hd ** -0.5, vt=G.make_vt(vd))
    within_one_ulp(o_k.reshape(T * N, D), o_ref, "flash attention (N=257, d=64)", p_noise(q, k, v, None, hd ** -0.5).transpose(1, 2).reshape(T * N, D))
    Wo, bo = sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"]
    h1 = R(h + o_ref @ Wo.t() + bo)
    within_one_ulp(G.gemm(G.dev(o_ref, bf), G.dev(Wo, bf), bias=G.dev(bo, bf), res=G.dev(h, bf)), h1, "out_proj + bias + residual")
    ln2 = R(F.layer_norm(h1, (D,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], 1e-5))
    W1, b1, W2, b2 = (sd[p + n] for n in ("mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias"))
    m = R(F.gelu(ln2 @ W1.t() + b1))
    within_one_ulp(G.gemm(G.dev(ln2, bf), G.dev(W1, bf), bias=G.dev(b1, bf), act=L.ACT_GELU_ERF), m, "fc1 + bias + gelu")
    within_one_ulp(G.gemm(G.dev(m, bf), G.dev(W2, bf), bias=G.dev(b2, bf), res=G.dev(h1, bf)), R(h1 + m @ W2.t() + b2), "fc2 + bias + residual")
