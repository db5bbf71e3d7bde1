"""tinyB LLaMA layer 0, primitive by primitive, bf16 kernels vs the boundary oracle on identical inputs."""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import teo_oracle as O
from teochat_amd import _lib as L
from teochat_amd.engine import interleave_gate_up, rope_tables
from tests import _gpu as G, _tiny as TY

bf = torch.bfloat16
R = lambda t: t.to(bf).float()


def rel(a, b, tag):
    a = a.float().cpu(); d = (a - b).abs()
    ulp = (2.0 ** -7) * b.abs() + 1e-3
    print(f"{tag:28s} max|d|/max|ref| {float(d.max()) / float(b.abs().max()):.2e}   elements beyond 1 ulp: {int((d > ulp).sum())} / {d.numel()}")


vcfg, lcfg, mm = TY.cfgs("tinyB")
sd = {k: R(v) for k, v in TY.state_dict("tinyB").items()}
S, D, H, hd, Fd = 534, lcfg.hidden_size, lcfg.num_attention_heads, lcfg.head_dim, lcfg.intermediate_size
g = torch.Generator().manual_seed(0)
h = R(torch.randn(S, D, generator=g))
p = "model.layers.0."
n1 = R(O.rmsnorm(h, sd[p + "input_layernorm.weight"], 1e-5))
k_n1 = G.rmsnorm(G.dev(h, bf), G.dev(sd[p + "input_layernorm.weight"], bf), 1e-5)
rel(k_n1, n1, "rmsnorm")
Wqkv = torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
qkv = R(n1 @ Wqkv.t())
k_qkv = G.gemm(G.dev(n1, bf), G.dev(Wqkv, bf))
rel(k_qkv, qkv, "qkv gemm (mfma)")
k_qkv_s = G.gemm(G.dev(n1, bf), G.dev(Wqkv, bf), flags=L.GEMM_FORCE_SIMPLE)
rel(k_qkv_s, qkv, "qkv gemm (generic)")
# rope
pos = torch.arange(S)
c, s_ = O.rope_cos_sin(pos, hd, 10000.0, torch.float32)
q = qkv[:, :H * hd].view(S, H, hd); k = qkv[:, H * hd:2 * H * hd].view(S, H, hd); v = qkv[:, 2 * H * hd:].view(S, H, hd)
qr = R(q * c[:, None] + O.rotate_half(q) * s_[:, None]); kr = R(k * c[:, None] + O.rotate_half(k) * s_[:, None])
cs, sn = rope_tables(hd, 10000.0, 1024)
d_qkv = G.dev(qkv, bf); S_max = 576
kc = torch.zeros(H, S_max, hd, dtype=bf, device="cuda"); vc = torch.zeros_like(kc); vtc = torch.zeros(H, hd, S_max, dtype=bf, device="cuda")
d_pos, d_cs, d_sn = pos.int().cuda(), cs.cuda(), sn.cuda()
L.check(G.lib().teo_rope_kv_append(G.p(d_qkv), 3 * H * hd, G.p(d_pos), G.p(d_cs), G.p(d_sn), G.p(kc), G.p(vc), G.p(vtc), S, 0, S_max, H, H, hd, L.TEO_BF16, G.stream()), "rope")
rel(d_qkv[:, :H * hd].view(S, H, hd), qr, "rope q")
rel(kc[:, :S].transpose(0, 1), kr, "rope k (cache)")
# attention
vis = (torch.arange(S).view(1, S) <= torch.arange(S).view(S, 1)).view(1, 1, S, S)
qq, kk, vv = qr.transpose(0, 1)[None], kr.transpose(0, 1)[None], v.transpose(0, 1)[None]
o_ref = R(O.attention_core(qq, kk, vv, vis, 1 / math.sqrt(hd), R, "flash64").transpose(1, 2).reshape(1, S, H * hd))[0]
a = L.AttnArgs()
o_k = torch.empty(S, H * hd, dtype=bf, device="cuda")
dq = G.dev(qr.reshape(S, H * hd), bf)
a.q, a.k, a.v, a.vt, a.o = dq.data_ptr(), kc.data_ptr(), vc.data_ptr(), vtc.data_ptr(), o_k.data_ptr()
a.q_hs, a.q_rs = hd, H * hd
a.k_hs, a.k_rs = S_max * hd, hd
a.v_hs, a.v_rs = S_max * hd, hd
a.vt_hs, a.vt_rs = hd * S_max, S_max
a.o_rs = H * hd
a.batch, a.heads, a.kv_heads, a.head_dim, a.q_len, a.kv_len, a.causal, a.scale = 1, H, H, hd, S, S, 1, 1 / math.sqrt(hd)
import ctypes as C
L.check(G.lib().teo_attention(C.byref(a), L.TEO_BF16, G.stream()), "attn")
rel(o_k, o_ref, "attention (mfma, cache)")
# o proj + residual
h1 = R(h + o_ref @ sd[p + "self_attn.o_proj.weight"].t())
k_h1 = G.gemm(G.dev(o_ref, bf), G.dev(sd[p + "self_attn.o_proj.weight"], bf), res=G.dev(h, bf))
rel(k_h1, h1, "o_proj + residual (mfma)")
n2 = R(O.rmsnorm(h1, sd[p + "post_attention_layernorm.weight"], 1e-5))
gate, up = sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]
act = R(torch.nn.functional.silu(n2 @ gate.t()) * (n2 @ up.t()))
k_act = G.gemm(G.dev(n2, bf), G.dev(interleave_gate_up(gate, up), bf), flags=L.GEMM_SWIGLU16)
rel(k_act, act, "gate/up swiglu (mfma)")
h2 = R(h1 + act @ sd[p + "mlp.down_proj.weight"].t())
k_h2 = G.gemm(G.dev(act, bf), G.dev(sd[p + "mlp.down_proj.weight"], bf), res=G.dev(h1, bf))
rel(k_h2, h2, "down + residual (mfma)")
