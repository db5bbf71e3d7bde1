"""Thin test helpers: call libteo_hip.so primitives on torch CUDA tensors through the C ABI (ctypes)."""
import ctypes as C

import torch

from teochat_amd import _lib as L

DT = {torch.float32: L.TEO_F32, torch.bfloat16: L.TEO_BF16, torch.float16: L.TEO_F16}


def lib():
    return L.load()


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def dev(t, dtype=None):
    return t.to(device="cuda", dtype=dtype or t.dtype).contiguous()


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def ulp16(ref, mant_bits=8):
    """Exact unit in the last place of a 16-bit float format at |ref| (bf16: 8 significand bits incl. the hidden one; fp16: 11):
    x = m * 2^e with m in [0.5, 1) (torch.frexp) sits in the binade [2^(e-1), 2^e) whose spacing is 2^(e - mant_bits).
    Below 2^-14 the spacing of that binade is used (a floor far below anything the path produces, not a tolerance knob)."""
    _, e = torch.frexp(ref.float().abs().clamp_min(2.0 ** -14))
    return torch.ldexp(torch.ones_like(ref, dtype=torch.float32), e - mant_bits)


def ulps_off(got, ref, mant_bits=8):
    """|got - ref| in exact ulps of ref (fp32 tensors, same shape)."""
    return (got.float() - ref.float()).abs() / ulp16(ref, mant_bits)


def layernorm(x, w, b, eps):
    y = torch.empty_like(x)
    L.check(lib().teo_layernorm(p(x), p(w), p(b), p(y), x.shape[0], x.shape[1], eps, DT[x.dtype], stream()), "layernorm")
    return y


def rmsnorm(x, w, eps):
    y = torch.empty_like(x)
    L.check(lib().teo_rmsnorm(p(x), p(w), p(y), x.shape[0], x.shape[1], eps, DT[x.dtype], stream()), "rmsnorm")
    return y


def gemm(A, W, bias=None, res=None, act=L.ACT_NONE, flags=0, out_dtype=None):
    M, K = A.shape
    N = W.shape[0]
    out_dtype = out_dtype or A.dtype
    Nc = N // 2 if flags & L.GEMM_SWIGLU16 else N
    Cc = torch.empty(M, Nc, dtype=out_dtype, device=A.device)
    L.check(lib().teo_gemm(p(A), p(W), p(bias), p(res), p(Cc), M, N, K, A.stride(0), Nc, act, flags, DT[A.dtype],
                           DT[out_dtype], stream()), "gemm")
    return Cc


def gemm_skinny(x, W, scale=None, res=None, flags=0, out_dtype=None, N=None, norm_w=None, eps=1e-5):
    """x [MB,K] bf16; W [N,K] bf16 or uint8 (fp8 e4m3 bits) with scale [N] fp32 (N: logical rows of a tiled W)."""
    MB, K = x.shape
    N = N or W.shape[0]
    out_dtype = out_dtype or x.dtype
    Nc = N // 2 if flags & (L.GEMM_SWIGLU16 | L.GEMM_SWIGLU8) else N
    out = torch.empty(MB, Nc, dtype=out_dtype, device=x.device)
    L.check(lib().teo_gemm_skinny(p(x), p(W), p(scale), 1 if scale is not None else 0, p(norm_w), eps, p(res), p(out), MB, N, K,
                                  x.stride(0), Nc,
                                  flags, DT[out_dtype], stream()), "gemm_skinny")
    return out


def gemv(x, W, norm_w=None, res=None, eps=1e-5, flags=0, out_dtype=None):
    N, K = W.shape
    out_dtype = out_dtype or x.dtype
    Ny = N // 2 if flags & L.GEMM_SWIGLU16 else N
    y = torch.empty(Ny, dtype=out_dtype, device=x.device)
    L.check(lib().teo_gemv(p(x), p(W), p(norm_w), p(res), p(y), N, K, eps, flags, DT[x.dtype], DT[out_dtype], stream()), "gemv")
    return y


def attention(q, k, v, causal, scale, vt=None, force_simple=False):
    """q [B,H,Sq,d], k/v [B,Hk,Sk,d] contiguous; vt [B,Hk,d,ldv] optional.  Returns [B,Sq,H*d]."""
    B, H, Sq, d = q.shape
    Hk, Sk = k.shape[1], k.shape[2]
    o = torch.empty(B, Sq, H * d, dtype=q.dtype, device=q.device)
    a = L.AttnArgs()
    a.q, a.k, a.v, a.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr()
    a.vt = vt.data_ptr() if vt is not None else None
    a.q_bs, a.q_hs, a.q_rs = q.stride(0), q.stride(1), q.stride(2)
    a.k_bs, a.k_hs, a.k_rs = k.stride(0), k.stride(1), k.stride(2)
    a.v_bs, a.v_hs, a.v_rs = v.stride(0), v.stride(1), v.stride(2)
    if vt is not None:
        a.vt_bs, a.vt_hs, a.vt_rs = vt.stride(0), vt.stride(1), vt.stride(2)
    a.o_bs, a.o_rs = o.stride(0), o.stride(1)
    a.batch, a.heads, a.kv_heads, a.head_dim, a.q_len, a.kv_len = B, H, Hk, d, Sq, Sk
    a.causal, a.scale = int(causal), scale
    a.flags = L.ATTN_FORCE_SIMPLE if force_simple else 0
    L.check(lib().teo_attention(C.byref(a), DT[q.dtype], stream()), "attention")
    return o


def make_vt(v, ldv=None):
    """[B,Hk,S,d] -> zero padded [B,Hk,d,ldv] with ldv a multiple of 64."""
    B, Hk, S, d = v.shape
    ldv = ldv or (S + 63) // 64 * 64
    vt = torch.zeros(B, Hk, d, ldv, dtype=v.dtype, device=v.device)
    vt[..., :S] = v.transpose(2, 3)
    return vt
