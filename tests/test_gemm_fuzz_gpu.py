"""Every 16-bit GEMM tile family against the register-staged 128 x 128 kernel on RANDOM shapes: the bit-identity claim of DESIGN section 5
(same LDS image, same fragment reads, same k-ascending MFMA chain per output element -> the choice of family is a performance decision only)
is tested at the model's shapes elsewhere; here it is fuzzed over ragged M / N (one row over a tile, N not a multiple of any tile width),
short and long K (one K tile .. more than any ring), strided A (lda > K), every epilogue and both 16-bit formats, for the nine forced
families incl. the two 256 x 160 forms of round 5 and the eight-wave 128 x 128 form of round 6.  The reference leg (plain 128 x 128 tile) is itself checked against fp64 on a sample."""
import os
import random

import pytest
import torch

from teochat_amd import _lib as L
from tests import _gpu as G

pytestmark = pytest.mark.gpu

PLAIN = {"gemm_wide": 0, "gemm_big": 0, "gemm_sk": 0, "gemm_narrow": 0, "gemm_quad": 0, "gemm_bm": 128}
FORCED = (("64x128 LDS-DMA", {"gemm_narrow": 2, "gemm_narrow_bm": 64}, "gemm_narrow_64"),
          ("128x128 LDS-DMA", {"gemm_narrow": 2, "gemm_narrow_bm": 128}, "gemm_narrow_128"),
          ("128x128 LDS-DMA, eight waves", {"gemm_narrow": 2, "gemm_narrow_bm": 128, "gemm_narrow_waves": 8}, "gemm_narrow_128w8"),      # round 6
          ("64x64 software-pipelined", {"gemm_narrow": 2, "gemm_narrow_bm": 64, "gemm_narrow_pipe": 2, "gemm_pipe_bn": 64}, "gemm_pipe_64x64"),            # round 6
          ("64x128 software-pipelined", {"gemm_narrow": 2, "gemm_narrow_bm": 64, "gemm_narrow_pipe": 2}, "gemm_pipe_64"),
          ("64x128 software-pipelined, ring of 4", {"gemm_narrow": 2, "gemm_narrow_bm": 64, "gemm_narrow_pipe": 2, "gemm_pipe_stages": 4}, "gemm_pipe_64_r4"),
          ("128x128 software-pipelined", {"gemm_narrow": 2, "gemm_narrow_bm": 128, "gemm_narrow_pipe": 2}, "gemm_pipe_128"),
          ("128x96 software-pipelined", {"gemm_narrow": 2, "gemm_narrow_bm": 128, "gemm_narrow_pipe": 2, "gemm_pipe_bn": 96}, "gemm_pipe_128x96"),
          ("128x96 software-pipelined, ring of 4", {"gemm_narrow": 2, "gemm_narrow_bm": 128, "gemm_narrow_pipe": 2, "gemm_pipe_bn": 96, "gemm_pipe_stages": 4}, "gemm_pipe_128x96"),
          ("256x160 eight waves", {"gemm_quad": 2}, "gemm_quad_160"),
          ("256x160 four waves", {"gemm_quad": 2, "gemm_quad_waves": 4}, "gemm_quad_160_w4"),
          ("128x256", {"gemm_wide": 2, "gemm_big": 0, "gemm_sk": 0, "gemm_narrow": 0, "gemm_quad": 0, "gemm_bm": 128}, "gemm_wide"),      # (both need
          ("256x256", {"gemm_big": 2, "gemm_big_hybrid": 0, "gemm_narrow": 0, "gemm_quad": 0, "gemm_bm": 128, "gemm_big_ragged": 0}, "gemm_big"),   # K >= 128)
          ("256x256 + 128x512 over a ragged last row block", {"gemm_big": 2, "gemm_big_hybrid": 0, "gemm_narrow": 0, "gemm_quad": 0, "gemm_bm": 128,
                                                               "gemm_big_ragged": 2}, "gemm_big"),                                            # round 6
          ("64-row register-staged", {"gemm_wide": 0, "gemm_big": 0, "gemm_sk": 0, "gemm_narrow": 0, "gemm_quad": 0, "gemm_bm": 64}, "gemm_mfma_64"),
          ("production dispatch", {}, None))


def _set(knobs):
    L.tune_reset()
    for k, v in knobs.items():
        assert L.tune_set(k.encode(), v) == 0, k


def _case(rng):
    M = rng.choice([1, 7, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 513, 640, 1000, 1031, 2056, 2168])
    N = 4 * rng.choice([1, 3, 8, 16, 31, 32, 33, 39, 40, 41, 64, 65, 79, 80, 81, 128, 160, 250, 256, 320, 321, 1024])
    K = 64 * rng.choice([1, 1, 2, 2, 3, 3, 4, 5, 6, 7, 8, 9, 16, 17, 32])
    extra = rng.choice(["", "bias", "res", "bias_res", "bias_gelu", "bias_quick", "bias_gelu_res", "f32out", "bias_f32out", "lda"])
    return M, N, K, extra


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("seed", range(int(os.environ.get("TEO_FUZZ_SEEDS", "6"))))       # soak: TEO_FUZZ_SEEDS=60 (profiles/r05_gemm_fuzz_soak.txt)
def test_every_tile_family_is_bitwise_the_plain_kernel_on_random_shapes(seed, dt):
    rng = random.Random(1000 * seed + (dt == torch.float16))
    lib = G.lib()
    seen = set()
    for _ in range(10):
        M, N, K, extra = _case(rng)
        g = torch.Generator().manual_seed(M * 31 + N * 7 + K + seed)
        lda = K + 64 if "lda" in extra else K
        Afull = torch.randn(M, lda, generator=g).to(dt).cuda()
        A = Afull[:, :K]                                   # row stride lda (16-byte aligned rows), as the engine's fused-buffer slices are
        W = (torch.randn(N, K, generator=g) * 0.05).to(dt).cuda()
        bias = (torch.randn(N, generator=g) * 0.1).to(dt).cuda() if "bias" in extra else None
        res = torch.randn(M, N, generator=g).to(dt).cuda() if "res" in extra else None
        act = L.ACT_GELU_ERF if "gelu" in extra else (L.ACT_QUICK_GELU if "quick" in extra else L.ACT_NONE)
        od = torch.float32 if "f32out" in extra else dt
        _set(PLAIN)
        want = G.gemm(A, W, bias=bias, res=res, act=act, out_dtype=od)
        assert lib.teo_last_kernel().decode() == "gemm_mfma_128"
        if M * N <= 70000:                                # the reference leg itself: fp64 on the host, one rounding of slack + fp32 summation
            ref = A.double().cpu() @ W.double().cpu().t()
            if bias is not None:
                ref = ref + bias.double().cpu()
            if act == L.ACT_GELU_ERF:
                ref = torch.nn.functional.gelu(ref)
            elif act == L.ACT_QUICK_GELU:
                ref = ref * torch.sigmoid(1.702 * ref)
            if res is not None:
                ref = ref + res.double().cpu()
            err = (want.double().cpu() - ref).abs()
            tol = (2.0 ** (-7 if dt == torch.bfloat16 else -10)) * ref.abs() + 1e-3 * (K ** 0.5) * 0.05 + 1e-6
            assert bool((err <= tol).all()), (M, N, K, extra, float(err.max()))
        for name, knobs, kernel in FORCED:
            _set(knobs)
            got = G.gemm(A, W, bias=bias, res=res, act=act, out_dtype=od)
            ran = lib.teo_last_kernel().decode()
            seen.add(ran)
            if kernel is not None and not (kernel in ("gemm_wide", "gemm_big") and K < 128):
                assert ran == kernel, (name, ran, M, N, K)
            assert torch.equal(got, want), (name, ran, M, N, K, extra, float((got.float() - want.float()).abs().max()))
    L.tune_reset()
    assert {"gemm_narrow_64", "gemm_narrow_128", "gemm_narrow_128w8", "gemm_pipe_64", "gemm_pipe_64_r4", "gemm_pipe_64x64", "gemm_pipe_128", "gemm_pipe_128x96", "gemm_quad_160", "gemm_quad_160_w4", "gemm_wide", "gemm_big", "gemm_mfma_64"} <= seen


SWIGLU_FORCED = (("128x256", {"gemm_wide": 2, "gemm_big": 0, "gemm_sk": 0, "gemm_narrow": 0, "gemm_quad": 0, "gemm_bm": 128}, "gemm_wide"),
                 ("256x256", {"gemm_big": 2, "gemm_big_hybrid": 0, "gemm_narrow": 0, "gemm_quad": 0, "gemm_bm": 128, "gemm_big_ragged": 0}, "gemm_big"),
                 ("64x64 software-pipelined", {"gemm_narrow": 2, "gemm_narrow_bm": 64, "gemm_narrow_pipe": 2, "gemm_pipe_bn": 64}, "gemm_pipe_64x64"),
                 ("64x128 software-pipelined, ring of 4", {"gemm_narrow": 2, "gemm_narrow_bm": 64, "gemm_narrow_pipe": 2, "gemm_pipe_stages": 4}, "gemm_pipe_64_r4"),
                 ("128x128 software-pipelined", {"gemm_narrow": 2, "gemm_narrow_bm": 128, "gemm_narrow_pipe": 2}, "gemm_pipe_128"),
                 ("production dispatch", {}, None))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_swiglu_epilogue_is_bitwise_the_same_in_every_family_that_carries_it(dt):
    """gate/up + SwiGLU (TEO_GEMM_SWIGLU16: W rows interleaved as 16 gate | 16 up, out[m, j] = silu(gate_j) * up_j; tf LlamaMLP reached from
    llava_llama.py:88-99) on random shapes incl. the M <= 128 ones the small software-pipelined tiles take in production (round 6): every family
    against the register-staged 128 x 128 kernel, bitwise, and that kernel against an fp64 evaluation."""
    rng = random.Random(77 + (dt == torch.float16))
    lib = G.lib()
    seen = set()
    for _ in range(8):
        M = rng.choice([1, 5, 64, 65, 100, 128, 129, 200, 257, 640])
        N = 32 * rng.choice([1, 2, 3, 5, 8, 9, 16, 33, 64, 172])
        K = 64 * rng.choice([2, 3, 4, 8, 32, 64])
        g = torch.Generator().manual_seed(M * 13 + N + K)
        A = torch.randn(M, K, generator=g).to(dt).cuda()
        W = (torch.randn(N, K, generator=g) * (2.0 / K ** 0.5)).to(dt).cuda()
        _set(PLAIN)
        want = G.gemm(A, W, flags=L.GEMM_SWIGLU16)
        assert lib.teo_last_kernel().decode() == "gemm_mfma_128" and want.shape == (M, N // 2)
        if M * N <= 70000:
            full = A.double().cpu() @ W.double().cpu().t()
            blk = full.view(M, N // 32, 2, 16)
            ref = (torch.nn.functional.silu(blk[:, :, 0]) * blk[:, :, 1]).reshape(M, N // 2)
            err = (want.double().cpu() - ref).abs()
            tol = (2.0 ** (-7 if dt == torch.bfloat16 else -10)) * ref.abs() + 2e-3
            assert bool((err <= tol).all()), (M, N, K, float(err.max()))
        for name, knobs, kernel in SWIGLU_FORCED:
            _set(knobs)
            got = G.gemm(A, W, flags=L.GEMM_SWIGLU16)
            ran = lib.teo_last_kernel().decode()
            seen.add(ran)
            if kernel is not None:
                assert ran == kernel, (name, ran, M, N, K)
            assert torch.equal(got, want), (name, ran, M, N, K, float((got.float() - want.float()).abs().max()))
    L.tune_reset()
    assert {"gemm_wide", "gemm_big", "gemm_pipe_64x64", "gemm_pipe_64_r4", "gemm_pipe_128"} <= seen, seen
