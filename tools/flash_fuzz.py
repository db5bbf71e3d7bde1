#!/usr/bin/env python3
"""Soak test of the flash-attention kernel's staging protocol (LDS-DMA into two slots, K one tile ahead of V^T, one barrier per tile): random
shapes -- ragged query / key counts, chunked prefill offsets, GQA, both head dims, causal or not -- each run several times in every form
(in-wave pipeline on / off, workgroup order 0 / 1).  Every form must give the same bits every time, stay finite with NaN behind kv_len in the
V^T padding, and agree with the generic (VALU) kernel.  A slot refilled too early or read too early shows up here as a mismatch.
usage: python tools/flash_fuzz.py [cases] [seed]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from teochat_amd import _lib as L  # noqa: E402
from tests import _gpu as G  # noqa: E402

lib = L.load()
bf = torch.bfloat16


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for case in range(cases):
        d = rng.choice((64, 128))
        causal = rng.random() < 0.7
        Hk = rng.choice((1, 2, 8))
        H = Hk * rng.choice((1, 1, 4))
        B = rng.choice((1, 1, 2, 3))
        Sq = rng.choice((1, 7, 64, 65, 127, 128, 129, 200, 257, 300, 511, 640, 1000, 1500))
        Sk = Sq + (rng.choice((0, 0, 0, 1, 63, 64, 100, 1000)) if causal else rng.choice((0, 0, 5, 64)))
        g = torch.Generator(device="cuda").manual_seed(case)
        q = torch.randn(B, H, Sq, d, generator=g, device="cuda").to(bf)
        k = torch.randn(B, Hk, Sk, d, generator=g, device="cuda").to(bf)
        v = torch.randn(B, Hk, Sk, d, generator=g, device="cuda").to(bf)
        vt = G.make_vt(v)
        vt[..., Sk:] = float("nan")
        outs = []
        for rep in range(3):
            for pipe, order in ((1, 1), (0, 0), (-1, 1), (1, 0)):
                L.tune_set(b"flash_pipe", pipe)
                L.tune_set(b"flash_order", order)
                outs.append(G.attention(q, k, v, causal, d ** -0.5, vt=vt))
        L.tune_reset()
        torch.cuda.synchronize()
        ref = G.attention(q, k, v, causal, d ** -0.5, force_simple=True)
        same = all(torch.equal(o, outs[0]) for o in outs)
        finite = bool(torch.isfinite(outs[0].float()).all())
        err = float((outs[0].float() - ref.float()).abs().max())
        ok = same and finite and err < 3e-2
        bad += 0 if ok else 1
        print(f"case {case:3d}: B={B} H={H} Hk={Hk} Sq={Sq} Sk={Sk} d={d} causal={int(causal)}  same bits {same}  finite {finite}  max |flash - generic| {err:.2e}"
              f"{'' if ok else '   <-- FAIL'}", flush=True)
    print(f"{cases - bad} / {cases} cases clean")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
