#!/usr/bin/env python3
"""Device sampler (teo_sample_topk) per draw: the register-resident bisection form (16-byte aligned rows of <= 32768 logits) against the
4-pass radix select (taken for a row that is 4 bytes off).  usage: python tools/sampler_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teochat_amd import _lib as L
from tests import _gpu as G
lib = L.load()
lg = (torch.randn(32000) * 2).cuda()
tok = torch.zeros(1, dtype=torch.int64, device="cuda")
def run(ptr, k, p, n=200):
    for d in range(10): L.check(lib.teo_sample_topk(ptr, G.p(tok), 32000, 0.2, k, p, 1, d, G.stream()), "s")
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for d in range(n): lib.teo_sample_topk(ptr, G.p(tok), 32000, 0.2, k, p, 1, d, G.stream())
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
pad = torch.empty(32001, device="cuda"); pad[1:] = lg
for k, p in ((50, 1.0), (50, 0.9), (1000, 1.0), (0, 1.0)):
    print(f"top_k {k} top_p {p}: register form {run(G.p(lg), k, p):.1f} us, radix form {run(pad[1:].data_ptr(), k, p):.1f} us")
