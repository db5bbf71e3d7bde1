#!/usr/bin/env python3
"""Soak test of the device sampler: the register-resident bisection form against the radix form (a row 4 bytes off alignment) over random
vocabulary sizes, top-k / top-p / temperature, ties at the threshold and -inf logits: the same token for every draw.  usage: python tools/sampler_fuzz.py"""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teochat_amd import _lib as L
from tests import _gpu as G
lib = L.load()
rng = random.Random(3)
tok = torch.zeros(1, dtype=torch.int64, device="cuda")
def draw(ptr, n, t, k, p, seed, d):
    L.check(lib.teo_sample_topk(ptr, G.p(tok), n, t, k, p, seed, d, G.stream()), "s")
    return int(tok.item())
bad = 0
for case in range(60):
    vocab = rng.choice((1000, 4096, 31999, 32000, 32001, 32768))
    k = rng.choice((1, 2, 10, 50, 200, 1024))
    p = rng.choice((1.0, 1.0, 0.95, 0.5))
    t = rng.choice((0.2, 0.7, 1.0, 2.5))
    g = torch.Generator().manual_seed(case)
    lg = torch.randn(vocab, generator=g) * rng.choice((0.5, 3.0, 10.0))
    if rng.random() < 0.5:
        lg[torch.randint(0, vocab, (30,), generator=g)] = float(lg.topk(min(k, vocab)).values[-1])
    if rng.random() < 0.2:
        lg[torch.randint(0, vocab, (50,), generator=g)] = float("-inf")
    a = lg.cuda()
    pad = torch.empty(vocab + 1, device="cuda"); pad[1:] = a; b = pad[1:]
    xs = [draw(a.data_ptr(), vocab, t, k, p, 11 + case, d) for d in range(24)]
    ys = [draw(b.data_ptr(), vocab, t, k, p, 11 + case, d) for d in range(24)]
    ok = xs == ys
    bad += 0 if ok else 1
    print(f"case {case}: vocab {vocab} k {k} p {p} T {t}: {'same' if ok else 'DIFFERENT'} ({len(set(xs))} distinct tokens)", flush=True)
print(f"{60 - bad} / 60 clean")
