#!/usr/bin/env python3
"""Probe: does hipExtAnyOrderLaunch (AQL barrier bit cleared) let consecutive kernels of ONE stream overlap on gfx950?
Times plain-launch decode steps of the 7B-shaped model with launch_flags 0 and 1 (flag 1 RACES: numbers only, results are garbage)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from teochat_amd import _lib as L  # noqa: E402
from teochat_amd.builder import load_pretrained_model  # noqa: E402

_, model, _, _ = load_pretrained_model("synthetic:teochat-7b", None, "synthetic:teochat-7b", device="cuda:0", dtype=torch.bfloat16, max_seq=2560)
eng = model.engine
emb = torch.randn(2168, 4096, device="cuda:0").to(torch.bfloat16) * 0.02
for flags in (0, 1, 0, 1):
    eng.reset_cache()
    lg = eng.prefill(emb, last_only=True)
    eng.decode_begin(5)
    L.check(eng.lib.teo_tune_set(b"launch_flags", flags), "tune")
    eng.decode_steps(8, use_graph=False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    eng.decode_steps(64, use_graph=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 64 * 1e3
    L.check(eng.lib.teo_tune_set(b"launch_flags", 0), "tune")
    print(f"launch_flags={flags}: {dt:.3f} ms per decode step (plain launches)", flush=True)
eng.reset_cache()
lg = eng.prefill(emb, last_only=True)
eng.decode_begin(5)
eng.decode_steps(8)
torch.cuda.synchronize(); t = time.perf_counter()
eng.decode_steps(64)
torch.cuda.synchronize()
print(f"graph replay: {(time.perf_counter() - t) / 64 * 1e3:.3f} ms per decode step")
