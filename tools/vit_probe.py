"""ViT tower + projector alone (T frames), for rocprofv3 --kernel-trace --stats: where the tower's time goes.
usage: python tools/vit_probe.py [T] [iters] [knob=value ...]      (knobs: the engine's teo_tune block, e.g. gemm_narrow_pipe=0)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from teochat_amd.builder import load_pretrained_model  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    tok, model, _, _ = load_pretrained_model("synthetic:teochat-7b", None, "synthetic:teochat-7b", device="cuda:0", dtype=torch.bfloat16,
                                             max_seq=512)
    eng = model.engine
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        eng.tune_set(k, int(v))
    px = torch.randn(T, 3, 224, 224, device="cuda:0", dtype=torch.bfloat16)
    for _ in range(3):
        f = eng.vit_features(px)
        eng.project(f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        f = eng.vit_features(px)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(iters):
        eng.project(f)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    flops_vit = T * 155.3e9
    print(f"T={T}{(' ' + ' '.join(sys.argv[3:])) if len(sys.argv) > 3 else ''}: tower {1e3 * (t1 - t0) / iters:.3f} ms ({flops_vit / ((t1 - t0) / iters) / 1e12:.0f} TFLOP/s), projector {1e3 * (t2 - t1) / iters:.3f} ms")


if __name__ == "__main__":
    main()
