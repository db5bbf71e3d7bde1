#!/usr/bin/env python3
"""Diagnostic (GPU box): full-size synthetic 7B, bf16 prefill / bf16 decode logits against an fp32 run on the same
(bf16-rounded) weights.  Shows how much of the prefill-vs-decode difference is bf16 noise over 32 layers."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import teo_oracle as O  # noqa: E402  (inputs only)
from teochat_amd.config import teochat_7b_config  # noqa: E402
from teochat_amd.engine import TeoEngine  # noqa: E402
from teochat_amd.model import LlavaLlamaForCausalLM  # noqa: E402
from teochat_amd.synthetic import synthetic_state_dict  # noqa: E402

cfg = teochat_7b_config()
sd = synthetic_state_dict(cfg, seed=2, std=0.02, dtype=torch.bfloat16, device="cuda:0")
ids = O.synthetic_prompt_ids(96, 0, 32000, seed=4).view(1, -1).cuda()
m16 = LlavaLlamaForCausalLM(cfg, TeoEngine(sd, cfg, dtype=torch.bfloat16, device="cuda:0", max_seq=1024))
out = m16.generate(input_ids=ids, images=None, do_sample=False, max_new_tokens=3, eos_token_id=None)
dec16 = m16.engine.d_logits.clone()
pre16 = m16(input_ids=out[:, :-1], images=None).logits[0, -1].clone()
del m16
torch.cuda.empty_cache()
sd32 = {k: v.float() for k, v in sd.items()}
m32 = LlavaLlamaForCausalLM(cfg, TeoEngine(sd32, cfg, dtype=torch.float32, device="cuda:0", max_seq=1024))
pre32 = m32(input_ids=out[:, :-1], images=None).logits[0, -1].clone()
mx = float(pre32.abs().max())
print(f"max |logit| (fp32) {mx:.3f}")
for name, x in (("bf16 prefill", pre16), ("bf16 decode", dec16)):
    print(f"{name:13s} vs fp32: max abs diff {float((x - pre32).abs().max()):.4f}  rel-to-max {float((x - pre32).abs().max()) / mx:.3e}  "
          f"argmax equal {int(x.argmax()) == int(pre32.argmax())}")
print(f"bf16 prefill vs bf16 decode: rel-to-max {float((pre16 - dec16).abs().max()) / mx:.3e}")
