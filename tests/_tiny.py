"""Tiny configurations shared by the golden generator and the tests (kept in sync with
tests/golden/make_golden.py::TINY; the state-dict checksum stored in each fixture guards drift)."""
import json
import os

import numpy as np
import torch

from oracle import teo_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
TINY_STD = 0.08

TINY = {
    "tinyA": dict(
        vit=dict(hidden_size=64, num_attention_heads=4, intermediate_size=128, num_hidden_layers=3,
                 hidden_act="quick_gelu"),
        llm=dict(hidden_size=64, num_attention_heads=4, num_key_value_heads=2, intermediate_size=128,
                 num_hidden_layers=2, vocab_size=300),
    ),
    "tinyB": dict(
        vit=dict(hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=3,
                 hidden_act="gelu"),
        llm=dict(hidden_size=256, num_attention_heads=2, num_key_value_heads=2, intermediate_size=512,
                 num_hidden_layers=2, vocab_size=512),
    ),
    # round 6: the "anchored" construction of teochat_amd/synthetic.py::anchor_gains at tiny size -- tinyA / tinyB's greedy streams
    # settle on one token after two steps, so token equality against the reference says little there; here 16 rows of embed_tokens /
    # lm_head form a successor cycle weak enough that the context decides some steps (the reference's stream: 8 distinct tokens in
    # 8 steps with one jump off the +1 walk, top-2 margins 0.09 .. 6.3 at max|logit| 9.3).  GQA with ONE kv head, head_dim 64.
    "tinyC": dict(
        vit=dict(hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=3,
                 hidden_act="gelu"),
        llm=dict(hidden_size=128, num_attention_heads=2, num_key_value_heads=1, intermediate_size=256,
                 num_hidden_layers=2, vocab_size=512),
        anchors=dict(count=16, base=40, embed_scale=0.3, gain=2.0, seed=77),
    ),
}


def apply_anchors(sd, spec, std=TINY_STD):
    """embed_tokens[a_k] = embed_scale * r_k, lm_head[a_(k+1)] = gain * std * r_k for the `count` anchor ids from `base` (r_k ~ N(0, 1),
    seeded): every other weight keeps its make_state_dict value."""
    g = torch.Generator().manual_seed(spec["seed"])
    D = sd["model.embed_tokens.weight"].shape[1]
    r = torch.randn(spec["count"], D, generator=g)
    ids = torch.arange(spec["base"], spec["base"] + spec["count"])
    out = dict(sd)
    e = sd["model.embed_tokens.weight"].clone()
    e[ids] = (spec["embed_scale"] * r).to(e.dtype)
    h = sd["lm_head.weight"].clone()
    h[ids.roll(-1)] = (spec["gain"] * std * r).to(h.dtype)
    out["model.embed_tokens.weight"], out["lm_head.weight"] = e, h
    return out


def prompt_ids(name, n_text, T, vocab, seed=1):
    """The golden prompt of a tiny config; an anchored config ends its prompt ON the first anchor, so step 0 is already on the cycle."""
    ids = O.synthetic_prompt_ids(n_text, T, vocab, seed=seed)
    a = TINY[name].get("anchors")
    if a:
        ids[-1] = a["base"]
    return ids


def cfgs(name):
    t = TINY[name]
    v = O.VitCfg(**t["vit"])
    l = O.LlamaCfg(**t["llm"])
    mm = O.MMCfg(mm_hidden_size=v.hidden_size)
    return v, l, mm


def state_dict(name, dtype=None):
    import torch
    v, l, mm = cfgs(name)
    sd = O.make_state_dict(v, l, mm, seed=2, std=TINY_STD, dtype=dtype or torch.float32)
    a = TINY[name].get("anchors")
    return apply_anchors(sd, a) if a else sd


def sd_checksum(sd):
    return float(sum(v.double().abs().sum() for v in sd.values()))


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def load_json(name):
    return json.load(open(os.path.join(GOLDEN, name + ".json")))


def train_batch(vocab, image_size):
    """Training-shape inputs (train.py:840-901 feeds the same forward): batch of 2, right padding, labels that mask the
    prompt part, flat image list consumed in order (llava_arch.py:284-285)."""
    a = O.synthetic_prompt_ids(20, 1, vocab, seed=7)
    b = O.synthetic_prompt_ids(14, 2, vocab, seed=8)
    W = 20
    ids = torch.zeros(2, W, dtype=torch.long)
    mask = torch.zeros(2, W, dtype=torch.long)
    labels = torch.full((2, W), -100, dtype=torch.long)
    for r, row in enumerate((a, b)):
        n = row.numel()
        ids[r, :n] = row
        mask[r, :n] = 1
        labels[r, 8:n] = row[8:]                      # supervise the answer part only
    labels[ids == -200] = -100
    frames = O.synthetic_frames(3, image_size, seed=3)
    return ids, mask, labels, frames
