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
}


def cfgs(name):
    t = TINY[name]
    v = O.VitCfg(**t["vit"])
    l = O.LlamaCfg(**t["llm"])
    mm = O.MMCfg(mm_hidden_size=v.hidden_size)
    return v, l, mm


def state_dict(name, dtype=None):
    import torch
    v, l, mm = cfgs(name)
    return O.make_state_dict(v, l, mm, seed=2, std=TINY_STD, dtype=dtype or torch.float32)


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
