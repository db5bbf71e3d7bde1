"""Deterministic byte-level tokenizer used where no sentencepiece model is available.

The reference uses the LLaMA sentencepiece tokenizer (model/builder.py:111, use_fast=False), which
stays third-party and host-side.  The path only needs an object with `__call__(str).input_ids`,
`.bos_token_id`, `.eos_token_id`, `.decode` and `.batch_decode` (mm_utils.py:44,79,94; inference.py:75);
this class provides exactly that so synthetic runs, tests and golden fixtures are reproducible offline.

ids: 0 = <unk>/pad, 1 = <s> (BOS), 2 = </s> (EOS), 3 + b = byte b.  "</s>" in text maps to id 2.
"""
from types import SimpleNamespace


class ByteTokenizer:
    bos_token_id = 1
    eos_token_id = 2
    pad_token_id = 0
    unk_token_id = 0
    vocab_size = 259

    def __init__(self, add_bos=True):
        self.add_bos = add_bos

    def __call__(self, text, **kwargs):
        ids = [self.bos_token_id] if self.add_bos else []
        parts = text.split("</s>")
        for i, part in enumerate(parts):
            if i > 0:
                ids.append(self.eos_token_id)
            ids.extend(3 + b for b in part.encode("utf-8"))
        return SimpleNamespace(input_ids=ids)

    def decode(self, ids, skip_special_tokens=False, **kwargs):
        if hasattr(ids, "tolist"):
            ids = ids.tolist()
        out = bytearray()
        text = ""
        for t in ids:
            t = int(t)
            if 3 <= t < 259:
                out.append(t - 3)
                continue
            text += out.decode("utf-8", errors="replace")
            out = bytearray()
            if not skip_special_tokens:
                if t == 1:
                    text += "<s>"
                elif t == 2:
                    text += "</s>"
                elif t == 0:
                    text += "<unk>"
            # ids outside the byte range (synthetic large-vocab runs) decode to nothing
        text += out.decode("utf-8", errors="replace")
        return text

    def batch_decode(self, rows, skip_special_tokens=False, **kwargs):
        if hasattr(rows, "tolist"):
            rows = rows.tolist()
        return [self.decode(r, skip_special_tokens=skip_special_tokens) for r in rows]
