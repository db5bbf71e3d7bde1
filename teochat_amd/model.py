"""LlavaLlamaForCausalLM for MI355X: the reference's model-level interface over TeoEngine.

Mirrors (names, argument order, defaults, error behaviour):
  language_model/llava_llama.py:40-108  LlavaLlamaForCausalLM.forward / prepare_inputs_for_generation / get_model
  llava_arch.py:125-346                 get_image_tower / encode_images / prepare_inputs_labels_for_multimodal
  generation: the greedy / sampled loop GenerationMixin.generate runs for eval/inference.py:64-72
There is no nn.Module underneath: weights live in the engine's device buffers and all arithmetic is HIP kernels.
The index logic of the embedding splice runs on the host on integers (bit-exact); the data movement is one
teo_embed_splice launch driven by the resulting int32 plan.
"""
from types import SimpleNamespace
from typing import List, Optional

import numpy as np
import torch

from . import _lib as L
from .constants import IGNORE_INDEX, IMAGE_TOKEN_INDEX
from .engine import TeoEngine


class CausalLMOutputWithPast(dict):
    """Minimal stand-in for transformers.modeling_outputs.CausalLMOutputWithPast (attribute + key access)."""

    def __init__(self, loss=None, logits=None, past_key_values=None, hidden_states=None, attentions=None):
        super().__init__(loss=loss, logits=logits, past_key_values=past_key_values, hidden_states=hidden_states,
                         attentions=attentions)
        self.__dict__ = self

    def to_tuple(self):
        return tuple(v for v in (self.loss, self.logits, self.past_key_values, self.hidden_states, self.attentions) if v is not None)


class TeoKVCache:
    """Handle to the engine's device KV cache (one sequence).  `[-1][-1].shape[-2]` reports the cached length so that
    code written against the legacy tuple cache (llava_arch.py:156) keeps working."""

    def __init__(self, engine):
        self.engine = engine

    def get_seq_length(self):
        return self.engine.cache_len

    def __len__(self):
        return self.engine.cfg.num_hidden_layers

    def __getitem__(self, i):
        c = self.engine.cfg
        shape = SimpleNamespace(shape=(1, c.num_key_value_heads, self.engine.cache_len, c.head_dim))
        return (shape, shape)


class TeoBatchKVCache:
    """Handle to the per-conversation device KV caches of a batched forward (the BatchDecoder's slots): what
    forward(input_ids [B, S], use_cache=True) returns at B > 1 and what forward(input_ids [B, 1], past_key_values=...) continues.
    `[-1][-1].shape[-2]` reports the longest cached sequence (the legacy tuple cache of a padded batch has that length,
    llava_arch.py:156)."""

    def __init__(self, engine, decoder):
        self.engine, self.decoder = engine, decoder

    def get_seq_length(self):
        return max(self.decoder.cache_len)

    def __len__(self):
        return self.engine.cfg.num_hidden_layers

    def __getitem__(self, i):
        c = self.engine.cfg
        shape = SimpleNamespace(shape=(self.decoder.B, c.num_key_value_heads, max(self.decoder.cache_len), c.head_dim))
        return (shape, shape)


class TeoImageTower:
    """The tower object `get_image_tower()` returns (LanguageBindImageTower surface, languagebind/__init__.py:94-173)."""

    def __init__(self, engine, processor=None):
        self._engine = engine
        self.is_loaded = True
        self.select_layer = engine.cfg.mm_vision_select_layer
        self.select_feature = engine.cfg.mm_vision_select_feature
        self.image_processor = processor
        self.config = engine.vcfg

    def load_model(self):
        self.is_loaded = True

    def __call__(self, images):
        return self.forward(images)

    def shard_frames(self, comm=None, group=None):
        """Config C4: from now on a batched forward() encodes only this rank's contiguous block of frames and all-gathers
        the visual tokens (teochat_amd/parallel.py).  comm: TeoComm (RCCL behind the C ABI); else torch.distributed `group`."""
        self._shard = (comm, group)

    @torch.no_grad()
    def forward(self, images):
        if type(images) is list:
            return [self._engine.vit_features(im.unsqueeze(0)).to(im.dtype) for im in images]
        shard = getattr(self, "_shard", None)
        if shard is not None:
            from .parallel import sharded_frame_features
            return sharded_frame_features(self._engine.vit_features, images, group=shard[1], comm=shard[0]).to(images.dtype)
        return self._engine.vit_features(images).to(images.dtype)

    @property
    def dtype(self):
        return self._engine.dtype

    @property
    def device(self):
        return self._engine.device

    @property
    def hidden_size(self):
        return self.config.hidden_size

    @property
    def num_patches(self):
        return self.config.num_patches

    @property
    def dummy_feature(self):
        return torch.zeros(1, self.hidden_size, device=self.device, dtype=self.dtype)


class _Projector:
    def __init__(self, engine):
        self._engine = engine

    def __call__(self, x):
        return self._engine.project(x)


class _Embedding:
    def __init__(self, engine):
        self._engine = engine

    @property
    def weight(self):
        return self._engine.embed

    def __call__(self, ids):
        flat = ids.reshape(-1)
        if flat.numel() and (int(flat.min()) < 0 or int(flat.max()) >= self._engine.cfg.vocab_size):
            raise IndexError("index out of range in self")        # what nn.Embedding raises (e.g. a stray -200 sentinel)
        return self._engine.splice(flat.to(torch.int32), None).view(*ids.shape, -1)


class LlavaLlamaModel:
    """`model.model`: holds image_tower / video_tower / mm_projector / embed_tokens like LlavaMetaModel (llava_arch.py:27-49)."""

    def __init__(self, engine, processor=None):
        self.image_tower = TeoImageTower(engine, processor)
        self.video_tower = None
        self.mm_projector = _Projector(engine)
        self.embed_tokens = _Embedding(engine)

    def get_image_tower(self):
        t = getattr(self, "image_tower", None)
        return t[0] if type(t) is list else t

    def get_video_tower(self):
        t = getattr(self, "video_tower", None)
        return t[0] if type(t) is list else t


def build_splice_plan(input_ids, attention_mask, labels, n_feature_rows: List[int], max_length, padding_side):
    """Integer half of prepare_inputs_labels_for_multimodal (llava_arch.py:248-329), on host numpy arrays.

    input_ids [B,n] int64, attention_mask [B,n] bool, labels [B,n] int64, n_feature_rows[i] = rows of image feature i.
    Returns plan [B,Lmax] int32 (>=0 vocab id, <0 -(global visual row)-1, INT32_MIN pad), labels [B,Lmax],
    mask [B,Lmax] bool, position_ids [B,Lmax], lengths.
    Raises IndexError when a sample needs more image features than were supplied (reference: llava_arch.py:284).
    """
    starts = np.concatenate(([0], np.cumsum(n_feature_rows))).astype(np.int64)
    plans, labs = [], []
    nxt = 0                                   # next unused image feature, consumed globally across the batch
    for b in range(input_ids.shape[0]):
        keep = attention_mask[b]
        ids = input_ids[b][keep]
        lab = labels[b][keep]
        if not (ids == IMAGE_TOKEN_INDEX).any():
            if nxt >= len(n_feature_rows):
                raise IndexError("list index out of range")     # the reference indexes image_features[cur_image_idx]
            nxt += 1                          # a text-only sample still consumes one (empty slice of a) feature
            plans.append(ids.astype(np.int64))
            labs.append(lab)
            continue
        p_parts, l_parts = [], []
        cuts = np.flatnonzero(ids == IMAGE_TOKEN_INDEX)
        lo = 0
        for cpos in cuts:
            p_parts.append(ids[lo:cpos].astype(np.int64))
            l_parts.append(lab[lo:cpos])
            if nxt >= len(n_feature_rows):
                raise IndexError("list index out of range")
            rows = np.arange(starts[nxt], starts[nxt + 1], dtype=np.int64)
            p_parts.append(-(rows + 1))
            l_parts.append(np.full(rows.shape[0], IGNORE_INDEX, dtype=lab.dtype))
            nxt += 1
            lo = cpos + 1
        p_parts.append(ids[lo:].astype(np.int64))
        l_parts.append(lab[lo:])
        plans.append(np.concatenate(p_parts))
        labs.append(np.concatenate(l_parts))
    if max_length is not None:
        plans = [p[:max_length] for p in plans]
        labs = [l[:max_length] for l in labs]
    lens = [p.shape[0] for p in plans]
    Lmax = max(lens)
    B = len(plans)
    plan = np.full((B, Lmax), L.INT32_MIN, dtype=np.int64)
    lab_out = np.full((B, Lmax), IGNORE_INDEX, dtype=np.int64)
    mask = np.zeros((B, Lmax), dtype=bool)
    pos = np.zeros((B, Lmax), dtype=np.int64)
    for b, (p, l) in enumerate(zip(plans, labs)):
        n = p.shape[0]
        if n == 0:
            continue
        sl = slice(Lmax - n, Lmax) if padding_side == "left" else slice(0, n)
        plan[b, sl] = p
        lab_out[b, sl] = l
        mask[b, sl] = True
        pos[b, sl] = np.arange(n)
    return plan.astype(np.int32), lab_out, mask, pos, lens


class LlavaLlamaForCausalLM:
    def __init__(self, config, engine: TeoEngine, processor=None):
        self.config = config
        self.engine = engine
        self.model = LlavaLlamaModel(engine, processor)
        self.vocab_size = config.vocab_size
        self.pretraining_tp = getattr(config, "pretraining_tp", 1)
        # HF GenerationConfig defaults; builder.load_generation_config overlays the checkpoint's generation_config.json
        self.generation_config = SimpleNamespace(eos_token_id=getattr(config, "eos_token_id", None), do_sample=False, top_k=50,
                                                 top_p=1.0, temperature=1.0)
        self.training = False

    # --- nn.Module-ish surface the harness touches
    @property
    def device(self):
        return self.engine.device

    @property
    def dtype(self):
        return self.engine.dtype

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def get_model(self):
        return self.model

    def get_image_tower(self):
        return self.get_model().get_image_tower()

    def get_video_tower(self):
        return self.get_model().get_video_tower()

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    # --- H12
    def encode_images(self, images):
        feats = self.get_model().get_image_tower()(images)
        return self.get_model().mm_projector(feats)

    # --- H13
    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels, images):
        image_tower, video_tower = self.get_image_tower(), self.get_video_tower()
        if (image_tower is None and video_tower is None) or images is None or input_ids.shape[1] == 1:
            if (past_key_values is not None and (image_tower is not None or video_tower is not None)
                    and images is not None and input_ids.shape[1] == 1):
                target = past_key_values[-1][-1].shape[-2] + 1
                pad = torch.ones((attention_mask.shape[0], target - attention_mask.shape[1]),
                                 dtype=attention_mask.dtype, device=attention_mask.device)
                attention_mask = torch.cat((attention_mask, pad), dim=1)
                position_ids = torch.sum(attention_mask, dim=1).unsqueeze(-1) - 1
            return input_ids, position_ids, attention_mask, past_key_values, None, labels

        if any(im.ndim == 4 for im in images):
            raise ValueError("4-D (video) items need a video tower; load_model removes it (eval/eval.py:31)")
        if any(im.ndim != 3 for im in images):
            raise ValueError("images must be a flat list of [3,H,W] tensors")
        if getattr(self.config, "tune_mm_mlp_adapter", False) and getattr(self.config, "mm_use_im_start_end", False):
            raise NotImplementedError

        feats = self.encode_images(torch.stack(list(images)))           # [n_img, NV, D], one batched tower call
        n_img, NV, D = feats.shape
        dev = input_ids.device
        ids_h = input_ids.detach().cpu().numpy()
        mask_h = (np.ones_like(ids_h, dtype=bool) if attention_mask is None
                  else attention_mask.detach().cpu().numpy().astype(bool))
        lab_h = (np.full_like(ids_h, IGNORE_INDEX) if labels is None else labels.detach().cpu().numpy())
        plan, lab_out, mask, pos, _ = build_splice_plan(
            ids_h, mask_h, lab_h, [NV] * n_img, getattr(self.config, "tokenizer_model_max_length", None),
            getattr(self.config, "tokenizer_padding_side", "right"))
        B, Lmax = plan.shape
        embeds = self.engine.splice(torch.from_numpy(plan.reshape(-1)), feats.reshape(n_img * NV, D)).view(B, Lmax, D)
        new_labels = None if labels is None else torch.from_numpy(lab_out).to(dev)
        if attention_mask is None:
            new_mask = None
        else:
            new_mask = torch.from_numpy(mask).to(device=dev, dtype=attention_mask.dtype)
        new_pos = None if position_ids is None else torch.from_numpy(pos).to(device=dev, dtype=position_ids.dtype)
        return None, new_pos, new_mask, past_key_values, embeds, new_labels

    # --- H14 + H15
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, images=None,
                return_dict=None):
        """LlavaLlamaForCausalLM.forward (llava_llama.py:56-99), same arguments in the same order.

        Contract where it differs from the reference (also in INTEGRATION.md):
          * PADDED positions (attention_mask == 0) are never computed: their `logits` rows and, with `output_hidden_states`, their rows
            in ALL L + 1 hidden-state tensors are ZERO, at B = 1 and at B > 1 alike.  The reference's LlamaModel returns the (masked)
            activations it computed there; a consumer that pools over the full [B, S, D] tensor without applying the attention mask
            gets different values.  Real positions equal the reference's (tests/golden/hidden_*.npz).
          * `output_attentions` (B = 1): one [1, H, S, past + S] map per layer in the model dtype, written by a plain kernel beside the
            unchanged fused forward (teo_llama_prefill_attentions) -- softmax statistics in fp32, one rounding, exact zeros for masked keys;
            at B > 1 or on a batched decode step it raises NotImplementedError.
          * B > 1: `use_cache=True` (explicit) keeps one device cache per conversation and returns a TeoBatchKVCache; the next call
            takes it with input_ids [B, 1] (one batched decode step, every row at its own true next position).  With use_cache
            left None (the training-shape forward) nothing is kept and `past_key_values` is None.
        """
        if inputs_embeds is None:
            (input_ids, position_ids, attention_mask, past_key_values, inputs_embeds, labels) = \
                self.prepare_inputs_labels_for_multimodal(input_ids, position_ids, attention_mask, past_key_values,
                                                          labels, images)
        if inputs_embeds is None:
            inputs_embeds = self.get_model().embed_tokens(input_ids)
        eng = self.engine
        B, S, _ = inputs_embeds.shape
        if output_attentions and (B > 1 or isinstance(past_key_values, TeoBatchKVCache)):
            # the maps come from a plain kernel beside the single-sequence prefill (teo_llama_prefill_attentions); the batched forms keep
            # the fused kernels only
            raise NotImplementedError("output_attentions is a single-sequence feature (B = 1)")
        if isinstance(past_key_values, TeoBatchKVCache):
            # batched continuation (HF generate over a batch re-enters forward with the cache and ONE new token per row,
            # llava_arch.py:154-163): one batched decode step over the BatchDecoder's slots
            return self._forward_batch_step(input_ids, past_key_values, labels, output_hidden_states, return_dict, B, S)
        if past_key_values is None:
            eng.reset_cache()
        elif not isinstance(past_key_values, TeoKVCache) or past_key_values.engine is not eng:
            raise ValueError("past_key_values must be the cache object returned by this model")
        if B > 1 and past_key_values is not None:
            raise ValueError("a one-sequence TeoKVCache cannot continue a batch: pass the TeoBatchKVCache that the batched forward returned")
        past = eng.cache_len
        # B > 1 with use_cache=True (explicitly: the training-shape forward leaves it None and keeps nothing): the prompts are
        # prefilled INTO the batch decoder's per-conversation cache slots and the returned TeoBatchKVCache continues them
        keep_batch = B > 1 and use_cache is True
        logits = torch.zeros(B, S, self.config.vocab_size, dtype=torch.float32, device=eng.device)
        # output_hidden_states (llava_llama.py:56-69 -> LlamaModel.forward): L + 1 tensors [B, S, D] in the model dtype -- the input
        # embeddings, the residual stream after each layer but the last, the final-normed states; padded positions stay zero
        hs_all = (torch.zeros(self.config.num_hidden_layers + 1, B, S, self.config.hidden_size, dtype=eng.dtype, device=eng.device)
                  if output_hidden_states else None)
        # the rows that exist: right / left padding leaves one contiguous run of real tokens per sample (llava_arch.py:310-329); padded
        # positions keep zero logits -- the reference's values there are never consumed (their labels are IGNORE_INDEX, :320-329)
        spans = []
        for b in range(B):
            if attention_mask is not None:
                m = attention_mask[b].to(torch.bool)
                m_new = m[-S:] if m.shape[0] >= S else m
                idx = torch.nonzero(m_new, as_tuple=False).flatten()
                if idx.numel() and int(idx[-1] - idx[0]) + 1 != idx.numel():
                    raise ValueError("attention_mask with holes inside the sequence is not supported")
                if m.shape[0] > S and not bool(m[:-S].all()):
                    raise ValueError("masked positions inside the cached prefix are not supported")
            else:
                idx = torch.arange(S, device=inputs_embeds.device)
            spans.append((int(idx[0]), int(idx[-1]) + 1) if idx.numel() else None)
        if B > 1:
            # every sample in ONE pass: rows of all samples concatenated for the norms / GEMMs, attention per sample on scratch KV
            # slots (teo_llama_prefill_batch).  Positions: 0 .. S_b - 1 per sample (what the default position_ids give; explicit
            # position_ids other than that are a single-sample feature)
            live = [b for b in range(B) if spans[b] is not None]
            if position_ids is not None:
                for b in live:
                    lo, hi = spans[b]
                    pr = position_ids[b] if position_ids.dim() == 2 else position_ids
                    pr = pr[lo:hi] if pr.shape[-1] == S else pr
                    if not torch.equal(pr.to(torch.long).cpu(), torch.arange(hi - lo)):
                        raise ValueError("batched forward supports the default position_ids (0 .. len - 1 per sample) only")
            if keep_batch and len(live) != B:
                raise ValueError("use_cache=True at B > 1 needs at least one real token in every row")
            if live:
                seqs = [inputs_embeds[b, spans[b][0]:spans[b][1]] for b in live]
                if keep_batch:
                    dec = self.batch_decoder(B, 64)
                    dec.reset()
                    out = dec.prefill_all(seqs, last_only=False, hidden_states=hs_all is not None)
                else:
                    out = eng.prefill_batch(seqs, hidden_states=hs_all is not None)
                out, hs = out if hs_all is not None else (out, None)
                r0 = 0
                for b in live:
                    lo, hi = spans[b]
                    logits[b, lo:hi] = out[r0:r0 + hi - lo]
                    if hs is not None:
                        hs_all[:, b, lo:hi] = hs[:, r0:r0 + hi - lo]
                    r0 += hi - lo
        elif spans[0] is not None:
            lo, hi = spans[0]
            pos = None
            if position_ids is not None:
                pr = position_ids[0] if position_ids.dim() == 2 else position_ids
                pos = pr[lo:hi] if pr.shape[-1] == S else pr
            out = eng.prefill(inputs_embeds[0, lo:hi], positions=pos, last_only=False, hidden_states=hs_all is not None,
                              attentions=bool(output_attentions))
            if output_attentions:
                # [layers, H, hi - lo, past + hi - lo] over the real rows; padded query rows / key columns of the caller's [S, past + S]
                # frame stay zero (the contract of padded positions everywhere in this forward)
                att = out[-1]
                out = out[0] if hs_all is None else out[:2]
                att_all = torch.zeros(att.shape[0], 1, att.shape[1], S, past + S, dtype=att.dtype, device=att.device)
                att_all[:, 0, :, lo:hi, :past] = att[..., :past]
                att_all[:, 0, :, lo:hi, past + lo:past + hi] = att[..., past:]
            if hs_all is not None:
                out, hs = out
                hs_all[:, 0, lo:hi] = hs
            logits[0, lo:hi] = out
        pkv = TeoKVCache(eng) if (use_cache is None or use_cache) and B == 1 else None
        if keep_batch:
            pkv = TeoBatchKVCache(eng, self._batch_decoder)
        attn_t = None
        if output_attentions:
            if B == 1 and spans[0] is not None:
                attn_t = tuple(att_all[i] for i in range(att_all.shape[0]))
            else:
                attn_t = tuple(torch.zeros(1, self.config.num_attention_heads, S, past + S, dtype=eng.dtype, device=eng.device)
                               for _ in range(self.config.num_hidden_layers))
        return self._forward_result(logits, labels, pkv, hs_all, return_dict, attn_t)

    def _forward_result(self, logits, labels, pkv, hs_all, return_dict, attentions=None):
        eng = self.engine
        loss = None
        if labels is not None:
            loss = self._shifted_loss(logits, labels)
        out = CausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=pkv, attentions=attentions,
                                     hidden_states=tuple(hs_all[i] for i in range(hs_all.shape[0])) if hs_all is not None else None)
        if return_dict is False:
            return out.to_tuple()
        return out

    def _shifted_loss(self, logits, labels):
        """N4: training-shape loss (CrossEntropyLoss over the shifted positions) on the device: row (b, s) pairs logits[b, s] with
        labels[b, s + 1]; the last position of every sample is ignored."""
        eng = self.engine
        V = self.config.vocab_size
        lab = labels.to(device=logits.device, dtype=torch.int64)
        if bool(((lab != IGNORE_INDEX) & ((lab < 0) | (lab >= V))).any()):
            raise IndexError("Target out of bounds")                    # what torch's cross_entropy reports
        shifted = torch.cat([lab[:, 1:], torch.full_like(lab[:, :1], IGNORE_INDEX)], dim=1).reshape(-1).contiguous()
        rows = shifted.numel()
        loss_row = torch.empty(rows, dtype=torch.float32, device=logits.device)
        out3 = torch.empty(3, dtype=torch.float32, device=logits.device)
        with eng.phase() as st:
            lg = logits.reshape(rows, V)
            L.check(eng.lib.teo_cross_entropy(lg.data_ptr(), V, shifted.data_ptr(), loss_row.data_ptr(), out3.data_ptr(), rows,
                                              V, IGNORE_INDEX, st), "teo_cross_entropy")
        return out3[0]

    def _forward_batch_step(self, input_ids, cache, labels, output_hidden_states, return_dict, B, S):
        """forward(input_ids [B, 1], past_key_values=TeoBatchKVCache): conversation b's new token at ITS OWN next position
        (cache_len[b]); logits [B, 1, V] and the same cache object.  Positions: the reference derives them from the text-level
        attention_mask (sum(mask) - 1, llava_arch.py:157-162), which equals the true next position only when no row of the batch is
        padded; here every row continues at its true length whatever the mask says (stated in INTEGRATION.md)."""
        eng = self.engine
        if cache.engine is not eng or cache.decoder is not getattr(self, "_batch_decoder", None):
            raise ValueError("past_key_values must be the TeoBatchKVCache returned by this model's last batched forward")
        if S != 1 or input_ids is None:
            raise ValueError("batched continuation takes ONE new token id per conversation (input_ids [B, 1])")
        if B != cache.decoder.B:
            raise ValueError(f"the cache holds {cache.decoder.B} conversations, input_ids has {B} rows")
        if output_hidden_states:
            raise NotImplementedError("output_hidden_states on a batched decode step (the fused step keeps no per-layer snapshots)")
        logits = cache.decoder.forward_step(input_ids[:, 0]).view(B, 1, -1)
        return self._forward_result(logits, labels, cache, None, return_dict)

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, inputs_embeds=None, **kwargs):
        images = kwargs.pop("images", None)
        if past_key_values is not None:
            input_ids = input_ids[:, -1:]
        inputs = {"input_ids": input_ids, "past_key_values": past_key_values, "use_cache": kwargs.get("use_cache"),
                  "attention_mask": kwargs.get("attention_mask")}
        if inputs_embeds is not None and past_key_values is None:
            inputs = {"inputs_embeds": inputs_embeds, **{k: v for k, v in inputs.items() if k != "input_ids"}}
        if images is not None:
            inputs["images"] = images
        return inputs

    # --- H16
    @torch.no_grad()
    def generate(self, *args, **kwargs):
        # one engine phase around the whole call: tower, projector, splice, prefill and the decode chunks run back to back on the
        # engine stream with a single hand-over to the caller's stream at the end (no stream edges between the pieces)
        with self.engine.phase():
            return self._generate(*args, **kwargs)

    def _generate(self, input_ids=None, images=None, do_sample=None, temperature=None, top_k=None, top_p=None,
                  max_new_tokens=20, use_cache=True, stopping_criteria=None, eos_token_id="config", attention_mask=None,
                  generator=None, chunk=16, **kwargs):
        """Greedy or sampled (temperature / top-k, device sampler) decoding of ONE sequence; the loop is device-resident
        and replayed from a hipGraph.

        Returns int64 [1, n_prompt + n_generated]; the prompt part still contains the -200 sentinels, as with the
        reference (eval/inference.py:75 slices at input_ids.shape[1]).  `eos_token_id=None` disables EOS stopping.
        """
        # unspecified sampling knobs come from the checkpoint's generation_config.json (HF GenerationMixin semantics)
        gc = self.generation_config
        do_sample = bool(getattr(gc, "do_sample", False)) if do_sample is None else do_sample
        temperature = float(getattr(gc, "temperature", 1.0) or 1.0) if temperature is None else temperature
        top_k = getattr(gc, "top_k", 50) if top_k is None else top_k
        top_p = getattr(gc, "top_p", 1.0) if top_p is None else top_p
        if input_ids.shape[0] != 1:
            # batch of conversations: rows are cut by attention_mask, `images` is a list with one entry per conversation
            B = input_ids.shape[0]
            rows = [input_ids[b][attention_mask[b].to(torch.bool)] if attention_mask is not None else input_ids[b]
                    for b in range(B)]
            if images is not None and (not isinstance(images, (list, tuple)) or len(images) != B):
                raise ValueError("batched generate(): `images` must be a list with one entry (frame list) per conversation")
            outs = self.generate_batch(rows, images, do_sample=do_sample, temperature=temperature, top_k=top_k, top_p=top_p,
                                       max_new_tokens=max_new_tokens, stopping_criteria=stopping_criteria,
                                       eos_token_id=eos_token_id, generator=generator, chunk=chunk)
            # what GenerationMixin.generate returns: the input rows exactly as given (left or right padding included) followed
            # by the new tokens of each row; rows that stopped early are filled up with pad_token_id
            pad = getattr(self.config, "pad_token_id", None)
            pad = 0 if pad is None else int(pad)
            news = [o[rows[b].numel():] for b, o in enumerate(outs)]
            n_new = max(int(t.numel()) for t in news)
            tail = torch.full((B, n_new), pad, dtype=input_ids.dtype, device=input_ids.device)
            for b, t in enumerate(news):
                tail[b, :t.numel()] = t.to(tail.device)
            return torch.cat([input_ids, tail], dim=1)
        eng = self.engine
        if eos_token_id == "config":
            eos_token_id = self._config_eos()
        crits = list(stopping_criteria or [])
        if max_new_tokens <= 0:
            return input_ids
        # ---- prefill
        (_, pos, mask, _, embeds, _) = self.prepare_inputs_labels_for_multimodal(input_ids, None, attention_mask, None,
                                                                               None, images)
        if embeds is None:
            embeds = self.get_model().embed_tokens(input_ids)
        eng.reset_cache()
        if embeds.shape[1] + max_new_tokens > eng.max_seq:
            raise ValueError(f"prompt ({embeds.shape[1]}) + max_new_tokens ({max_new_tokens}) exceeds max_seq {eng.max_seq}")
        logits = eng.prefill(embeds[0], last_only=True)
        if do_sample:
            tp = 1.0 if top_p is None else float(top_p)
            k = int(top_k or 0)                    # 0 = top-k filter off (HF: top_k=0 / None disables TopKLogitsWarper)
            seed = generator.initial_seed() if generator is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
            first = eng.sample(logits[0], temperature, k, seed, 0, top_p=tp)
        else:
            k, seed, tp = 0, 0, 1.0
            first = self._argmax(logits[0])
        new_tokens = [first]

        def done(tokens):
            if eos_token_id is not None and tokens[-1] == eos_token_id:
                return True
            if crits:
                row = torch.cat([input_ids[0].cpu(), torch.tensor(tokens, dtype=torch.long)]).unsqueeze(0)
                return any(bool(c(row, None)) for c in crits)
            return False

        if done(new_tokens) or max_new_tokens == 1:
            return self._finish(input_ids, new_tokens)
        # ---- decode: the whole loop (incl. the sampler) runs on the device, replayed from a hipGraph in chunks;
        # the host only looks at the tokens once per chunk to apply EOS / stopping criteria at the exact token.
        cands = [ids for c in crits for ids in getattr(c, "keyword_id_lists", []) if ids]
        if eos_token_id is not None:
            cands.append([int(eos_token_id)])
        # the reference's default call stops on the keyword "</s>" AND on EOS (eval/inference.py:57-72), which are the same id
        # sequence [2]: duplicates are folded so that this default arms the device-side stop
        uniq = []
        for ids in cands:
            ids = [int(t) for t in ids]
            if ids not in uniq:
                uniq.append(ids)
        stop_ids = uniq[0] if len(uniq) == 1 else None
        eng.decode_begin(first, stop_ids, do_sample=do_sample, temperature=temperature, top_k=k, seed=seed, draws_done=1, top_p=tp)
        remaining = max_new_tokens - 1
        while remaining > 0:
            n = min(chunk, remaining)
            eng.decode_steps(n, use_graph=True)
            got = eng.generated().tolist()
            fresh = got[len(new_tokens) - 1:]
            stop_here = False
            for t in fresh:
                new_tokens.append(int(t))
                remaining -= 1
                if done(new_tokens):
                    stop_here = True
                    break
            if stop_here:
                break
        return self._finish(input_ids, new_tokens)

    # --- batched decode (config C5's variant): B conversations, one pass over the weights per generated token
    def batch_decoder(self, batch, max_new=1024):
        from .batch import BatchDecoder
        cur = getattr(self, "_batch_decoder", None)
        if cur is None or cur.B != batch or cur.max_new < max_new:
            self._batch_decoder = None          # free the old caches first
            cur = BatchDecoder(self.engine, batch, max_new=max(max_new, 64))
            self._batch_decoder = cur
        return cur

    @torch.no_grad()
    def generate_batch(self, *args, **kwargs):
        with self.engine.phase():
            return self._generate_batch(*args, **kwargs)

    def _generate_batch(self, input_ids_list, images_list=None, do_sample=False, temperature=None, top_k=None, top_p=None,
                        max_new_tokens=20, stopping_criteria=None, eos_token_id="config", generator=None, chunk=16):
        """Decode B conversations together (B <= 16).  input_ids_list: B 1-D id tensors (with -200 sentinels);
        images_list: per conversation what generate() takes as `images`.  stopping_criteria: None, or one list of
        criteria per conversation.  Returns B 1-D tensors prompt + generated, each cut at its own EOS / stop keyword.

        One multimodal preparation and ONE prefill pass over the concatenated rows of all conversations (per-sequence RoPE,
        KV append and causal attention); the decode loop is batched: per step every weight matrix is streamed once for all
        conversations (teo_llama_decode_batch_step)."""
        B = len(input_ids_list)
        if eos_token_id == "config":
            eos_token_id = self._config_eos()
        if max_new_tokens <= 0:
            return [ids.clone() for ids in input_ids_list]
        gc = self.generation_config
        temperature = float(getattr(gc, "temperature", 1.0) or 1.0) if temperature is None else temperature
        top_k = getattr(gc, "top_k", 50) if top_k is None else top_k
        top_p = getattr(gc, "top_p", 1.0) if top_p is None else top_p
        tp = 1.0 if (top_p is None or not do_sample) else float(top_p)
        crits = list(stopping_criteria) if stopping_criteria is not None else [[] for _ in range(B)]
        if len(crits) != B:
            raise ValueError("stopping_criteria must hold one list per conversation")
        eng = self.engine
        dec = self.batch_decoder(B, max_new_tokens)
        dec.reset()
        k = int(top_k or 0) if do_sample else 0
        base_seed = 0
        if do_sample:
            base_seed = generator.initial_seed() if generator is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
        seeds = [(base_seed + 0x9E3779B97F4A7C15 * b) & (2 ** 63 - 1) for b in range(B)]
        # one multimodal preparation for the whole batch: every frame of every conversation goes through the tower and
        # the projector in ONE call (flat image list consumed in order, llava_arch.py:284-285), one splice launch
        dev_ids = input_ids_list[0].device
        width = max(int(ids.numel()) for ids in input_ids_list)
        ids_p = torch.zeros(B, width, dtype=torch.long, device=dev_ids)
        mask_p = torch.zeros(B, width, dtype=torch.long, device=dev_ids)
        for b, ids in enumerate(input_ids_list):
            ids_p[b, :ids.numel()] = ids.view(-1)
            mask_p[b, :ids.numel()] = 1
        flat = []
        for b in range(B):
            imgs = images_list[b] if images_list is not None else None
            if imgs is not None:
                flat.extend(list(imgs))
        (_, _, new_mask, _, embeds, _) = self.prepare_inputs_labels_for_multimodal(ids_p, None, mask_p, None, None, flat or None)
        if embeds is None:
            embeds, new_mask = self.get_model().embed_tokens(ids_p), mask_p
        seqs = []
        for b in range(B):
            rows = torch.nonzero(new_mask[b].to(torch.bool), as_tuple=False).flatten()
            lo, hi = int(rows[0]), int(rows[-1]) + 1
            if hi - lo + max_new_tokens > eng.max_seq:
                raise ValueError(f"prompt ({hi - lo}) + max_new_tokens ({max_new_tokens}) exceeds max_seq {eng.max_seq}")
            seqs.append(embeds[b, lo:hi])
        logits = dec.prefill_all(seqs)             # one pass over the concatenated rows of all conversations
        firsts = [eng.sample(logits[b], temperature, k, seeds[b], 0, top_p=tp) if do_sample else self._argmax(logits[b])
                  for b in range(B)]
        new_tokens = [[t] for t in firsts]
        finished = [False] * B

        def done(b):
            toks = new_tokens[b]
            if eos_token_id is not None and toks[-1] == eos_token_id:
                return True
            if crits[b]:
                row = torch.cat([input_ids_list[b].cpu().view(-1), torch.tensor(toks, dtype=torch.long)]).unsqueeze(0)
                return any(bool(c(row, None)) for c in crits[b])
            return False

        for b in range(B):
            finished[b] = done(b) or max_new_tokens == 1
        if not all(finished):
            stop_ids = [int(eos_token_id)] if (eos_token_id is not None and not any(crits)) else None
            dec.begin(firsts, stop_ids, do_sample=do_sample, temperature=temperature, top_k=k, seeds=seeds, draws_done=1, top_p=tp)
            remaining = max_new_tokens - 1
            while remaining > 0 and not all(finished):
                n = min(chunk, remaining)
                dec.steps(n, use_graph=True)
                got = dec.generated()[:, -n:].tolist()
                for b in range(B):
                    if finished[b]:
                        continue
                    for t in got[b]:
                        new_tokens[b].append(int(t))
                        if done(b):
                            finished[b] = True
                            break
                remaining -= n
        return [torch.cat([input_ids_list[b].view(-1), torch.tensor(new_tokens[b], dtype=input_ids_list[b].dtype,
                                                                     device=input_ids_list[b].device)]) for b in range(B)]

    def _config_eos(self):
        """GenerationMixin semantics: generation_config.eos_token_id (generation_config.json of the checkpoint, or of model_base
        on the LoRA / projector branches) wins; config.json's eos_token_id is what generation_config is seeded from."""
        eos = getattr(self.generation_config, "eos_token_id", None)
        if eos is None:
            eos = getattr(self.config, "eos_token_id", None)
        if isinstance(eos, (list, tuple)):
            eos = eos[0] if eos else None
        return eos

    def _finish(self, input_ids, new_tokens):
        tail = torch.tensor([new_tokens], dtype=input_ids.dtype, device=input_ids.device)
        return torch.cat([input_ids, tail], dim=1)

    def _argmax(self, logits):
        tok = torch.empty(1, dtype=torch.int64, device=self.engine.device)
        with self.engine.phase() as st:
            lg = logits.contiguous()
            L.check(self.engine.lib.teo_argmax(lg.data_ptr(), tok.data_ptr(), 1, lg.numel(), st), "teo_argmax")
        return int(tok.item())
