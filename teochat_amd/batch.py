"""Batched decode: B conversations share one pass over the weights per generated token (config C5's batched variant).

Reference behaviour being reproduced: `GenerationMixin.generate` over a batch of prompts through
`LlavaLlamaForCausalLM.forward` (videollava/model/language_model/llava_llama.py:88-99) -- every conversation has its
own KV cache and position, finished conversations keep being stepped and are cut at their stop token by the caller.

`BatchDecoder` borrows the weights of a `TeoEngine` and owns
  * KV caches [layers][B][Hkv][S][hd] (+ V^T), conversation b = slot b,
  * per-slot descriptors for the (unchanged, one conversation at a time) prefill kernels,
  * a copy of the decode weight matrices in the TEO_GEMM_WTILED layout (include/teo_hip.h) when the activations are bf16,
  * the device-resident batch state and the hipGraph of one batched step.
"""
import ctypes as C

import torch

from . import _lib as L
from .engine import _p, reinterleave_gate_up, tile_weights


class BatchDecoder:
    def __init__(self, engine, batch, max_new=1024, tiled=True):
        if not 1 <= batch <= L.MAX_DECODE_BATCH:
            raise ValueError(f"batch {batch} outside 1..{L.MAX_DECODE_BATCH}")
        self.eng, self.B, self.lib = engine, int(batch), engine.lib
        self.tune = engine.tune                   # the descriptor copies below carry its raw pointer: keep the block alive as long as they live
        self.max_new = int(max_new)
        c = engine.cfg
        dev, dt = engine.device, engine.dtype
        Lr, Hk, hd, S = c.num_hidden_layers, c.num_key_value_heads, c.head_dim, engine.max_seq
        B = self.B
        self.k_cache = torch.zeros(Lr, B, Hk, S, hd, dtype=dt, device=dev)
        self.v_cache = torch.zeros(Lr, B, Hk, S, hd, dtype=dt, device=dev)
        self.vt_cache = torch.zeros(Lr, B, Hk, hd, S, dtype=dt, device=dev)
        self.cache_len = [0] * B
        self._keep = []
        # one prefill descriptor per slot: the engine's descriptor with the cache pointers of that slot
        self.slot_desc = []
        for b in range(B):
            d = L.LlamaDesc.from_buffer_copy(engine.llama_desc)
            d.k_cache = self._arr([self.k_cache[i, b] for i in range(Lr)])
            d.v_cache = self._arr([self.v_cache[i, b] for i in range(Lr)])
            d.vt_cache = self._arr([self.vt_cache[i, b] for i in range(Lr)])
            self.slot_desc.append(d)
        # batched-step descriptor: slot 0's caches + (optionally) operand-tiled weights
        d = L.LlamaDesc.from_buffer_copy(self.slot_desc[0])
        fp8 = engine.llama_w8 is not None
        ks = 64 if fp8 else 32
        self.tiled = bool(tiled) and dt in (torch.bfloat16, torch.float16) and c.hidden_size % ks == 0 and c.intermediate_size % ks == 0 \
            and (c.num_attention_heads * hd) % ks == 0
        self.tiled_w = None
        self.block8 = False
        if self.tiled:
            src = engine.llama_w8[0] if fp8 else engine.llama_w
            # gate/up: pairs re-interleaved in blocks of 8 rows so a gate row and its up row share one 16-row tile
            self.block8 = c.intermediate_size % 16 == 0
            tw = {k: [tile_weights(reinterleave_gate_up(w, 8) if (k == "gateup" and self.block8) else w) for w in src[k]]
                  for k in ("qkv", "o", "gateup", "down")}
            if fp8 and self.block8:
                self.gateup_s8 = [reinterleave_gate_up(s_.view(-1, 1), 8).view(-1).contiguous() for s_ in engine.llama_w8[1]["gateup"]]
                d.gateup_s = self._arr(self.gateup_s8)
            head = tile_weights(engine.lm_head8 if fp8 else engine.lm_head)
            self.tiled_w = (tw, head)
            if fp8:
                d.qkv_w8, d.o_w8 = self._arr(tw["qkv"]), self._arr(tw["o"])
                d.gateup_w8, d.down_w8 = self._arr(tw["gateup"]), self._arr(tw["down"])
                d.lm_head8 = head.data_ptr()
            else:
                d.qkv_w, d.o_w = self._arr(tw["qkv"]), self._arr(tw["o"])
                d.gateup_w, d.down_w = self._arr(tw["gateup"]), self._arr(tw["down"])
                d.lm_head = head.data_ptr()
        self.desc = d
        # the engine's per-engine options (TeoEngine.set_options) reach the copies made above
        import weakref
        me = weakref.ref(self)

        def _sync_options(src):
            o = me()
            if o is not None:
                for dd in o.slot_desc + [o.desc]:
                    dd.prefill_fp8, dd.rope_in_attn = src.prefill_fp8, src.rope_in_attn
        engine._option_hooks.append(_sync_options)

        def _knob_changed():                      # TeoEngine.tune_set: the captured batched step keeps the choices of its capture
            o = me()
            if o is not None:
                o._drop_graph()
        engine._tune_hooks.append(_knob_changed)
        # device state
        self.d_token = torch.zeros(B, dtype=torch.int64, device=dev)
        self.d_pos = torch.zeros(B, dtype=torch.int32, device=dev)
        self.d_out = torch.zeros(B, self.max_new, dtype=torch.int64, device=dev)
        self.d_count = torch.zeros(B, dtype=torch.int32, device=dev)
        self.d_stop = torch.zeros(B, dtype=torch.int32, device=dev)
        self.d_stop_ids = torch.zeros(16, dtype=torch.int64, device=dev)
        self.d_logits = torch.zeros(B, c.vocab_size, dtype=torch.float32, device=dev)
        self.d_rng = torch.zeros(B, 2, dtype=torch.int64, device=dev)
        s = L.DecodeBatchState()
        s.batch, s.out_stride = B, self.max_new
        s.cache_stride = self.k_cache.stride(1)
        assert self.v_cache.stride(1) == s.cache_stride and self.vt_cache.stride(1) == s.cache_stride
        s.w_tiled = 1 if self.tiled else 0
        s.gateup_block8 = 1 if self.block8 else 0
        s.d_token, s.d_pos, s.d_out_tokens = self.d_token.data_ptr(), self.d_pos.data_ptr(), self.d_out.data_ptr()
        s.d_out_count, s.d_stop = self.d_count.data_ptr(), self.d_stop.data_ptr()
        s.d_stop_ids, s.n_stop_ids, s.d_logits = self.d_stop_ids.data_ptr(), 0, self.d_logits.data_ptr()
        s.do_sample, s.top_k, s.temperature, s.d_rng, s.top_p = 0, 0, 1.0, self.d_rng.data_ptr(), 1.0
        self.state = s
        self._graph = None
        self._steps_done = 0
        self._armed = False

    def _arr(self, tensors):
        arr, pp = L.ptr_array([t.data_ptr() for t in tensors])
        self._keep.append((arr, tensors))
        return pp

    # ------------------------------------------------------------------ prefill (one conversation at a time)
    def reset(self):
        self.cache_len = [0] * self.B
        self._armed = False

    def prefill(self, slot, embeds, last_only=True):
        """Append embeds [S, D] to conversation `slot`; returns fp32 logits ([1, V] with last_only)."""
        eng = self.eng
        S = embeds.shape[0]
        past = self.cache_len[slot]
        if past + S > eng.max_seq:
            raise ValueError(f"sequence length {past + S} exceeds the engine's max_seq {eng.max_seq}")
        d = self.slot_desc[slot]
        with eng.phase() as st:
            e = embeds.to(device=eng.device, dtype=eng.dtype).contiguous()
            pos = torch.arange(past, past + S, dtype=torch.int32, device=eng.device)
            rows = 1 if last_only else S
            logits = torch.empty(rows, eng.cfg.vocab_size, dtype=torch.float32, device=eng.device)
            eng._flush_handoff_checks("prefill")
            ws = eng._workspace("prefill", self.lib.teo_llama_prefill_workspace_bytes(C.byref(d), S))
            L.check(self.lib.teo_llama_prefill(C.byref(d), _p(e), _p(pos), S, past, 1 if last_only else 0, _p(logits), _p(ws),
                                               ws.numel(), st, None), "teo_llama_prefill")
            sid = C.c_void_p(eng.stream.cuda_stream)
            eng._check_handoffs("prefill", lambda f: self.lib.teo_llama_prefill_workspace_status(C.byref(d), S, _p(ws), ws.numel(), C.byref(f), sid),
                                "teo_llama_prefill")
        self.cache_len[slot] = past + S
        return logits

    def prefill_all(self, embeds_list, last_only=True, hidden_states=False):
        """Prefill every slot at once from fresh caches: embeds_list[b] is [S_b, D].  The rows are concatenated so norms
        and GEMMs run over sum(S_b) rows (teo_llama_prefill_batch).  Returns fp32 logits [B, V] (last positions), or with
        last_only=False [sum(S_b), V] (every position, rows in the order of the list: what a forward() that keeps its cache
        returns); hidden_states=True additionally returns the [layers + 1, sum(S_b), D] residual-stream snapshots."""
        eng = self.eng
        if len(embeds_list) != self.B:
            raise ValueError(f"need {self.B} sequences")
        lens = [int(e.shape[0]) for e in embeds_list]
        if max(lens) > eng.max_seq:
            raise ValueError(f"sequence length {max(lens)} exceeds the engine's max_seq {eng.max_seq}")
        total = sum(lens)
        with eng.phase() as st:
            rows = torch.cat([e.to(device=eng.device, dtype=eng.dtype) for e in embeds_list], dim=0).contiguous()
            logits = torch.empty(self.B if last_only else total, eng.cfg.vocab_size, dtype=torch.float32, device=eng.device)
            hs = (torch.empty(eng.cfg.num_hidden_layers + 1, total, eng.cfg.hidden_size, dtype=eng.dtype, device=eng.device)
                  if hidden_states else None)
            eng._flush_handoff_checks("prefill")
            d0 = self.slot_desc[0]
            ws = eng._workspace("prefill", self.lib.teo_llama_prefill_workspace_bytes(C.byref(d0), total))
            arr = (C.c_int * self.B)(*lens)
            L.check(self.lib.teo_llama_prefill_batch(C.byref(d0), _p(rows), arr, self.B, self.k_cache.stride(1), 1 if last_only else 0,
                                                     _p(logits), _p(ws), ws.numel(), st, _p(hs) if hs is not None else None),
                    "teo_llama_prefill_batch")
            sid = C.c_void_p(eng.stream.cuda_stream)
            eng._check_handoffs("prefill", lambda f: self.lib.teo_llama_prefill_workspace_status(C.byref(d0), total, _p(ws), ws.numel(), C.byref(f), sid),
                                "teo_llama_prefill_batch")
        self.cache_len = list(lens)
        self._armed = False
        return (logits, hs) if hidden_states else logits

    # ------------------------------------------------------------------ decode
    def _workspace(self):
        return self.eng._workspace("decode_batch", self.lib.teo_llama_decode_batch_workspace_bytes(C.byref(self.desc), self.B))

    def _drop_graph(self):
        if self._graph is not None:
            self.lib.teo_graph_destroy(self._graph)
            self._graph = None

    def begin(self, first_tokens, stop_ids=None, do_sample=False, temperature=1.0, top_k=0, seeds=None, draws_done=1, top_p=1.0):
        """Arm the loop: first_tokens[b] is the input of conversation b's next step, at position cache_len[b]."""
        eng = self.eng
        if len(first_tokens) != self.B:
            raise ValueError(f"need {self.B} first tokens")
        with eng.phase():
            self.d_token.copy_(torch.tensor([int(t) for t in first_tokens], dtype=torch.int64))
            self.d_pos.copy_(torch.tensor(self.cache_len, dtype=torch.int32))
            self.d_count.zero_()
            self.d_stop.zero_()
            seeds = list(seeds) if seeds is not None else [0] * self.B
            self.d_rng.copy_(torch.tensor([[int(sd) & (2 ** 63 - 1), int(draws_done)] for sd in seeds], dtype=torch.int64))
            n = 0
            if stop_ids:
                n = min(len(stop_ids), 16)
                self.d_stop_ids[:n] = torch.tensor(list(stop_ids)[-n:], dtype=torch.int64, device=eng.device)
            s = self.state
            key = (n, int(bool(do_sample)), int(top_k or 0), C.c_float(float(temperature)).value, C.c_float(float(top_p or 1.0)).value)
            if key != (s.n_stop_ids, s.do_sample, s.top_k, float(s.temperature), float(s.top_p)):
                s.n_stop_ids, s.do_sample, s.top_k, s.temperature, s.top_p = key
                self._drop_graph()
        ws = self._workspace()
        with eng.phase() as st:
            L.check(self.lib.teo_llama_decode_batch_begin(C.byref(self.desc), C.byref(self.state), _p(ws), ws.numel(), st),
                    "teo_llama_decode_batch_begin")
        self._steps_done = 0

    def steps(self, n, use_graph=True):
        """n batched steps (every conversation advances n tokens) on the device."""
        eng = self.eng
        if max(self.cache_len) + n > eng.max_seq:
            raise ValueError(f"decode would exceed max_seq {eng.max_seq}")
        if self._steps_done + n > self.max_new:
            raise ValueError(f"decode would exceed the output buffer ({self.max_new} tokens)")
        ws = self._workspace()
        with eng.phase() as st:
            if use_graph:
                if self._graph is None or self._graph_ws != ws.data_ptr():
                    self._drop_graph()
                    g = C.c_void_p()
                    L.check(self.lib.teo_llama_decode_batch_graph_create(C.byref(self.desc), C.byref(self.state), _p(ws),
                                                                         ws.numel(), st, C.byref(g)),
                            "teo_llama_decode_batch_graph_create")
                    self._graph, self._graph_ws = g, ws.data_ptr()
                L.check(self.lib.teo_graph_launch(self._graph, n, st), "teo_graph_launch")
                eng.stream.synchronize()        # drain the replays before anything is queued behind them (engine.py _Phase.__exit__)
            else:
                for _ in range(n):
                    L.check(self.lib.teo_llama_decode_batch_step(C.byref(self.desc), C.byref(self.state), _p(ws), ws.numel(),
                                                                 st), "teo_llama_decode_batch_step")
        self.cache_len = [x + n for x in self.cache_len]
        self._steps_done += n

    def steps_profiled(self, n):
        """n batched steps as plain launches, every kernel timed by its own dispatch timestamps (teo_llama_decode_batch_step_profile).
        Returns {class: (launches per step, mean microseconds per launch)}; advances the caches like steps()."""
        from .engine import TeoEngine
        eng = self.eng
        if max(self.cache_len) + n > eng.max_seq or self._steps_done + n > self.max_new:
            raise ValueError("steps_profiled: would exceed max_seq / the output buffer")
        ws = self._workspace()
        names = TeoEngine.PROF_CLASSES
        K = len(names)
        tot, cnt = [0.0] * K, [0] * K
        ms, ct = (C.c_float * K)(), (C.c_int * K)()
        with eng.phase() as st:
            for _ in range(n):
                L.check(self.lib.teo_llama_decode_batch_step_profile(C.byref(self.desc), C.byref(self.state), _p(ws), ws.numel(), ms, ct, st),
                        "teo_llama_decode_batch_step_profile")
                for k in range(K):
                    tot[k] += ms[k]
                    cnt[k] += ct[k]
        self.cache_len = [x + n for x in self.cache_len]
        self._steps_done += n
        return {name: (cnt[k] // n, tot[k] / cnt[k] * 1e3) for k, name in enumerate(names) if cnt[k]}

    def forward_step(self, tokens):
        """One batched step fed with CALLER-chosen tokens (the continuation of LlavaLlamaForCausalLM.forward(past_key_values=...) at
        B > 1, llava_arch.py:154-163): conversation b takes tokens[b] at position cache_len[b], its K / V rows are appended, and the
        fp32 logits [B, V] of that position are returned.  The device loop's own pick (the tail kernel's argmax) is overwritten by
        the next call's tokens, so greedy streams driven from here equal generate_batch()'s token for token."""
        toks = [int(t) for t in (tokens.view(-1).tolist() if torch.is_tensor(tokens) else tokens)]
        if len(toks) != self.B:
            raise ValueError(f"need {self.B} tokens")
        s = self.state
        fresh = (not getattr(self, "_armed", False)) or self._steps_done + 1 > self.max_new \
            or (s.n_stop_ids, s.do_sample) != (0, 0)
        if fresh:
            self.begin(toks)                       # greedy, no device stop: positions from cache_len, counters zeroed
            self._armed = True
        else:
            with self.eng.phase():
                self.d_token.copy_(torch.tensor(toks, dtype=torch.int64))
                self.d_stop.zero_()
            ws = self._workspace()                 # (the step reads the token's embedding row itself: teo_llama_decode_batch_begin
            with self.eng.phase() as st:           #  re-embeds d_token)
                L.check(self.lib.teo_llama_decode_batch_begin(C.byref(self.desc), C.byref(self.state), _p(ws), ws.numel(), st),
                        "teo_llama_decode_batch_begin")
        self.steps(1, use_graph=True)
        return self.d_logits.clone()

    def generated(self):
        """[B, steps] int64: the tokens produced by the steps so far (the first token of each conversation excluded)."""
        n = self._steps_done
        return self.d_out[:, :n].clone()

    def __del__(self):
        try:
            self._drop_graph()
        except Exception:  # noqa: BLE001
            pass
