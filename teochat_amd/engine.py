"""TeoEngine: device-resident weights + KV cache + workspaces, driving libteo_hip.so through the C ABI.

PyTorch is used here only as a container: device allocation, host->device copies, dtype casts at load time and
stream handles.  Every arithmetic step of the forward path is a teo_* call (include/teo_hip.h).

Weights arrive keyed by the reference's state-dict names and are re-laid-out once:
  * ViT q/k/v Linear weights+biases fused to one [3D, D] GEMM; patch conv weight flattened to [D, 3*P*P] and
    zero-padded along K to a multiple of 64;
  * LLaMA q/k/v fused to [(H + 2*Hkv)*hd, D]; gate/up fused to [2F, D] with rows interleaved in blocks of 16
    (gate rows 16b..16b+15, then up rows 16b..16b+15) so the SwiGLU epilogue finds both operands in one lane;
  * RoPE cos/sin tables [max_pos, hd/2] fp32, built on the host exactly as tf LlamaRotaryEmbedding computes them;
  * KV cache per layer: K [Hkv][S_max][hd], V [Hkv][S_max][hd] (decode streams rows) and V^T [Hkv][hd][S_max]
    (prefill MFMA attention wants key-contiguous values) -- 288 GB of HBM3E pays for the second V layout.
"""
import ctypes as C
import math
import warnings

import torch

from . import _lib as L

_ACT = {"quick_gelu": L.ACT_QUICK_GELU, "gelu": L.ACT_GELU_ERF}
VIT_PREFIX = "model.image_tower.image_tower."
SEED_MASK = 2 ** 63 - 1            # d_rng is int64 on the host side: one mask for the first draw and the device loop


def _dt(dtype):
    if dtype == torch.float32:
        return L.TEO_F32
    if dtype == torch.bfloat16:
        return L.TEO_BF16
    if dtype == torch.float16:
        return L.TEO_F16
    raise ValueError(f"unsupported engine dtype {dtype}; use torch.float32, torch.bfloat16 or torch.float16")


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def tile_weights(W):
    """Row-major [N, K] (bf16, or uint8 holding fp8-e4m3 bits) -> the TEO_GEMM_WTILED layout of include/teo_hip.h:
    ceil(N/16) * (K/KS) tiles of 1 KB in v_mfma_f32_16x16x32_bf16 operand order (rows past N are zero)."""
    N, K = W.shape
    ks, ch = (64, 16) if W.element_size() == 1 else (32, 8)
    if K % ks:
        raise ValueError(f"tile_weights: K={K} must be a multiple of {ks}")
    npad = (N + 15) // 16 * 16
    if npad != N:
        W = torch.cat([W, torch.zeros(npad - N, K, dtype=W.dtype, device=W.device)])
    return W.view(npad // 16, 16, K // ks, 4, ch).permute(0, 2, 3, 1, 4).contiguous().view(npad, K)


def reinterleave_gate_up(gu, block):
    """gate/up rows interleaved in blocks of 16 (the engine's layout) -> blocks of `block` rows (same pairs)."""
    n, k = gu.shape
    f = n // 2
    g16 = gu.view(f // 16, 2, 16, k)
    gate, up = g16[:, 0].reshape(f, k), g16[:, 1].reshape(f, k)
    return torch.stack([gate.view(f // block, block, k), up.view(f // block, block, k)], dim=1).reshape(n, k)


def interleave_gate_up(gate, up):
    """[F, D], [F, D] -> [2F, D] with 16-row blocks alternating gate/up."""
    F_, D = gate.shape
    assert F_ % 16 == 0, "intermediate_size must be a multiple of 16"
    return torch.stack((gate.view(F_ // 16, 16, D), up.view(F_ // 16, 16, D)), dim=1).reshape(2 * F_, D).contiguous()


def quantize_fp8_rows(w):
    """Per-output-row fp8 (OCP e4m3fn) quantisation with POWER-OF-TWO scales: W ~= q * s[:, None].

    Returns (q uint8 [N,K], s fp32 [N], dq = q*s in w's dtype).  With power-of-two scales q*s is exactly representable
    in bf16 (e4m3 has 3 mantissa bits), so a bf16 GEMM on dq and an fp8 GEMV on (q, s) see the same weights."""
    wf = w.float()
    amax = wf.abs().amax(dim=1).clamp_min(1e-30)
    s = torch.exp2(torch.ceil(torch.log2(amax / 448.0)))
    q = (wf / s[:, None]).to(torch.float8_e4m3fn)
    dq = (q.float() * s[:, None]).to(w.dtype)
    return q.view(torch.uint8).contiguous(), s.contiguous(), dq.contiguous()


def rope_tables(head_dim, theta, max_pos):
    """cos/sin [max_pos, hd/2] fp32; inv_freq and pos*inv_freq in fp32 on the host (tf LlamaRotaryEmbedding)."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv[None, :]
    return fr.cos().contiguous(), fr.sin().contiguous()


class TeoEngine:
    def __init__(self, state_dict, config, dtype=torch.bfloat16, device="cuda:0", max_seq=None, weight_format=None):
        self.lib = L.load()                       # raises TeoLibraryError when the HIP library is missing
        if not torch.cuda.is_available():
            raise RuntimeError("TeoEngine needs an MI355X (no CPU fallback exists for the product path)")
        self.cfg = config
        self.vcfg = config.vision_config
        self.dtype = dtype
        self.dt = _dt(dtype)
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.max_seq = int(math.ceil((max_seq or config.max_position_embeddings) / 64.0) * 64)
        self.weight_format = weight_format or "native"          # "fp8": decode streams fp8-e4m3 copies (config C5)
        if self.weight_format not in ("native", "fp8"):
            raise ValueError(f"unknown weight_format {weight_format!r}")
        if self.weight_format == "fp8" and dtype != torch.bfloat16:
            # the power-of-two row scales make every dequantised e4m3 weight an exact bfloat16; in binary16 the smallest ones fall
            # into the subnormal range and would not be exact (and the fp8 GEMV / skinny kernels convert e4m3 -> bf16)
            raise ValueError("weight_format='fp8' needs dtype=torch.bfloat16")
        self._keep = []                           # host pointer arrays referenced by the descriptors
        self._ws = {}
        self._phase_depth = 0
        self._pending_status = {}                 # workspace key -> deferred hand-off check (see _check_handoffs)
        self._option_hooks = []                   # callables(desc): keep copies of the LLaMA descriptor in step with set_options
        self._tune_hooks = []                     # callables(): a knob changed -- captured graphs bake kernel choices in and are dropped
        self._graph = None
        # performance knobs of THIS engine: a teo_tune block every descriptor points at (include/teo_hip.h); nothing process-wide
        self.tune = L.Tune()
        self._load_vit(state_dict)
        self._load_projector(state_dict)
        self._load_llama(state_dict)
        self._alloc_cache()
        self._alloc_decode_state(max_new=max(4096, self.max_seq))

    # ------------------------------------------------------------------ weights
    def _dev(self, t):
        return t.to(device=self.device, dtype=self.dtype).contiguous()

    def _arr(self, tensors):
        arr, pp = L.ptr_array([t.data_ptr() for t in tensors])
        self._keep.append((arr, tensors))
        return pp

    def _load_vit(self, sd):
        v = self.vcfg
        n_states = v.num_hidden_layers + 1
        sel = self.cfg.mm_vision_select_layer
        idx = sel if sel >= 0 else n_states + sel
        if not 0 <= idx <= v.num_hidden_layers:
            raise ValueError(f"mm_vision_select_layer {sel} out of range")
        self.vit_layers_run = idx                  # layers after the selected hidden state are dead code
        if v.hidden_act not in _ACT:
            raise ValueError(f"unsupported vision hidden_act {v.hidden_act}")
        D, P = v.hidden_size, v.patch_size
        K = v.num_channels * P * P
        self.vit_kpad = (K + 63) // 64 * 64
        pw = sd[VIT_PREFIX + "embeddings.patch_embedding.weight"].reshape(D, K)
        pw_pad = torch.zeros(D, self.vit_kpad, dtype=pw.dtype, device=pw.device)
        pw_pad[:, :K] = pw
        w = {"patch_w": self._dev(pw_pad),
             "cls": self._dev(sd[VIT_PREFIX + "embeddings.class_embedding"]),
             "pos": self._dev(sd[VIT_PREFIX + "embeddings.position_embedding.weight"]),
             "pre_w": self._dev(sd[VIT_PREFIX + "pre_layrnorm.weight"]),
             "pre_b": self._dev(sd[VIT_PREFIX + "pre_layrnorm.bias"])}
        per = {k: [] for k in ("ln1_w", "ln1_b", "qkv_w", "qkv_b", "out_w", "out_b", "ln2_w", "ln2_b", "fc1_w", "fc1_b",
                               "fc2_w", "fc2_b")}
        for i in range(self.vit_layers_run):
            p = VIT_PREFIX + f"encoder.layers.{i}."
            per["ln1_w"].append(self._dev(sd[p + "layer_norm1.weight"]))
            per["ln1_b"].append(self._dev(sd[p + "layer_norm1.bias"]))
            per["qkv_w"].append(self._dev(torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], dim=0)))
            per["qkv_b"].append(self._dev(torch.cat([sd[p + f"self_attn.{n}_proj.bias"] for n in "qkv"], dim=0)))
            per["out_w"].append(self._dev(sd[p + "self_attn.out_proj.weight"]))
            per["out_b"].append(self._dev(sd[p + "self_attn.out_proj.bias"]))
            per["ln2_w"].append(self._dev(sd[p + "layer_norm2.weight"]))
            per["ln2_b"].append(self._dev(sd[p + "layer_norm2.bias"]))
            per["fc1_w"].append(self._dev(sd[p + "mlp.fc1.weight"]))
            per["fc1_b"].append(self._dev(sd[p + "mlp.fc1.bias"]))
            per["fc2_w"].append(self._dev(sd[p + "mlp.fc2.weight"]))
            per["fc2_b"].append(self._dev(sd[p + "mlp.fc2.bias"]))
        self.vit_w = (w, per)
        d = L.VitDesc()
        d.hidden, d.heads, d.inter, d.layers_run = D, v.num_attention_heads, v.intermediate_size, self.vit_layers_run
        d.image, d.patch, d.channels = v.image_size, v.patch_size, v.num_channels
        d.act, d.eps, d.dtype, d.k_pad = _ACT[v.hidden_act], v.layer_norm_eps, self.dt, self.vit_kpad
        d.patch_w, d.cls, d.pos = w["patch_w"].data_ptr(), w["cls"].data_ptr(), w["pos"].data_ptr()
        d.pre_ln_w, d.pre_ln_b = w["pre_w"].data_ptr(), w["pre_b"].data_ptr()
        for k in per:
            setattr(d, k, self._arr(per[k]))
        sf = getattr(self.cfg, "mm_vision_select_feature", "patch")
        if sf not in ("patch", "cls_patch"):
            raise ValueError(f"Unexpected select feature: {sf}")        # languagebind/__init__.py:128
        d.keep_cls = 1 if sf == "cls_patch" else 0
        self.vit_tokens = v.num_patches + d.keep_cls                    # rows per frame the tower returns
        d.tune = self.tune.ptr
        self.vit_desc = d

    def _load_projector(self, sd):
        import re
        pt = getattr(self.cfg, "mm_projector_type", "linear")
        pre = "model.mm_projector."
        d = L.ProjDesc()
        d.in_dim, d.out_dim, d.dtype = self.cfg.mm_hidden_size, self.cfg.hidden_size, self.dt
        ws, bs = [], []
        if pt == "linear":
            ws.append(self._dev(sd[pre + "weight"])); bs.append(self._dev(sd[pre + "bias"]))
        elif pt == "identity":
            pass
        else:
            m = re.match(r"^mlp(\d+)x_gelu$", pt)
            if not m:
                raise ValueError(f"Unknown projector type: {pt}")
            depth = int(m.group(1))
            if depth > 4:
                raise ValueError("projector depth > 4 not supported")
            for j in range(depth):
                ws.append(self._dev(sd[pre + f"{2 * j}.weight"])); bs.append(self._dev(sd[pre + f"{2 * j}.bias"]))
        d.depth = len(ws)
        for j, (w, b) in enumerate(zip(ws, bs)):
            d.w[j], d.b[j] = w.data_ptr(), b.data_ptr()
        self.proj_w = (ws, bs)
        d.tune = self.tune.ptr
        self.proj_desc = d

    def _load_llama(self, sd):
        c = self.cfg
        H, Hk, hd = c.num_attention_heads, c.num_key_value_heads, c.head_dim
        self.embed = self._dev(sd["model.embed_tokens.weight"])
        self.final_norm = self._dev(sd["model.norm.weight"])
        self.lm_head = self._dev(sd["lm_head.weight"])
        self.max_pos = max(c.max_position_embeddings, self.max_seq)
        cs, sn = rope_tables(hd, c.rope_theta, self.max_pos)
        self.rope_cos, self.rope_sin = cs.to(self.device), sn.to(self.device)
        per = {k: [] for k in ("in_norm", "qkv", "o", "post_norm", "gateup", "down")}
        for i in range(c.num_hidden_layers):
            p = f"model.layers.{i}."
            per["in_norm"].append(self._dev(sd[p + "input_layernorm.weight"]))
            per["qkv"].append(self._dev(torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], dim=0)))
            per["o"].append(self._dev(sd[p + "self_attn.o_proj.weight"]))
            per["post_norm"].append(self._dev(sd[p + "post_attention_layernorm.weight"]))
            per["gateup"].append(self._dev(interleave_gate_up(sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"])))
            per["down"].append(self._dev(sd[p + "mlp.down_proj.weight"]))
        self.llama_w = per
        self.llama_w8 = None
        if self.weight_format == "fp8":
            # quantise the (already fused / interleaved) Linear weights row-wise; prefill keeps using the bf16 tensors,
            # which are replaced by the exactly-dequantised values so both paths see identical weights
            w8 = {k: [] for k in ("qkv", "o", "gateup", "down")}
            s8 = {k: [] for k in ("qkv", "o", "gateup", "down")}
            for k in w8:
                for i in range(c.num_hidden_layers):
                    q, s, dq = quantize_fp8_rows(per[k][i])
                    per[k][i] = dq
                    w8[k].append(q)
                    s8[k].append(s)
            q, s, dq = quantize_fp8_rows(self.lm_head)
            self.lm_head, self.lm_head8, self.lm_head_s = dq, q, s
            self.llama_w8 = (w8, s8)

    def _alloc_cache(self):
        c = self.cfg
        Lr, Hk, hd, S = c.num_hidden_layers, c.num_key_value_heads, c.head_dim, self.max_seq
        self.k_cache = torch.zeros(Lr, Hk, S, hd, dtype=self.dtype, device=self.device)
        self.v_cache = torch.zeros(Lr, Hk, S, hd, dtype=self.dtype, device=self.device)
        self.vt_cache = torch.zeros(Lr, Hk, hd, S, dtype=self.dtype, device=self.device)
        self.cache_len = 0
        d = L.LlamaDesc()
        d.hidden, d.heads, d.kv_heads, d.head_dim = c.hidden_size, c.num_attention_heads, Hk, hd
        d.inter, d.layers, d.vocab, d.eps = c.intermediate_size, Lr, c.vocab_size, c.rms_norm_eps
        d.dtype, d.max_seq = self.dt, S
        d.embed, d.final_norm_w, d.lm_head = self.embed.data_ptr(), self.final_norm.data_ptr(), self.lm_head.data_ptr()
        d.rope_cos, d.rope_sin, d.max_pos = self.rope_cos.data_ptr(), self.rope_sin.data_ptr(), self.max_pos
        d.in_norm_w = self._arr(self.llama_w["in_norm"])
        d.qkv_w = self._arr(self.llama_w["qkv"])
        d.o_w = self._arr(self.llama_w["o"])
        d.post_norm_w = self._arr(self.llama_w["post_norm"])
        d.gateup_w = self._arr(self.llama_w["gateup"])
        d.down_w = self._arr(self.llama_w["down"])
        d.k_cache = self._arr([self.k_cache[i] for i in range(Lr)])
        d.v_cache = self._arr([self.v_cache[i] for i in range(Lr)])
        d.vt_cache = self._arr([self.vt_cache[i] for i in range(Lr)])
        if self.llama_w8 is not None:
            w8, s8 = self.llama_w8
            d.qkv_w8, d.qkv_s = self._arr(w8["qkv"]), self._arr(s8["qkv"])
            d.o_w8, d.o_s = self._arr(w8["o"]), self._arr(s8["o"])
            d.gateup_w8, d.gateup_s = self._arr(w8["gateup"]), self._arr(s8["gateup"])
            d.down_w8, d.down_s = self._arr(w8["down"]), self._arr(s8["down"])
            d.lm_head8, d.lm_head_s = self.lm_head8.data_ptr(), self.lm_head_s.data_ptr()
        d.tune = self.tune.ptr
        self.llama_desc = d

    def _alloc_decode_state(self, max_new=4096):
        dev = self.device
        self.d_token = torch.zeros(1, dtype=torch.int64, device=dev)
        self.d_pos = torch.zeros(1, dtype=torch.int32, device=dev)
        self.d_out = torch.zeros(max_new, dtype=torch.int64, device=dev)
        self.d_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.d_stop = torch.zeros(1, dtype=torch.int32, device=dev)
        self.d_stop_ids = torch.zeros(16, dtype=torch.int64, device=dev)
        self.d_logits = torch.zeros(self.cfg.vocab_size, dtype=torch.float32, device=dev)
        self.d_rng = torch.zeros(2, dtype=torch.int64, device=dev)       # {seed, draws so far} of the device sampler
        self.max_new_cap = max_new
        s = L.DecodeState()
        s.d_token, s.d_pos, s.d_out_tokens = self.d_token.data_ptr(), self.d_pos.data_ptr(), self.d_out.data_ptr()
        s.d_out_count, s.d_stop = self.d_count.data_ptr(), self.d_stop.data_ptr()
        s.d_stop_ids, s.n_stop_ids, s.d_logits = self.d_stop_ids.data_ptr(), 0, self.d_logits.data_ptr()
        s.do_sample, s.top_k, s.temperature, s.d_rng, s.top_p = 0, 0, 1.0, self.d_rng.data_ptr(), 1.0
        self.decode_state = s

    # ------------------------------------------------------------------ plumbing
    def _workspace(self, key, nbytes):
        t = self._ws.get(key)
        if t is None or t.numel() < nbytes:
            t = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)
            self._ws[key] = t
        return t

    class _Phase:
        def __init__(self, eng):
            self.eng = eng

        def __enter__(self):
            self.nested = self.eng._phase_depth > 0          # inside an outer phase (generate()): already on the engine stream, no edges
            self.eng._phase_depth += 1
            if self.nested:
                return C.c_void_p(self.eng.stream.cuda_stream)
            cur = torch.cuda.current_stream(self.eng.device)
            self.cur = cur
            self.eng.stream.wait_stream(cur)
            self.ctx = torch.cuda.stream(self.eng.stream)
            self.ctx.__enter__()
            return C.c_void_p(self.eng.stream.cuda_stream)

        def __exit__(self, *a):
            self.eng._phase_depth -= 1
            if self.nested:
                return False
            self.ctx.__exit__(*a)
            # Drain the engine stream BEFORE the cross-stream edge.  Measured (tools/chunk_probe.py): an event record + wait on
            # another stream issued while hipGraph launches are still outstanding puts the not-yet-executed tail (up to ~88 decode
            # steps) into a mode where every step takes 2.93 instead of 2.71 ms (+1.1 us per kernel); with the stream drained
            # first every step runs at 2.71 ms, whatever the chunk size.  The host needs the results of a phase anyway.
            if a[0] is None:
                self.eng.stream.synchronize()
            self.cur.wait_stream(self.eng.stream)
            if a[0] is None:
                self.eng._flush_handoff_checks()     # deferred by the nested phases of generate() / generate_batch()
            else:
                self.eng._pending_status.clear()
            return False

    def phase(self):
        return TeoEngine._Phase(self)

    # ------------------------------------------------------------------ vision
    def vit_features(self, pixels):
        """[T,3,H,W] -> hidden_states[select_layer][:, 1:]  ([T, n_patches, Dv]); H7-H11."""
        v = self.vcfg
        if pixels.dim() != 4 or pixels.shape[1] != v.num_channels or pixels.shape[2] != v.image_size or pixels.shape[3] != v.image_size:
            raise ValueError(f"Input image size ({tuple(pixels.shape)}) doesn't match model "
                             f"({v.num_channels}x{v.image_size}x{v.image_size}).")
        T = pixels.shape[0]
        with self.phase() as st:
            px = pixels.to(device=self.device, dtype=self.dtype).contiguous()
            out = torch.empty(T, self.vit_tokens, v.hidden_size, dtype=self.dtype, device=self.device)
            need = self.lib.teo_vit_workspace_bytes(C.byref(self.vit_desc), T)
            self._flush_handoff_checks("vit")         # an unread status of this workspace: read it before the call re-arms the word
            ws = self._workspace("vit", need)
            L.check(self.lib.teo_vit_encode(C.byref(self.vit_desc), _p(px), T, _p(out), _p(ws), ws.numel(), st), "teo_vit_encode")
            sid = C.c_void_p(self.stream.cuda_stream)
            self._check_handoffs("vit", lambda f: self.lib.teo_vit_workspace_status(C.byref(self.vit_desc), T, _p(ws), ws.numel(), C.byref(f), sid),
                                 "teo_vit_encode")
        return out

    def _check_handoffs(self, key, status_call, what):
        """The persistent GEMM forms hand partial tiles over through workspace flags; a hand-off that timed out sets a sticky error
        word (cleared when the next call on that workspace re-arms it) instead of continuing silently.  Reading the word
        synchronises the stream, so inside a nested phase (generate() / generate_batch() wrap tower, prefill and decode in one outer
        phase) the check is DEFERRED: it runs when the outermost phase ends -- the stream is drained there anyway -- or right before
        the same workspace is used again, whichever comes first.  A timed-out hand-off therefore always raises."""
        if self._phase_depth > 1:
            self._pending_status[key] = (status_call, what)
            return
        self._run_handoff_check(status_call, what)

    def _run_handoff_check(self, status_call, what):
        flag = C.c_int(0)
        L.check(status_call(flag), what + " workspace status")
        if flag.value:
            raise RuntimeError(f"{what}: a stream-K / hybrid GEMM hand-off timed out (results invalid)")

    def _flush_handoff_checks(self, key=None):
        keys = [key] if key is not None else list(self._pending_status)
        for k in keys:
            item = self._pending_status.pop(k, None)
            if item is not None:
                self._run_handoff_check(*item)

    def project(self, feats):
        """[..., Dv] -> [..., D] through the mm_projector (H12)."""
        if self.proj_desc.depth == 0:
            return feats
        shape = feats.shape
        rows = feats.numel() // shape[-1]
        with self.phase() as st:
            x = feats.to(device=self.device, dtype=self.dtype).contiguous()
            y = torch.empty(rows, self.cfg.hidden_size, dtype=self.dtype, device=self.device)
            need = self.lib.teo_projector_workspace_bytes(C.byref(self.proj_desc), rows)
            ws = self._workspace("proj", need)
            L.check(self.lib.teo_projector(C.byref(self.proj_desc), _p(x), rows, _p(y), _p(ws), ws.numel(), st), "teo_projector")
        return y.view(*shape[:-1], self.cfg.hidden_size)

    def encode_images(self, pixels):
        return self.project(self.vit_features(pixels))

    # ------------------------------------------------------------------ splice
    def splice(self, plan, visual):
        """plan int32 [rows] (host or device) -> embeds [rows, D] (H13 data movement)."""
        with self.phase() as st:
            plan_d = plan.to(device=self.device, dtype=torch.int32).contiguous()
            rows = plan_d.numel()
            out = torch.empty(rows, self.cfg.hidden_size, dtype=self.dtype, device=self.device)
            vis = visual.to(device=self.device, dtype=self.dtype).contiguous() if visual is not None else None
            L.check(self.lib.teo_embed_splice(_p(plan_d), _p(self.embed), _p(vis), _p(out), rows, self.cfg.hidden_size,
                                              self.dt, st), "teo_embed_splice")
        return out

    # ------------------------------------------------------------------ LLaMA
    def reset_cache(self):
        self.cache_len = 0

    def prefill(self, embeds, positions=None, last_only=False, hidden_states=False, attentions=False):
        """embeds [S, D] appended to the cache; returns fp32 logits [S, V] (or [1, V]).  With hidden_states: (logits, hs) where hs is
        [layers + 1, S, D] in the model dtype -- the input embeddings, the residual stream after every layer but the last, and the
        final-normed states (HF's `output_hidden_states` tuple; llava_llama.py:88-99).  With attentions: the attention maps
        [layers, heads, S, past + S] in the model dtype are appended to the returned tuple (HF's `output_attentions`; a plain kernel
        beside the unchanged forward: teo_llama_prefill_attentions)."""
        S = embeds.shape[0]
        past = self.cache_len
        if past + S > self.max_seq:
            raise ValueError(f"sequence length {past + S} exceeds the engine's max_seq {self.max_seq}")
        with self.phase() as st:
            e = embeds.to(device=self.device, dtype=self.dtype).contiguous()
            if positions is None:
                pos = torch.arange(past, past + S, dtype=torch.int32, device=self.device)
            else:
                pos = positions.to(device=self.device, dtype=torch.int32).contiguous()
            rows = 1 if last_only else S
            logits = torch.empty(rows, self.cfg.vocab_size, dtype=torch.float32, device=self.device)
            need = self.lib.teo_llama_prefill_workspace_bytes(C.byref(self.llama_desc), S)
            self._flush_handoff_checks("prefill")
            ws = self._workspace("prefill", need)
            hs = torch.empty(self.cfg.num_hidden_layers + 1, S, self.cfg.hidden_size, dtype=self.dtype, device=self.device) if hidden_states else None
            att = None
            if attentions:
                att = torch.empty(self.cfg.num_hidden_layers, self.cfg.num_attention_heads, S, past + S, dtype=self.dtype, device=self.device)
                L.check(self.lib.teo_llama_prefill_attentions(C.byref(self.llama_desc), _p(e), _p(pos), S, past, 1 if last_only else 0,
                                                              _p(logits), _p(ws), ws.numel(), st, _p(hs) if hs is not None else None, _p(att)),
                        "teo_llama_prefill_attentions")
            else:
                L.check(self.lib.teo_llama_prefill(C.byref(self.llama_desc), _p(e), _p(pos), S, past, 1 if last_only else 0,
                                                   _p(logits), _p(ws), ws.numel(), st, _p(hs) if hs is not None else None), "teo_llama_prefill")
            sid = C.c_void_p(self.stream.cuda_stream)
            self._check_handoffs("prefill", lambda f: self.lib.teo_llama_prefill_workspace_status(C.byref(self.llama_desc), S, _p(ws), ws.numel(),
                                                                                                  C.byref(f), sid), "teo_llama_prefill")
        self.cache_len = past + S
        if attentions:
            return (logits, hs, att) if hidden_states else (logits, att)
        return (logits, hs) if hidden_states else logits

    def prefill_batch(self, embeds_list, hidden_states=False):
        """Training-shape forward of B independent sequences in ONE pass (teo_llama_prefill_batch, last_only = 0): embeds_list[b]
        is [S_b, D]; the rows are concatenated for the norms / GEMMs, RoPE + causal attention run per sequence on scratch KV slots
        owned by the engine (one [B, Hkv, S64, hd] buffer x 3 shared by all layers, grown on demand).  Returns fp32 logits
        [sum(S_b), V], rows in the order of the list.  The single-conversation cache (self.cache_len) is untouched."""
        B = len(embeds_list)
        lens = [int(e.shape[0]) for e in embeds_list]
        if max(lens) > self.max_seq:
            raise ValueError(f"sequence length {max(lens)} exceeds the engine's max_seq {self.max_seq}")
        c = self.cfg
        Lr, Hk, hd = c.num_hidden_layers, c.num_key_value_heads, c.head_dim
        # Scratch K / V / V^T for the causal attention of this pass: the forward never reads a layer's keys after that layer, so EVERY
        # layer's cache pointer aliases ONE [B, Hkv, S64, hd] buffer (S64 = the longest sequence rounded up to the 64-key tile of the
        # flash kernel), sized by what is asked for, not by the engine's max_seq: 7B at B = 16, S = 4096 is 3 x 0.5 GB instead of 3 x 17 GB
        S64 = (max(lens) + 63) // 64 * 64
        slot = getattr(self, "_fwd_slots", None)
        if slot is None or slot["B"] < B or slot["S"] < S64:
            # regrow to the maximum of the old and the new shape (alternating (B = 4, S = 64) / (B = 2, S = 128) calls must not
            # reallocate every time); the pointer arrays of the slot live in the slot, not in the engine-lifetime `_keep` list, so a
            # regrow frees the old ones, and the slot holds the Tune object its descriptor copy points at
            Bn, Sn = (B, S64) if slot is None else (max(slot["B"], B), max(slot["S"], S64))
            self._fwd_slots = slot = None
            kv = [torch.zeros(Bn, Hk, Sn, hd, dtype=self.dtype, device=self.device) for _ in range(2)]
            vt = torch.zeros(Bn, Hk, hd, Sn, dtype=self.dtype, device=self.device)
            d = L.LlamaDesc.from_buffer_copy(self.llama_desc)
            d.max_seq = Sn
            arrs = [L.ptr_array([t.data_ptr()] * Lr) for t in (kv[0][0], kv[1][0], vt[0])]
            d.k_cache, d.v_cache, d.vt_cache = arrs[0][1], arrs[1][1], arrs[2][1]
            slot = self._fwd_slots = {"B": Bn, "S": Sn, "k": kv[0], "v": kv[1], "vt": vt, "desc": d, "arrs": arrs, "tune": self.tune}

            def _sync(src, eng=self):                 # ONE hook for whatever descriptor copy is current (a regrow replaces the copy, not the hook)
                cur = getattr(eng, "_fwd_slots", None)
                if cur is not None:
                    cur["desc"].prefill_fp8, cur["desc"].rope_in_attn = src.prefill_fp8, src.rope_in_attn
            if not getattr(self, "_fwd_hooked", False):
                self._option_hooks.append(_sync)
                self._fwd_hooked = True
        d = slot["desc"]
        total = sum(lens)
        with self.phase() as st:
            rows = torch.cat([e.to(device=self.device, dtype=self.dtype) for e in embeds_list], dim=0).contiguous()
            logits = torch.empty(total, c.vocab_size, dtype=torch.float32, device=self.device)
            self._flush_handoff_checks("prefill")
            ws = self._workspace("prefill", self.lib.teo_llama_prefill_workspace_bytes(C.byref(d), total))
            arr = (C.c_int * B)(*lens)
            hs = torch.empty(c.num_hidden_layers + 1, total, c.hidden_size, dtype=self.dtype, device=self.device) if hidden_states else None
            L.check(self.lib.teo_llama_prefill_batch(C.byref(d), _p(rows), arr, B, slot["k"].stride(0), 0, _p(logits), _p(ws), ws.numel(), st,
                                                     _p(hs) if hs is not None else None), "teo_llama_prefill_batch")
            sid = C.c_void_p(self.stream.cuda_stream)
            self._check_handoffs("prefill", lambda f: self.lib.teo_llama_prefill_workspace_status(C.byref(d), total, _p(ws), ws.numel(), C.byref(f), sid),
                                 "teo_llama_prefill_batch")
        return (logits, hs) if hidden_states else logits

    def sample(self, logits, temperature, top_k, seed, draw, top_p=1.0):
        """One draw of the device sampler (temperature -> top-k -> softmax -> multinomial) from fp32 logits [V]."""
        tok = torch.empty(1, dtype=torch.int64, device=self.device)
        with self.phase() as st:
            lg = logits.to(device=self.device, dtype=torch.float32).contiguous()
            L.check(self.lib.teo_sample_topk(_p(lg), _p(tok), lg.numel(), float(temperature), int(top_k or 0), float(top_p or 1.0),
                                             int(seed) & SEED_MASK, int(draw), st), "teo_sample_topk")
        return int(tok.item())

    def decode_begin(self, first_token, stop_ids=None, do_sample=False, temperature=1.0, top_k=0, seed=0, draws_done=0, top_p=1.0):
        """Arm the device-side decode loop: first_token is the input of the next step, at position cache_len.
        With do_sample the steps draw from the device sampler (counter-based RNG: draw index = draws so far)."""
        with self.phase():
            self.d_token.fill_(int(first_token))
            self.d_pos.fill_(self.cache_len)
            self.d_count.zero_()
            self.d_stop.zero_()
            self.d_rng[0] = int(seed) & SEED_MASK            # the same 63-bit seed teo_sample_topk gets for draw 0
            self.steps_since_begin = 0
            self.d_rng[1] = int(draws_done)
            n = 0
            if stop_ids:
                n = min(len(stop_ids), 16)
                self.d_stop_ids[:n] = torch.tensor(list(stop_ids)[-n:], dtype=torch.int64, device=self.device)
            s = self.decode_state
            key = (n, int(bool(do_sample)), int(top_k or 0), C.c_float(float(temperature)).value, C.c_float(float(top_p or 1.0)).value)
            if key != (s.n_stop_ids, s.do_sample, s.top_k, float(s.temperature), float(s.top_p)):
                s.n_stop_ids, s.do_sample, s.top_k, s.temperature, s.top_p = key     # baked into the captured launch: re-capture
                self._drop_graph()
        ws = self._workspace("decode", self.lib.teo_llama_decode_workspace_bytes(C.byref(self.llama_desc)))
        with self.phase() as st:
            L.check(self.lib.teo_llama_decode_begin(C.byref(self.llama_desc), C.byref(self.decode_state), _p(ws), ws.numel(),
                                                    st), "teo_llama_decode_begin")

    def set_options(self, prefill_fp8=None, rope_in_attn=None):
        """Per-engine options of the LLaMA descriptor (include/teo_hip.h teo_llama_desc): `prefill_fp8` = w8a8 prefill on the fp8
        MFMA (lossy, needs weight_format='fp8'), `rope_in_attn` = RoPE + KV append inside the decode attention kernel instead of
        the QKV GEMV epilogue (same values).  Not process-global: two engines in one process can differ."""
        d = self.llama_desc
        if prefill_fp8 is not None:
            if prefill_fp8 and self.llama_w8 is None:
                raise ValueError("prefill_fp8 needs weight_format='fp8' (the e4m3 weight copies)")
            d.prefill_fp8 = 1 if prefill_fp8 else 0
        if rope_in_attn is not None and int(bool(rope_in_attn)) != d.rope_in_attn:
            d.rope_in_attn = 1 if rope_in_attn else 0
            self._drop_graph()                    # the placement is baked into the captured decode step
        for hook in getattr(self, "_option_hooks", []):
            hook(d)

    def tune_set(self, key, value):
        """A performance knob of THIS engine (teo_tune_set on the engine's own block; keys in include/teo_hip.h).  Captured decode
        graphs keep the kernel choices of their capture, so they are dropped."""
        self.tune.set(key, value)                 # raises TeoError on an unknown key / rejected value
        self._drop_graph()
        for hook in self._tune_hooks:
            hook()

    def tune_reset(self):
        self.tune.reset()
        self._drop_graph()
        for hook in self._tune_hooks:
            hook()

    def _drop_graph(self):
        if self._graph is not None:
            self.lib.teo_graph_destroy(self._graph)
            self._graph = None

    def decode_steps(self, n, use_graph=True):
        """Run n greedy steps on the device (hipGraph replay by default).  The caller bounds n by max_seq."""
        if self.cache_len + n > self.max_seq:
            raise ValueError(f"decode would exceed max_seq {self.max_seq}")
        if getattr(self, "steps_since_begin", 0) + n > self.max_new_cap:
            raise ValueError(f"decode_steps: {self.steps_since_begin + n} tokens since decode_begin exceed the output buffer "
                             f"({self.max_new_cap}); call decode_begin again or build the engine with a larger max_new")
        ws = self._workspace("decode", self.lib.teo_llama_decode_workspace_bytes(C.byref(self.llama_desc)))
        with self.phase() as st:
            if use_graph:
                if self._graph is None or self._graph_ws != ws.data_ptr():
                    self._drop_graph()
                    g = C.c_void_p()
                    L.check(self.lib.teo_llama_decode_graph_create(C.byref(self.llama_desc), C.byref(self.decode_state),
                                                                   _p(ws), ws.numel(), st, C.byref(g)),
                            "teo_llama_decode_graph_create")
                    self._graph, self._graph_ws = g, ws.data_ptr()
                L.check(self.lib.teo_graph_launch(self._graph, n, st), "teo_graph_launch")
                # drain before anything else is queued behind the replays (an event record, a D2H copy of the token buffer):
                # replays still outstanding when such a command is enqueued run 8 % slower (see _Phase.__exit__)
                self.stream.synchronize()
            else:
                for _ in range(n):
                    L.check(self.lib.teo_llama_decode_step(C.byref(self.llama_desc), C.byref(self.decode_state), _p(ws),
                                                           ws.numel(), st), "teo_llama_decode_step")
        self.cache_len += n
        self.steps_since_begin = getattr(self, "steps_since_begin", 0) + n

    PROF_CLASSES = ("qkv_rope_gemv", "attn_decode_partial", "attn_decode_combine", "o_gemv", "gateup_gemv", "down_gemv", "lm_head_gemv",
                    "decode_tail")

    def decode_steps_profiled(self, n):
        """n decode steps as plain launches, every kernel timed by its own dispatch timestamps (teo_llama_decode_step_profile).
        Returns {class: (launches per step, mean microseconds per launch)}; advances the cache like decode_steps."""
        if self.cache_len + n > self.max_seq:
            raise ValueError(f"decode would exceed max_seq {self.max_seq}")
        if getattr(self, "steps_since_begin", 0) + n > self.max_new_cap:
            raise ValueError(f"decode_steps_profiled: {self.steps_since_begin + n} tokens since decode_begin exceed the output buffer "
                             f"({self.max_new_cap})")
        ws = self._workspace("decode", self.lib.teo_llama_decode_workspace_bytes(C.byref(self.llama_desc)))
        K = len(self.PROF_CLASSES)
        tot, cnt = [0.0] * K, [0] * K
        ms, ct = (C.c_float * K)(), (C.c_int * K)()
        with self.phase() as st:
            for _ in range(n):
                L.check(self.lib.teo_llama_decode_step_profile(C.byref(self.llama_desc), C.byref(self.decode_state), _p(ws), ws.numel(),
                                                               ms, ct, st), "teo_llama_decode_step_profile")
                for k in range(K):
                    tot[k] += ms[k]
                    cnt[k] += ct[k]
        self.cache_len += n
        self.steps_since_begin = getattr(self, "steps_since_begin", 0) + n
        return {name: (cnt[k] // n, tot[k] / cnt[k] * 1e3) for k, name in enumerate(self.PROF_CLASSES) if cnt[k]}

    def generated(self):
        n = int(self.d_count.item())
        return self.d_out[:n].clone()

    def __del__(self):
        try:
            self._drop_graph()
        except Exception:  # noqa: BLE001
            pass
