"""ctypes binding of libteo_hip.so (C ABI declared in include/teo_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  If it cannot be loaded, importing
callers get a TeoLibraryError telling them to run `python -c "import __graft_entry__ as g; g.build()"`.
"""
import ctypes as C
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libteo_hip.so")

TEO_F32, TEO_BF16, TEO_F16 = 0, 1, 2
ACT_NONE, ACT_GELU_ERF, ACT_QUICK_GELU = 0, 1, 2
GEMM_SWIGLU16, GEMM_FORCE_SIMPLE, GEMM_WTILED, GEMM_SWIGLU8, GEMM_F16 = 1, 2, 4, 8, 16
ATTN_FORCE_SIMPLE = 1
INT32_MIN = -(2 ** 31)


class TeoLibraryError(RuntimeError):
    pass


class TeoError(RuntimeError):
    pass


class AttnArgs(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("vt", C.c_void_p), ("o", C.c_void_p),
                ("q_bs", C.c_longlong), ("q_hs", C.c_longlong), ("q_rs", C.c_longlong),
                ("k_bs", C.c_longlong), ("k_hs", C.c_longlong), ("k_rs", C.c_longlong),
                ("v_bs", C.c_longlong), ("v_hs", C.c_longlong), ("v_rs", C.c_longlong),
                ("vt_bs", C.c_longlong), ("vt_hs", C.c_longlong), ("vt_rs", C.c_longlong),
                ("o_bs", C.c_longlong), ("o_rs", C.c_longlong),
                ("batch", C.c_int), ("heads", C.c_int), ("kv_heads", C.c_int), ("head_dim", C.c_int),
                ("q_len", C.c_int), ("kv_len", C.c_int), ("causal", C.c_int), ("scale", C.c_float),
                ("flags", C.c_uint)]


PP = C.POINTER(C.c_void_p)


class VitDesc(C.Structure):
    _fields_ = [("hidden", C.c_int), ("heads", C.c_int), ("inter", C.c_int), ("layers_run", C.c_int),
                ("image", C.c_int), ("patch", C.c_int), ("channels", C.c_int), ("act", C.c_int),
                ("eps", C.c_float), ("dtype", C.c_int), ("k_pad", C.c_int),
                ("patch_w", C.c_void_p), ("cls", C.c_void_p), ("pos", C.c_void_p),
                ("pre_ln_w", C.c_void_p), ("pre_ln_b", C.c_void_p),
                ("ln1_w", PP), ("ln1_b", PP), ("qkv_w", PP), ("qkv_b", PP), ("out_w", PP), ("out_b", PP),
                ("ln2_w", PP), ("ln2_b", PP), ("fc1_w", PP), ("fc1_b", PP), ("fc2_w", PP), ("fc2_b", PP),
                ("keep_cls", C.c_int), ("tune", C.c_void_p)]


class ProjDesc(C.Structure):
    _fields_ = [("in_dim", C.c_int), ("out_dim", C.c_int), ("depth", C.c_int), ("dtype", C.c_int),
                ("w", C.c_void_p * 4), ("b", C.c_void_p * 4), ("tune", C.c_void_p)]


class LlamaDesc(C.Structure):
    _fields_ = [("hidden", C.c_int), ("heads", C.c_int), ("kv_heads", C.c_int), ("head_dim", C.c_int),
                ("inter", C.c_int), ("layers", C.c_int), ("vocab", C.c_int), ("eps", C.c_float),
                ("dtype", C.c_int), ("max_seq", C.c_int),
                ("embed", C.c_void_p), ("final_norm_w", C.c_void_p), ("lm_head", C.c_void_p),
                ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p), ("max_pos", C.c_int),
                ("in_norm_w", PP), ("qkv_w", PP), ("o_w", PP), ("post_norm_w", PP), ("gateup_w", PP),
                ("down_w", PP), ("k_cache", PP), ("v_cache", PP), ("vt_cache", PP),
                ("qkv_w8", PP), ("qkv_s", PP), ("o_w8", PP), ("o_s", PP), ("gateup_w8", PP), ("gateup_s", PP),
                ("down_w8", PP), ("down_s", PP), ("lm_head8", C.c_void_p), ("lm_head_s", C.c_void_p),
                ("prefill_fp8", C.c_int), ("rope_in_attn", C.c_int), ("tune", C.c_void_p)]


class DecodeState(C.Structure):
    _fields_ = [("d_token", C.c_void_p), ("d_pos", C.c_void_p), ("d_out_tokens", C.c_void_p),
                ("d_out_count", C.c_void_p), ("d_stop", C.c_void_p), ("d_stop_ids", C.c_void_p),
                ("n_stop_ids", C.c_int), ("d_logits", C.c_void_p),
                ("do_sample", C.c_int), ("top_k", C.c_int), ("temperature", C.c_float), ("d_rng", C.c_void_p),
                ("top_p", C.c_float)]


MAX_DECODE_BATCH = 16
COMM_ID_BYTES = 128


class DecodeBatchState(C.Structure):
    _fields_ = [("batch", C.c_int), ("out_stride", C.c_int), ("cache_stride", C.c_longlong), ("w_tiled", C.c_int), ("gateup_block8", C.c_int),
                ("d_token", C.c_void_p), ("d_pos", C.c_void_p), ("d_out_tokens", C.c_void_p),
                ("d_out_count", C.c_void_p), ("d_stop", C.c_void_p), ("d_stop_ids", C.c_void_p),
                ("n_stop_ids", C.c_int), ("d_logits", C.c_void_p),
                ("do_sample", C.c_int), ("top_k", C.c_int), ("temperature", C.c_float), ("d_rng", C.c_void_p),
                ("top_p", C.c_float)]


_SIGS = {
    "teo_version": (C.c_int, []),
    "teo_last_error": (C.c_char_p, []),
    "teo_last_kernel": (C.c_char_p, []),
    "teo_sizeof": (C.c_size_t, [C.c_char_p]),
    "teo_tune_create": (C.c_void_p, []),
    "teo_tune_destroy": (C.c_int, [C.c_void_p]),
    "teo_tune_set": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "teo_tune_get": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    "teo_tune_reset": (C.c_int, [C.c_void_p]),
    "teo_tune_bind": (C.c_int, [C.c_void_p]),
    "teo_tune_keys": (C.c_char_p, []),
    "teo_gemm_uses_mfma": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint]),
    "teo_layernorm": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "teo_rmsnorm": (C.c_int, [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "teo_gemm": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 6 + [C.c_uint, C.c_int, C.c_int, C.c_void_p]),
    "teo_gemm_fp8": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_uint, C.c_int, C.c_void_p]),
    "teo_gemm_fp8_ws": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_uint, C.c_int, C.c_void_p, C.c_void_p]),
    "teo_quant_rows_fp8": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "teo_gemm_workspace_bytes": (C.c_size_t, []),
    "teo_gemm_workspace_init": (C.c_int, [C.c_void_p, C.c_void_p]),
    "teo_gemm_workspace_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_void_p]),
    "teo_vit_workspace_status": (C.c_int, [C.POINTER(VitDesc), C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.c_void_p]),
    "teo_llama_prefill_workspace_status": (C.c_int, [C.POINTER(LlamaDesc), C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.c_void_p]),
    "teo_gemm_ws": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 6 + [C.c_uint, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "teo_im2col_patches": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]),
    "teo_patch_embed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p]),
    "teo_vit_embed_ln": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "teo_attention": (C.c_int, [C.POINTER(AttnArgs), C.c_int, C.c_void_p]),
    "teo_vit_value_transpose": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]),
    "teo_rope_kv_append": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_int] * 7 + [C.c_void_p]),
    "teo_embed_splice": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "teo_drop_cls": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]),
    "teo_argmax": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "teo_sample_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_float, C.c_ulonglong, C.c_ulonglong,
                                  C.c_void_p]),
    "teo_gemv": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_float, C.c_uint, C.c_int, C.c_int, C.c_void_p]),
    "teo_gemv_w8": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_float, C.c_uint, C.c_int, C.c_void_p]),
    "teo_vit_workspace_bytes": (C.c_size_t, [C.POINTER(VitDesc), C.c_int]),
    "teo_vit_encode": (C.c_int, [C.POINTER(VitDesc), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "teo_projector_workspace_bytes": (C.c_size_t, [C.POINTER(ProjDesc), C.c_int]),
    "teo_projector": (C.c_int, [C.POINTER(ProjDesc), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "teo_llama_prefill_workspace_bytes": (C.c_size_t, [C.POINTER(LlamaDesc), C.c_int]),
    "teo_llama_prefill": (C.c_int, [C.POINTER(LlamaDesc), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "teo_llama_prefill_attentions": (C.c_int, [C.POINTER(LlamaDesc), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "teo_llama_decode_workspace_bytes": (C.c_size_t, [C.POINTER(LlamaDesc)]),
    "teo_llama_decode_begin": (C.c_int, [C.POINTER(LlamaDesc), C.POINTER(DecodeState), C.c_void_p, C.c_size_t, C.c_void_p]),
    "teo_llama_decode_step": (C.c_int, [C.POINTER(LlamaDesc), C.POINTER(DecodeState), C.c_void_p, C.c_size_t, C.c_void_p]),
    "teo_llama_decode_step_profile": (C.c_int, [C.POINTER(LlamaDesc), C.POINTER(DecodeState), C.c_void_p, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_int),
                                                C.c_void_p]),
    "teo_llama_decode_graph_create": (C.c_int, [C.POINTER(LlamaDesc), C.POINTER(DecodeState), C.c_void_p, C.c_size_t,
                                                C.c_void_p, C.POINTER(C.c_void_p)]),
    "teo_llama_prefill_batch": (C.c_int, [C.POINTER(LlamaDesc), C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_longlong, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "teo_llama_decode_batch_workspace_bytes": (C.c_size_t, [C.POINTER(LlamaDesc), C.c_int]),
    "teo_llama_decode_batch_begin": (C.c_int, [C.POINTER(LlamaDesc), C.POINTER(DecodeBatchState), C.c_void_p, C.c_size_t, C.c_void_p]),
    "teo_llama_decode_batch_step": (C.c_int, [C.POINTER(LlamaDesc), C.POINTER(DecodeBatchState), C.c_void_p, C.c_size_t, C.c_void_p]),
    "teo_llama_decode_batch_step_profile": (C.c_int, [C.POINTER(LlamaDesc), C.POINTER(DecodeBatchState), C.c_void_p, C.c_size_t,
                                                      C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_void_p]),
    "teo_llama_decode_batch_graph_create": (C.c_int, [C.POINTER(LlamaDesc), C.POINTER(DecodeBatchState), C.c_void_p, C.c_size_t,
                                                      C.c_void_p, C.POINTER(C.c_void_p)]),
    "teo_graph_launch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "teo_graph_destroy": (C.c_int, [C.c_void_p]),
    "teo_attn_decode_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "teo_attn_decode": (C.c_int, [C.c_void_p] * 9 + [C.c_int] * 4 + [C.c_float, C.c_int, C.c_int, C.c_longlong, C.c_longlong,
                                                                      C.c_longlong, C.c_void_p]),
    "teo_cross_entropy": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_longlong,
                                    C.c_void_p]),
    "teo_preprocess_frames": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                        C.POINTER(C.c_float), C.c_int, C.c_void_p]),
    "teo_preprocess_frames_pad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                            C.POINTER(C.c_float), C.POINTER(C.c_ubyte), C.c_int, C.c_void_p]),
    "teo_gemm_skinny": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
                        + [C.c_int] * 5 + [C.c_uint, C.c_int, C.c_void_p]),
    "teo_comm_unique_id": (C.c_int, [C.c_void_p]),
    "teo_ctx_create": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "teo_ctx_destroy": (C.c_int, [C.c_void_p]),
    "teo_ctx_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "teo_allgather_visual": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "teo_ctx_tune": (C.c_void_p, [C.c_void_p]),
}

ABI_VERSION = 2            # TEO_ABI_VERSION of include/teo_hip.h this binding was written against
# (the ctypes mirrors of the header's structs are checked against the library's own sizeof at load: a stale or newer .so must fail
# THERE, not by reading shifted fields)


EXPORTS = tuple(_SIGS.keys())
_lib = None


def load():
    """Load libteo_hip.so and attach signatures.  Raises TeoLibraryError if it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (libamdhip64.so); it must be loaded first so that this library binds to the
    # SAME runtime instance (streams and device pointers are shared with torch).  Loading /opt/rocm's copy first
    # leaves two runtimes in the process and every launch fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise TeoLibraryError(
            f"{LIB_PATH} not found: the HIP library is required (no CPU fallback). Build it with "
            "`python -c \"import __graft_entry__ as g; g.build()\"` or `make -C teochat_amd/csrc`.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise TeoLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise TeoLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    if lib.teo_version() != ABI_VERSION:
        raise TeoLibraryError(f"ABI version mismatch: library {lib.teo_version()}, binding {ABI_VERSION}; rebuild {LIB_PATH}")
    mirrors = {"teo_vit_desc": VitDesc, "teo_proj_desc": ProjDesc, "teo_llama_desc": LlamaDesc, "teo_decode_state": DecodeState,
               "teo_decode_batch_state": DecodeBatchState, "teo_attn_args": AttnArgs}
    for name, cls in mirrors.items():
        want = lib.teo_sizeof(name.encode())
        if want != C.sizeof(cls):
            raise TeoLibraryError(f"struct layout mismatch: {name} is {want} bytes in {LIB_PATH}, {C.sizeof(cls)} in this binding; rebuild")
    _lib = lib
    return lib


# ---- performance knobs (teo_tune blocks; include/teo_hip.h) -----------------------------------------------------------------------
class Tune:
    """One block of performance knobs (teo_tune).  An engine owns one and points its descriptors at it; code that calls the
    primitive operators binds one to its thread (`bind()`, or the `tuned()` context manager below)."""

    def __init__(self):
        self._lib = load()
        self.ptr = self._lib.teo_tune_create()
        if not self.ptr:
            raise TeoError("teo_tune_create failed")

    def try_set(self, key, value):
        """The raw status of teo_tune_set (0, or TEO_ERR_ARG for an unknown key / a value outside the knob's set): for callers that
        test the rejection itself."""
        key = key if isinstance(key, bytes) else key.encode()
        return self._lib.teo_tune_set(self.ptr, key, int(value))

    def set(self, key, value):
        """Set a knob; raises TeoError when the library rejects the key or the value -- a sweep must never report the default
        dispatch under the label of a knob that was silently dropped.  Returns 0."""
        rc = self.try_set(key, value)
        check(rc, f"teo_tune_set({key!r}, {value})")
        return rc

    def get(self, key):
        key = key if isinstance(key, bytes) else key.encode()
        v = C.c_int(0)
        check(self._lib.teo_tune_get(self.ptr, key, C.byref(v)), "teo_tune_get")
        return v.value

    def reset(self):
        rc = self._lib.teo_tune_reset(self.ptr)
        check(rc, "teo_tune_reset")
        return rc

    def bind(self):
        return self._lib.teo_tune_bind(self.ptr)

    def close(self):
        if self.ptr:
            self._lib.teo_tune_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_tls = threading.local()


def thread_tune():
    """The calling thread's own block, created and bound on first use (what `tune_set` / `tune_reset` act on)."""
    t = getattr(_tls, "tune", None)
    if t is None:
        t = _tls.tune = Tune()
        t.bind()
    return t


def tune_set(key, value):
    """Set a knob in the calling thread's bound block (primitive operators called from this thread see it; engines have their own).
    Raises TeoError on an unknown key or a rejected value (`Tune.try_set` is the non-raising form)."""
    return thread_tune().set(key, value)


def tune_reset():
    t = getattr(_tls, "tune", None)
    return t.reset() if t is not None else 0


def tune_release():
    """Unbind and destroy the calling thread's block (tests call this after each test: nothing outlives the test that set it)."""
    t = getattr(_tls, "tune", None)
    if t is not None:
        load().teo_tune_bind(None)
        t.close()
        _tls.tune = None


def tune_keys():
    return load().teo_tune_keys().decode().split()


def check(rc, what=""):
    if rc != 0:
        msg = load().teo_last_error()
        raise TeoError(f"{what} failed with status {rc}: {msg.decode() if msg else ''}")


def ptr_array(ptrs):
    """Host array of device pointers (kept alive by the caller)."""
    arr = (C.c_void_p * len(ptrs))(*ptrs)
    return arr, C.cast(arr, PP)
