"""CPU oracle for the TEOChat temporal-image -> LLM forward path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product path
(teochat_amd/) never does and fails loudly when its HIP library is missing.

What it is: a plain torch-CPU restatement (fp32 by default, fp64 on request) of the
reference's algorithm for the path SURVEY.md section 8(a) lists (rows H2..H16).  The
reference delegates all dense arithmetic to un-vendored `transformers==4.31.0`
(pyproject.toml:17); this file restates that published algorithm and anchors it on
the reference's own call sites.  Each function cites the reference file:line it follows
(paths relative to /root/reference; "tf" = the transformers CLIP/LLaMA modelling files).

Pinning: the reference owns no tests or golden vectors for this path.
The oracle is pinned instead against outputs of the reference itself, imported and run
in the build container by tests/golden/make_golden.py (fixtures in tests/golden/*.npz,
checked by tests/test_oracle_golden.py).  The arithmetic executed while generating
those fixtures is transformers 5.15.0 (the only version installable here), not 4.31.0:
same math, different op order.

`rounding`: None  -> no intermediate rounding (fp32/fp64 reference semantics)
            "bf16" -> round activations to bf16 at the kernel boundaries listed in
                     DESIGN.md section "Precision contract" (what the HIP bf16 path stores).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch
import torch.nn.functional as F

IGNORE_INDEX = -100          # videollava/constants.py:7
IMAGE_TOKEN_INDEX = -200     # videollava/constants.py:9
DEFAULT_IMAGE_TOKEN = "<image>"   # videollava/constants.py:10
DEFAULT_VIDEO_TOKEN = "<video>"   # videollava/constants.py:17


# --------------------------------------------------------------------------------------
# configs
# --------------------------------------------------------------------------------------
@dataclass
class VitCfg:
    hidden_size: int = 1024
    num_attention_heads: int = 16
    intermediate_size: int = 4096
    num_hidden_layers: int = 24
    patch_size: int = 14
    image_size: int = 224
    num_channels: int = 3
    hidden_act: str = "quick_gelu"      # configuration_image.py:191 default; "gelu" also supported
    layer_norm_eps: float = 1e-5

    @property
    def num_positions(self):
        return (self.image_size // self.patch_size) ** 2 + 1


@dataclass
class LlamaCfg:
    hidden_size: int = 4096
    num_attention_heads: int = 32
    num_key_value_heads: int = 32
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    vocab_size: int = 32000
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads


@dataclass
class MMCfg:
    mm_hidden_size: int = 1024
    mm_projector_type: str = "mlp2x_gelu"
    mm_vision_select_layer: int = -2
    mm_vision_select_feature: str = "patch"
    tokenizer_model_max_length: Optional[int] = None
    tokenizer_padding_side: str = "right"


VIT_PREFIX = "model.image_tower.image_tower."


def _rounder(rounding):
    if rounding is None:
        return lambda t: t
    if rounding == "bf16":
        return lambda t: t.to(torch.bfloat16).to(t.dtype)
    if rounding == "fp16":
        return lambda t: t.to(torch.float16).to(t.dtype)
    raise ValueError(f"unknown rounding {rounding!r}")


# fp32 summation order of the Linear layers (test infrastructure for the bf16 noise-floor control, tests/test_noise_floor.py):
# None = one matmul over the whole K; (n, descending) = K cut into n equal chunks whose partial products are added in
# ascending or descending chunk order.  Nothing else changes -- same weights, same rounding points.
K_ORDER = None


# Op ORDER of the attention / RoPE arithmetic (round 5, information level only; VERDICT r04 "What's missing" #4).  None = the order of the
# stack the fixtures were made with (transformers 5.15: scores scaled after q k^T, fp32 softmax statistics, fp32 RoPE angles) and of the
# kernels.  "tf431" = the reference's PINNED stack (pyproject.toml:17, transformers 4.31.0) as its 16-bit run executes it:
#   * CLIPAttention (constructed at languagebind/image/modeling_image.py:69; 4.31 modeling_clip.py CLIPAttention.forward): q is scaled
#     BEFORE q k^T (`self.q_proj(hidden_states) * self.scale`, a rounding of its own), q k^T is a 16-bit bmm output, the softmax runs in
#     the working dtype (normalised probabilities rounded to 16 bit), P V is a 16-bit bmm output;
#   * LlamaAttention (4.31 modeling_llama.py): q k^T is a 16-bit matmul output, divided by sqrt(d) (rounded again), softmax in fp32 but
#     cast back to 16 bit NORMALISED, P V rounded; LlamaRotaryEmbedding keeps cos / sin caches that are cast to the model dtype, and
#     apply_rotary_pos_emb is three 16-bit ops (q cos, rotate_half(q) sin, their sum).
# Only meaningful together with rounding="bf16" / "fp16"; in fp32 both orders agree to round-off (asserted by the test that uses it).
OP_ORDER = None


def _lin(x, w):
    """x [..., K] . w[N, K]^T -- every nn.Linear of the path (bias added by the caller)."""
    wt = w.to(x.dtype).t()
    if K_ORDER is None:
        return x @ wt
    n, descending = K_ORDER
    K = x.shape[-1]
    step = -(-K // n)
    starts = list(range(0, K, step))
    if descending:
        starts.reverse()
    acc = None
    for k0 in starts:
        part = x[..., k0:k0 + step] @ wt[k0:k0 + step]
        acc = part if acc is None else acc + part
    return acc


# --------------------------------------------------------------------------------------
# H2/H3: prompt construction
# --------------------------------------------------------------------------------------
V1_SYSTEM = ("A chat between a curious user and an artificial intelligence assistant. "
             "The assistant gives helpful, detailed, and polite answers to the user's questions.")


def conv_v1_prompt(user_msg: str) -> str:
    """conversation.py:51-60 (SeparatorStyle.TWO) with conv_vicuna_v1 (conversation.py:252-262):
    system + sep, then "USER: msg" + sep, then "ASSISTANT:" for the empty assistant turn."""
    sep, roles = " ", ("USER", "ASSISTANT")
    out = V1_SYSTEM + sep
    out += roles[0] + ": " + user_msg + sep      # i = 0 -> seps[0]
    out += roles[1] + ":"                        # message None
    return out


def replace_video_token(prompt: str, n_images: int, prompt_strategy) -> str:
    """eval/inference.py:11-20."""
    if prompt_strategy is None:
        rep = DEFAULT_IMAGE_TOKEN * n_images
    elif prompt_strategy == "interleave":
        rep = "".join("Image %d: %s" % (i + 1, DEFAULT_IMAGE_TOKEN) for i in range(n_images))
    else:
        raise ValueError(f"Unknown prompt strategy: {prompt_strategy}")
    return prompt.replace(DEFAULT_VIDEO_TOKEN, rep)


def build_prompt(inp: str, n_images: int, prompt_strategy="interleave", chronological_prefix=True) -> str:
    """eval/inference.py:37-43,55."""
    p = conv_v1_prompt(inp)
    if chronological_prefix:
        p = p.replace("times:", "times in chronological order:")
    return replace_video_token(p, n_images, prompt_strategy)


# --------------------------------------------------------------------------------------
# H4: image-token packing of the tokenized prompt (bit-exact integer work)
# --------------------------------------------------------------------------------------
def tokenizer_image_token(prompt: str, tokenizer, image_token_index: int = IMAGE_TOKEN_INDEX) -> List[int]:
    """mm_utils.py:43-62.  Chunks between '<image>' are tokenized independently; a leading
    BOS of the first chunk is kept once, every later chunk has its BOS dropped, and a single
    sentinel is placed between chunks."""
    chunks = [list(tokenizer(c).input_ids) for c in prompt.split("<image>")]
    ids: List[int] = []
    has_bos = bool(chunks) and len(chunks[0]) > 0 and chunks[0][0] == tokenizer.bos_token_id
    skip = 1 if has_bos else 0
    if has_bos:
        ids.append(chunks[0][0])
    for ci, chunk in enumerate(chunks):
        if ci > 0:
            # the reference's separator is [sentinel]*(skip+1) sliced by [skip:] -> exactly one sentinel
            ids.append(image_token_index)
        ids.extend(chunk[skip:])
    return ids


# --------------------------------------------------------------------------------------
# H5: stopping criterion
# --------------------------------------------------------------------------------------
class KeywordsStop:
    """mm_utils.py:73-104."""

    def __init__(self, keywords: Sequence[str], tokenizer, n_prompt: int):
        self.keywords = list(keywords)
        self.keyword_ids = []
        self.max_keyword_len = 0
        for kw in keywords:
            kid = list(tokenizer(kw).input_ids)
            if len(kid) > 1 and kid[0] == tokenizer.bos_token_id:
                kid = kid[1:]
            self.max_keyword_len = max(self.max_keyword_len, len(kid))
            self.keyword_ids.append(kid)
        self.tokenizer = tokenizer
        self.start_len = n_prompt

    def one(self, row: List[int]) -> bool:
        offset = min(len(row) - self.start_len, self.max_keyword_len)
        for kid in self.keyword_ids:
            if row[-len(kid):] == kid:
                return True
        tail = row[-offset:]          # note: offset == 0 -> whole row, as in the reference slice [-0:]
        text = self.tokenizer.batch_decode([tail], skip_special_tokens=True)[0]
        return any(kw in text for kw in self.keywords)

    def __call__(self, rows: List[List[int]]) -> bool:
        return all(self.one(list(r)) for r in rows)


# --------------------------------------------------------------------------------------
# H6: preprocessing (ToTensor -> Resize(224,bicubic) -> CenterCrop(224) -> Normalize)
# --------------------------------------------------------------------------------------
OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)   # processing_image.py:7
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)   # processing_image.py:8


def preprocess_uint8(img_hwc_u8: torch.Tensor) -> torch.Tensor:
    """processing_image.py:15-25 for an image that is already 224x224 (Resize and CenterCrop are
    identities then): ToTensor (/255, HWC->CHW) and Normalize."""
    x = img_hwc_u8.to(torch.float32).div(255.0).permute(2, 0, 1)
    mean = torch.tensor(OPENAI_DATASET_MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(OPENAI_DATASET_STD, dtype=torch.float32).view(3, 1, 1)
    return (x - mean) / std



def resize_geometry(h: int, w: int, size: int = 224):
    """torchvision 0.17.1 (pyproject.toml:16) `Resize(size)` with an int size: the shorter edge becomes `size`, the other
    int(size * long / short); then `CenterCrop(size)`: offsets int(round((n - size) / 2.0)) (Python round = half-to-even).
    Returns (new_h, new_w, crop_top, crop_left)."""
    if h <= w:
        nh, nw = size, int(size * w / h)
    else:
        nh, nw = int(size * h / w), size
    return nh, nw, int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))


def preprocess_image(img_hwc_u8: torch.Tensor, size: int = 224) -> torch.Tensor:
    """processing_image.py:15-25 get_image_transform for an arbitrary uint8 HWC RGB image:
        ToTensor()                                      uint8 HWC -> float32 CHW / 255
        Resize(224, interpolation=BICUBIC)              on a TENSOR: torchvision 0.17.1 calls
                                                        F.interpolate(mode="bicubic", align_corners=False, antialias=True)
                                                        (antialias defaults to True since 0.17; float input is not clamped)
        CenterCrop(224)
        Normalize(OPENAI_DATASET_MEAN, OPENAI_DATASET_STD)
    torchvision is not installed here; the ATen op it dispatches to is (same torch call).  Pinned by
    tests/test_oracle_golden.py: identity / analytic cases and a PIL-bicubic cross-check."""
    x = img_hwc_u8.to(torch.float32).div(255.0).permute(2, 0, 1)
    _, h, w = x.shape
    nh, nw, top, left = resize_geometry(h, w, size)
    if (nh, nw) != (h, w):
        x = F.interpolate(x[None], size=(nh, nw), mode="bicubic", align_corners=False, antialias=True)[0]
    x = x[:, top:top + size, left:left + size]
    mean = torch.tensor(OPENAI_DATASET_MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(OPENAI_DATASET_STD, dtype=torch.float32).view(3, 1, 1)
    return (x - mean) / std


def expand2square_u8(img_hwc_u8: torch.Tensor, background_rgb) -> torch.Tensor:
    """mm_utils.py:14-25 expand2square on an array: square canvas of the longer side filled with `background_rgb`, the image
    pasted centred (offset (side - n) // 2 along the shorter axis)."""
    h, w, _ = img_hwc_u8.shape
    if h == w:
        return img_hwc_u8
    side = max(h, w)
    canvas = torch.tensor(list(background_rgb), dtype=torch.uint8).view(1, 1, 3).expand(side, side, 3).clone()
    oy, ox = (side - h) // 2, (side - w) // 2
    canvas[oy:oy + h, ox:ox + w] = img_hwc_u8
    return canvas


def pad_fill_from_mean(image_mean):
    """mm_utils.py:34: tuple(int(x * 255) for x in image_processor.image_mean)."""
    return tuple(int(x * 255) for x in image_mean)


# --------------------------------------------------------------------------------------
# attention core shared by the ViT and LLaMA restatements
# --------------------------------------------------------------------------------------
LOG2E = 1.44269504088896340736


def attention_core(q, k, v, visible, scale, R, mode="exact"):
    """softmax(scale * q k^T restricted to `visible`) v   with q [B,H,Sq,d], k/v [B,H,Sk,d], visible bool [B,1,Sq,Sk] or None.

    mode "exact"    : global row max, P rounded by R for the PV product, normaliser from unrounded P (the reference's
                      eager attention when R is the identity; also what the generic HIP kernel does).
    mode "flash64"  : arithmetic of the MFMA flash kernel (csrc/attention.hip attn_mfma_kernel): 64-key tiles from key
                      0, running max, P = exp2(s*scale*log2e - m_run) rounded by R per tile, fp32 rescale.
    mode "splitN"   : arithmetic of the decode kernel: independent N-key chunks (chunk max, rounded P), fp32 combine
                      ("split128" is what the model-level restatement uses; the kernel's chunk is a tuning knob, 64 by default
                      for one conversation, and any N gives the same value up to the bf16 rounding of P).
    The bf16 parity tests use the emulating modes so that P is rounded at exactly the kernels' points.
    """
    B, H, Sq, d = q.shape
    Sk = k.shape[2]
    neg = float("-inf")
    if mode == "tf431_clip":          # q arrives pre-scaled (and rounded); 16-bit bmm output, softmax output rounded NORMALISED, 16-bit P V
        s = R(q @ k.transpose(-1, -2))
        p = R(torch.softmax(s, dim=-1))
        return R(p @ v)
    if mode == "tf431_llama":         # 16-bit q k^T, / sqrt(d) rounded, additive mask, fp32 softmax cast back NORMALISED, 16-bit P V
        s = R(R(q @ k.transpose(-1, -2)) / (1.0 / scale))
        if visible is not None:
            s = s.masked_fill(~visible, neg)
        p = R(torch.softmax(s, dim=-1))
        return R(p @ v)
    if mode == "exact":
        s = (q @ k.transpose(-1, -2)) * scale
        if visible is not None:
            s = s.masked_fill(~visible, neg)
        m = s.max(dim=-1, keepdim=True).values
        p = torch.exp(s - m)
        return (R(p) @ v) / p.sum(dim=-1, keepdim=True)
    if mode == "flash64":
        sl2 = (torch.tensor(scale, dtype=torch.float32) * torch.tensor(LOG2E, dtype=torch.float32)).to(q.dtype)
        m = torch.full((B, H, Sq, 1), neg, dtype=q.dtype)
        l = torch.zeros((B, H, Sq, 1), dtype=q.dtype)
        acc = torch.zeros((B, H, Sq, d), dtype=q.dtype)
        for j0 in range(0, Sk, 64):
            j1 = min(Sk, j0 + 64)
            s = (q @ k[:, :, j0:j1].transpose(-1, -2)) * sl2
            if visible is not None:
                s = s.masked_fill(~visible[..., j0:j1], neg)
            m_new = torch.maximum(m, s.max(dim=-1, keepdim=True).values)
            m_use = torch.where(torch.isinf(m_new), torch.zeros_like(m_new), m_new)
            alpha = torch.exp2(m - m_use)
            p = torch.exp2(s - m_use)
            l = l * alpha + p.sum(dim=-1, keepdim=True)
            acc = acc * alpha + R(p) @ v[:, :, j0:j1]
            m = m_new
        return acc / l
    if mode.startswith("split"):
        chunk = int(mode[5:])                     # keys per independent chunk ("split128", "split64", ...)
        ms, ls, os_ = [], [], []
        for j0 in range(0, Sk, chunk):
            j1 = min(Sk, j0 + chunk)
            s = (q @ k[:, :, j0:j1].transpose(-1, -2)) * scale
            if visible is not None:
                s = s.masked_fill(~visible[..., j0:j1], neg)
            mc = s.max(dim=-1, keepdim=True).values
            p = torch.exp(s - torch.where(torch.isinf(mc), torch.zeros_like(mc), mc))
            ms.append(mc); ls.append(p.sum(dim=-1, keepdim=True)); os_.append(R(p) @ v[:, :, j0:j1])
        M = torch.stack(ms).max(dim=0).values
        L = sum(l_ * torch.exp(m_ - M) for m_, l_ in zip(ms, ls))
        O = sum(o_ * torch.exp(m_ - M) for m_, o_ in zip(ms, os_))
        return O / L
    raise ValueError(f"unknown attention mode {mode}")


FORCE_KERNEL_MODES = False     # tests/test_oracle_golden.py: run the tile-wise modes with R = identity against the goldens


def kernel_attention_mode(rounding, head_dim, q_len, decode_kernel=False):
    """Which arithmetic the HIP path uses for this call (bf16 path only; fp32 always runs the exact generic kernel).
    With FORCE_KERNEL_MODES the tile-wise modes are chosen without rounding too: that is how they are pinned to the
    reference's own outputs (they must reproduce the fp32 goldens to fp32 round-off)."""
    if rounding is None and not FORCE_KERNEL_MODES:
        return "exact"
    if decode_kernel and q_len == 1:
        return "split128"
    return "flash64" if head_dim in (64, 128) else "exact"


# --------------------------------------------------------------------------------------
# H9-H11, H7/H8: CLIP ViT
# --------------------------------------------------------------------------------------
def _act(name: str):
    if name == "quick_gelu":
        return lambda x: x * torch.sigmoid(1.702 * x)      # tf activations.py QuickGELUActivation
    if name == "gelu":
        return lambda x: F.gelu(x)                          # erf form
    raise ValueError(f"unsupported hidden_act {name}")


def vit_embeddings(pixels: torch.Tensor, sd, cfg: VitCfg, R) -> torch.Tensor:
    """tf clip/modeling_clip.py CLIPVisionEmbeddings.forward (used at modeling_image.py:602,645):
    conv(k=stride=patch, no bias) == GEMM over flattened (c,kh,kw) patches; prepend class_embedding;
    add position_embedding[0..N]."""
    B = pixels.shape[0]
    P, D = cfg.patch_size, cfg.hidden_size
    w = sd[VIT_PREFIX + "embeddings.patch_embedding.weight"].to(pixels.dtype)     # [D, 3, P, P]
    g = cfg.image_size // P
    # [B,3,g,P,g,P] -> [B,g,g,3,P,P] -> [B*g*g, 3*P*P]
    cols = pixels.view(B, cfg.num_channels, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B * g * g, -1)
    patches = R(_lin(cols, w.reshape(D, -1))).view(B, g * g, D)
    cls = sd[VIT_PREFIX + "embeddings.class_embedding"].to(pixels.dtype).view(1, 1, D).expand(B, 1, D)
    pos = sd[VIT_PREFIX + "embeddings.position_embedding.weight"].to(pixels.dtype)
    return R(torch.cat([cls, patches], dim=1) + pos.unsqueeze(0))


def _layernorm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w.to(x.dtype), b.to(x.dtype), eps)


def vit_attention(x, sd, pre, cfg: VitCfg, R, mode="exact"):
    """tf CLIPAttention (constructed at modeling_image.py:69): q/k/v/out Linear with bias,
    heads x head_dim, softmax(q k^T * d^-1/2) v, no mask on the vision path."""
    B, N, D = x.shape
    H = cfg.num_attention_heads
    d = D // H

    def lin(name):
        return R(_lin(x, sd[pre + name + ".weight"]) + sd[pre + name + ".bias"].to(x.dtype))

    q = lin("self_attn.q_proj").view(B, N, H, d).transpose(1, 2)
    k = lin("self_attn.k_proj").view(B, N, H, d).transpose(1, 2)
    v = lin("self_attn.v_proj").view(B, N, H, d).transpose(1, 2)
    if OP_ORDER == "tf431":
        o = attention_core(R(q * (d ** -0.5)), k, v, None, 1.0, R, "tf431_clip")
    else:
        o = attention_core(q, k, v, None, d ** -0.5, R, mode)
    return R(o.transpose(1, 2).reshape(B, N, D))


def vit_layer(h, sd, i, cfg: VitCfg, R, mode="exact"):
    """modeling_image.py:136-151 (spatial branch; add_time_attn is False for the image tower):
    h += Attn(LN1(h)); h += MLP(LN2(h))."""
    pre = VIT_PREFIX + f"encoder.layers.{i}."
    a = R(_layernorm(h, sd[pre + "layer_norm1.weight"], sd[pre + "layer_norm1.bias"], cfg.layer_norm_eps))
    a = vit_attention(a, sd, pre, cfg, R, mode)
    h = R(h + _lin(a, sd[pre + "self_attn.out_proj.weight"]) + sd[pre + "self_attn.out_proj.bias"].to(h.dtype))
    m = R(_layernorm(h, sd[pre + "layer_norm2.weight"], sd[pre + "layer_norm2.bias"], cfg.layer_norm_eps))
    m = R(_act(cfg.hidden_act)(_lin(m, sd[pre + "mlp.fc1.weight"]) + sd[pre + "mlp.fc1.bias"].to(h.dtype)))
    h = R(h + _lin(m, sd[pre + "mlp.fc2.weight"]) + sd[pre + "mlp.fc2.bias"].to(h.dtype))
    return h


def vit_hidden_states(pixels, sd, cfg: VitCfg, rounding=None, n_layers=None):
    """modeling_image.py:610-672 + CLIPEncoder loop :400-425 with output_hidden_states=True:
    states[0] = pre_layrnorm(embeddings); states[i+1] = layer_i(states[i])."""
    R = _rounder(rounding)
    x = R(pixels)
    h = vit_embeddings(x, sd, cfg, R)
    h = R(_layernorm(h, sd[VIT_PREFIX + "pre_layrnorm.weight"], sd[VIT_PREFIX + "pre_layrnorm.bias"], cfg.layer_norm_eps))
    states = [h]
    L = cfg.num_hidden_layers if n_layers is None else n_layers
    mode = kernel_attention_mode(rounding, cfg.hidden_size // cfg.num_attention_heads, cfg.num_positions)
    for i in range(L):
        h = vit_layer(h, sd, i, cfg, R, mode)
        states.append(h)
    return states


def vit_features(pixels, sd, cfg: VitCfg, select_layer=-2, select_feature="patch", rounding=None):
    """languagebind/__init__.py:121-146 (twin: clip_encoder.py:29-51): hidden_states[select_layer],
    drop CLS for 'patch'.  Layers after the selected one are dead code and are skipped."""
    n_states = cfg.num_hidden_layers + 1
    idx = select_layer if select_layer >= 0 else n_states + select_layer
    states = vit_hidden_states(pixels, sd, cfg, rounding, n_layers=idx)
    f = states[idx]
    if select_feature == "patch":
        return f[:, 1:]
    if select_feature == "cls_patch":
        return f
    raise ValueError(f"Unexpected select feature: {select_feature}")


# --------------------------------------------------------------------------------------
# H12: projector
# --------------------------------------------------------------------------------------
def projector(x, sd, projector_type="mlp2x_gelu", rounding=None):
    """multimodal_projector/builder.py:33-51: linear | mlpNx_gelu | identity."""
    import re
    R = _rounder(rounding)
    pre = "model.mm_projector."
    if projector_type == "identity":
        return x
    if projector_type == "linear":
        return R(_lin(x, sd[pre + "weight"]) + sd[pre + "bias"].to(x.dtype))
    m = re.match(r"^mlp(\d+)x_gelu$", projector_type)
    if not m:
        raise ValueError(f"Unknown projector type: {projector_type}")
    depth = int(m.group(1))
    h = _lin(x, sd[pre + "0.weight"]) + sd[pre + "0.bias"].to(x.dtype)
    for j in range(1, depth):
        h = R(F.gelu(h))
        h = _lin(h, sd[pre + f"{2 * j}.weight"]) + sd[pre + f"{2 * j}.bias"].to(x.dtype)
    return R(h)


def encode_images(pixels, sd, vcfg: VitCfg, mm: MMCfg, rounding=None):
    """llava_arch.py:137-140."""
    f = vit_features(pixels, sd, vcfg, mm.mm_vision_select_layer, mm.mm_vision_select_feature, rounding)
    return projector(f, sd, mm.mm_projector_type, rounding)


# --------------------------------------------------------------------------------------
# H13: embedding splice
# --------------------------------------------------------------------------------------
def prepare_inputs_labels_for_multimodal(input_ids, position_ids, attention_mask, past_key_values, labels,
                                         image_features: Optional[List[torch.Tensor]], embed_weight, mm: MMCfg,
                                         past_len: Optional[int] = None):
    """llava_arch.py:148-346.  `image_features` is the flat per-image list the reference builds at
    :212-222 (None plays the role of images=None).  Returns the reference's 6-tuple."""
    if image_features is None or input_ids.shape[1] == 1:
        # decode-step early-out, llava_arch.py:154-163
        if past_key_values is not None and image_features is not None and input_ids.shape[1] == 1:
            target = (past_len if past_len is not None else past_key_values[-1][-1].shape[-2]) + 1
            attention_mask = torch.cat(
                (attention_mask, torch.ones((attention_mask.shape[0], target - attention_mask.shape[1]),
                                            dtype=attention_mask.dtype)), dim=1)
            position_ids = torch.sum(attention_mask, dim=1).unsqueeze(-1) - 1
        return input_ids, position_ids, attention_mask, past_key_values, None, labels

    _labels, _pos, _mask = labels, position_ids, attention_mask
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids, dtype=torch.bool)
    else:
        attention_mask = attention_mask.bool()
    if position_ids is None:
        position_ids = torch.arange(0, input_ids.shape[1], dtype=torch.long)
    if labels is None:
        labels = torch.full_like(input_ids, IGNORE_INDEX)

    ids_list = [row[m] for row, m in zip(input_ids, attention_mask)]          # :248
    lab_list = [row[m] for row, m in zip(labels, attention_mask)]             # :249

    new_embeds, new_labels = [], []
    cur = 0
    for b, ids in enumerate(ids_list):
        n_img = int((ids == IMAGE_TOKEN_INDEX).sum())
        if n_img == 0:                                                          # :257-264
            feat = image_features[cur]
            e = torch.cat([embed_weight[ids], feat[0:0].to(embed_weight.dtype)], dim=0)
            new_embeds.append(e)
            new_labels.append(lab_list[b])
            cur += 1
            continue
        cut = [-1] + torch.where(ids == IMAGE_TOKEN_INDEX)[0].tolist() + [ids.shape[0]]   # :266
        pieces_e, pieces_l = [], []
        for i in range(len(cut) - 1):
            seg = ids[cut[i] + 1:cut[i + 1]]
            pieces_e.append(embed_weight[seg])                                  # :274 (one gather, split back)
            pieces_l.append(lab_list[b][cut[i] + 1:cut[i + 1]])
            if i < n_img:
                feat = image_features[cur]                                      # IndexError here mirrors :284
                cur += 1
                pieces_e.append(feat.to(embed_weight.dtype))
                pieces_l.append(torch.full((feat.shape[0],), IGNORE_INDEX, dtype=lab_list[b].dtype))
        new_embeds.append(torch.cat(pieces_e))
        new_labels.append(torch.cat(pieces_l))

    if mm.tokenizer_model_max_length is not None:                               # :295-299
        new_embeds = [x[:mm.tokenizer_model_max_length] for x in new_embeds]
        new_labels = [x[:mm.tokenizer_model_max_length] for x in new_labels]

    max_len = max(x.shape[0] for x in new_embeds)
    B = len(new_embeds)
    lab_pad = torch.full((B, max_len), IGNORE_INDEX, dtype=new_labels[0].dtype)
    mask = torch.zeros((B, max_len), dtype=attention_mask.dtype)
    pos = torch.zeros((B, max_len), dtype=position_ids.dtype)
    padded = []
    for i, (e, l) in enumerate(zip(new_embeds, new_labels)):                    # :310-329
        n = e.shape[0]
        z = torch.zeros((max_len - n, e.shape[1]), dtype=e.dtype)
        if mm.tokenizer_padding_side == "left":
            padded.append(torch.cat((z, e), dim=0))
            if n > 0:
                lab_pad[i, -n:] = l
                mask[i, -n:] = True
                pos[i, -n:] = torch.arange(0, n, dtype=pos.dtype)
        else:
            padded.append(torch.cat((e, z), dim=0))
            if n > 0:
                lab_pad[i, :n] = l
                mask[i, :n] = True
                pos[i, :n] = torch.arange(0, n, dtype=pos.dtype)
    embeds = torch.stack(padded, dim=0)
    out_labels = None if _labels is None else lab_pad
    out_mask = None if _mask is None else mask.to(dtype=_mask.dtype)
    out_pos = None if _pos is None else pos
    return None, out_pos, out_mask, past_key_values, embeds, out_labels


# --------------------------------------------------------------------------------------
# H15: LLaMA stack
# --------------------------------------------------------------------------------------
def rmsnorm(x, w, eps):
    """tf llama/modeling_llama.py LlamaRMSNorm: x * rsqrt(mean(x^2) + eps) * w."""
    var = x.pow(2).mean(-1, keepdim=True)
    return x * torch.rsqrt(var + eps) * w.to(x.dtype)


def rope_cos_sin(position_ids, head_dim, theta, dtype):
    """tf LlamaRotaryEmbedding: inv_freq = 1/theta^(2i/d) and freqs = pos * inv_freq, both in fp32
    (the reference forces fp32 there); emb = cat(freqs, freqs); cos/sin then cast to the model dtype."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = position_ids.to(torch.float32)[..., None] * inv                        # [..., d/2]
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


@dataclass
class KVCache:
    """tf DynamicCache / the legacy tuple cache: per layer K and V [B, H_kv, S, d], grown along S.
    capacity > 0: the buffers are allocated once at [B, H_kv, capacity, d] and k[i] / v[i] are views of the filled part
    (what a CPU run that is being TIMED should do: `torch.cat` per layer per step re-copies the whole cache)."""
    k: List[torch.Tensor] = field(default_factory=list)     # per layer [B, H_kv, S, d]
    v: List[torch.Tensor] = field(default_factory=list)
    capacity: int = 0
    _kbuf: List[torch.Tensor] = field(default_factory=list)
    _vbuf: List[torch.Tensor] = field(default_factory=list)

    @property
    def length(self):
        return 0 if not self.k else self.k[0].shape[2]

    def append(self, i, k, v):
        """Add the new positions of layer i; returns the full (K, V) of that layer."""
        if not self.capacity:
            if len(self.k) > i:
                self.k[i] = torch.cat((self.k[i], k), dim=2)
                self.v[i] = torch.cat((self.v[i], v), dim=2)
            else:
                self.k.append(k)
                self.v.append(v)
            return self.k[i], self.v[i]
        if len(self._kbuf) <= i:
            B, Hk, _, d = k.shape
            self._kbuf.append(torch.empty(B, Hk, self.capacity, d, dtype=k.dtype))
            self._vbuf.append(torch.empty(B, Hk, self.capacity, d, dtype=v.dtype))
            self.k.append(self._kbuf[i][:, :, :0])
            self.v.append(self._vbuf[i][:, :, :0])
        n0, S = self.k[i].shape[2], k.shape[2]
        if n0 + S > self.capacity:
            raise ValueError(f"KVCache capacity {self.capacity} exceeded")
        self._kbuf[i][:, :, n0:n0 + S] = k
        self._vbuf[i][:, :, n0:n0 + S] = v
        self.k[i] = self._kbuf[i][:, :, :n0 + S]
        self.v[i] = self._vbuf[i][:, :, :n0 + S]
        return self.k[i], self.v[i]


def quant_rows_e4m3(x):
    """Per-token (per-row) activation quantisation of the w8a8 prefill path (config C5; csrc/gemm_fp8.hip quant_rows_fp8): the
    reference has no counterpart (its 8-bit option is bitsandbytes, eval.py:52-53) -- this restates the DEFINITION the kernels
    implement: s = max|x| * (1/448) in fp32 (1 for an all-zero row), q = e4m3_rne(x * (1/s)), value used by the GEMM = q * s.
    Returns the dequantised tensor (same shape / dtype as x)."""
    xf = x.to(torch.float32)
    amax = xf.abs().amax(dim=-1, keepdim=True)
    s = torch.where(amax > 0, amax * torch.tensor(1.0 / 448.0, dtype=torch.float32), torch.ones_like(amax))
    q = (xf * (1.0 / s)).to(torch.float8_e4m3fn).to(torch.float32)
    return (q * s).to(x.dtype)


def attention_probs(q, k, visible, scale, R):
    """`output_attentions` of the kept forward signature (llava_llama.py:65,95 -> tf llama/modeling_llama.py eager_attention_forward): the
    maps a caller gets back -- softmax over the visible keys of scale * q k^T, statistics in fp32, rounded ONCE to the working type (R);
    masked keys are exactly 0.  q [B,H,Sq,d], k [B,H,Sk,d] (the rotated, 16-bit-rounded operands the attention kernels read)."""
    s = (q.float() @ k.float().transpose(-1, -2)) * scale
    if visible is not None:
        s = s.masked_fill(~visible, float("-inf"))
    return R(torch.softmax(s, dim=-1))


def llama_layer(h, i, sd, cfg: LlamaCfg, cos, sin, visible, cache: KVCache, R, mode="exact", act_quant=None, attentions=None):
    """tf LlamaDecoderLayer/LlamaAttention/LlamaMLP (called from llava_llama.py:88-99).
    act_quant="e4m3": the input of every Linear layer goes through quant_rows_e4m3 first (w8a8 prefill of config C5; the weights
    in `sd` are then expected to be the dequantised e4m3 weights)."""
    if act_quant not in (None, "e4m3"):
        raise ValueError(f"unknown act_quant {act_quant!r}")
    if act_quant:
        return _llama_layer_w8a8(h, i, sd, cfg, cos, sin, visible, cache, R, mode)
    B, S, D = h.shape
    H, Hk, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    pre = f"model.layers.{i}."
    n1 = R(rmsnorm(h, sd[pre + "input_layernorm.weight"], cfg.rms_norm_eps))
    q = R(_lin(n1, sd[pre + "self_attn.q_proj.weight"])).view(B, S, H, d).transpose(1, 2)
    k = R(_lin(n1, sd[pre + "self_attn.k_proj.weight"])).view(B, S, Hk, d).transpose(1, 2)
    v = R(_lin(n1, sd[pre + "self_attn.v_proj.weight"])).view(B, S, Hk, d).transpose(1, 2)
    c, s = cos.unsqueeze(1), sin.unsqueeze(1)
    if OP_ORDER == "tf431":                                                      # 16-bit cos / sin caches, three 16-bit ops
        c, s = R(c), R(s)
        q = R(R(q * c) + R(rotate_half(q) * s))
        k = R(R(k * c) + R(rotate_half(k) * s))
    else:
        q = R(q * c + rotate_half(q) * s)
        k = R(k * c + rotate_half(k) * s)
    kk, vv = cache.append(i, k, v)
    if Hk != H:                                                                  # repeat_kv
        rep = H // Hk
        kk = kk[:, :, None].expand(B, Hk, rep, kk.shape[2], d).reshape(B, H, -1, d)
        vv = vv[:, :, None].expand(B, Hk, rep, vv.shape[2], d).reshape(B, H, -1, d)
    if attentions is not None:
        attentions.append(attention_probs(q, kk, visible, 1.0 / math.sqrt(d), R))
    o = attention_core(q, kk, vv, visible, 1.0 / math.sqrt(d), R, "tf431_llama" if OP_ORDER == "tf431" else mode)
    o = R(o.transpose(1, 2).reshape(B, S, D))
    h = R(h + _lin(o, sd[pre + "self_attn.o_proj.weight"]))
    n2 = R(rmsnorm(h, sd[pre + "post_attention_layernorm.weight"], cfg.rms_norm_eps))
    g = _lin(n2, sd[pre + "mlp.gate_proj.weight"])
    u = _lin(n2, sd[pre + "mlp.up_proj.weight"])
    a = R(F.silu(g) * u)
    h = R(h + _lin(a, sd[pre + "mlp.down_proj.weight"]))
    return h


def _llama_layer_w8a8(h, i, sd, cfg: LlamaCfg, cos, sin, visible, cache: KVCache, R, mode):
    """llama_layer with per-token e4m3 activations in front of the four Linear layers (same rounding boundaries otherwise)."""
    B, S, D = h.shape
    H, Hk, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    pre = f"model.layers.{i}."
    Q = quant_rows_e4m3
    n1 = Q(R(rmsnorm(h, sd[pre + "input_layernorm.weight"], cfg.rms_norm_eps)))
    q = R(_lin(n1, sd[pre + "self_attn.q_proj.weight"])).view(B, S, H, d).transpose(1, 2)
    k = R(_lin(n1, sd[pre + "self_attn.k_proj.weight"])).view(B, S, Hk, d).transpose(1, 2)
    v = R(_lin(n1, sd[pre + "self_attn.v_proj.weight"])).view(B, S, Hk, d).transpose(1, 2)
    c, s = cos.unsqueeze(1), sin.unsqueeze(1)
    q = R(q * c + rotate_half(q) * s)
    k = R(k * c + rotate_half(k) * s)
    kk, vv = cache.append(i, k, v)
    if Hk != H:
        rep = H // Hk
        kk = kk[:, :, None].expand(B, Hk, rep, kk.shape[2], d).reshape(B, H, -1, d)
        vv = vv[:, :, None].expand(B, Hk, rep, vv.shape[2], d).reshape(B, H, -1, d)
    o = attention_core(q, kk, vv, visible, 1.0 / math.sqrt(d), R, mode)
    o = Q(R(o.transpose(1, 2).reshape(B, S, D)))
    h = R(h + _lin(o, sd[pre + "self_attn.o_proj.weight"]))
    n2 = Q(R(rmsnorm(h, sd[pre + "post_attention_layernorm.weight"], cfg.rms_norm_eps)))
    g = _lin(n2, sd[pre + "mlp.gate_proj.weight"])
    u = _lin(n2, sd[pre + "mlp.up_proj.weight"])
    a = Q(R(F.silu(g) * u))
    h = R(h + _lin(a, sd[pre + "mlp.down_proj.weight"]))
    return h


def llama_forward(inputs_embeds, position_ids, attention_mask, cache: Optional[KVCache], sd, cfg: LlamaCfg,
                  rounding=None, last_only=False, return_hidden=False, decode_kernel=False, act_quant=None, hidden_states=None,
                  attentions=None):
    """LlamaModel + lm_head.  inputs_embeds [B,S,D]; position_ids [B,S] or None (-> past..past+S);
    attention_mask [B, past+S] of 0/1 or None.  Returns fp logits [B,S,V] (or [B,1,V]).
    hidden_states: a list that receives what LlamaModel.forward collects under output_hidden_states=True (llava_llama.py:56-69 ->
    tf llama/modeling_llama.py LlamaModel.forward): the input embeddings, the residual stream after every layer but the last, and
    the final-normed states -- num_hidden_layers + 1 tensors [B,S,D]."""
    R = _rounder(rounding)
    if cache is None:
        cache = KVCache()
    B, S, D = inputs_embeds.shape
    past = cache.length
    if position_ids is None:
        position_ids = torch.arange(past, past + S, dtype=torch.long).unsqueeze(0).expand(B, S)
    h = R(inputs_embeds)
    cos, sin = rope_cos_sin(position_ids, cfg.head_dim, cfg.rope_theta, h.dtype)
    T = past + S
    qpos = torch.arange(past, T).view(S, 1)
    kpos = torch.arange(0, T).view(1, T)
    visible = (kpos <= qpos).view(1, 1, S, T).expand(B, 1, S, T)
    if attention_mask is not None:
        visible = visible & (attention_mask[:, None, None, :T] != 0)
    mode = kernel_attention_mode(rounding, cfg.head_dim, S, decode_kernel)
    for i in range(cfg.num_hidden_layers):
        if hidden_states is not None:
            hidden_states.append(h)
        h = llama_layer(h, i, sd, cfg, cos, sin, visible, cache, R, mode, act_quant, attentions=attentions)
    if hidden_states is not None:
        hidden_states.append(R(rmsnorm(h, sd["model.norm.weight"], cfg.rms_norm_eps)))
    if last_only:
        h = h[:, -1:, :]
    hn = R(rmsnorm(h, sd["model.norm.weight"], cfg.rms_norm_eps))
    logits = _lin(hn, sd["lm_head.weight"])
    if return_hidden:
        return logits, cache, hn
    return logits, cache


# --------------------------------------------------------------------------------------
# H14/H16: multimodal forward and greedy generation
# --------------------------------------------------------------------------------------
def mm_forward(input_ids, images: List[torch.Tensor], sd, vcfg: VitCfg, lcfg: LlamaCfg, mm: MMCfg,
               attention_mask=None, rounding=None, dtype=torch.float32, hidden_states=None, attentions=None):
    """llava_llama.py:56-99 prefill: encode + splice + LLaMA.  images = flat list of [3,H,W]."""
    pix = torch.stack([im.to(dtype) for im in images])                           # llava_arch.py:194
    feats = encode_images(pix, sd, vcfg, mm, rounding)
    flat = [feats[i] for i in range(feats.shape[0])]
    emb_w = sd["model.embed_tokens.weight"].to(dtype)
    _, pos, mask, _, embeds, _ = prepare_inputs_labels_for_multimodal(
        input_ids, None, attention_mask, None, None, flat, emb_w, mm)
    logits, cache = llama_forward(embeds, pos, mask, None, sd, lcfg, rounding, hidden_states=hidden_states, attentions=attentions)
    return logits, cache, embeds


def logit_stats(a, b):
    """(max, p99, median) of |a - b| as fractions of max|b|."""
    d = (a.double() - b.double()).abs().flatten()
    scale = float(b.abs().max())
    k99 = max(1, int(0.99 * d.numel()))
    return float(d.max()) / scale, float(d.kthvalue(k99).values) / scale, float(d.median()) / scale


SELF_DIFF_ORDERS = ((8, False), (8, True), (4, True), (2, False))


def self_difference(input_ids, images, sd, vcfg: VitCfg, lcfg: LlamaCfg, mm: MMCfg, rounding, base=None):
    """The 16-bit noise floor of THIS configuration, measured (test infrastructure; tests/test_noise_floor.py is the demonstration at
    7B width): the oracle's prefill logits under `rounding`, re-run with only the fp32 summation order of its Linear layers changed
    (K_ORDER: K in 8 / 4 / 2 chunks, ascending or descending) -- same weights, same rounding points.  Returns (max, p99, median) of
    the pairwise differences between the runs (and `base`, the plain single-matmul run, if given) as fractions of max|logit|: the
    LARGEST of the pairs for the max statistic (a maximum over a few 10^5 logits is itself noisy), the mean of the pairs for p99 and
    median.  A kernel path that differs from the oracle by rounding only lands at ~1 x these numbers; the parity tests assert
    <= 1.25 x."""
    global K_ORDER
    runs = [] if base is None else [base]
    try:
        for order in SELF_DIFF_ORDERS:
            K_ORDER = order
            lg, _, _ = mm_forward(input_ids, images, sd, vcfg, lcfg, mm, None, rounding, torch.float32)
            runs.append(lg[0])
    finally:
        K_ORDER = None
    st = [logit_stats(runs[i], runs[j]) for i in range(len(runs)) for j in range(i + 1, len(runs))]
    n = float(len(st))
    return max(x[0] for x in st), sum(x[1] for x in st) / n, sum(x[2] for x in st) / n


def training_loss(input_ids, attention_mask, labels, images, sd, vcfg: VitCfg, lcfg: LlamaCfg, mm: MMCfg):
    """Training-shape forward with labels: llava_llama.py:56-99 -> LlamaForCausalLM loss = CrossEntropyLoss over the
    shifted positions of the SPLICED sequence (labels come back from prepare_inputs_labels_for_multimodal with
    IGNORE_INDEX on visual rows and padding, llava_arch.py:251-340).  Returns (loss, logits, spliced labels)."""
    pix = torch.stack([im.to(torch.float32) for im in images])
    feats = encode_images(pix, sd, vcfg, mm, None)
    flat = [feats[i] for i in range(feats.shape[0])]
    emb_w = sd["model.embed_tokens.weight"].to(torch.float32)
    _, pos, mask, _, embeds, lab = prepare_inputs_labels_for_multimodal(input_ids, None, attention_mask, None, labels, flat, emb_w, mm)
    logits, _ = llama_forward(embeds, pos, mask, None, sd, lcfg, None)
    V = logits.shape[-1]
    loss = torch.nn.functional.cross_entropy(logits[:, :-1].reshape(-1, V), lab[:, 1:].reshape(-1), ignore_index=IGNORE_INDEX)
    return loss, logits, lab


def greedy_generate(input_ids, images, sd, vcfg, lcfg, mm, max_new_tokens, rounding=None, dtype=torch.float32,
                    eos_token_id=None):
    """inference.py:64-72 with do_sample=False: prefill, then 1-token steps with the KV cache
    (decode branch llava_arch.py:154-163: position = past_len)."""
    logits, cache, _ = mm_forward(input_ids, images, sd, vcfg, lcfg, mm, None, rounding, dtype)
    emb_w = sd["model.embed_tokens.weight"].to(dtype)
    out, step_logits = [], []
    nxt = int(torch.argmax(logits[0, -1]))
    step_logits.append(logits[0, -1].clone())
    out.append(nxt)
    for _ in range(max_new_tokens - 1):
        if eos_token_id is not None and nxt == eos_token_id:
            break
        e = emb_w[torch.tensor([[nxt]])]
        logits, cache = llama_forward(e, None, None, cache, sd, lcfg, rounding, decode_kernel=True)
        nxt = int(torch.argmax(logits[0, -1]))
        step_logits.append(logits[0, -1].clone())
        out.append(nxt)
    return out, torch.stack(step_logits), cache


# --------------------------------------------------------------------------------------
# deterministic synthetic weights (shared by tests, smoke and bench's cpu_baseline)
# --------------------------------------------------------------------------------------
def make_state_dict(vcfg: VitCfg, lcfg: LlamaCfg, mm: MMCfg, seed=2, std=0.02, dtype=torch.float32,
                    vit_layers=None, llm_layers=None):
    """N(0, std^2) weights with the H17 key names; norm weights 1 + small noise."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, s=std):
        return (torch.randn(*shape, generator=g) * s).to(dtype)

    sd = {}
    D, F_, V = lcfg.hidden_size, lcfg.intermediate_size, lcfg.vocab_size
    d = lcfg.head_dim
    sd["model.embed_tokens.weight"] = rn(V, D)
    for i in range(lcfg.num_hidden_layers if llm_layers is None else llm_layers):
        p = f"model.layers.{i}."
        sd[p + "self_attn.q_proj.weight"] = rn(lcfg.num_attention_heads * d, D)
        sd[p + "self_attn.k_proj.weight"] = rn(lcfg.num_key_value_heads * d, D)
        sd[p + "self_attn.v_proj.weight"] = rn(lcfg.num_key_value_heads * d, D)
        sd[p + "self_attn.o_proj.weight"] = rn(D, D)
        sd[p + "mlp.gate_proj.weight"] = rn(F_, D)
        sd[p + "mlp.up_proj.weight"] = rn(F_, D)
        sd[p + "mlp.down_proj.weight"] = rn(D, F_)
        sd[p + "input_layernorm.weight"] = (1.0 + rn(D, s=0.1)).to(dtype)
        sd[p + "post_attention_layernorm.weight"] = (1.0 + rn(D, s=0.1)).to(dtype)
    sd["model.norm.weight"] = (1.0 + rn(D, s=0.1)).to(dtype)
    sd["lm_head.weight"] = rn(V, D)
    Dv, Fv = vcfg.hidden_size, vcfg.intermediate_size
    sd[VIT_PREFIX + "embeddings.class_embedding"] = rn(Dv)
    sd[VIT_PREFIX + "embeddings.patch_embedding.weight"] = rn(Dv, vcfg.num_channels, vcfg.patch_size, vcfg.patch_size)
    sd[VIT_PREFIX + "embeddings.position_embedding.weight"] = rn(vcfg.num_positions, Dv)
    sd[VIT_PREFIX + "pre_layrnorm.weight"] = (1.0 + rn(Dv, s=0.1)).to(dtype)
    sd[VIT_PREFIX + "pre_layrnorm.bias"] = rn(Dv)
    for i in range(vcfg.num_hidden_layers if vit_layers is None else vit_layers):
        p = VIT_PREFIX + f"encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"self_attn.{nm}.weight"] = rn(Dv, Dv)
            sd[p + f"self_attn.{nm}.bias"] = rn(Dv)
        sd[p + "layer_norm1.weight"] = (1.0 + rn(Dv, s=0.1)).to(dtype)
        sd[p + "layer_norm1.bias"] = rn(Dv)
        sd[p + "layer_norm2.weight"] = (1.0 + rn(Dv, s=0.1)).to(dtype)
        sd[p + "layer_norm2.bias"] = rn(Dv)
        sd[p + "mlp.fc1.weight"] = rn(Fv, Dv)
        sd[p + "mlp.fc1.bias"] = rn(Fv)
        sd[p + "mlp.fc2.weight"] = rn(Dv, Fv)
        sd[p + "mlp.fc2.bias"] = rn(Dv)
    sd["model.mm_projector.0.weight"] = rn(D, mm.mm_hidden_size)
    sd["model.mm_projector.0.bias"] = rn(D)
    sd["model.mm_projector.2.weight"] = rn(D, D)
    sd["model.mm_projector.2.bias"] = rn(D)
    return sd


def make_state_dict_for_timing(vcfg: VitCfg, lcfg: LlamaCfg, mm: MMCfg, seed=2, std=0.02, dtype=torch.float32, base=None):
    """Full-depth synthetic weights for the CPU-baseline TIMING run (bench.py cpu_baseline): drawing 7e9 normals on one host
    thread takes minutes, so ONE encoder layer and ONE decoder layer are drawn like make_state_dict draws them and the
    other layers are rolled copies of those (torch.roll: distinct memory for every layer, same value statistics) -- a
    timing run streams exactly as many distinct bytes as with independent weights; its numbers mean nothing."""
    sd = dict(base) if base is not None else make_state_dict(vcfg, lcfg, mm, seed=seed, std=std, dtype=dtype, vit_layers=1,
                                                               llm_layers=1)
    for i in range(1, lcfg.num_hidden_layers):
        for k in [k for k in sd if k.startswith("model.layers.0.")]:
            w = sd[k]
            sd[k.replace("model.layers.0.", f"model.layers.{i}.")] = torch.roll(w, shifts=977 * i, dims=-1) if w.dim() > 1 else w.clone()
    p0 = VIT_PREFIX + "encoder.layers.0."
    for i in range(1, vcfg.num_hidden_layers):
        for k in [k for k in sd if k.startswith(p0)]:
            w = sd[k]
            sd[k.replace(p0, VIT_PREFIX + f"encoder.layers.{i}.")] = torch.roll(w, shifts=131 * i, dims=-1) if w.dim() > 1 else w.clone()
    return sd


def synthetic_prompt_ids(n_text: int, T: int, vocab: int, seed=1):
    """[BOS] + uniform{3..vocab-1}, exactly T sentinels at evenly spaced positions >= 8
    (>= 2 for prompts shorter than 16)."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, vocab, (n_text,), generator=g, dtype=torch.long)
    ids[0] = 1
    lo = 8 if n_text >= 16 + T else 2
    span = n_text - lo - 1
    for j in range(T):
        ids[lo + (j * span) // max(T, 1)] = IMAGE_TOKEN_INDEX
    assert int((ids == IMAGE_TOKEN_INDEX).sum()) == T
    return ids


def synthetic_frames(T: int, image_size=224, seed=0):
    """uint8-like U{0..255} frames -> H6 normalisation; list of T fp32 [3,H,W] tensors."""
    g = torch.Generator().manual_seed(seed)
    raw = torch.randint(0, 256, (T, image_size, image_size, 3), generator=g, dtype=torch.uint8)
    return [preprocess_uint8(raw[t]) for t in range(T)]
