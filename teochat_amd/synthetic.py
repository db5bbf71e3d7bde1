"""Synthetic checkpoints (there is no network for real weights): N(0, std^2) tensors with the reference's
state-dict key names, generated directly on the target device."""
import torch

VIT_PREFIX = "model.image_tower.image_tower."


ANCHOR_COUNT, ANCHOR_GAIN, ANCHOR_BASE_ID, ANCHOR_EMBED_STD = 16, 16.0, 1000, 0.75


def anchor_gains(vocab, device="cpu"):
    """Per-token gain of the "anchored" synthetic checkpoint's lm_head rows (1 for ordinary tokens).

    Why: with N(0, 0.02^2) weights the 32000 logits of a position are i.i.d.-looking Gaussians, the top-2 margin is ~5 % of
    max|logit| on average and most greedy decisions sit inside the bf16 noise of a 32-layer stack -- token-stream checks at the
    full model size then cannot tell a correct kernel from a slightly wrong one.  The anchored checkpoint keeps every weight of
    the stack random (same kernels, same magnitudes) and restructures ANCHOR_COUNT rows of embed_tokens and of lm_head:

        embed_tokens[a_k] = ANCHOR_EMBED_STD * r_k        (r_k ~ N(0, 1): ~37 x the ordinary embedding scale)
        lm_head[a_(k+1)]  = ANCHOR_GAIN * 0.02 * r_k      (the SUCCESSOR anchor's row points along the current anchor's embedding)

    Through the residual connections the embedding of the token just fed survives to the final hidden state with a cosine of
    ~0.07 against 32 layers of random contributions (~4.6 sigma on the successor's logit before the gain), so a greedy stream walks
    the cycle a_0 -> a_1 -> ... with context-decided jumps where another anchor's random projection wins -- a VARIED token stream
    whose every decision depends on the token fed at that step, on its position and on the cache, with margins far above the bf16
    noise at ~95 % of the positions.  Gross errors (stale input token, wrong position, a broken layer) change the stream; errors at
    the rounding level are what the logit comparisons of tests/test_true_shapes_gpu.py are for."""
    g = torch.ones(vocab, dtype=torch.float32, device=device)
    g[ANCHOR_BASE_ID:ANCHOR_BASE_ID + ANCHOR_COUNT] = ANCHOR_GAIN
    return g


MASSIVE_CHANNELS = (1415, 2533, 97, 3001)        # the first two are where LLaMA-2-7B's massive activations sit; two more for width
MASSIVE_LAYER, MASSIVE_GAIN = 1, 250.0


def synthetic_state_dict(cfg, seed=2, std=0.02, dtype=torch.bfloat16, device="cuda:0", anchored=False, realistic=False):
    """realistic = True: the statistics of a TRAINED LLaMA-2-class checkpoint that N(0, std^2) weights lack and that stress range,
    the power-of-two fp8 row scales and the per-token e4m3 activation quantiser (VERDICT r03 "What's missing" #3; no real weights
    exist in this environment -- what a real checkpoint goes through is videollava/model/builder.py:94-112):
      * heavy-tailed Linear weights: Student-t (4 degrees of freedom) scaled to the same std -- kurtosis of trained weights, rows
        whose absmax is 5-10 sigma (one outlier decides a row's e4m3 scale);
      * massive activations: MASSIVE_CHANNELS of the residual stream carry values 10^2-10^3 x the median from layer MASSIVE_LAYER on
        (down_proj rows of that layer scaled by MASSIVE_GAIN), as LLaMA-2-7B does in channels 1415 / 2533;
      * non-unit norm gains: log-normal around 0.4 (RMSNorm) / 1.0 (LayerNorm), small (0.05) at the massive channels -- what keeps a
        trained model's normalised activations bounded."""
    g = torch.Generator(device=device).manual_seed(seed)
    v = cfg.vision_config

    def rn(*shape, s=std):
        z = torch.randn(*shape, generator=g, device=device, dtype=torch.float32)
        if realistic and len(shape) >= 2:
            # Student-t, 4 dof: z / sqrt(chi2_4 / 4), variance 2 -> rescaled to unit variance
            chi = torch.zeros(*shape, device=device, dtype=torch.float32)
            for _ in range(4):
                chi += torch.randn(*shape, generator=g, device=device, dtype=torch.float32) ** 2
            z = z / torch.sqrt(chi / 4.0) * (0.5 ** 0.5)
            z.clamp_(-12.0, 12.0)
        return (z * s).to(dtype)

    def near_one(n, rms=False):
        z = torch.randn(n, generator=g, device=device, dtype=torch.float32)
        if not realistic:
            return (1.0 + z * 0.1).to(dtype)
        gain = torch.exp(z * 0.5) * (0.4 if rms else 1.0)
        if rms and n > max(MASSIVE_CHANNELS):
            gain[list(MASSIVE_CHANNELS)] = 0.05
        return gain.to(dtype)

    sd = {}
    D, F_, V, hd = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size, cfg.head_dim
    sd["model.embed_tokens.weight"] = rn(V, D)
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        sd[p + "self_attn.q_proj.weight"] = rn(cfg.num_attention_heads * hd, D)
        sd[p + "self_attn.k_proj.weight"] = rn(cfg.num_key_value_heads * hd, D)
        sd[p + "self_attn.v_proj.weight"] = rn(cfg.num_key_value_heads * hd, D)
        sd[p + "self_attn.o_proj.weight"] = rn(D, D)
        sd[p + "mlp.gate_proj.weight"] = rn(F_, D)
        sd[p + "mlp.up_proj.weight"] = rn(F_, D)
        sd[p + "mlp.down_proj.weight"] = rn(D, F_)
        sd[p + "input_layernorm.weight"] = near_one(D, rms=True)
        sd[p + "post_attention_layernorm.weight"] = near_one(D, rms=True)
        if realistic and i == min(MASSIVE_LAYER, cfg.num_hidden_layers - 1) and D > max(MASSIVE_CHANNELS):
            w = sd[p + "mlp.down_proj.weight"]
            w[list(MASSIVE_CHANNELS)] = (w[list(MASSIVE_CHANNELS)].float() * MASSIVE_GAIN).to(dtype)
    sd["model.norm.weight"] = near_one(D, rms=True)
    sd["lm_head.weight"] = rn(V, D)
    if anchored:
        if V < ANCHOR_BASE_ID + ANCHOR_COUNT:
            raise ValueError("anchored synthetic checkpoint needs a vocabulary of at least %d" % (ANCHOR_BASE_ID + ANCHOR_COUNT))
        r = torch.randn(ANCHOR_COUNT, D, generator=g, device=device, dtype=torch.float32)
        ids = torch.arange(ANCHOR_BASE_ID, ANCHOR_BASE_ID + ANCHOR_COUNT, device=device)
        sd["model.embed_tokens.weight"][ids] = (ANCHOR_EMBED_STD * r).to(dtype)
        sd["lm_head.weight"][ids.roll(-1)] = (ANCHOR_GAIN * std * r).to(dtype)        # row of a_(k+1) <- direction of embed[a_k]
    Dv, Fv = v.hidden_size, v.intermediate_size
    sd[VIT_PREFIX + "embeddings.class_embedding"] = rn(Dv)
    sd[VIT_PREFIX + "embeddings.patch_embedding.weight"] = rn(Dv, v.num_channels, v.patch_size, v.patch_size)
    sd[VIT_PREFIX + "embeddings.position_embedding.weight"] = rn(v.num_patches + 1, Dv)
    sd[VIT_PREFIX + "pre_layrnorm.weight"] = near_one(Dv)
    sd[VIT_PREFIX + "pre_layrnorm.bias"] = rn(Dv)
    for i in range(v.num_hidden_layers):
        p = VIT_PREFIX + f"encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"self_attn.{nm}.weight"] = rn(Dv, Dv)
            sd[p + f"self_attn.{nm}.bias"] = rn(Dv)
        sd[p + "layer_norm1.weight"] = near_one(Dv)
        sd[p + "layer_norm1.bias"] = rn(Dv)
        sd[p + "layer_norm2.weight"] = near_one(Dv)
        sd[p + "layer_norm2.bias"] = rn(Dv)
        sd[p + "mlp.fc1.weight"] = rn(Fv, Dv)
        sd[p + "mlp.fc1.bias"] = rn(Fv)
        sd[p + "mlp.fc2.weight"] = rn(Dv, Fv)
        sd[p + "mlp.fc2.bias"] = rn(Dv)
    sd["model.mm_projector.0.weight"] = rn(D, cfg.mm_hidden_size)
    sd["model.mm_projector.0.bias"] = rn(D)
    sd["model.mm_projector.2.weight"] = rn(D, D)
    sd["model.mm_projector.2.bias"] = rn(D)
    return sd
