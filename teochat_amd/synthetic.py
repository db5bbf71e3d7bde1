"""Synthetic checkpoints (there is no network for real weights): N(0, std^2) tensors with the reference's
state-dict key names, generated directly on the target device."""
import torch

VIT_PREFIX = "model.image_tower.image_tower."


def synthetic_state_dict(cfg, seed=2, std=0.02, dtype=torch.bfloat16, device="cuda:0"):
    g = torch.Generator(device=device).manual_seed(seed)
    v = cfg.vision_config

    def rn(*shape, s=std):
        return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * s).to(dtype)

    def near_one(n):
        return (1.0 + torch.randn(n, generator=g, device=device, dtype=torch.float32) * 0.1).to(dtype)

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
        sd[p + "input_layernorm.weight"] = near_one(D)
        sd[p + "post_attention_layernorm.weight"] = near_one(D)
    sd["model.norm.weight"] = near_one(D)
    sd["lm_head.weight"] = rn(V, D)
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
