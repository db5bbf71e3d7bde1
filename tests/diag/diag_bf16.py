"""Where does the bf16 path diverge from the boundary-rounding oracle?  Stage-by-stage max|diff|/max|ref|."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import teo_oracle as O
from tests import _tiny as TY
from tests.test_model_gpu import build, inputs

bf = torch.bfloat16
R = lambda t: t.to(bf).float()


def rel(a, b):
    return float((a.float().cpu() - b).abs().max()) / float(b.abs().max())


for name in ("tinyA", "tinyB"):
    g = TY.load_npz(name)
    model, sd = build(name, bf)
    eng = model.engine
    frames, ids = inputs(name, g)
    vcfg, lcfg, mm = TY.cfgs(name)
    sd16 = {k: v.to(bf).float() for k, v in sd.items()}
    pix = torch.stack(frames)
    # ViT stage by stage: run the engine with fewer layers by editing layers_run
    for nl in range(0, eng.vit_layers_run + 1):
        eng.vit_desc.layers_run = nl
        got = eng.vit_features(pix.cuda().to(bf))
        st = O.vit_hidden_states(pix, sd16, vcfg, rounding="bf16", n_layers=nl)[-1][:, 1:]
        st32 = O.vit_hidden_states(pix, sd16, vcfg, rounding=None, n_layers=nl)[-1][:, 1:]
        print(f"[{name}] vit after {nl} layers: vs boundary oracle {rel(got, st):.2e}   (oracle bf16 vs its own fp32: {rel(st, st32):.2e})")
    eng.vit_desc.layers_run = eng.vit_layers_run
    feats_o = O.vit_features(pix, sd16, vcfg, -2, "patch", rounding="bf16")
    proj = eng.project(feats_o.cuda().to(bf))
    proj_o = O.projector(feats_o, sd16, mm.mm_projector_type, rounding="bf16")
    print(f"[{name}] projector on oracle features: {rel(proj, proj_o):.2e}")
    # LLaMA alone on oracle embeddings
    flat = [proj_o[i] for i in range(proj_o.shape[0])]
    _, pos, mask, _, emb, _ = O.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, flat, sd16["model.embed_tokens.weight"], mm)
    eng.reset_cache()
    lg = eng.prefill(R(emb[0]).cuda().to(bf))
    lo, _ = O.llama_forward(R(emb), None, None, None, sd16, lcfg, rounding="bf16")
    lo32, _ = O.llama_forward(R(emb), None, None, None, sd16, lcfg, rounding=None)
    print(f"[{name}] llama logits on oracle embeddings: {rel(lg, lo[0]):.2e}   (oracle bf16 vs fp32: {rel(lo[0], lo32[0]):.2e})")
    d = (lg.float().cpu() - lo[0]).abs()
    print(f"[{name}]   rows with diff > 1e-3*max: {int((d.max(-1).values > 1e-3 * float(lo.abs().max())).sum())} / {d.shape[0]}; median row max diff {float(d.max(-1).values.median()):.2e}")
