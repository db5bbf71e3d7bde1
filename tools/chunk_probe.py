"""Where does the product-default chunk=16 loop lose 5 % against chunk=n_out?  Times the pieces of one chunk."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_inputs
from teochat_amd.builder import load_pretrained_model

dev = "cuda:0"
tok, model, _, _ = load_pretrained_model("synthetic:teochat-7b", None, "synthetic:teochat-7b", device=dev, dtype=torch.bfloat16, max_seq=2560)
eng = model.engine
frames, ids = synthetic_inputs(8, 128, model.config.vocab_size, seed=0, device=dev, dtype=torch.bfloat16)
for chunk in (256, 256, 16, 16, 64, 256, 128, 255):
    torch.cuda.synchronize(); t = time.perf_counter()
    model.generate(input_ids=ids, images=frames, do_sample=False, max_new_tokens=256, eos_token_id=None, chunk=chunk)
    torch.cuda.synchronize()
    print(f"generate chunk={chunk}: {(time.perf_counter() - t) * 1e3:.1f} ms", flush=True)
import time as _t
from teochat_amd import _lib as L
sync = torch.cuda.synchronize
def prep():
    model.generate(input_ids=ids, images=frames, do_sample=False, max_new_tokens=2, eos_token_id=None)
    sync()
def burst(label, groups=16, per=16, between=None, before=None):
    prep()
    if before:
        before()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(groups + 1)]
    ph = eng.phase(); st = ph.__enter__()
    evs[0].record()
    for i in range(groups):
        L.check(eng.lib.teo_graph_launch(eng._graph, per, st), "launch")
        evs[i + 1].record()
        if between == "phase":
            ph.__exit__(None, None, None); ph = eng.phase(); st = ph.__enter__()
        elif between == "throttle" and i >= 2:
            evs[i - 1].synchronize()                    # at most ~3 groups outstanding, the GPU never idles
        elif between == "sync":
            sync()
        elif between == "streamsync":
            torch.cuda.current_stream().synchronize()
        elif between == "eventsync":
            evs[i + 1].synchronize()
        elif between == "sync+phase":
            sync(); ph.__exit__(None, None, None); ph = eng.phase(); st = ph.__enter__()
        elif between == "streamsync_every4" and i % 4 == 3:
            torch.cuda.current_stream().synchronize()
    if "stream-sync before leaving" in label:
        torch.cuda.current_stream().synchronize()
    ph.__exit__(None, None, None)
    sync()
    print(label, [round(evs[i].elapsed_time(evs[i + 1]) / per, 3) for i in range(groups)], flush=True)

def regraph():
    eng._drop_graph()
    eng.decode_steps(1)
    sync()
for n in (256, 128, 32):
    prep()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with eng.phase() as st:
        e0.record()
        L.check(eng.lib.teo_graph_launch(eng._graph, n, st), "launch")
        e1.record()
        torch.cuda.current_stream().synchronize()
    sync()
    print(f"Y1 one call of {n} launches, stream-sync before leaving the phase:", round(e0.elapsed_time(e1) / n, 3), "ms/step", flush=True)
burst("Y2 one burst 16x16, stream-sync before leaving the phase", between=None)
