"""hipGraph replay modes of the decode step ("Graph replays must not be left outstanding ...").
Per-16-step device times of 256 replays of the captured decode step at C3 shapes, under different host-side command patterns,
then whole generate() calls at several chunk sizes.  Usage: python tools/chunk_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_inputs
from teochat_amd import _lib as L
from teochat_amd.builder import load_pretrained_model

dev = "cuda:0"
tok, model, _, _ = load_pretrained_model("synthetic:teochat-7b", None, "synthetic:teochat-7b", device=dev, dtype=torch.bfloat16, max_seq=2560)
eng = model.engine
frames, ids = synthetic_inputs(8, 128, model.config.vocab_size, seed=0, device=dev, dtype=torch.bfloat16)
sync = torch.cuda.synchronize


def prep():
    model.generate(input_ids=ids, images=frames, do_sample=False, max_new_tokens=2, eos_token_id=None)
    sync()


def burst(label, groups=16, per=16, between=None, drain_before_edge=False):
    """`groups` x `per` replays on the engine stream, an event after every group; then the cross-stream hand-over a phase ends with."""
    prep()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(groups + 1)]
    cur = torch.cuda.current_stream()
    eng.stream.wait_stream(cur)
    with torch.cuda.stream(eng.stream):
        st = eng.stream.cuda_stream
        evs[0].record()
        for i in range(groups):
            L.check(eng.lib.teo_graph_launch(eng._graph, per, st), "launch")
            evs[i + 1].record()
            if between == "edge":                                  # what leaving and re-entering an engine phase used to do
                cur.wait_stream(eng.stream); eng.stream.wait_stream(cur)
            elif between == "throttle" and i >= 2:
                evs[i - 1].synchronize()                           # <= 3 groups outstanding, the GPU never idles
            elif between == "streamsync":
                eng.stream.synchronize()
            elif between == "streamsync4" and i % 4 == 3:
                eng.stream.synchronize()
        if drain_before_edge:
            eng.stream.synchronize()
    cur.wait_stream(eng.stream)                                    # event record on the engine stream + wait on the caller's
    sync()
    print(f"{label:66s}", [round(evs[i].elapsed_time(evs[i + 1]) / per, 3) for i in range(groups)], flush=True)


burst("one burst of 256 replays, then the cross-stream edge")
burst("cross-stream edge after every 16 replays", between="edge")
burst("throttled: <= 3 groups outstanding (event sync), edge at the end", between="throttle")
burst("stream sync after every 16 replays", between="streamsync")
burst("stream sync after every 64 replays", between="streamsync4")
burst("one burst of 256 replays, stream drained BEFORE the edge", drain_before_edge=True)
for chunk in (256, 256, 128, 64, 16, 16):
    sync(); t = time.perf_counter()
    model.generate(input_ids=ids, images=frames, do_sample=False, max_new_tokens=256, eos_token_id=None, chunk=chunk)
    sync()
    print(f"generate(chunk={chunk}): {(time.perf_counter() - t) * 1e3:.1f} ms", flush=True)
