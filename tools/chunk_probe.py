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
for chunk in (256, 16, 16, 32, 64):
    torch.cuda.synchronize(); t = time.perf_counter()
    model.generate(input_ids=ids, images=frames, do_sample=False, max_new_tokens=256, eos_token_id=None, chunk=chunk)
    torch.cuda.synchronize()
    print(f"generate chunk={chunk}: {(time.perf_counter() - t) * 1e3:.1f} ms", flush=True)
# pieces
model.generate(input_ids=ids, images=frames, do_sample=False, max_new_tokens=2, eos_token_id=None)
sync = torch.cuda.synchronize
for rep in range(3):
    sync(); t0 = time.perf_counter()
    eng.decode_steps(16, use_graph=True)
    t1 = time.perf_counter()
    sync(); t2 = time.perf_counter()
    got = eng.generated()
    t3 = time.perf_counter()
    lst = got.tolist()
    t4 = time.perf_counter()
    print(f"decode_steps(16) enqueue {1e3 * (t1 - t0):.2f} ms, wait {1e3 * (t2 - t1):.2f} ms, generated() {1e3 * (t3 - t2):.3f} ms, tolist {1e3 * (t4 - t3):.3f} ms", flush=True)
for n in (1, 1, 2, 4, 8, 32):
    sync(); t0 = time.perf_counter()
    eng.decode_steps(n, use_graph=True)
    t1 = time.perf_counter()
    sync(); t2 = time.perf_counter()
    print(f"decode_steps({n}) from idle: enqueue {1e3 * (t1 - t0):.2f} ms, total {1e3 * (t2 - t0):.2f} ms = {1e3 * (t2 - t0) / n:.3f} ms/step", flush=True)
# per-step device timestamps inside one chunk
evs = [torch.cuda.Event(enable_timing=True) for _ in range(17)]
sync()
evs[0].record()
for i in range(16):
    eng.decode_steps(1, use_graph=True)
    evs[i + 1].record()
sync()
print("per-step device ms inside a chunk from idle:", [round(evs[i].elapsed_time(evs[i + 1]), 3) for i in range(16)], flush=True)
