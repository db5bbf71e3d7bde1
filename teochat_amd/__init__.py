"""teochat_amd: MI355X-native implementation of TEOChat's temporal-image -> LLM forward path.

Host side mirrors the reference's videollava API for that path (load_model, run_inference_single,
LlavaLlamaForCausalLM.forward/generate/prepare_inputs_labels_for_multimodal, mm_utils token packing); the
arithmetic is hand-written HIP for gfx950 in libteo_hip.so behind the C ABI of include/teo_hip.h.
"""
__version__ = "0.1.0"
