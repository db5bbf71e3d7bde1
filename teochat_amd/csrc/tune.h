// Performance knobs of libteo_hip as ONE plain struct (teo_tune): no knob is process-wide state.  A block belongs to whoever created it
// (teo_tune_create, or the block inside a teo_ctx); the block in effect for a call is the descriptor's (`tune` field of teo_vit_desc /
// teo_proj_desc / teo_llama_desc) when that is set, else the block the calling thread bound with teo_tune_bind, else the built-in
// defaults below (= what ships).  Every key selects among kernels / geometries that compute the same values (see include/teo_hip.h).
#pragma once

// X(key / field name, default, validity of a candidate value `v`)
#define TEO_TUNE_KEYS(X)                                                                                     \
    X(gemv_variant, -1, true)                                                                                \
    X(gemv_nt, 1, true)                                                                                      \
    X(gemv_max_blocks, 1024, v > 0)                                                                          \
    X(gemv_small_k, 1, true)                                                                                 \
    X(gemv_splitk_u, 0, (v >= 0 && v <= 6 && v != 5))                                                        \
    X(gemv_splitk_r, 0, (v == 0 || v == 2 || v == 4))                                                        \
    X(gemm_bm, 0, (v == 0 || v == 64 || v == 128))                                                           \
    X(gemm_depth, 0, true)                                                                                   \
    X(gemm_sk, 1, (v >= 0 && v <= 2))                                                                        \
    X(gemm_wide, 1, (v >= 0 && v <= 2))                                                                      \
    X(gemm_wide_sched, 1, true)                                                                              \
    X(gemm_wide_group, 0, v >= 0)                                                                            \
    X(gemm_big, 1, (v >= 0 && v <= 2))                                                                       \
    X(gemm_big_group, 0, v >= 0)                                                                             \
    X(gemm_big_hybrid, 1, (v >= 0 && v <= 2))                                                                \
    X(gemm_big_cohort, -1, (v == -1 || v == 0 || v == 8 || v == 16 || v == 32))                              \
    X(gemm_big_ragged, 1, (v >= 0 && v <= 2))                                                                \
    X(gemm_narrow, 1, (v >= 0 && v <= 2))                                                                    \
    X(gemm_narrow_bm, 0, (v == 0 || v == 64 || v == 128))                                                                \
    X(gemm_narrow_waves, 0, (v == 0 || v == 4 || v == 8))                                                    \
    X(gemm_narrow_pipe, 1, (v >= 0 && v <= 2))                                                               \
    X(gemm_pipe_stages, 0, (v == 0 || v == 3 || v == 4))                                                    \
    X(rope_vt_fused, 1, (v == 0 || v == 1))                                                                  \
    X(gemm_pipe_bn, 0, (v == 0 || v == 64 || v == 96 || v == 128))                                                 \
    X(gemm_quad, 1, (v >= 0 && v <= 2))                                                                      \
    X(gemm_quad_waves, 8, (v == 4 || v == 8))                                                                \
    X(gemm_fp8_wide, 1, (v >= 0 && v <= 3))                                                                  \
    X(gemm_fp8_big, 1, (v >= 0 && v <= 2))                                                                   \
    X(flash_order, 1, (v >= 0 && v <= 1))                                                                    \
    X(flash_pipe, -1, (v >= -1 && v <= 1))                                                                   \
    X(attn_chunk, 0, (v == 0 || v == 32 || v == 64 || v == 128 || v == 256))                                 \
    X(attn_whole, 1, (v >= 0 && v <= 2))                                                                     \
    X(skinny_tiles, 0, (v == 0 || v == 1 || v == 2 || v == 4 || v == 8))                                     \
    X(skinny_nt, 1, true)                                                                                    \
    X(skinny_stream, 1, (v >= 0 && v <= 2))                                                                  \
    X(skinny_ring, 0, (v == 0 || v == 1))                                                                    \
    X(skinny_unr, 0, (v == 0 || v == 4 || v == 8))                                                           \
    X(skinny_waves, 0, (v == 0 || v == 8 || v == 16))                                                        \
    X(skinny_grid, 0, (v >= 0 && v <= 3))

struct teo_tune {
#define TEO_TUNE_FIELD(name, def, ok) int name = def;
    TEO_TUNE_KEYS(TEO_TUNE_FIELD)
#undef TEO_TUNE_FIELD
};

namespace teo {
// the block in effect on this thread (abi.hip)
const teo_tune& tune();
// descriptor-driven entry points put the descriptor's block in effect for their duration (nullptr: leave the thread's choice)
struct TuneScope {
    const teo_tune* prev;
    bool active;
    explicit TuneScope(const teo_tune* t);
    ~TuneScope();
    TuneScope(const TuneScope&) = delete;
    TuneScope& operator=(const TuneScope&) = delete;
};
}  // namespace teo
