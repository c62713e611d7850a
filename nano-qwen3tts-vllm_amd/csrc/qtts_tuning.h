/*
 * qtts_tuning.h — entry points that exist only in a -DQTTS_TUNING build of the library (csrc/build.py --tuning ->
 * libqtts_hip_tuning.so), used by tools/ for A/B measurements.  None of this is part of the shipped C ABI
 * (include/qtts_hip.h): the setters change kernel dispatch process-wide and are not thread-safe.
 */
#ifndef QTTS_TUNING_H
#define QTTS_TUNING_H

#include "qtts_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

void qtts_tune_linear(int ksw, int tiles_per_cu);   /* k-steps per wave / workgroups per CU of the decode GEMM; 0 = automatic */
void qtts_tune_subtiles(int min_bytes);             /* smallest weight slice of a workgroup that shares a tile */
void qtts_tune_rows(int min_rows);                  /* most rows that still take single-tile workgroups */
void qtts_tune_rows_tm(int tm);                     /* row tiles per workgroup of the prompt GEMM (2, 3, 4); 0 = per shape */
void qtts_tune_shared_prologue(int on);             /* lane halves share the activation prologue (<= 8 rows) */
void qtts_tune_prompt_attention(int min_rows);      /* rows from which prompts take the MFMA attention kernel */
void qtts_debug_linear(int flags);                  /* bit 0: every tile reads weight tile 0; bit 1: no prompt GEMM; bit 5: no fused norm */
void qtts_debug_linear_trace(long long *buf);       /* device buffer of per-workgroup timestamps, or NULL */
void qtts_tune_attn_wide(int mode);                 /* decode attention: 0 per shape, 1 = 128-token workgroups, 2 = 256-token ones */
void qtts_tune_rows_tm1(int on);                    /* narrow prompt GEMMs may take one row tile per workgroup */
void qtts_tune_rows_norm(int on, int max_rows);     /* prompt GEMMs normalise their rows from the producer's partial sums (up to max_rows rows) */
int qtts_debug_linear_rows(const void *x, int ldx, int xpacked, const float *ssq_in, const void *norm_w, float eps,
                           const void *w_packed, const void *residual, int ldr, int rpks, void *out, int ldo, int opks0,
                           float *ssq_out, int M, int N, int K, int swiglu, qtts_stream_t stream);   /* one prompt GEMM, any form */
int qtts_debug_grid_barrier(int *mem, int iters, int nwg, int threads, int mode, int gsize, int sleep, qtts_stream_t stream);

/* ---------------------------------------------------------------- resident burst
 * The code predictor's whole burst for up to 8 sequences — the 2-position prefill and
 * n_gen-1 decode steps with head, pick and token feedback
 * (P/engine/model_runner/predictor_model_runner.py:134-157 driven 15 times by
 * P/workers/predictor_worker.py:128-134) — as ONE kernel that stays on the chip and walks
 * a table of stages with a device-wide barrier between them, instead of one launch per
 * stage.  Same arithmetic and rounding points as the per-stage entry points above.  (On
 * MI355X the graph of per-stage launches measured faster; the host keeps this opt-in.)
 *   qtts_burst_program_bytes : device bytes for the stage table + barrier words
 *   qtts_burst_build         : fills the table (synchronous upload; call outside graph
 *                              capture, once per batch size / buffer set).  `inputs` are the
 *                              2*bs prompt rows [talker_hidden] (pairs per sequence); rows of
 *                              positions / slot_mapping / context_lens are laid out as 2*bs
 *                              prefill rows followed by n_gen-1 groups of bs decode rows;
 *                              emb_tables / heads_packed are HOST arrays of n_gen device
 *                              pointers; tokens_out[step*bs + i] receives the ids.
 *   qtts_burst_launch        : clears the barrier words and launches n_workgroups (<= CU
 *                              count: they must all be on the chip at once) x 512 threads.
 *                              barrier_group >= n_workgroups: one flat arrival counter.
 *   qtts_burst_status        : 0 after a burst that ran to its end, 1 if a workgroup gave up
 *                              waiting at a barrier (synchronous read).
 * Limits: bs <= 8, q/kv head ratio 1 or 2, K of every GEMM <= 3072 and hidden <= 2048. */
size_t qtts_burst_program_bytes(const qtts_model_t *m, int n_gen, int has_projection);
int qtts_burst_build(qtts_model_t *m, void *program, size_t program_bytes, int bs, const void *inputs,
                     int talker_hidden, const void *proj_w_packed, const void *proj_bias, void *x_buf,
                     const void *const *emb_tables, int emb_rows, const void *const *heads_packed, int n_gen,
                     int vocab, const int64_t *positions, const int32_t *slot_mapping,
                     const int32_t *context_lens, const int32_t *block_tables, int max_blocks, void *logits,
                     int64_t *tokens_out);
int qtts_burst_launch(qtts_model_t *m, void *program, int n_gen, int has_projection, int n_workgroups,
                      int barrier_group, const float *temperatures, const uint8_t *greedy_flags, int top_k,
                      uint64_t seed, const uint64_t *rng_offset, int max_blocks, qtts_stream_t stream);
int qtts_burst_status(const qtts_model_t *m, const void *program, int n_gen, int has_projection);

#ifdef __cplusplus
}
#endif
#endif /* QTTS_TUNING_H */
