/*
 * qtts_tuning.h — entry points that exist only in a -DQTTS_TUNING build of the library (csrc/build.py --tuning ->
 * tools/libqtts_hip_tuning.so), used by tools/ for A/B measurements.  None of this is part of the shipped C ABI
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

#ifdef __cplusplus
}
#endif
#endif /* QTTS_TUNING_H */
