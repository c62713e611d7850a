// qtts_common.h - what the translation units of libqtts_hip share: error reporting, the bf16 / lane helpers every
// kernel uses, the activation layouts, the dispatch thresholds, the GEMM launcher's argument record and the few
// host functions one unit calls in another (gemm: qtts_gemm.hip, attention: qtts_attention.hip, sampling and the
// per-frame glue: qtts_glue.hip, the decoder stack and graphs: qtts_model.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <mutex>
#include <type_traits>
#include <vector>

#include "qtts_hip.h"

typedef uint16_t bf16_t;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));

extern thread_local char g_err[512];
#define QFAIL(code, ...) do { snprintf(g_err, sizeof(g_err), __VA_ARGS__); return (code); } while (0)
#define QCHECK(cond, ...) do { if (!(cond)) QFAIL(QTTS_EINVAL, __VA_ARGS__); } while (0)
#define QLAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { \
    snprintf(g_err, sizeof(g_err), "%s: %s", __func__, hipGetErrorString(e_)); return (int)e_; } } while (0)

// Kernels that ask for more dynamic LDS than the default limit: the limit is raised once per (kernel, device).
// Returns false (with the error text set) when the device refuses.
bool allow_dynamic_lds(const void *kernel, size_t bytes);

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;                       // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }   // round through bf16
// x * sigmoid(x) on the hardware exp2 and reciprocal (a few fp32 ulp, far inside the bf16 rounding that follows; the
// library expf and IEEE division cost ten times the instructions on the serial path of a GEMM epilogue).
// exp overflow gives 1/inf = 0, i.e. -0 for very negative x, as the exact quotient does.
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// Cross-lane sums inside a 16-lane row on the DPP path (no LDS crossbar round trip).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f32<0xB1>(v);      // quad_perm [1,0,3,2]   (lane ^ 1)
    v += dpp_f32<0x4E>(v);      // quad_perm [2,3,0,1]   (lane ^ 2)
    v += dpp_f32<0x141>(v);     // row_half_mirror: the other quad of the 8-lane half
    v += dpp_f32<0x140>(v);     // row_mirror: the other half of the 16-lane row
    return v;
}
__device__ __forceinline__ float row16_xor8(float v) { return dpp_f32<0x128>(v); }   // row_ror:8 == lane ^ 8

__device__ __forceinline__ uint4 dpp_u4_ror8(uint4 v) {      // every dword from lane ^ 8 of the 16-lane row
    return make_uint4(__builtin_amdgcn_update_dpp(0, v.x, 0x128, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, v.y, 0x128, 0xf, 0xf, true),
                      __builtin_amdgcn_update_dpp(0, v.z, 0x128, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, v.w, 0x128, 0xf, 0xf, true));
}

union Frag {
    uint4 u;
    u32x4_t n;
    bf16x8_t v;
    bf16_t h[8];
};

__device__ __forceinline__ void unpack2(uint32_t w, float &a, float &b) {
    a = __uint_as_float(w << 16);
    b = __uint_as_float(w & 0xffff0000u);
}
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    return (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
}

// RMSNorm of two neighbouring elements (one dword of bf16 pairs): bf16(bf16(x * rstd) * w), the reference's rounding
// points (P/layers/layernorm.py:17-27), on the packed fp32 multiplier and the packed fp32 -> bf16 conversion (half the
// instructions of the element-by-element form, same bits).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2v(float2_t v) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));       // v_cvt_pk_bf16_f32: RNE
}
__device__ __forceinline__ float2_t unpack2v(uint32_t w) {
    return float2_t{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
}
__device__ __forceinline__ uint32_t norm_pair(uint32_t xw, float rstd, float2_t w2) {
    const float2_t v = unpack2v(xw) * float2_t{rstd, rstd};
    return pack2v(unpack2v(pack2v(v)) * w2);
}
__device__ __forceinline__ uint4 norm_frag(uint4 x, float rstd, uint4 w) {
    return make_uint4(norm_pair(x.x, rstd, unpack2v(w.x)), norm_pair(x.y, rstd, unpack2v(w.y)),
                      norm_pair(x.z, rstd, unpack2v(w.z)), norm_pair(x.w, rstd, unpack2v(w.w)));
}

// Where element (row m, column k) of an activation matrix lives.  Row-major (pks == 0): m * ld + k.  Fragment order
// (pks = K / 32 > 0): [row tile m / 16][k-step k / 32][lane = 16 * ((k % 32) / 8) + m % 16][k % 8] - the order in which
// the MFMA B operand of a 16-row tile is consumed, so that the prompt GEMM reads one contiguous KiB per request (as
// it does for the packed weights) instead of sixty-four bytes from each of sixteen rows.  Used between the kernels of
// a prompt's layer only (normalised rows, attention output, SwiGLU output); buffers hold whole 16-row tiles.
__device__ __forceinline__ size_t act_index(int m, int k, int ld, int pks) {
    if (pks == 0) return (size_t)m * ld + k;
    return ((((size_t)(m >> 4) * pks + (k >> 5)) * 64) + (((k & 31) >> 3) << 4) + (m & 15)) * 8 + (k & 7);
}

// Dispatch thresholds.  The shipped library holds them as constants; a -DQTTS_TUNING build (tools/ only, see
// csrc/build.py --tuning) turns them into process-wide variables with setters for A/B measurements (qtts_gemm.hip).
#ifdef QTTS_TUNING
extern int g_tune_ksw, g_tune_t, g_dbg_linear, g_rows_min, g_share_prologue, g_sub_min_bytes, g_mfma_attn_rows, g_rows_tm,
           g_attn_wide, g_rows_tm1, g_rows_norm, g_rows_norm_max;
extern long long *g_trace;                 // timestamp buffer of traced GEMM launches
extern int g_trace_launch;
#else
static constexpr int g_tune_ksw = 0, g_tune_t = 0, g_dbg_linear = 0;
static constexpr int g_rows_min = 16;              // most rows that still take single-tile workgroups instead of the prompt GEMM
static constexpr int g_share_prologue = 1;         // norm prologue shared between the lane halves for up to 8 rows
static constexpr int g_sub_min_bytes = 32 << 10;   // smallest weight slice per workgroup when a tile is split over workgroups
static constexpr int g_mfma_attn_rows = 112;       // rows from which qtts_paged_attention takes the MFMA kernel for prompts
static constexpr int g_rows_tm = 0;                // row tiles per workgroup of the prompt GEMM (0 = chosen per shape)
static constexpr int g_attn_wide = 0;              // decode attention: workgroup size chosen per shape
static constexpr int g_rows_tm1 = 2;               // prompt GEMMs of up to four row tiles may take one row tile per workgroup
static constexpr int g_rows_norm = 1;              // prompt GEMMs normalise their rows themselves
static constexpr int g_rows_norm_max = 1 << 30;    // most rows for which they do
#endif

// ------------------------------------------------------------------ linear (skinny GEMM on MFMA)
// One kernel shape for every projection.  Workgroup = NW <= 8 waves, wave w owns
// k-steps [w*KSW, (w+1)*KSW); the workgroup walks `tiles_per_wg` consecutive 16-row weight tiles, the
// next tile's fragments already in flight (double buffer) while the current tile is multiplied, reduced
// across waves through LDS and written by wave 0.  Activation prologues (PRO):
//   0  x as is (optionally gathered rows: x row m = x[ids[m]])
//   1  RMSNorm(x) * norm_w, the reference's rounding points; blockIdx.x == 0 may write the normalised rows
//   2  x holds [gate | up] (2K columns): silu(gate) -> bf16, * up -> bf16 (the MLP activation, moved from the
//      producing GEMM's epilogue into the consuming GEMM's prologue so that every GEMM is plain tiles)
struct LinArgs {
    const bf16_t *x; int ldx;
    const int64_t *ids; int id_rows;
    const uint4 *wp;
    const bf16_t *bias, *res; int ldr;
    const bf16_t *nw; float eps;
    bf16_t *xnorm_out; int ldxn;
    bf16_t *out; int ldo;
    int M, N, K, tiles_per_wg;
    const bf16_t *next_nw = nullptr; bf16_t *next_xn = nullptr; int ldnn = 0;   // K-split prompt GEMM only: also RMSNorm(out) * next_nw
    const float *wscale = nullptr;   // non-null: wp holds fp8 (e4m3) weights, two k-steps per 16-byte fragment; scale per packed row
    const int64_t *res_ids = nullptr; int res_id_rows = 0;   // non-null: residual row m = res[res_ids[m]] (decode GEMM only)
    // prompt GEMMs between the kernels of one layer: x / the SwiGLU output / the fused norm's output in fragment
    // order (act_index) instead of row-major
    int xpacked = 0, opks = 0, nnpks = 0;
    // prompt GEMMs, plain epilogue: the residual rows / `out` in fragment order (k-steps = N / 32); ssq_out: the stored
    // rows' sums of squares per 16 columns, [M][N / 16]
    int rpks = 0, opks0 = 0;
    float *ssq_out = nullptr;
    // prompt GEMMs, prologue 3: RMSNorm(x) * nw with the rows' sums of squares taken from ssq_in [M][K / 16]
    const float *ssq_in = nullptr;
    int dbg;      // bit0: tuning experiment (every tile reads weight tile 0); bit3: `ids` select the residual rows, not the
                  // x rows; bits 4-5: log2 of the workgroups per tile; bit 7: x in fragment order (one row tile);
                  // bits 8+: row-tile count of a prefill launch
};

// Measurement hook (qtts_model_set_layer_events): when set, the next decode GEMM launch of this thread carries the
// two events as the kernel's start / stop events.
struct KernelEvents { hipEvent_t start, stop; };
extern thread_local KernelEvents g_kernel_events;
#define QTTS_LAUNCH_MAYBE_TIMED(kern, grid, block, shmem, s, ...) do { \
        if (g_kernel_events.start) { \
            hipExtLaunchKernelGGL(kern, grid, block, (uint32_t)(shmem), s, g_kernel_events.start, g_kernel_events.stop, 0, __VA_ARGS__); \
            g_kernel_events = {nullptr, nullptr}; \
        } else { \
            hipLaunchKernelGGL(kern, grid, block, shmem, s, __VA_ARGS__); \
        } } while (0)

// ---- host functions of one unit that another calls
// qtts_gemm.hip
int launch_linear2(LinArgs a, int pro, hipStream_t s, bool silu_pairs = false, float *scratch = nullptr, size_t scratch_bytes = 0);
size_t linear_rows_scratch_bytes(int M, int N, int K);
bool linear_rows_is_ksplit(int M, int N, int K, size_t scratch_bytes);
bool linear_rows_covers(int M, int N, int K, bool silu_pairs, bool has_bias, bool has_scratch, size_t scratch_bytes);
bool linear_rows_norm_covers(int K);
int rmsnorm_impl(const void *x, int ldx, const void *w, float eps, void *y, int ldy, int M, int H, int ypks, qtts_stream_t stream,
                 int xpks = 0);
// qtts_attention.hip
int paged_attention_impl(const void *q, int ldq, const void *k_cache, const void *v_cache, const int32_t *block_tables,
                         int max_blocks, const int32_t *row_seq, const int32_t *context_lens, int max_context, float scale,
                         void *out, int ldo, int N, int Hq, int Hkv, void *workspace, qtts_stream_t stream, int opks);
int attention_decode_fused_impl(const void *qkv, int ldqkv, const void *q_norm_w, const void *k_norm_w, float eps,
                                const void *rope_cos, const void *rope_sin, int max_pos, const int64_t *positions,
                                const int32_t *slot_mapping, void *k_cache, void *v_cache, const int32_t *block_tables,
                                int max_blocks, const int32_t *context_lens, int max_context, float scale, void *out, int ldo,
                                int B, int Hq, int Hkv, void *partials, int *counters, hipStream_t s,
                                const int64_t *row_ids = nullptr, int id_rows = 0, int opks = 0);
int attention_pairs_impl(const void *qkv, int ldqkv, const void *q_norm_w, const void *k_norm_w, float eps, const void *rope_cos,
                         const void *rope_sin, int max_pos, const int64_t *positions, const int32_t *slot_mapping,
                         void *k_cache, void *v_cache, float scale, void *out, int ldo, int N, int Hq, int Hkv, hipStream_t s,
                         int opks = 0);
size_t attn2_partial_bytes(int B, int Hq, int max_context);
