// libqtts_hip — MI355X (gfx950 / CDNA4) kernels for the Qwen3-TTS talker and
// code-predictor decode path.  C ABI in include/qtts_hip.h.
//
// Design notes
//  * Decode is weight streaming: every linear layer reads its weight once per
//    step.  Weights are re-laid at load time into MFMA fragment order so that a
//    wave-instruction reads 1 KiB contiguous and feeds
//    v_mfma_f32_16x16x32_bf16 directly (A = 16 weight rows, B = up to 16 token
//    rows, fp32 accumulate).  K is split over the waves of a workgroup and
//    reduced through LDS in a fixed order, so results are bitwise reproducible.
//  * RMSNorm is folded into the consuming GEMM's prologue, SwiGLU and the
//    residual add into the producing GEMM's epilogue, with the reference's
//    bf16 rounding points kept (see qtts_hip.h).
//  * Attention over the paged KV cache is HBM/L2 bound with two query heads
//    per KV head: K/V rows go straight to VGPRs (16 lanes x 16 B per token),
//    fp32 online softmax per 16-lane group, 128-token splits merged by a
//    combine pass.
//  * Nothing here allocates or synchronises; everything is graph-capturable.
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

static thread_local char g_err[512] = "";
#define QFAIL(code, ...) do { snprintf(g_err, sizeof(g_err), __VA_ARGS__); return (code); } while (0)
#define QCHECK(cond, ...) do { if (!(cond)) QFAIL(QTTS_EINVAL, __VA_ARGS__); } while (0)
#define QLAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { \
    snprintf(g_err, sizeof(g_err), "%s: %s", __func__, hipGetErrorString(e_)); return (int)e_; } } while (0)

// Kernels that ask for more dynamic LDS than the default limit: the limit is raised once per (kernel, device).
// Returns false (with the error text set) when the device refuses.
static bool allow_dynamic_lds(const void *kernel, size_t bytes) {
    struct Seen { const void *k; int dev; };
    static thread_local std::vector<Seen> seen;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    for (const Seen &e : seen)
        if (e.k == kernel && e.dev == dev) return true;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "cannot raise the dynamic LDS limit to %zu bytes: %s", bytes, hipGetErrorString(e));
        return false;
    }
    seen.push_back({kernel, dev});
    return true;
}

extern "C" int qtts_version(void) { return 100; }
extern "C" const char *qtts_last_error(void) { return g_err; }

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

// ------------------------------------------------------------------ weight packing
__global__ void pack_weight_kernel(const bf16_t *__restrict__ w, uint4 *__restrict__ packed,
                                   int N, int K, int tile_stride, int tile_offset) {
    // one thread per (nt, ks, lane)
    const int KS = K >> 5;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)(N >> 4) * KS * 64;
    if (idx >= total) return;
    int lane = idx & 63;
    long f = idx >> 6;
    int ks = f % KS;
    int nt = f / KS;
    int r = lane & 15, g = lane >> 4;
    const uint4 *src = reinterpret_cast<const uint4 *>(w + (size_t)(16 * nt + r) * K + 32 * ks + 8 * g);
    size_t dst_tile = (size_t)nt * tile_stride + tile_offset;
    packed[(dst_tile * KS + ks) * 64 + lane] = *src;
}

extern "C" size_t qtts_packed_weight_bytes(int N, int K) { return (size_t)N * K * 2; }

extern "C" int qtts_pack_weight(const void *w, void *packed, int N, int K, qtts_stream_t stream) {
    QCHECK(w && packed, "pack_weight: null pointer");
    QCHECK(N > 0 && K > 0 && N % 16 == 0 && K % 32 == 0, "pack_weight: need N%%16==0, K%%32==0 (N=%d K=%d)", N, K);
    long total = (long)(N / 16) * (K / 32) * 64;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t *)w, (uint4 *)packed, N, K, 1, 0);
    QLAUNCH();
    return QTTS_OK;
}

// Tile t of the packed gate/up matrix holds gate rows 8t..8t+7 in its rows 0..7 and up rows 8t..8t+7 in
// its rows 8..15, so that every 16-row MFMA tile carries both operands of 8 SwiGLU outputs.
__global__ void pack_gate_up_kernel(const bf16_t *__restrict__ gate, const bf16_t *__restrict__ up,
                                    uint4 *__restrict__ packed, int I, int K) {
    const int KS = K >> 5;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)(I >> 3) * KS * 64;
    if (idx >= total) return;
    int lane = idx & 63;
    long f = idx >> 6;
    int ks = f % KS;
    int t = f / KS;
    int r = lane & 15, g = lane >> 4;
    const bf16_t *src = (r < 8 ? gate : up) + (size_t)(8 * t + (r & 7)) * K + 32 * ks + 8 * g;
    packed[((size_t)t * KS + ks) * 64 + lane] = *reinterpret_cast<const uint4 *>(src);
}

extern "C" int qtts_pack_gate_up(const void *gate, const void *up, void *packed, int I, int K, qtts_stream_t stream) {
    QCHECK(gate && up && packed, "pack_gate_up: null pointer");
    QCHECK(I > 0 && K > 0 && I % 8 == 0 && K % 32 == 0, "pack_gate_up: need I%%8==0, K%%32==0 (I=%d K=%d)", I, K);
    long total = (long)(I / 8) * (K / 32) * 64;
    hipLaunchKernelGGL(pack_gate_up_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t *)gate, (const bf16_t *)up, (uint4 *)packed, I, K);
    QLAUNCH();
    return QTTS_OK;
}

// ------------------------------------------------------------------ fp8 (e4m3) weights
// A weight row is stored as q * scale with q in e4m3 and scale a power of two (so q * scale is a bf16 number and
// the GEMM over fp8 weights equals, bit for bit, the bf16 GEMM over the dequantised weights).  Layout:
// [N/16 tiles][K/64][64 lanes][16 bytes]; lane (g, r) of a fragment holds row 16*tile + r, bytes 0..7 the k values
// 64*j + 8*g .. +8 (the even k-step of the pair), bytes 8..15 the k values 64*j + 32 + 8*g .. +8 (the odd one).
// scales[packed row].  With `up` != null the packed rows are the gate/up interleave of qtts_pack_gate_up.
__global__ __launch_bounds__(256) void pack_fp8_kernel(const bf16_t *__restrict__ w, const bf16_t *__restrict__ up,
                                                       unsigned char *__restrict__ packed, float *__restrict__ scales, int K) {
    __shared__ float red[4];
    const int pr = blockIdx.x, tile = pr >> 4, r = pr & 15;
    const bf16_t *src = !up ? w + (size_t)pr * K : (r < 8 ? w + (size_t)(8 * tile + r) * K : up + (size_t)(8 * tile + r - 8) * K);
    float amax = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) amax = fmaxf(amax, fabsf(bf2f(src[k])));
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    // smallest power of two with amax / scale <= 448
    float scale = 1.0f;
    if (amax > 0.f) {
        int e;
        const float mant = frexpf(amax / 448.0f, &e);            // amax/448 = mant * 2^e, mant in [0.5, 1)
        scale = ldexpf(1.0f, mant == 0.5f ? e - 1 : e);
    }
    if (threadIdx.x == 0) scales[pr] = scale;
    const float inv = 1.0f / scale;                              // exact
    for (int k4 = threadIdx.x * 4; k4 < K; k4 += 1024) {
        int word = 0;
        word = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(src[k4]) * inv, bf2f(src[k4 + 1]) * inv, word, false);
        word = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(src[k4 + 2]) * inv, bf2f(src[k4 + 3]) * inv, word, true);
        const int j = k4 >> 6, within = k4 & 63, half = within >> 5, g = (within & 31) >> 3, e = within & 7;
        const size_t off = (((size_t)tile * (K >> 6) + j) * 64 + (g * 16 + r)) * 16 + half * 8 + e;
        *reinterpret_cast<int *>(packed + off) = word;
    }
}

extern "C" size_t qtts_packed_weight_fp8_bytes(int N, int K) { return (size_t)N * K; }

extern "C" int qtts_pack_weight_fp8(const void *w_rowmajor, void *packed, float *scales, int N, int K, qtts_stream_t stream) {
    QCHECK(w_rowmajor && packed && scales, "pack_weight_fp8: null pointer");
    QCHECK(N > 0 && K > 0 && N % 16 == 0 && K % 64 == 0, "pack_weight_fp8: need N%%16==0, K%%64==0 (N=%d K=%d)", N, K);
    hipLaunchKernelGGL(pack_fp8_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)w_rowmajor, nullptr,
                       (unsigned char *)packed, scales, K);
    QLAUNCH();
    return QTTS_OK;
}

extern "C" int qtts_pack_gate_up_fp8(const void *gate, const void *up, void *packed, float *scales, int I, int K,
                                     qtts_stream_t stream) {
    QCHECK(gate && up && packed && scales, "pack_gate_up_fp8: null pointer");
    QCHECK(I > 0 && K > 0 && I % 8 == 0 && K % 64 == 0, "pack_gate_up_fp8: need I%%8==0, K%%64==0 (I=%d K=%d)", I, K);
    hipLaunchKernelGGL(pack_fp8_kernel, dim3(2 * I), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)gate,
                       (const bf16_t *)up, (unsigned char *)packed, scales, K);
    QLAUNCH();
    return QTTS_OK;
}

// Dispatch thresholds.  The shipped library holds them as constants; a -DQTTS_TUNING build (tools/ only, see
// csrc/build.py --tuning) turns them into process-wide variables with setters for A/B measurements.
#ifdef QTTS_TUNING
static int g_tune_ksw = 0;   // 0 = automatic
static int g_tune_t = 0;
static int g_dbg_linear = 0;
static int g_rows_min = 16;              // most rows that still take single-tile workgroups instead of the prompt GEMM
static int g_share_prologue = 1;         // norm prologue shared between the lane halves for up to 8 rows (0: tuning)
static int g_sub_min_bytes = 32 << 10;   // smallest weight slice per workgroup when a tile is split over workgroups
static int g_mfma_attn_rows = 112;       // rows from which qtts_paged_attention takes the MFMA kernel for prompts
static int g_rows_tm = 0;                // row tiles per workgroup of the prompt GEMM (0 = chosen per shape)
static int g_attn_wide = 0;              // decode attention: 0 = per shape, 1 = 128-token workgroups, 2 = 256-token ones
extern "C" void qtts_tune_attn_wide(int mode) { g_attn_wide = mode; }
static int g_rows_tm1 = 2;               // prompt GEMMs that may take one row tile per workgroup (0 none, 1 narrow plain outputs, 2 any)
extern "C" void qtts_tune_rows_tm1(int on) { g_rows_tm1 = on; }
static int g_rows_norm = 1;              // prompt GEMMs normalise their rows themselves (0: a norm launch in front, for A/B runs)
static int g_rows_norm_max = 1 << 30;    // most rows for which they do
extern "C" void qtts_tune_rows_norm(int on, int max_rows) { g_rows_norm = on; g_rows_norm_max = max_rows; }
extern "C" void qtts_tune_rows_tm(int tm) { g_rows_tm = tm; }
extern "C" void qtts_tune_linear(int ksw, int t) { g_tune_ksw = ksw; g_tune_t = t; }
extern "C" void qtts_tune_subtiles(int min_bytes) { g_sub_min_bytes = min_bytes; }
extern "C" void qtts_tune_rows(int min_rows) { g_rows_min = min_rows; }
extern "C" void qtts_tune_shared_prologue(int on) { g_share_prologue = on; }
extern "C" void qtts_tune_prompt_attention(int min_rows) { g_mfma_attn_rows = min_rows; }
extern "C" void qtts_debug_linear(int flags) { g_dbg_linear = flags; }
static long long *g_trace = nullptr;       // timestamp buffer of traced GEMM launches
static int g_trace_launch = 0;
extern "C" void qtts_debug_linear_trace(long long *buf) { g_trace = buf; g_trace_launch = 0; }
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

// 16 fp8 (e4m3, OCP) values -> two bf16 fragments, exactly (every e4m3 value is a bf16 value).  Bytes 0..7 belong
// to the even k-step of the pair, bytes 8..15 to the odd one.
typedef float float2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t fp8x2_to_bf16x2(uint32_t w, bool hi) {
    const float2_t f = hi ? __builtin_amdgcn_cvt_pk_f32_fp8(w, true) : __builtin_amdgcn_cvt_pk_f32_fp8(w, false);
    return (__float_as_uint(f[0]) >> 16) | (__float_as_uint(f[1]) & 0xffff0000u);
}
__device__ __forceinline__ void fp8x16_to_bf16(const uint4 q, Frag &even, Frag &odd) {
    even.u = make_uint4(fp8x2_to_bf16x2(q.x, false), fp8x2_to_bf16x2(q.x, true), fp8x2_to_bf16x2(q.y, false), fp8x2_to_bf16x2(q.y, true));
    odd.u = make_uint4(fp8x2_to_bf16x2(q.z, false), fp8x2_to_bf16x2(q.z, true), fp8x2_to_bf16x2(q.w, false), fp8x2_to_bf16x2(q.w, true));
}

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

// EPI: 0 = bias/residual epilogue; 1 = SwiGLU: a tile's rows 0..7 are gate rows and rows 8..15 the up rows of the
// same 8 outputs (qtts_pack_gate_up order), out[m][8*t + j] = bf16(bf16(silu(bf16 gate_j)) * bf16 up_j).
// What the epilogue needs; everything the first loads depend on is passed as leading scalar arguments, which
// the hardware can hand to a wave in registers at launch (kernel-argument preload) instead of through a
// dependent scalar load.
struct LinTail {
    const bf16_t *bias, *res; int ldr;
    bf16_t *xnorm_out; int ldxn;
    bf16_t *out; int ldo;
    const float *wscale;  // fp8 weights: one power-of-two scale per packed weight row
    int opks;             // EPI 1: > 0 = `out` in fragment order with this many k-steps (the next GEMM's K / 32)
#ifdef QTTS_TUNING
    long long *trace;     // 4 timestamps per workgroup
#endif
};
#ifdef QTTS_TUNING
#define QTTS_TRACE(slot) do { if (tl.trace && threadIdx.x == 0) tl.trace[(size_t)blockIdx.x * 4 + (slot)] = wall_clock64(); } while (0)
#else
#define QTTS_TRACE(slot) do { } while (0)
#endif

// H8 (at most 8 rows, one row tile; with or without the norm): MFMA columns 8..15 carry no row, so their lanes share the
// prologue's work instead of repeating it on a copy of row 0: lane (g, r < 8) loads and normalises the EVEN k-steps
// of row r, lane (g, r + 8) the ODD k-steps of the same row, and a row rotate hands each its partner's result.  Half
// the activation requests and half the normalisation arithmetic per lane; the sum of squares is added up in the same
// order in both forms (even k-steps, odd k-steps, then their sum), so both return the same bits.
// Where element (row m, column k) of an activation matrix lives.  Row-major (pks == 0): m * ld + k.  Fragment order
// (pks = K / 32 > 0): [row tile m / 16][k-step k / 32][lane = 16 * ((k % 32) / 8) + m % 16][k % 8] - the order in which
// the MFMA B operand of a 16-row tile is consumed, so that the prompt GEMM reads one contiguous KiB per request (as
// it does for the packed weights) instead of sixty-four bytes from each of sixteen rows.  Used between the kernels of
// a prompt's layer only (normalised rows, attention output, SwiGLU output); buffers hold whole 16-row tiles.
__device__ __forceinline__ size_t act_index(int m, int k, int ld, int pks) {
    if (pks == 0) return (size_t)m * ld + k;
    return ((((size_t)(m >> 4) * pks + (k >> 5)) * 64) + (((k & 31) >> 3) << 4) + (m & 15)) * 8 + (k & 7);
}

// Sum of the waves' partial results kept in LDS, base[w * stride] for w = 0 .. nw-1, added in that order.  For the
// wave counts the launches use, every read is issued before the first addition: a loop over a run-time count
// compiles to a chain of dependent reads, each waiting out the LDS latency (about 0.3 us for eight waves on the
// critical path of every GEMM launch).  The order of the additions, and with it every bit, is the loop's.
// BATCH: reads in flight at a time (a kernel short of registers takes the reads in two batches).
template <typename T, int NWC, int BATCH = NWC>
__device__ __forceinline__ T sum_partials_fixed(const T *base, int stride) {
    static_assert(NWC % BATCH == 0, "whole batches");
    T acc;
#pragma unroll
    for (int b = 0; b < NWC; b += BATCH) {
        T part[BATCH];
#pragma unroll
        for (int w = 0; w < BATCH; ++w) part[w] = base[(b + w) * stride];
#pragma unroll
        for (int w = 0; w < BATCH; ++w) acc = (b + w == 0) ? part[0] : acc + part[w];
    }
    return acc;
}
template <typename T>
__device__ __forceinline__ T sum_partials(const T *base, int stride, int nw) {
    switch (nw) {
    case 8: return sum_partials_fixed<T, 8>(base, stride);
    case 4: return sum_partials_fixed<T, 4>(base, stride);
    case 2: return sum_partials_fixed<T, 2>(base, stride);
    default: break;
    }
    T acc = base[0];
    for (int w = 1; w < nw; ++w) acc += base[w * stride];
    return acc;
}

template <int KSW, int PRO, bool DB, int EPI, bool SUB = false, bool W8 = false, bool H8 = false>
__global__ __launch_bounds__(512) void linear2_kernel(const bf16_t *x, const uint4 *wp, const int64_t *ids, int ldx, int K,
                                                      int N, int M, int tiles_per_wg, int dbg, int id_rows,
                                                      const bf16_t *nw, float eps, const LinTail tl) {
    // argument order: the 14 dwords the hardware preloads into SGPRs are the ones the first loads need
    QTTS_TRACE(0);
    LinArgs p;
    p.x = x; p.ldx = ldx; p.ids = ids; p.id_rows = id_rows; p.wp = wp; p.bias = tl.bias; p.res = tl.res; p.ldr = tl.ldr;
    p.nw = nw; p.eps = eps; p.xnorm_out = tl.xnorm_out; p.ldxn = tl.ldxn; p.out = tl.out; p.ldo = tl.ldo;
    p.M = M; p.N = N; p.K = K; p.tiles_per_wg = tiles_per_wg; p.dbg = dbg;
    extern __shared__ float4_t smem4[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int NW = blockDim.x >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int KS = p.K >> 5;
    // W8: the weights are fp8 (e4m3), one 16-byte fragment per lane covers TWO k-steps; NLD loads per tile
    constexpr int NLD = W8 ? KSW / 2 : KSW;
    const int KSP = W8 ? KS >> 1 : KS;
    // Several row tiles (prefill): the launch is one-dimensional and the workgroups that share a weight tile are
    // placed on the same XCD, next to each other in dispatch order (workgroup i runs on XCD i % 8), so the tile
    // comes from HBM once and the other row tiles find it in that XCD's L2.
    int bx = blockIdx.x, by = 0;
    const int mtiles = p.dbg >> 8;
    if (mtiles > 1) {
        const int per = 8 * mtiles, grp = blockIdx.x / per, rem = blockIdx.x % per;
        bx = grp * 8 + (rem & 7);
        by = rem >> 3;
        if (bx * p.tiles_per_wg >= (p.N >> 4)) return;
    }
    // Narrow outputs with a long K (fewer tiles than CUs, a large slice per workgroup): a 16-row weight tile is
    // shared by 2 or 4 workgroups, each loading and storing only its 8 or 4 rows (the other A rows are zero), so
    // that more CUs stream a share of the matrix.
    const int ssh = SUB ? (p.dbg >> 4) & 3 : 0;            // log2 of the workgroups per tile (single-tile launches only)
    const int sub = bx & ((1 << ssh) - 1);
    bx >>= ssh;
    const bool arow = !SUB || (r >> (4 - ssh)) == sub;      // this lane's weight row belongs to the workgroup
    const bool orow = !SUB || (g >> (2 - ssh)) == sub;      // this lane's four output columns do
    const int m = by * 16 + r;
    const bool mvalid = m < p.M;
    const int ks0 = wave * KSW;
    const int ntiles = p.N >> 4;
    const int t_begin = bx * p.tiles_per_wg;
    const int t_end = min(t_begin + p.tiles_per_wg, ntiles);

    // ---- activation prologue.  Rows beyond M read row 0 instead of being predicated off: their MFMA columns
    // are never stored, and unconditional loads keep the instruction stream ahead of the weight requests short.
    static_assert(!H8 || (PRO != 2 && KSW % 2 == 0), "shared prologue: plain or norm prologue with an even number of k-steps");
    const int hi = H8 ? r >> 3 : 0;                         // H8: this lane's k-step parity
    const int xm = H8 ? (r & 7) : m;                        // the activation row this lane loads
    const bool xvalid = H8 ? xm < p.M : mvalid;
    Frag xb[KSW];
    size_t xrow = 0;
    {
        long rid = xvalid ? xm : 0;
        if (p.ids && !(p.dbg & 8)) {
            rid = p.ids[rid];
            rid = rid < 0 ? 0 : (rid >= p.id_rows ? p.id_rows - 1 : rid);
        }
        xrow = (size_t)rid * p.ldx;
    }
    // Residual rows gathered from a table (bit 3 of dbg: `ids` are theirs): the id is requested here, in front of
    // everything else and without a branch (a launch without such ids reads 8 bytes of x instead), and first used
    // when the residual words are requested, behind the weight requests.  Everything this depends on arrives in
    // SGPRs with the wave.
    long rres = 0;
    if (EPI == 0)
        rres = *(((p.dbg & 8) && p.ids) ? p.ids + (mvalid ? m : 0) : reinterpret_cast<const int64_t *>(p.x));
    const bf16_t *xp = p.x + xrow + 32 * ks0 + 8 * g;
    // bit 7 of dbg: x is one 16-row tile in fragment order (act_index) - a request is one contiguous KiB instead of
    // 64 bytes from each of 16 rows (plain prologue without shared lanes only: decode steps of 9..16 rows)
    int xstep = 32;
    if constexpr (!H8 && PRO == 0) {
        if (p.dbg & 128) {
            xp = p.x + ((size_t)ks0 * 64 + lane) * 8;
            xstep = 512;
        }
    }
#pragma unroll
    for (int i = 0; i < KSW; ++i) {
        if (PRO == 2) {
            Frag ga, up;
            ga.u = *reinterpret_cast<const uint4 *>(xp + 32 * i);
            up.u = *reinterpret_cast<const uint4 *>(xp + p.K + 32 * i);
            uint32_t gw[4] = {ga.u.x, ga.u.y, ga.u.z, ga.u.w}, uw[4] = {up.u.x, up.u.y, up.u.z, up.u.w}, ow[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float g0, g1, u0, u1;
                unpack2(gw[j], g0, g1);
                unpack2(uw[j], u0, u1);
                const float s0 = rbf(silu_f(g0)), s1 = rbf(silu_f(g1));
                ow[j] = pack2(s0 * u0, s1 * u1);
            }
            xb[i].u = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        } else if (!H8) {
            xb[i].u = *reinterpret_cast<const uint4 *>(xp + xstep * i);
        } else if (i % 2 == 0) {
            xb[i].u = *reinterpret_cast<const uint4 *>(xp + 32 * (i + hi));
        }
    }
    // The (small) activation loads above and the norm weights go out first: vector memory requests of a wave are
    // served in order, so behind KSW KiB of weight requests they would arrive last although everything else
    // waits for them.
    uint4 nwv[PRO == 1 ? KSW : 1];
    if (PRO == 1) {
#pragma unroll
        for (int i = 0; i < KSW; ++i)
            if (!H8 || i % 2 == 0) nwv[i] = *reinterpret_cast<const uint4 *>(p.nw + 32 * (ks0 + i + hi) + 8 * g);
    }
    const int kp0 = W8 ? ks0 >> 1 : ks0;
    Frag a[DB ? 2 : 1][NLD];
    {
        const uint4 *wptr = p.wp + ((size_t)((p.dbg & 1) ? 0 : t_begin) * KSP + kp0) * 64 + lane;
        if constexpr (SUB && !W8 && KSW % 2 == 0) {
            // Two workgroups per tile (ssh == 1): a workgroup needs 8 of the tile's 16 rows, i.e. half the lanes of a
            // k-step's 1 KiB fragment.  One full-width request serves TWO k-steps: a lane that owns its row reads
            // its own 16 bytes of the even k-step, the other lanes read the 16 bytes of their partner (lane ^ 8, the
            // same g, row +- 8) of the odd k-step, and a row rotate hands those to the partner afterwards.  Half
            // the requests of the masked form, every one 64 lanes wide.  Rows the workgroup does not own hold data
            // of other rows / k-steps: an MFMA output row depends on its own A row only, and those outputs are
            // never stored.
            const uint4 *wsub = p.wp + ((size_t)((p.dbg & 1) ? 0 : t_begin) * KSP + kp0 + (arow ? 0 : 1)) * 64 + (arow ? lane : lane ^ 8);
#pragma unroll
            for (int i = 0; i < NLD; i += 2) {
                a[0][i].n = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(wsub + i * 64));
            }
        } else {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                if (arow) a[0][i].n = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(wptr + i * 64));
                else a[0][i].u = make_uint4(0, 0, 0, 0);
            }
        }
        if (DB) {
            // tile loop: two tiles are in flight from the start; a fragment register is refilled with the tile
            // after next as soon as its multiplication is issued (the last tile is requested again at the end,
            // which keeps the code free of branches so that the compiler's wait counts stay exact)
            const bool two = t_begin + 1 < t_end;
            const uint4 *wptr1 = p.wp + ((size_t)((p.dbg & 1) || !two ? 0 : t_begin + 1) * KSP + (two ? kp0 : 0)) * 64 + lane;
            const int wstep1 = two ? 64 : 0;
#pragma unroll
            for (int i = 0; i < NLD; ++i)
                a[DB ? 1 : 0][i].n = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(wptr1 + i * wstep1));
        }
    }

    size_t rrow = (size_t)m * p.ldr;
    if (EPI == 0 && (p.dbg & 8)) {
        rres = rres < 0 ? 0 : (rres >= p.id_rows ? p.id_rows - 1 : rres);
        rrow = (size_t)rres * p.ldr;
    }
    // Single-tile launches: the tile's residual / bias words are requested HERE, directly behind the weight requests
    // and before anything waits for the activations.  Requested inside the tile body they went out only after the
    // activations had arrived and been prepared, i.e. one round trip late, and since the compiler must assume that
    // a request under a condition may not have been made, its wait for the last weight fragment also waited for
    // them: the launch then took two round trips one after the other instead of one.
    uint2 resw_first = make_uint2(0, 0), biasw_first = make_uint2(0, 0);
    if (!DB && EPI == 0 && wave == 0 && orow) {
        if (p.res && mvalid) resw_first = *reinterpret_cast<const uint2 *>(p.res + rrow + 16 * t_begin + 4 * g);
        if (p.bias) biasw_first = *reinterpret_cast<const uint2 *>(p.bias + 16 * t_begin + 4 * g);
    }
    float *normred = reinterpret_cast<float *>(smem4 + (size_t)2 * NW * 64);
    if (PRO == 1) {
        // sum of squares: even k-steps, odd k-steps, then their sum (H8: the odd ones are the partner lane's)
        float sp[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KSW; ++i) {
            if (H8 && i % 2) continue;
            const uint32_t w4[4] = {xb[i].u.x, xb[i].u.y, xb[i].u.z, xb[i].u.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v0, v1;
                unpack2(w4[j], v0, v1);
                sp[i & 1] += v0 * v0;
                sp[i & 1] += v1 * v1;
            }
        }
        float ss = H8 ? sp[0] + row16_xor8(sp[0]) : sp[0] + sp[1];
        ss += __shfl_xor(ss, 16);
        ss += __shfl_xor(ss, 32);
        if (g == 0) normred[wave * 16 + r] = ss;
        __syncthreads();
        const float tot = sum_partials(normred + r, 16, NW);
        const float rstd = 1.0f / sqrtf(tot / (float)p.K + p.eps);
#pragma unroll
        for (int i = 0; i < KSW; ++i) {
            if (H8 && i % 2) continue;
            const int k = 32 * (ks0 + i + hi) + 8 * g;
            xb[i].u = norm_frag(xb[i].u, rstd, nwv[i]);
            if (p.xnorm_out && bx == 0 && sub == 0 && xvalid)
                *reinterpret_cast<uint4 *>(p.xnorm_out + (size_t)xm * p.ldxn + k) = xb[i].u;
            // H8: the partner lane's k-step (lanes r >= 8 end up with a copy of row r - 8: columns never stored)
            if (H8) xb[i + 1].u = dpp_u4_ror8(xb[i].u);
        }
    }

    if (H8 && PRO == 0) {
        // no norm: the lane halves still split the activation requests (one request, two k-steps)
#pragma unroll
        for (int i = 0; i < KSW; i += 2) xb[i + 1].u = dpp_u4_ror8(xb[i].u);
    }

    // ---- tile loop
    auto tile_body = [&](const int cur, const int t, const int it) {
        // residual / bias words of this tile are requested up front by the wave that will need them, so the
        // epilogue does not start another memory round trip after the reduction
        // The wave that finishes a tile (cross-wave sum, epilogue, store) alternates between waves 0 and 1: the
        // finishing of tile t then overlaps the multiplications of tile t + 1 instead of preceding them in one wave.
        const bool finisher = wave == (NW > 1 ? (it & 1) : 0);
        uint2 resw = resw_first, biasw = biasw_first;
        if (DB && EPI == 0 && finisher && orow) {
            if (p.res && mvalid) resw = *reinterpret_cast<const uint2 *>(p.res + rrow + 16 * t + 4 * g);
            if (p.bias) biasw = *reinterpret_cast<const uint2 *>(p.bias + 16 * t + 4 * g);
        }
        // fp8 weights: the rows' power-of-two scales, requested up front for the same reason
        float4_t wsc = {1.f, 1.f, 1.f, 1.f}, wsc_up = {1.f, 1.f, 1.f, 1.f};
        if (W8 && finisher) {
            wsc = *reinterpret_cast<const float4_t *>(tl.wscale + 16 * t + 4 * g);
            if (EPI == 1 && g < 2) wsc_up = *reinterpret_cast<const float4_t *>(tl.wscale + 16 * t + 8 + 4 * g);
        }
        // past the last tile the refills still execute (no branch) but all read one 1 KiB fragment that is in
        // the cache already
        const bool more = t + 2 < t_end;
        const uint4 *wnext = p.wp + ((size_t)((p.dbg & 1) || !more ? 0 : t + 2) * KSP + (more ? kp0 : 0)) * 64 + lane;
        const int wstep = more ? 64 : 0;
        float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if constexpr (SUB && !W8 && KSW % 2 == 0) {
                if (i % 2 == 0)      // the odd k-step's rows arrived in the partner lanes (row_ror:8 == lane ^ 8)
                    a[0][i + 1].u = make_uint4(__builtin_amdgcn_update_dpp(0, a[0][i].u.x, 0x128, 0xf, 0xf, true),
                                               __builtin_amdgcn_update_dpp(0, a[0][i].u.y, 0x128, 0xf, 0xf, true),
                                               __builtin_amdgcn_update_dpp(0, a[0][i].u.z, 0x128, 0xf, 0xf, true),
                                               __builtin_amdgcn_update_dpp(0, a[0][i].u.w, 0x128, 0xf, 0xf, true));
            }
            if (W8) {
                Frag f0, f1;
                fp8x16_to_bf16(a[DB ? cur : 0][i].u, f0, f1);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0.v, xb[2 * i].v, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1.v, xb[W8 ? 2 * i + 1 : 0].v, acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[DB ? cur : 0][i].v, xb[i].v, acc, 0, 0, 0);
            }
            if (DB) a[DB ? cur : 0][i].n = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(wnext + i * wstep));
        }
        float4_t *red = smem4 + (size_t)(it & 1) * NW * 64;
        red[wave * 64 + lane] = acc;
        QTTS_TRACE(1);
        __syncthreads();
        QTTS_TRACE(2);
        if (finisher) {
            float4_t sum = sum_partials(red + lane, 64, NW);
            // SwiGLU: lanes 0..31 hold gate rows, the up rows of the same outputs sit 32 lanes on - their sums are
            // what those lanes have just added up (same partials, same order), so they are fetched from there with
            // a lane permute instead of a second pass over the partials
            float4_t ups = sum;
            if (EPI == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) ups[j] = __shfl(sum[j], (lane + 32) & 63);
            }
            // fp8 weights: the row's power-of-two scale, exact in fp32 (same result as bf16 weights q * scale)
            if (W8) sum *= wsc;
            if (EPI == 1) {
                if (g < 2) {
                    if (W8) ups *= wsc_up;
                    if (mvalid) {
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float gv = rbf(sum[j]);
                            const float sv = rbf(silu_f(gv));
                            v[j] = sv * rbf(ups[j]);
                        }
                        *reinterpret_cast<uint2 *>(p.out + act_index(m, 8 * t + 4 * g, p.ldo, H8 ? 0 : tl.opks)) =
                            make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
                    }
                }
                return;
            }
            const int n = 16 * t + 4 * g;
            if (mvalid && orow) {
                uint32_t o2[2];
                float v[4], rf[4], bfv[4];
                unpack2(resw.x, rf[0], rf[1]);
                unpack2(resw.y, rf[2], rf[3]);
                unpack2(biasw.x, bfv[0], bfv[1]);
                unpack2(biasw.y, bfv[2], bfv[3]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float q = sum[j];
                    if (p.bias) q += bfv[j];
                    q = rbf(q);
                    if (p.res) q = q + rf[j];
                    v[j] = q;
                }
                o2[0] = pack2(v[0], v[1]);
                o2[1] = pack2(v[2], v[3]);
                *reinterpret_cast<uint2 *>(p.out + (size_t)m * p.ldo + n) = make_uint2(o2[0], o2[1]);
            }
        }
    };
    if (!DB) {
        // one tile per workgroup: straight-line code, every load of the launch is in flight before the first wait
        tile_body(0, t_begin, 0);
        QTTS_TRACE(3);
        return;
    }
    int t = t_begin;
    for (; t + 4 <= t_end; t += 4) {
        tile_body(0, t, 0);
        tile_body(1, t + 1, 1);
        tile_body(0, t + 2, 0);
        tile_body(1, t + 3, 1);
    }
    if (t < t_end) tile_body(0, t, 0);
    if (t + 1 < t_end) tile_body(1, t + 1, 1);
    if (t + 2 < t_end) tile_body(0, t + 2, 0);
    QTTS_TRACE(3);
}

// ------------------------------------------------------------------ linear over several row tiles (prompts)
// The same split-K workgroup, but the activations of TM 16-row tiles stay in registers while the workgroup
// streams `tiles_per_wg` weight tiles past them (double-buffered): a weight fragment is multiplied into TM
// accumulators, so a weight tile is fetched once per TM row tiles instead of once per row tile, and the
// activations once per workgroup instead of once per weight tile.  EPI 0 / 1 as above; EPI 2 writes fp32 partial
// sums of one K part (blockIdx.y) for linear_parts_kernel to finish — the long-K projection runs as K parts,
// since TM x K/8 activation fragments per wave must fit the register file.
struct RowsArgs {
    const bf16_t *bias, *res; int ldr;
    bf16_t *out; int ldo;
    float *part; int mpad;       // EPI 2: part[kpart][mpad][N]
    int rgroups, KS;             // row groups of TM tiles; k-steps of a whole packed weight tile
    const float *wscale;         // fp8 weights: scale per packed weight row
    int xpacked;                 // x is in fragment order (act_index): whole 16-row tiles
    int opks;                    // EPI 1: > 0 = `out` in fragment order with this many k-steps (the next GEMM's K / 32)
    int xnpks;                   // K-split: > 0 = the fused norm's output in fragment order
    int rpks, opks0;             // EPI 0: > 0 = the residual rows / `out` in fragment order with this many k-steps (N / 32)
    float *ssq_out;              // EPI 0: non-null = sum of squares of every stored row over the 16 columns of a weight
                                 // tile, ssq_out[row][N / 16] (what the RMSNorm of the finished rows starts from)
    // NORM: x holds raw rows; RMSNorm(x) * nw is applied in registers, the rows' sums of squares come from the
    // producer's partials ssq_in[row][ssq_parts] (ssq_parts = K / 16: a multiple of 8, ssq_parts / 8 a power of two <= 16)
    const float *ssq_in; int ssq_parts;
    const bf16_t *nw; float eps;
    bf16_t *xnorm_out; int ldxn; // NORM: non-null = the normalised rows, row-major (written by the first column block)
#ifdef QTTS_TUNING
    long long *trace;            // 32 timestamps per workgroup
#endif
};

// Sum over the q = 1, 2, 4, 8 or 16 neighbouring lanes of a 16-lane row that hold the parts of one value; every lane ends
// up with the same bits (each step adds the two halves symmetrically).
__device__ __forceinline__ float lanes_sum_q(float v, int q) {
    if (q >= 2) v += dpp_f32<0xB1>(v);
    if (q >= 4) v += dpp_f32<0x4E>(v);
    if (q >= 8) v += dpp_f32<0x141>(v);
    if (q >= 16) v += dpp_f32<0x140>(v);
    return v;
}

template <int KSW, int TM, int EPI, bool W8 = false, bool NORM = false>
__global__ __launch_bounds__(512) void linear_rows_kernel(const bf16_t *x, const uint4 *wp, int ldx, int N, int M,
                                                          int tiles_per_wg, const RowsArgs p) {
#ifdef QTTS_TUNING
#define QTTS_RTRACE(slot) do { if (p.trace && threadIdx.x == 0 && (slot) < 32) p.trace[(size_t)blockIdx.x * 32 + (slot)] = wall_clock64(); } while (0)
#else
#define QTTS_RTRACE(slot) do { } while (0)
#endif
    QTTS_RTRACE(0);
    extern __shared__ float4_t smem4[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int NW = blockDim.x >> 6;
    const int g = lane >> 4, r = lane & 15;
    // workgroups that stream the same weight tiles (one per row group) sit next to each other in dispatch order
    // on one XCD (workgroup i runs on XCD i % 8): the tiles come from HBM once, the other groups hit that L2
    const int per = 8 * p.rgroups, grp = blockIdx.x / per, rem = blockIdx.x % per;
    const int bx = grp * 8 + (rem & 7), by = rem >> 3;
    const int ntiles = N >> 4;
    const int t_begin = bx * tiles_per_wg;
    if (t_begin >= ntiles) return;
    const int t_end = min(t_begin + tiles_per_wg, ntiles);
    const int ks0 = blockIdx.y * (NW * KSW) + wave * KSW;
    // W8: fp8 weights, one 16-byte fragment per lane covers two k-steps
    constexpr int NLD = W8 ? KSW / 2 : KSW;
    const int KSP = W8 ? p.KS >> 1 : p.KS, kp0 = W8 ? ks0 >> 1 : ks0;

    // NORM: the rows arrive raw and are normalised here.  Their sums of squares were left by the producing GEMM as one
    // partial per (row, 16 columns); thread (row, j) of the workgroup adds 8 of a row's partials, the q = parts / 8
    // threads of a row add theirs up across lanes, the row's total goes through LDS.  The order of these additions
    // depends on the partial count only, not on the rows of the launch.  These small requests and the norm weights go
    // out first: a wave's requests are served in order, and everything else waits for them.
    constexpr int NPASS = NORM ? (TM * 16 * 16 + 511) / 512 : 1;     // passes of 512 threads over TM * 16 rows x (at most) 16 threads
    float4_t sq[NPASS][2];
    uint4 nwv[NORM ? KSW : 1];
    const int sq_q = NORM ? p.ssq_parts >> 3 : 1;
    if constexpr (NORM) {
        const int qsh = 31 - __builtin_clz(sq_q);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int idx = ps * 512 + threadIdx.x;
            const int rl = min(idx >> qsh, TM * 16 - 1), j = idx & (sq_q - 1);
            const int mrow = min(by * TM * 16 + rl, M - 1);
            const float *sp = p.ssq_in + (size_t)mrow * p.ssq_parts + 8 * j;
            sq[ps][0] = *reinterpret_cast<const float4_t *>(sp);
            sq[ps][1] = *reinterpret_cast<const float4_t *>(sp + 4);
        }
#pragma unroll
        for (int k = 0; k < KSW; ++k) nwv[k] = *reinterpret_cast<const uint4 *>(p.nw + 32 * (ks0 + k) + 8 * g);
    }
    // rows beyond M read row 0: their accumulator columns are never stored
    Frag xb[TM][KSW];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mi = (by * TM + i) * 16 + r;
        // row-major: 16 rows x 64 bytes per request; fragment order: one contiguous KiB per request (a row tile past
        // the last one re-reads the last: never stored)
        const int mtl = (M + 15) >> 4, rt = min(by * TM + i, mtl - 1);
        const bf16_t *xp = p.xpacked ? x + (((size_t)rt * p.KS + ks0) * 64 + lane) * 8
                                     : x + (size_t)(mi < M ? mi : 0) * ldx + 32 * ks0 + 8 * g;
        const int kstep = p.xpacked ? 512 : 32;
#pragma unroll
        for (int k = 0; k < KSW; ++k) xb[i][k].u = *reinterpret_cast<const uint4 *>(xp + (size_t)kstep * k);
    }
    // Two weight tiles are in flight: a fragment register is refilled with the tile after next as soon as its
    // multiplications are issued, so the loads of tile t+2 go out while tile t is still being multiplied.
    const int nt = t_end - t_begin;
    Frag a[2][NLD];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const bool have = j < nt;
        const uint4 *wptr = wp + ((size_t)(have ? t_begin + j : 0) * KSP + (have ? kp0 : 0)) * 64 + lane;
        const int wstep0 = have ? 64 : 0;
#pragma unroll
        for (int k = 0; k < NLD; ++k) a[j][k].u = wptr[k * wstep0];
    }
    if constexpr (NORM) {
        float *rowsum = reinterpret_cast<float *>(smem4 + (size_t)2 * NW * TM * 64);
        const int qsh = 31 - __builtin_clz(sq_q);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int idx = ps * 512 + threadIdx.x;
            float v = sq[ps][0][0];
            v += sq[ps][0][1]; v += sq[ps][0][2]; v += sq[ps][0][3];
            v += sq[ps][1][0]; v += sq[ps][1][1]; v += sq[ps][1][2]; v += sq[ps][1][3];
            v = lanes_sum_q(v, sq_q);
            if ((idx & (sq_q - 1)) == 0 && (idx >> qsh) < TM * 16) rowsum[idx >> qsh] = v;
        }
        __syncthreads();
        const float kf = (float)(p.KS * 32);
        float rstd[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) rstd[i] = 1.0f / sqrtf(rowsum[i * 16 + r] / kf + p.eps);
#pragma unroll
        for (int k = 0; k < KSW; ++k) {
            const float2_t w2[4] = {unpack2v(nwv[k].x), unpack2v(nwv[k].y), unpack2v(nwv[k].z), unpack2v(nwv[k].w)};
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const uint4 xv = xb[i][k].u;
                xb[i][k].u = make_uint4(norm_pair(xv.x, rstd[i], w2[0]), norm_pair(xv.y, rstd[i], w2[1]),
                                        norm_pair(xv.z, rstd[i], w2[2]), norm_pair(xv.w, rstd[i], w2[3]));
                const int mi = (by * TM + i) * 16 + r;
                if (p.xnorm_out && bx == 0 && mi < M)
                    *reinterpret_cast<uint4 *>(p.xnorm_out + (size_t)mi * p.ldxn + 32 * (ks0 + k) + 8 * g) = xb[i][k].u;
            }
        }
    }
    // Plain epilogue: where this lane's four columns of weight tile t lie in the residual rows and in `out`, as
    // base + t * stride.  Row-major: m * ld + 4 g, 16 columns per tile; fragment order (act_index): tile t is half a
    // k-step, 256 elements on, and the lane's four columns sit in lane (2 (t & 1) + g / 2, r) at 4 (g & 1).
    int res_base = 0, res_step = 16, out_base = 0, out_step = 16;
    if (EPI == 0) {
        const int mt = by * TM + wave, m = mt * 16 + r;
        const int fbase = (((g >> 1) << 4) + r) * 8 + 4 * (g & 1);
        res_base = p.rpks ? mt * p.rpks * 512 + fbase : m * p.ldr + 4 * g;
        out_base = p.opks0 ? mt * p.opks0 * 512 + fbase : m * p.ldo + 4 * g;
        res_step = p.rpks ? 256 : 16;
        out_step = p.opks0 ? 256 : 16;
    }
    auto tile_body = [&](const int cur, const int it) {
        const int t = t_begin + it;
        uint2 resw = make_uint2(0, 0);
        if (EPI == 0 && wave < TM) {
            const int m = (by * TM + wave) * 16 + r;
            if (p.res && m < M) resw = *reinterpret_cast<const uint2 *>(p.res + (size_t)(res_base + t * res_step));
        }
        // fp8 weights: the rows' scales, requested here and not in the epilogue (a late load is waited for together
        // with the weight tiles requested after it)
        float4_t wsc = {1.f, 1.f, 1.f, 1.f}, wsc_up = {1.f, 1.f, 1.f, 1.f};
        if (W8 && wave < TM) {
            wsc = *reinterpret_cast<const float4_t *>(p.wscale + 16 * t + 4 * g);
            if (EPI == 1 && g < 2) wsc_up = *reinterpret_cast<const float4_t *>(p.wscale + 16 * t + 8 + 4 * g);
        }
        // unconditional: with straight-line code the compiler's wait counts let exactly the younger tile's loads
        // stay in flight; past the last tile the refills all read one 1 KiB fragment that is cached already
        const bool more = it + 2 < nt;
        const uint4 *wnext = wp + ((size_t)(more ? t + 2 : 0) * KSP + (more ? kp0 : 0)) * 64 + lane;
        const int wstep = more ? 64 : 0;
        float4_t acc[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            if (W8) {
                Frag f0, f1;
                fp8x16_to_bf16(a[cur][k].u, f0, f1);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0.v, xb[i][2 * k].v, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1.v, xb[i][W8 ? 2 * k + 1 : 0].v, acc[i], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[cur][k].v, xb[i][k].v, acc[i], 0, 0, 0);
            }
            a[cur][k].u = wnext[k * wstep];
        }
        float4_t *red = smem4 + (size_t)(it & 1) * NW * TM * 64;
#pragma unroll
        for (int i = 0; i < TM; ++i) red[(wave * TM + i) * 64 + lane] = acc[i];
        QTTS_RTRACE(1 + 2 * it);
        __syncthreads();
        QTTS_RTRACE(2 + 2 * it);
        if (wave >= TM) return;
        // wave i finishes row tile i of this weight tile (all eight waves' partial sums, added in wave order)
        const float4_t *mine = red + wave * 64;
        const int m = (by * TM + wave) * 16 + r;
        // (four row tiles of activations leave few registers: the reads go out in two batches of four)
        constexpr int RB = (TM > 2 && KSW >= 8) ? 4 : 8;
        float4_t sum = sum_partials_fixed<float4_t, 8, RB>(mine + lane, TM * 64);
        // SwiGLU: the up rows' sums are what the lanes 32 on have just added up (fetched with a lane permute)
        float4_t ups = sum;
        if (EPI == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ups[j] = __shfl(sum[j], (lane + 32) & 63);
        }
        if (W8) sum *= wsc;                                // exact: powers of two
        if (EPI == 2) {
            if (m < M)
                *reinterpret_cast<float4_t *>(p.part + ((size_t)blockIdx.y * p.mpad + m) * N + 16 * t + 4 * g) = sum;
            return;
        }
        if (EPI == 1) {
            if (g < 2) {
                if (W8) ups *= wsc_up;
                if (m < M) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float gv = rbf(sum[j]);
                        const float sv = rbf(silu_f(gv));
                        v[j] = sv * rbf(ups[j]);
                    }
                    *reinterpret_cast<uint2 *>(p.out + act_index(m, 8 * t + 4 * g, p.ldo, p.opks)) =
                        make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
                }
            }
            return;
        }
        // (no bias here: a load this late is waited for with every younger request, i.e. with the weight tiles
        // in flight for the next bodies; GEMMs with a bias take the single-tile kernel or the K-split epilogue)
        float v[4], rf[4];
        unpack2(resw.x, rf[0], rf[1]);
        unpack2(resw.y, rf[2], rf[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float q = rbf(sum[j]);
            if (p.res) q = q + rf[j];
            v[j] = rbf(q);                                 // the value as it is stored
        }
        if (m < M)
            *reinterpret_cast<uint2 *>(p.out + (size_t)(out_base + t * out_step)) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
        if (p.ssq_out) {
            // the row's sum of squares over this tile's 16 columns: four columns per lane, then the four lanes of the row
            float s2 = v[0] * v[0];
            s2 += v[1] * v[1]; s2 += v[2] * v[2]; s2 += v[3] * v[3];
            s2 += __shfl_xor(s2, 16);
            s2 += __shfl_xor(s2, 32);
            if (g == 0 && m < M) p.ssq_out[(size_t)m * (N >> 4) + t] = s2;
        }
    };
    // groups of four tiles as straight-line code: inside a group the compiler counts outstanding loads exactly
    // and the younger tile stays in flight; at a loop head it waits for everything
    int it = 0;
    for (; it + 4 <= nt; it += 4) {
        tile_body(0, it);
        tile_body(1, it + 1);
        tile_body(0, it + 2);
        tile_body(1, it + 3);
    }
    if (it < nt) tile_body(0, it);
    if (it + 1 < nt) tile_body(1, it + 1);
    if (it + 2 < nt) tile_body(0, it + 2);
    QTTS_RTRACE(31);
#undef QTTS_RTRACE
}

// Sum of the K parts of one output quad, part 0 first.  Up to four parts are requested together (a loop over a
// run-time count waits for each part before it asks for the next).
__device__ __forceinline__ float4_t sum_k_parts(const float *part, size_t first, size_t part_stride, int kparts) {
    if (kparts <= 4) {
        float4_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            v[k] = *reinterpret_cast<const float4_t *>(part + first + (size_t)(k < kparts ? k : kparts - 1) * part_stride);
        float4_t sum = v[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) sum = k < kparts ? sum + v[k] : sum;
        return sum;
    }
    float4_t sum = *reinterpret_cast<const float4_t *>(part + first);
    for (int k = 1; k < kparts; ++k) sum += *reinterpret_cast<const float4_t *>(part + first + (size_t)k * part_stride);
    return sum;
}

// Sum of the K parts + the bias/residual epilogue (same rounding points as the single-launch epilogue).
__global__ __launch_bounds__(256) void linear_parts_kernel(const float *part, int kparts, int mpad, int M, int N,
                                                           const bf16_t *bias, const bf16_t *res, int ldr, bf16_t *out, int ldo) {
    const int n4 = N >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)M * n4) return;
    const int m = (int)(idx / n4), n = (int)(idx % n4) * 4;
    const float4_t sum = sum_k_parts(part, (size_t)m * N + n, (size_t)mpad * N, kparts);
    float v[4], rf[4] = {0.f, 0.f, 0.f, 0.f}, bfv[4] = {0.f, 0.f, 0.f, 0.f};
    if (res) {
        const uint2 w = *reinterpret_cast<const uint2 *>(res + (size_t)m * ldr + n);
        unpack2(w.x, rf[0], rf[1]);
        unpack2(w.y, rf[2], rf[3]);
    }
    if (bias) {
        const uint2 w = *reinterpret_cast<const uint2 *>(bias + n);
        unpack2(w.x, bfv[0], bfv[1]);
        unpack2(w.y, bfv[2], bfv[3]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float q = sum[j];
        if (bias) q += bfv[j];
        q = rbf(q);
        if (res) q = q + rf[j];
        v[j] = q;
    }
    *reinterpret_cast<uint2 *>(out + (size_t)m * ldo + n) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
}

// The same, one workgroup per row, followed by the RMSNorm of the finished row with the weight of the layer that
// consumes it (the arithmetic and summation order of rmsnorm_kernel): saves the norm's own launch.  N % 8 == 0.
__global__ __launch_bounds__(256) void linear_parts_norm_kernel(const float *part, int kparts, int mpad, int N,
                                                                const bf16_t *bias, const bf16_t *res, int ldr, bf16_t *out,
                                                                int ldo, const bf16_t *nw, float eps, bf16_t *xn, int ldxn,
                                                                int xnpks, int rpks, int opks0) {
    // rpks / opks0 > 0: the residual rows / `out` in fragment order (act_index; a thread's 8 columns stay contiguous)
    __shared__ float red[4];
    const int m = blockIdx.x;
    float ss = 0.f;
    for (int n = threadIdx.x * 8; n < N; n += 256 * 8) {
        float q[8];
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const float4_t sum = sum_k_parts(part, (size_t)m * N + n + 4 * hlf, (size_t)mpad * N, kparts);
#pragma unroll
            for (int j = 0; j < 4; ++j) q[4 * hlf + j] = sum[j];
        }
        Frag rf, bfv, o;
        rf.u = res ? *reinterpret_cast<const uint4 *>(res + act_index(m, n, ldr, rpks)) : make_uint4(0, 0, 0, 0);
        bfv.u = bias ? *reinterpret_cast<const uint4 *>(bias + n) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = q[j];
            if (bias) v += bf2f(bfv.h[j]);
            v = rbf(v);
            if (res) v = v + bf2f(rf.h[j]);
            o.h[j] = f2bf(v);
            const float r = bf2f(o.h[j]);
            ss += r * r;
        }
        *reinterpret_cast<uint4 *>(out + act_index(m, n, ldo, opks0)) = o.u;
    }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const float rstd = 1.0f / sqrtf(tot / (float)N + eps);
    for (int n = threadIdx.x * 8; n < N; n += 256 * 8) {
        const uint4 f = *reinterpret_cast<const uint4 *>(out + act_index(m, n, ldo, opks0));          // this thread's own stores
        const uint4 wv = *reinterpret_cast<const uint4 *>(nw + n);
        *reinterpret_cast<uint4 *>(xn + act_index(m, n, ldxn, xnpks)) = norm_frag(f, rstd, wv);
    }
}

// Measurement hook (qtts_model_set_layer_events): when set, the next decode GEMM launch of this thread carries the
// two events as the kernel's start / stop events.
struct KernelEvents { hipEvent_t start, stop; };
static thread_local KernelEvents g_kernel_events = {nullptr, nullptr};
#define QTTS_LAUNCH_MAYBE_TIMED(kern, grid, block, shmem, s, ...) do { \
        if (g_kernel_events.start) { \
            hipExtLaunchKernelGGL(kern, grid, block, (uint32_t)(shmem), s, g_kernel_events.start, g_kernel_events.stop, 0, __VA_ARGS__); \
            g_kernel_events = {nullptr, nullptr}; \
        } else { \
            hipLaunchKernelGGL(kern, grid, block, shmem, s, __VA_ARGS__); \
        } } while (0)

template <int KSW, int PRO, bool DB, int EPI, bool SUB = false>
static void launch_linear2_inst(dim3 grid, int NW, hipStream_t s, const LinArgs &a) {
    constexpr bool CAN_H8 = PRO != 2 && KSW % 2 == 0;
    size_t shmem = (size_t)2 * NW * 64 * sizeof(float4_t) + (size_t)NW * 16 * sizeof(float);
    LinTail tl;
    tl.bias = a.bias; tl.res = a.res; tl.ldr = a.ldr; tl.xnorm_out = a.xnorm_out; tl.ldxn = a.ldxn; tl.out = a.out; tl.ldo = a.ldo;
#ifdef QTTS_TUNING
    tl.trace = nullptr;
    if (g_trace) {
        tl.trace = g_trace + (size_t)g_trace_launch * 4 * 1024;      // up to 1024 workgroups per traced launch
        ++g_trace_launch;
    }
#endif
    tl.wscale = a.wscale;
    tl.opks = a.opks;
    if constexpr (KSW % 2 == 0 && PRO != 2) {
        if (a.wscale) {
            // fp8 weights: the activation prologue is the bf16 kernels', shared lanes for at most 8 rows included
            if (CAN_H8 && a.M <= 8 && !(a.dbg >> 8) && g_share_prologue) {
                QTTS_LAUNCH_MAYBE_TIMED((linear2_kernel<KSW, PRO, DB, EPI, SUB, true, CAN_H8>), grid, dim3(NW * 64), shmem, s, a.x,
                                        a.wp, a.ids, a.ldx, a.K, a.N, a.M, a.tiles_per_wg, a.dbg, a.id_rows, a.nw, a.eps, tl);
                return;
            }
            QTTS_LAUNCH_MAYBE_TIMED((linear2_kernel<KSW, PRO, DB, EPI, SUB, true>), grid, dim3(NW * 64), shmem, s, a.x, a.wp, a.ids,
                                    a.ldx, a.K, a.N, a.M, a.tiles_per_wg, a.dbg, a.id_rows, a.nw, a.eps, tl);
            return;
        }
    }
    if constexpr (CAN_H8) {
        if (a.M <= 8 && !(a.dbg >> 8) && g_share_prologue) {          // one row tile of at most 8 rows
            QTTS_LAUNCH_MAYBE_TIMED((linear2_kernel<KSW, PRO, DB, EPI, SUB, false, true>), grid, dim3(NW * 64), shmem, s, a.x, a.wp,
                                    a.ids, a.ldx, a.K, a.N, a.M, a.tiles_per_wg, a.dbg, a.id_rows, a.nw, a.eps, tl);
            return;
        }
    }
    QTTS_LAUNCH_MAYBE_TIMED((linear2_kernel<KSW, PRO, DB, EPI, SUB>), grid, dim3(NW * 64), shmem, s, a.x, a.wp, a.ids, a.ldx,
                            a.K, a.N, a.M, a.tiles_per_wg, a.dbg, a.id_rows, a.nw, a.eps, tl);
}

template <int KSW, int TM, int EPI, bool NORM = false>
static bool launch_rows_inst(dim3 grid, hipStream_t s, const LinArgs &a, int tpw, const RowsArgs &r) {
    const size_t shmem = (size_t)2 * 8 * TM * 64 * sizeof(float4_t) + (NORM ? (size_t)TM * 16 * sizeof(float) : 0);
    if (a.wscale) {
        if (!allow_dynamic_lds(reinterpret_cast<const void *>(&linear_rows_kernel<KSW, TM, EPI, true, NORM>), shmem)) return false;
        hipLaunchKernelGGL((linear_rows_kernel<KSW, TM, EPI, true, NORM>), grid, dim3(512), shmem, s, a.x, a.wp, a.ldx, a.N, a.M, tpw, r);
        return true;
    }
    if (!allow_dynamic_lds(reinterpret_cast<const void *>(&linear_rows_kernel<KSW, TM, EPI, false, NORM>), shmem)) return false;
    hipLaunchKernelGGL((linear_rows_kernel<KSW, TM, EPI, false, NORM>), grid, dim3(512), shmem, s, a.x, a.wp, a.ldx, a.N, a.M, tpw, r);
    return true;
}

// Whether the prompt GEMM can normalise its input rows itself from the producer's sum-of-squares partials (prologue
// 3): K = 1024 or 2048 in one part, K / 16 partials per row handled by 1..16 threads of 8.
static bool linear_rows_norm_covers(int K) {
    const int ksw = K / 256, parts = K / 16;
    return K % 256 == 0 && (ksw == 4 || ksw == 8) && parts % 8 == 0 && ((parts / 8) & (parts / 8 - 1)) == 0 && parts / 8 <= 16;
}

// Bytes of fp32 scratch the K-split form of a prompt GEMM needs (0 when the GEMM runs in one part).
static size_t linear_rows_scratch_bytes(int M, int N, int K) {
    const int KS = K / 32;
    if (M <= 16 || KS % 8 != 0 || KS / 8 <= 12) return 0;
    return (size_t)(KS / 64) * M * N * sizeof(float);
}

// True when launch_linear_rows runs this shape as K parts (and can therefore take the fused norm of the result).
static bool linear_rows_is_ksplit(int M, int N, int K, size_t scratch_bytes) {
    const size_t need = linear_rows_scratch_bytes(M, N, K);
    return M > g_rows_min && need > 0 && scratch_bytes >= need && (K / 32) % 64 == 0 && N % 8 == 0 && !(g_dbg_linear & (2 | 32));   // bit 5: tests switch the fusion off
}

// Prompt GEMM (more than one 16-row tile, plain activations).  Returns false when the shape is not covered and the
// caller should take the single-tile path.
// Whether launch_linear_rows takes this shape (the decision a caller needs before it lays activations out in fragment
// order, which only that kernel reads).
static bool linear_rows_covers(int M, int N, int K, bool silu_pairs, bool has_bias, bool has_scratch, size_t scratch_bytes) {
    if (M <= g_rows_min || (g_dbg_linear & 2) || K % 32 != 0 || N % 16 != 0) return false;
    const int KS = K / 32;
    if (KS % 8 != 0) return false;
    int ksw = KS / 8, kparts = 1;
    if (ksw > 12) {
        if (ksw % 8 != 0 || silu_pairs) return false;
        kparts = ksw / 8;
        ksw = 8;
        if (!has_scratch || scratch_bytes < (size_t)kparts * M * N * sizeof(float)) return false;
    }
    if (ksw != 4 && ksw != 8 && ksw != 12) return false;
    if (ksw == 12 && silu_pairs) return false;
    if (has_bias && kparts == 1) return false;
    return true;
}

static bool launch_linear_rows(const LinArgs &a, bool silu_pairs, float *scratch, size_t scratch_bytes, hipStream_t s,
                               bool norm3 = false) {
    const int KS = a.K / 32;
    if (KS % 8 != 0) return false;
    if (norm3 && (!linear_rows_norm_covers(a.K) || !a.ssq_in || !a.nw)) return false;
    int ksw = KS / 8, kparts = 1;
    if (ksw > 12) {                                       // long K: parts of 8 k-steps per wave
        if (ksw % 8 != 0 || silu_pairs) return false;
        kparts = ksw / 8;
        ksw = 8;
        if (!scratch || scratch_bytes < (size_t)kparts * a.M * a.N * sizeof(float)) return false;
    }
    if (ksw != 4 && ksw != 8 && ksw != 12) return false;
    if (ksw == 12 && silu_pairs) return false;
    if (a.bias && kparts == 1) return false;              // the single-pass epilogue carries no bias
    if (kparts > 1 && !a.next_nw && (a.rpks || a.opks0)) return false;   // (the plain K-part finish is row-major only)
    const int mtiles = (a.M + 15) / 16;
    const int ntiles = a.N / 16;
    // Row tiles per workgroup (TM).  More row tiles per workgroup mean fewer passes over the weights (fewer, longer
    // column blocks in the one round of workgroups) but more activation tiles to pull in before the first
    // multiplication, and those are the expensive ones: every workgroup of an XCD asks its L2 for the same rows at the
    // same moment, nothing is multiplied until they are there, and they hold the registers the weight stream could
    // use.  Measured on the talker's four GEMMs at 40..400 rows (activations in fragment order) with TM forced to
    // 2 / 3 / 4: a row tile costs a workgroup about what two weight tiles do (qkv at 96 rows: 9.6 us with four row
    // tiles, 8.0 with three; gate/up at 48 rows 15.1 -> 13.0 with three, at 200 rows 22.6 with four against 30.4 with
    // two; o_proj is fastest with two up to 256 rows).  The plan with the lowest 2 * TM + weight tiles wins; ties go
    // to the larger TM (fewer re-reads of the weights from L2).  The K-split form takes two or four (three measured
    // 2-3 us slower at 64..96 rows with the same tiles per workgroup).
    auto plan = [&](int tm, int *tpw_out) {
        const int rg = (mtiles + tm - 1) / tm;
        const int gxm = 8 * std::max(1, 32 / (rg * kparts));
        const int tpw_ = (ntiles + gxm - 1) / gxm;
        if (tpw_out) *tpw_out = tpw_;
        // (row groups x K parts beyond the 32 CUs of an XCD: the launch takes several rounds of workgroups)
        const int gx_ = (ntiles + tpw_ - 1) / tpw_;
        const int rounds = (rg * kparts * ((gx_ + 7) / 8) + 31) / 32;
        return rounds * (4 * tm + (a.wscale ? tpw_ : 2 * tpw_));
    };
    int TM = 2;
    {
        // (one row tile per workgroup: narrow outputs only - up to 128 weight tiles, plain epilogue, one K part - where a
        // workgroup streams a single weight tile and the rows it has to pull in first are most of its bytes)
        const int cands[] = {4, 3, 2, 1};
        int best = 1 << 30;
        for (int tm : cands) {
            if (tm == 3 && (ksw != 8 || kparts > 1)) continue;   // three row tiles: built for K = 2048 in one part
            if (tm > 2 && (mtiles <= 2 || ksw == 12)) continue;
            if (norm3 && tm > 2 && ksw == 8) continue;           // rows normalised in registers: two row tiles at K = 2048
            // (g_rows_tm1 == 1: narrow plain outputs only; 2: any single-part GEMM of up to four row tiles)
            if (tm == 1 && (kparts > 1 || mtiles > 4 || g_rows_tm1 == 0 || (g_rows_tm1 == 1 && (ntiles > 128 || silu_pairs)))) continue;
            if (tm == 1 && silu_pairs && ksw != 4 && ksw != 8) continue;
            const int c = plan(tm, nullptr);
            if (c < best) { best = c; TM = tm; }
        }
    }
    if (!norm3 && (g_rows_tm == 2 || (g_rows_tm == 4 && ksw != 12 && mtiles > 2) || (g_rows_tm == 3 && ksw == 8 && kparts == 1 && mtiles > 2))) TM = g_rows_tm;
    RowsArgs r;
    r.bias = a.bias; r.res = a.res; r.ldr = a.ldr; r.out = a.out; r.ldo = a.ldo;
    r.rpks = a.rpks; r.opks0 = a.opks0; r.ssq_out = kparts == 1 && !silu_pairs ? a.ssq_out : nullptr;
    r.ssq_in = norm3 ? a.ssq_in : nullptr; r.ssq_parts = a.K / 16; r.nw = a.nw; r.eps = a.eps;
    r.xnorm_out = norm3 ? a.xnorm_out : nullptr; r.ldxn = a.ldxn;
    r.part = scratch; r.mpad = a.M; r.rgroups = (mtiles + TM - 1) / TM; r.KS = KS;
#ifdef QTTS_TUNING
    r.trace = g_trace;
#endif
    r.wscale = a.wscale;
    r.xpacked = (g_dbg_linear & 64) != 0 ? 1 : a.xpacked; r.opks = a.opks; r.xnpks = a.nnpks;   // (bit 6: timing experiments read x in fragment order)
    // One round of workgroups (a CU holds one, an XCD 32): workgroup i runs on XCD i % 8 and the launch is laid out
    // in groups of 8 column blocks x all row groups (x K parts), so the groups of 8 that fit one round are
    // 32 / (row groups x K parts); each workgroup streams `tpw` weight tiles.
    const int gx_max = 8 * std::max(1, 32 / (r.rgroups * kparts));
    const int tpw = (ntiles + gx_max - 1) / gx_max;
    const int gx = (ntiles + tpw - 1) / tpw;
    dim3 grid(((gx + 7) / 8) * 8 * r.rgroups, kparts);
    const int epi = kparts > 1 ? 2 : (silu_pairs ? 1 : 0);
    bool ok = true;
#define QR(KSW_, TM_, EPI_) ok = launch_rows_inst<KSW_, TM_, EPI_>(grid, s, a, tpw, r)
#define QRN(KSW_, TM_, EPI_) ok = launch_rows_inst<KSW_, TM_, EPI_, true>(grid, s, a, tpw, r)
    if (norm3) {
        if (TM == 1) { if (ksw == 8) { if (epi == 0) QRN(8, 1, 0); else QRN(8, 1, 1); } else { if (epi == 0) QRN(4, 1, 0); else QRN(4, 1, 1); } }
        else if (ksw == 8) { if (epi == 0) QRN(8, 2, 0); else QRN(8, 2, 1); }
        else if (TM == 2) { if (epi == 0) QRN(4, 2, 0); else QRN(4, 2, 1); }
        else { if (epi == 0) QRN(4, 4, 0); else QRN(4, 4, 1); }
    } else
    if (TM == 1) {
        if (ksw == 12) QR(12, 1, 0);
        else if (ksw == 8) { if (epi == 0) QR(8, 1, 0); else QR(8, 1, 1); }
        else { if (epi == 0) QR(4, 1, 0); else QR(4, 1, 1); }
    }
    else if (ksw == 12) QR(12, 2, 0);
    else if (ksw == 8) {
        if (TM == 2) { if (epi == 0) QR(8, 2, 0); else if (epi == 1) QR(8, 2, 1); else QR(8, 2, 2); }
        else if (TM == 3) { if (epi == 0) QR(8, 3, 0); else QR(8, 3, 1); }
        else { if (epi == 0) QR(8, 4, 0); else if (epi == 1) QR(8, 4, 1); else QR(8, 4, 2); }
    } else {
        if (TM == 2) { if (epi == 0) QR(4, 2, 0); else QR(4, 2, 1); }
        else { if (epi == 0) QR(4, 4, 0); else QR(4, 4, 1); }
    }
#undef QR
#undef QRN
    if (!ok) return false;                               // (the caller's single-tile path still works)
    if (kparts > 1) {
        const long quads = (long)a.M * (a.N / 4);
        if (a.next_nw)
            hipLaunchKernelGGL(linear_parts_norm_kernel, dim3(a.M), dim3(256), 0, s, scratch, kparts, a.M, a.N, a.bias, a.res,
                               a.ldr, a.out, a.ldo, a.next_nw, a.eps, a.next_xn, a.ldnn, a.nnpks, a.rpks, a.opks0);
        else
            hipLaunchKernelGGL(linear_parts_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, scratch, kparts, a.M,
                               a.M, a.N, a.bias, a.res, a.ldr, a.out, a.ldo);
    }
    return true;
}

static int launch_linear2(LinArgs a, int pro, hipStream_t s, bool silu_pairs = false, float *scratch = nullptr,
                          size_t scratch_bytes = 0) {
    QCHECK(a.x && a.wp && a.out, "linear: null pointer");
    QCHECK(a.M > 0 && a.N > 0 && a.K > 0, "linear: empty shape M=%d N=%d K=%d", a.M, a.N, a.K);
    QCHECK(a.N % 16 == 0 && a.K % 32 == 0, "linear: need N%%16==0 and K%%32==0 (N=%d K=%d)", a.N, a.K);
    QCHECK(a.ldx % 8 == 0 && a.ldx >= (pro == 2 ? 2 * a.K : a.K), "linear: x row stride %d too small or not a multiple of 8", a.ldx);
    QCHECK(a.ldo % 4 == 0 && a.ldo >= (silu_pairs ? a.N / 2 : a.N), "linear: bad out row stride %d", a.ldo);
    QCHECK(!a.res || a.ldr >= a.N, "linear: residual row stride %d < N", a.ldr);
    QCHECK(!silu_pairs || (!a.bias && !a.res && pro != 2), "linear: bad SwiGLU arguments");
    QCHECK(((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.wp & 15) == 0 && ((uintptr_t)a.out & 7) == 0, "linear: misaligned pointer");
    QCHECK((pro != 1 && pro != 3) || a.nw, "linear: norm prologue without weight");
    QCHECK(!a.xnorm_out || (a.ldxn % 8 == 0 && a.ldxn >= a.K), "linear: bad xnorm_out stride");
    // from two row tiles on (two row tiles per workgroup in the prompt kernel against single-tile workgroups: frames at
    // 20 / 24 / 32 streams 4.41 / 4.42 / 4.52 -> 4.13 / 4.20 / 4.31 ms, a 17..32-token prompt 2.07 -> 1.75 ms)
    QCHECK(!a.res_ids || (a.M <= 16 && a.res && a.res_id_rows > 0 && !a.ids && !a.bias), "linear: gathered residual rows need M <= 16, a residual table and plain x rows");
    if (a.M > g_rows_min && (pro == 0 || pro == 3) && !a.ids && !(g_dbg_linear & 2) &&
        launch_linear_rows(a, silu_pairs, scratch, scratch_bytes, s, pro == 3)) {
        QLAUNCH();
        return QTTS_OK;
    }
    if (pro == 3 || a.rpks || a.opks0 || a.ssq_out)
        QFAIL(QTTS_ESTATE, "linear: rows normalised from partial sums / fragment-order residual rows are forms of the prompt GEMM, which does not take M=%d N=%d K=%d", a.M, a.N, a.K);
    // One row tile of 9..16 rows (a decode step; up to 8 rows the lanes share the prologue instead): x of a plain
    // prologue and the SwiGLU output may be in fragment order as well
    const bool tile_frag = a.M > 8 && a.M <= 16 && !a.nnpks && (!a.xpacked || (pro == 0 && !a.ids)) && (!a.opks || silu_pairs);
    if ((a.xpacked || a.opks || a.nnpks) && !tile_frag)
        QFAIL(QTTS_ESTATE, "linear: activations in fragment order but neither the prompt GEMM nor the single-tile form takes M=%d N=%d K=%d", a.M, a.N, a.K);
    QCHECK(!a.next_nw, "linear: the fused norm of the result exists for the K-split prompt GEMM only");
    const int KS = a.K / 32;
    static const int ksws[] = {1, 2, 4, 6, 8, 12, 24};
    int KSW = 0;
    for (int c : ksws)
        if (KS % c == 0 && KS / c <= 8) { KSW = c; break; }
    // K = 1024: four waves of 8 k-steps beat eight waves of 4 (fewer waves to start, synchronise and reduce; the
    // predictor's qkv GEMM 4.6 -> 3.8 us, gate/up 5.8 -> 5.3, head 3.3 -> 3.0); K = 2048 measured the same either way
    if (KSW < 8 && KS % 8 == 0 && KS / 8 >= 4) KSW = 8;
    if (g_tune_ksw > 0 && KS % g_tune_ksw == 0 && KS / g_tune_ksw <= 8)
        for (int c : ksws) if (c == g_tune_ksw) KSW = c;
    QCHECK(KSW > 0, "linear: K=%d not supported (K/32 must factor as <=8 waves x {1,2,4,6,8,12,24})", a.K);
    QCHECK(!(pro == 1 && KSW == 24), "linear: RMSNorm prologue not supported for K=%d", a.K);
    QCHECK(!a.wscale || (KSW % 2 == 0 && pro != 2 && a.K % 64 == 0), "linear: fp8 weights need K %% 64 == 0, an even number of k-steps per wave and no SwiGLU prologue (K=%d)", a.K);
    const int NW = KS / KSW;
    const int ntiles = a.N / 16;
    int tpw = 1;
    if (KSW <= 12) {
        const int target = g_tune_t > 0 ? g_tune_t * 64 : 256;
        tpw = (ntiles + target - 1) / target;
        // one and a half rounds of single-tile workgroups beat 192 workgroups looping over two tiles (measured on
        // the predictor's gate/up GEMM: 6.9 -> 6.7 us); beyond that the tile loop with its double buffer wins
        if (g_tune_t == 0 && ntiles <= 384) tpw = 1;
    }
    // prefill (several row tiles): single-tile workgroups keep the register count low, so that three or four
    // workgroups share a CU and cover each other's memory latency
    if (a.M > 16) tpw = 1;
    const bool db = tpw > 1;                              // tile loop with a register double buffer
    if (silu_pairs) QCHECK(KSW <= 12, "linear: the SwiGLU epilogue needs K <= 3072 (K=%d)", a.K);
    a.tiles_per_wg = tpw;
    a.dbg = g_dbg_linear & ~8;
    if (a.xpacked) a.dbg |= 128;
    if (a.res_ids) {                                       // the kernel's id arguments carry the residual's row ids
        a.ids = a.res_ids; a.id_rows = a.res_id_rows; a.dbg |= 8;
    }
    int gx = (ntiles + tpw - 1) / tpw;
    const int gy = (a.M + 15) / 16;
    int ssh = 0;
    if (tpw == 1 && gy == 1 && !silu_pairs && pro != 2) {
        // 2 or 4 workgroups per tile while not every CU has one and a workgroup's slice stays above the threshold
        // (bf16 weights: at most two per tile, the kernel's paired requests are built for that; no model shape asks for four)
        const int ssh_max = a.wscale ? 2 : 1;
        while (ssh < ssh_max && (gx << (ssh + 1)) <= 256 && ((a.K * 32) >> (ssh + 1)) >= g_sub_min_bytes) ++ssh;
        a.dbg |= ssh << 4;
        gx <<= ssh;
    }
    dim3 grid(gx, 1);
    if (gy > 1) {
        a.dbg |= gy << 8;                                  // row-tile count rides above the experiment bits
        grid = dim3(((gx + 7) / 8) * 8 * gy, 1);
    }
#define QL2D(KSW_, DB_) \
        if (silu_pairs) { if (pro == 1) launch_linear2_inst<KSW_, 1, DB_, 1>(grid, NW, s, a); \
                          else launch_linear2_inst<KSW_, 0, DB_, 1>(grid, NW, s, a); } \
        else if (pro == 0) launch_linear2_inst<KSW_, 0, DB_, 0>(grid, NW, s, a); \
        else if (pro == 1) launch_linear2_inst<KSW_, 1, DB_, 0>(grid, NW, s, a); \
        else launch_linear2_inst<KSW_, 2, DB_, 0>(grid, NW, s, a);
#define QL2S(KSW_) if (pro == 0) launch_linear2_inst<KSW_, 0, false, 0, true>(grid, NW, s, a); \
                   else launch_linear2_inst<KSW_, 1, false, 0, true>(grid, NW, s, a);
#define QL2(KSW_) case KSW_: if (ssh) { QL2S(KSW_) } else if (db) { QL2D(KSW_, true) } else { QL2D(KSW_, false) } break;
    switch (KSW) {
        QL2(1) QL2(2) QL2(4) QL2(6) QL2(8) QL2(12)
        case 24: if (ssh) launch_linear2_inst<24, 0, false, 0, true>(grid, NW, s, a);
                 else if (pro == 0) launch_linear2_inst<24, 0, false, 0>(grid, NW, s, a);
                 else QFAIL(QTTS_EINVAL, "linear: activation prologues are not supported for K=%d", a.K);
                 break;
        default: QFAIL(QTTS_EINVAL, "linear: unsupported KSW %d", KSW);
    }
#undef QL2
#undef QL2S
#undef QL2D
    QLAUNCH();
    return QTTS_OK;
}

extern "C" int qtts_linear_fused(const void *x, int ldx, const int64_t *gather_ids, int gather_rows,
                                 const void *w_packed, const void *bias, const void *residual, int ldr,
                                 const void *norm_w, float eps, void *xnorm_out, int ldxn, int swiglu_input,
                                 void *out, int ldo, int M, int N, int K, qtts_stream_t stream) {
    QCHECK(!(swiglu_input && norm_w), "linear_fused: swiglu_input and norm_w are exclusive");
    QCHECK(!gather_ids || gather_rows > 0, "linear_fused: gather without a row count");
    LinArgs a;
    a.x = (const bf16_t *)x; a.ldx = ldx; a.ids = gather_ids; a.id_rows = gather_rows;
    a.wp = (const uint4 *)w_packed; a.bias = (const bf16_t *)bias; a.res = (const bf16_t *)residual; a.ldr = ldr;
    a.nw = (const bf16_t *)norm_w; a.eps = eps; a.xnorm_out = (bf16_t *)xnorm_out; a.ldxn = ldxn;
    a.out = (bf16_t *)out; a.ldo = ldo; a.M = M; a.N = N; a.K = K; a.tiles_per_wg = 1;
    return launch_linear2(a, swiglu_input ? 2 : (norm_w ? 1 : 0), (hipStream_t)stream);
}

extern "C" size_t qtts_linear_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 32 != 0) return 0;
    return linear_rows_scratch_bytes(M, N, K);
}

extern "C" int qtts_linear_ws(const void *x, int ldx, const void *w_packed, const void *bias, const void *residual, int ldr,
                              void *out, int ldo, int M, int N, int K, void *workspace, size_t workspace_bytes,
                              qtts_stream_t stream) {
    QCHECK(!workspace || ((uintptr_t)workspace & 15) == 0, "linear_ws: workspace must be 16-byte aligned");
    LinArgs a;
    a.x = (const bf16_t *)x; a.ldx = ldx; a.ids = nullptr; a.id_rows = 0;
    a.wp = (const uint4 *)w_packed; a.bias = (const bf16_t *)bias; a.res = (const bf16_t *)residual; a.ldr = ldr;
    a.nw = nullptr; a.eps = 0.f; a.xnorm_out = nullptr; a.ldxn = 0;
    a.out = (bf16_t *)out; a.ldo = ldo; a.M = M; a.N = N; a.K = K; a.tiles_per_wg = 1;
    return launch_linear2(a, 0, (hipStream_t)stream, false, (float *)workspace, workspace_bytes);
}

extern "C" int qtts_linear(const void *x, int ldx, const void *w_packed, const void *bias, const void *residual,
                           int ldr, const void *norm_w, float eps, void *out, int ldo, int M, int N, int K,
                           qtts_stream_t stream) {
    return qtts_linear_fused(x, ldx, nullptr, 0, w_packed, bias, residual, ldr, norm_w, eps, nullptr, 0, 0, out, ldo, M,
                             N, K, stream);
}

extern "C" int qtts_linear_swiglu(const void *x, int ldx, const void *w_gate_up_packed, const void *norm_w, float eps,
                                  void *out, int ldo, int M, int I, int K, qtts_stream_t stream) {
    QCHECK(I > 0 && I % 8 == 0, "linear_swiglu: I=%d must be a multiple of 8", I);
    QCHECK(ldo >= I, "linear_swiglu: out row stride %d < I=%d", ldo, I);
    LinArgs a;
    a.x = (const bf16_t *)x; a.ldx = ldx; a.ids = nullptr; a.id_rows = 0;
    a.wp = (const uint4 *)w_gate_up_packed; a.bias = nullptr; a.res = nullptr; a.ldr = 0;
    a.nw = (const bf16_t *)norm_w; a.eps = eps; a.xnorm_out = nullptr; a.ldxn = 0;
    a.out = (bf16_t *)out; a.ldo = ldo; a.M = M; a.N = 2 * I; a.K = K; a.tiles_per_wg = 1;
    return launch_linear2(a, norm_w ? 1 : 0, (hipStream_t)stream, true);
}

extern "C" int qtts_linear_fp8(const void *x, int ldx, const void *w_fp8_packed, const float *w_scales, const void *bias,
                               const void *residual, int ldr, const void *norm_w, float eps, void *out, int ldo, int M, int N,
                               int K, void *workspace, size_t workspace_bytes, qtts_stream_t stream) {
    QCHECK(w_scales, "linear_fp8: null scales");
    QCHECK(!workspace || ((uintptr_t)workspace & 15) == 0, "linear_fp8: workspace must be 16-byte aligned");
    LinArgs a;
    a.x = (const bf16_t *)x; a.ldx = ldx; a.ids = nullptr; a.id_rows = 0;
    a.wp = (const uint4 *)w_fp8_packed; a.wscale = w_scales; a.bias = (const bf16_t *)bias; a.res = (const bf16_t *)residual;
    a.ldr = ldr; a.nw = (const bf16_t *)norm_w; a.eps = eps; a.xnorm_out = nullptr; a.ldxn = 0;
    a.out = (bf16_t *)out; a.ldo = ldo; a.M = M; a.N = N; a.K = K; a.tiles_per_wg = 1;
    return launch_linear2(a, norm_w ? 1 : 0, (hipStream_t)stream, false, (float *)workspace, workspace_bytes);
}

extern "C" int qtts_linear_swiglu_fp8(const void *x, int ldx, const void *w_gate_up_fp8_packed, const float *w_scales,
                                      const void *norm_w, float eps, void *out, int ldo, int M, int I, int K,
                                      qtts_stream_t stream) {
    QCHECK(w_scales, "linear_swiglu_fp8: null scales");
    QCHECK(I > 0 && I % 8 == 0, "linear_swiglu_fp8: I=%d must be a multiple of 8", I);
    QCHECK(ldo >= I, "linear_swiglu_fp8: out row stride %d < I=%d", ldo, I);
    LinArgs a;
    a.x = (const bf16_t *)x; a.ldx = ldx; a.ids = nullptr; a.id_rows = 0;
    a.wp = (const uint4 *)w_gate_up_fp8_packed; a.wscale = w_scales; a.bias = nullptr; a.res = nullptr; a.ldr = 0;
    a.nw = (const bf16_t *)norm_w; a.eps = eps; a.xnorm_out = nullptr; a.ldxn = 0;
    a.out = (bf16_t *)out; a.ldo = ldo; a.M = M; a.N = 2 * I; a.K = K; a.tiles_per_wg = 1;
    return launch_linear2(a, norm_w ? 1 : 0, (hipStream_t)stream, true);
}

// ------------------------------------------------------------------ rmsnorm (stand-alone)
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t *__restrict__ x, int ldx, const bf16_t *__restrict__ w,
                                                      float eps, bf16_t *__restrict__ y, int ldy, int H, int ypks, int xpks) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    float ss = 0.f;
    for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
        Frag f;
        f.u = *reinterpret_cast<const uint4 *>(x + act_index(row, i, ldx, xpks));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = bf2f(f.h[j]);
            ss += v * v;
        }
    }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const float rstd = 1.0f / sqrtf(tot / (float)H + eps);
    for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
        const uint4 f = *reinterpret_cast<const uint4 *>(x + act_index(row, i, ldx, xpks));
        const uint4 wv = *reinterpret_cast<const uint4 *>(w + i);
        *reinterpret_cast<uint4 *>(y + act_index(row, i, ldy, ypks)) = norm_frag(f, rstd, wv);
    }
}

// ypks > 0: y in fragment order (act_index) for a prompt GEMM that reads it so; xpks > 0: x in fragment order
static int rmsnorm_impl(const void *x, int ldx, const void *w, float eps, void *y, int ldy, int M, int H, int ypks,
                        qtts_stream_t stream, int xpks = 0) {
    QCHECK(x && w && y, "rmsnorm: null pointer");
    QCHECK(M > 0 && H > 0 && H % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= H && ldy >= H,
           "rmsnorm: bad shape M=%d H=%d ldx=%d ldy=%d", M, H, ldx, ldy);
    QCHECK((ypks == 0 || ypks * 32 == H) && (xpks == 0 || xpks * 32 == H), "rmsnorm: fragment-order rows need H = 32 * k-steps");
    hipLaunchKernelGGL(rmsnorm_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)x, ldx,
                       (const bf16_t *)w, eps, (bf16_t *)y, ldy, H, ypks, xpks);
    QLAUNCH();
    return QTTS_OK;
}

extern "C" int qtts_rmsnorm(const void *x, int ldx, const void *w, float eps, void *y, int ldy, int M, int H,
                            qtts_stream_t stream) {
    return rmsnorm_impl(x, ldx, w, eps, y, ldy, M, H, 0, stream);
}

// ------------------------------------------------------------------ q/k norm + RoPE + KV store
// One wave per (token, head); lane i owns dims i and i+64 (the rotate_half pair).
__global__ __launch_bounds__(64) void qknorm_rope_store_kernel(
    bf16_t *__restrict__ qkv, int ld, const bf16_t *__restrict__ qw, const bf16_t *__restrict__ kw, float eps,
    const bf16_t *__restrict__ cosT, const bf16_t *__restrict__ sinT, const int64_t *__restrict__ positions,
    const int32_t *__restrict__ slots, bf16_t *__restrict__ kc, bf16_t *__restrict__ vc, int Hq, int Hkv, int max_pos) {
    const int tok = blockIdx.x, head = blockIdx.y, i = threadIdx.x;
    bf16_t *p = qkv + (size_t)tok * ld + (size_t)head * 128;
    const int slot = slots ? slots[tok] : -1;
    if (head >= Hq + Hkv) {                        // value head: cache copy only
        if (vc && slot >= 0) {
            bf16_t *dst = vc + ((size_t)slot * Hkv + (head - Hq - Hkv)) * 128;
            dst[i] = p[i];
            dst[i + 64] = p[i + 64];
        }
        return;
    }
    const bool isq = head < Hq;
    const bf16_t *w = isq ? qw : kw;
    const float x0 = bf2f(p[i]), x1 = bf2f(p[i + 64]);
    float ss = x0 * x0 + x1 * x1;
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float rstd = 1.0f / sqrtf(ss / 128.0f + eps);
    const float n0 = rbf(rbf(x0 * rstd) * bf2f(w[i]));
    const float n1 = rbf(rbf(x1 * rstd) * bf2f(w[i + 64]));
    long pos = positions[tok];
    pos = pos < 0 ? 0 : (pos >= max_pos ? max_pos - 1 : pos);          // metadata is device data: never index past the table
    const float c = bf2f(cosT[pos * 64 + i]), s = bf2f(sinT[pos * 64 + i]);
    // q*cos + rotate_half(q)*sin, each op rounded to bf16
    const bf16_t o0 = f2bf(rbf(n0 * c) + rbf(-n1 * s));
    const bf16_t o1 = f2bf(rbf(n1 * c) + rbf(n0 * s));
    p[i] = o0;
    p[i + 64] = o1;
    if (!isq && kc && slot >= 0) {
        bf16_t *dst = kc + ((size_t)slot * Hkv + (head - Hq)) * 128;
        dst[i] = o0;
        dst[i + 64] = o1;
    }
}

extern "C" int qtts_qknorm_rope_store(void *qkv, int ldqkv, const void *q_norm_w, const void *k_norm_w, float eps,
                                      const void *rope_cos, const void *rope_sin, int max_pos,
                                      const int64_t *positions, const int32_t *slot_mapping, void *k_cache,
                                      void *v_cache, int N, int Hq, int Hkv, qtts_stream_t stream) {
    QCHECK(qkv && q_norm_w && k_norm_w && rope_cos && rope_sin && positions, "qknorm_rope_store: null pointer");
    QCHECK(N > 0 && Hq > 0 && Hkv > 0 && ldqkv >= (Hq + 2 * Hkv) * 128, "qknorm_rope_store: bad shape");
    QCHECK(max_pos > 0, "qknorm_rope_store: empty rope table");
    hipLaunchKernelGGL(qknorm_rope_store_kernel, dim3(N, Hq + 2 * Hkv), dim3(64), 0, (hipStream_t)stream,
                       (bf16_t *)qkv, ldqkv, (const bf16_t *)q_norm_w, (const bf16_t *)k_norm_w, eps,
                       (const bf16_t *)rope_cos, (const bf16_t *)rope_sin, positions, slot_mapping, (bf16_t *)k_cache,
                       (bf16_t *)v_cache, Hq, Hkv, max_pos);
    QLAUNCH();
    return QTTS_OK;
}

__global__ __launch_bounds__(128) void store_kvcache_kernel(const bf16_t *__restrict__ k, int ldk,
                                                            const bf16_t *__restrict__ v, int ldv,
                                                            bf16_t *__restrict__ kc, bf16_t *__restrict__ vc,
                                                            const int32_t *__restrict__ slots, int D) {
    const int tok = blockIdx.x;
    const int slot = slots[tok];
    if (slot < 0) return;
    for (int i = threadIdx.x * 8; i < D; i += 128 * 8) {
        *reinterpret_cast<uint4 *>(kc + (size_t)slot * D + i) = *reinterpret_cast<const uint4 *>(k + (size_t)tok * ldk + i);
        *reinterpret_cast<uint4 *>(vc + (size_t)slot * D + i) = *reinterpret_cast<const uint4 *>(v + (size_t)tok * ldv + i);
    }
}

extern "C" int qtts_store_kvcache(const void *k, int ldk, const void *v, int ldv, void *k_cache, void *v_cache,
                                  const int32_t *slot_mapping, int N, int Hkv, qtts_stream_t stream) {
    QCHECK(k && v && k_cache && v_cache && slot_mapping, "store_kvcache: null pointer");
    QCHECK(N > 0 && Hkv > 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldk >= Hkv * 128 && ldv >= Hkv * 128,
           "store_kvcache: bad shape");
    hipLaunchKernelGGL(store_kvcache_kernel, dim3(N), dim3(128), 0, (hipStream_t)stream, (const bf16_t *)k, ldk,
                       (const bf16_t *)v, ldv, (bf16_t *)k_cache, (bf16_t *)v_cache, slot_mapping, Hkv * 128);
    QLAUNCH();
    return QTTS_OK;
}

// ------------------------------------------------------------------ paged attention
#define ATT_SPLIT 128   // tokens per workgroup
static_assert(QTTS_KV_BLOCK % ATT_SPLIT == 0, "a workgroup's tokens must lie in one KV block");
// grid (Hkv, rows, nsplit); 256 threads = 16 groups of 16 lanes; a group owns
// tokens split*128 + i*16 + grp, i < 8; a lane owns 8 of the 128 head dims.
template <int G>
__global__ __launch_bounds__(256) void paged_attn_kernel(
    const bf16_t *__restrict__ q, int ldq, const bf16_t *__restrict__ kc, const bf16_t *__restrict__ vc,
    const int32_t *__restrict__ block_tables, int max_blocks, const int32_t *__restrict__ row_seq,
    const int32_t *__restrict__ context_lens, float scale, bf16_t *__restrict__ out, int ldo, int Hkv,
    float *__restrict__ part_o, float *__restrict__ part_ml, int nsplit, int opks) {
    __shared__ float sm_acc[16][G][128];
    __shared__ float sm_m[16][G];
    __shared__ float sm_l[16][G];
    // grid (Hkv, rows, splits): workgroups are dealt to the XCDs in turn along x, so with 8 KV heads every workgroup of
    // head h runs on XCD h and that head's K/V rows, which all rows of a sequence read, are fetched into one L2 once
    // instead of into eight
    const int kvh = blockIdx.x, row = blockIdx.y, split = blockIdx.z;
    const int ctx = context_lens[row];
    const int Hq = Hkv * G;
    const int t0 = split * ATT_SPLIT;
    if (ctx <= 0) {
        if (split == 0 && nsplit == 1)
            for (int e = threadIdx.x; e < G * 128; e += 256) out[act_index(row, kvh * G * 128 + e, ldo, opks)] = 0;
        return;
    }
    if (t0 >= ctx) return;
    const int seq = row_seq ? row_seq[row] : row;
    const int32_t *bt = block_tables + (size_t)seq * max_blocks;
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;

    Frag kk[8], vv[8];
    // no branch around the loads (a conditional load is waited for inside its block, one round trip per token):
    // lanes past the context repeat its last token, which the `valid` test below keeps out of the sums
    const int blk_t0 = bt[t0 >> 8];                       // ATT_SPLIT divides the KV block: one table entry per workgroup
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int t = min(t0 + i * 16 + grp, ctx - 1);
        const size_t base = (((size_t)blk_t0 * QTTS_KV_BLOCK + (t & 255)) * Hkv + kvh) * 128 + 8 * l16;
        kk[i].u = *reinterpret_cast<const uint4 *>(kc + base);
        vv[i].u = *reinterpret_cast<const uint4 *>(vc + base);
    }
    float qf[G][8];
#pragma unroll
    for (int h = 0; h < G; ++h) {
        Frag f;
        f.u = *reinterpret_cast<const uint4 *>(q + (size_t)row * ldq + (size_t)(kvh * G + h) * 128 + 8 * l16);
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[h][j] = bf2f(f.h[j]);
    }
    float mx[G], ls[G], acc[G][8];
#pragma unroll
    for (int h = 0; h < G; ++h) {
        mx[h] = -INFINITY;
        ls[h] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[h][j] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int t = t0 + i * 16 + grp;
        const bool valid = t < ctx;                 // uniform within a 16-lane group
        float kf[8], vf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            kf[j] = bf2f(kk[i].h[j]);
            vf[j] = bf2f(vv[i].h[j]);
        }
#pragma unroll
        for (int h = 0; h < G; ++h) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += qf[h][j] * kf[j];
            s = row16_sum(s) * scale;
            if (valid) {
                const float mn = fmaxf(mx[h], s);
                const float corr = __expf(mx[h] - mn);
                const float p = __expf(s - mn);
                ls[h] = ls[h] * corr + p;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[h][j] = acc[h][j] * corr + p * vf[j];
                mx[h] = mn;
            }
        }
    }
#pragma unroll
    for (int h = 0; h < G; ++h) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sm_acc[grp][h][8 * l16 + j] = acc[h][j];
        if (l16 == 0) {
            sm_m[grp][h] = mx[h];
            sm_l[grp][h] = ls[h];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < G * 128; e += 256) {
        const int h = e >> 7, d = e & 127;
        float M = -INFINITY;
#pragma unroll
        for (int gI = 0; gI < 16; ++gI) M = fmaxf(M, sm_m[gI][h]);
        float L = 0.f, o = 0.f;
#pragma unroll
        for (int gI = 0; gI < 16; ++gI) {
            const float wgt = __expf(sm_m[gI][h] - M);      // exp(-inf) = 0 for empty groups
            L += sm_l[gI][h] * wgt;
            o += sm_acc[gI][h][d] * wgt;
        }
        const int hq = kvh * G + h;
        if (nsplit == 1) {
            out[act_index(row, hq * 128 + d, ldo, opks)] = f2bf(o / L);
        } else {
            const size_t pi = ((size_t)row * Hq + hq) * nsplit + split;
            part_o[pi * 128 + d] = o;
            if (d == 0) {
                part_ml[pi * 2 + 0] = M;
                part_ml[pi * 2 + 1] = L;
            }
        }
    }
}

__global__ __launch_bounds__(128) void attn_combine_kernel(const float *__restrict__ part_o,
                                                           const float *__restrict__ part_ml,
                                                           const int32_t *__restrict__ context_lens,
                                                           bf16_t *__restrict__ out, int ldo, int Hq, int nsplit,
                                                           int split_tokens, int opks) {
    const int row = blockIdx.x, hq = blockIdx.y, d = threadIdx.x;
    const int ctx = context_lens[row];
    if (ctx <= 0) {
        out[act_index(row, hq * 128 + d, ldo, opks)] = 0;
        return;
    }
    const int ns = (ctx + split_tokens - 1) / split_tokens;
    const size_t p0 = ((size_t)row * Hq + hq) * nsplit;
    float M = -INFINITY;
    for (int s = 0; s < ns; ++s) M = fmaxf(M, part_ml[(p0 + s) * 2]);
    float L = 0.f, o = 0.f;
    for (int s = 0; s < ns; ++s) {
        const float wgt = __expf(part_ml[(p0 + s) * 2] - M);
        L += part_ml[(p0 + s) * 2 + 1] * wgt;
        o += part_o[(p0 + s) * 128 + d] * wgt;
    }
    out[act_index(row, hq * 128 + d, ldo, opks)] = f2bf(o / L);
}

// ------------------------------------------------------------------ prompt attention on MFMA
// Rows of a prompt, 16 at a time: grid (Hkv, row tiles), 8 waves per workgroup; wave (tg, h) owns query head
// kvh*G + h of the tile's 16 rows and every (8/G)-th 32-token step of the context (a step is a chain of dependent
// round trips, so the steps of a context run side by side rather than one after the other), the token groups merge
// through LDS at the end (no partials in global memory, no combine launch).
//   S^T[token][row] = K Q^T   A = 16 cached K rows (one 16-byte fragment per lane straight from the cache),
//                             B = the 16 query rows; lane (g, r) ends up with row r's scores of tokens 4g..4g+3:
//                             a query row lives in 4 lanes, its softmax statistics need two cross-lane steps.
//   O^T[dim][row]  = V^T P^T  B = the lane's own 8 probabilities of the two 16-token tiles (bf16), A = V^T, read
//                             out of the wave's private LDS copy of the 32 V rows, two bytes at a time (the
//                             contraction runs over tokens, the cache holds a token's 128 dims contiguously).
// The rescale factor of the running maximum belongs to the lane's own row in both layouts.
// Rows of one tile may belong to different sequences (packed prompts): the sequences present in the tile are
// walked one after the other with the other rows masked.  Causality is context_lens[row] = position + 1.
#define PA_VROW 272           // bytes of one V row in LDS: 4 rows apart = 16 banks, conflict-free transposed reads
#define PA_WAVE_LDS (32 * PA_VROW)
// Token groups (waves per query head) of a workgroup: eight waves in all, so that a wave may hold a step's K and V
// fragments, the query rows and the accumulators at once (with sixteen waves the 128 registers each force the V
// rows through one temporary, a round trip per row).
#define PA_NTG(G) (8 / (G))
template <int G>
__global__ __launch_bounds__(512) void prefill_attn_mfma_kernel(
    const bf16_t *__restrict__ q, int ldq, const bf16_t *__restrict__ kc, const bf16_t *__restrict__ vc,
    const int32_t *__restrict__ block_tables, int max_blocks, const int32_t *__restrict__ row_seq,
    const int32_t *__restrict__ context_lens, float scale, bf16_t *__restrict__ out, int ldo, int N, int Hkv, int opks) {
    constexpr int NTG = PA_NTG(G);
    constexpr float NEG = -1e30f;
    extern __shared__ unsigned char pa_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int tg = wave / G, h = wave % G;
    // grid (Hkv, row tiles): the row tiles of a KV head share an XCD and its L2 (see paged_attn_kernel); the last
    // row tiles of a prompt walk the most tokens and are dispatched first, so that no long workgroup starts when the
    // others are about to end
    const int kvh = blockIdx.x, hq = kvh * G + h;
    const int m = ((int)gridDim.y - 1 - (int)blockIdx.y) * 16 + r;
    const bool rvalid = m < N;
    const int mc = rvalid ? m : N - 1;
    const int ctx = rvalid ? context_lens[mc] : 0;
    const int seq = row_seq ? row_seq[mc] : mc;
    unsigned char *vs = pa_smem + (size_t)wave * PA_WAVE_LDS;

    Frag qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks].u = *reinterpret_cast<const uint4 *>(q + (size_t)mc * ldq + (size_t)hq * 128 + 32 * ks + 8 * g);

    float mrun = NEG, lrun = 0.f;
    float4_t o[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) o[dt] = float4_t{0.f, 0.f, 0.f, 0.f};

    bool done = !(rvalid && ctx > 0);
    for (int pass = 0; pass < 16; ++pass) {
        const unsigned long long pend = __ballot(!done) & 0xffffull;       // lanes 0..15 carry the 16 rows
        if (pend == 0) break;
        const int lead = __ffsll((long long)pend) - 1;
        const int cur = __shfl(seq, lead);
        const bool act = !done && seq == cur;
        done = done || act;
        int cmax = act ? ctx : 0;
        cmax = max(cmax, __shfl_xor(cmax, 1));
        cmax = max(cmax, __shfl_xor(cmax, 2));
        cmax = max(cmax, __shfl_xor(cmax, 4));
        cmax = max(cmax, __shfl_xor(cmax, 8));
        const int32_t *bt = block_tables + (size_t)cur * max_blocks;
        const int nsteps = (cmax + 31) >> 5;
        const int climit = act ? ctx : 0;                                  // tokens this lane's row may see
        // A 32-token step lies in one KV block (32 divides the block size), tokens clamped to the longest row's
        // last one included (t0 <= cmax - 1): one table entry per step, requested one step ahead.
        int blk_next = bt[min((min(tg, max(nsteps - 1, 0)) * 32) >> 8, max_blocks - 1)];
        for (int step = tg; step < nsteps; step += NTG) {
            const int t0 = step * 32;
            const size_t bbase = (size_t)blk_next * QTTS_KV_BLOCK;
            blk_next = bt[min((min(step + NTG, nsteps - 1) * 32) >> 8, max_blocks - 1)];
            // K fragments of the two 16-token tiles (tokens past the longest row re-read its last token: masked)
            Frag ka[4], kb[4];
            {
                const int ta = min(t0 + r, cmax - 1), tb = min(t0 + 16 + r, cmax - 1);
                const bf16_t *pa = kc + ((bbase + (ta & 255)) * Hkv + kvh) * 128 + 8 * g;
                const bf16_t *pb = kc + ((bbase + (tb & 255)) * Hkv + kvh) * 128 + 8 * g;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    ka[ks].u = *reinterpret_cast<const uint4 *>(pa + 32 * ks);
                    kb[ks].u = *reinterpret_cast<const uint4 *>(pb + 32 * ks);
                }
            }
            // the 32 V rows into this wave's LDS region, as they lie in the cache: all eight requests first, then the
            // eight LDS writes (a write right behind its load makes every load a round trip of its own)
            Frag vrow[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int tv = min(t0 + 4 * i + g, cmax - 1);
                vrow[i].u = *reinterpret_cast<const uint4 *>(vc + ((bbase + (tv & 255)) * Hkv + kvh) * 128 + 8 * r);
            }
            // every row is in its registers before the first write: left to itself the compiler walks the rows
            // through one temporary (load, wait, write, next load)
            asm volatile("" : "+v"(vrow[0].v), "+v"(vrow[1].v), "+v"(vrow[2].v), "+v"(vrow[3].v), "+v"(vrow[4].v),
                              "+v"(vrow[5].v), "+v"(vrow[6].v), "+v"(vrow[7].v));
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<uint4 *>(vs + (4 * i + g) * PA_VROW + 16 * r) = vrow[i].u;
            // the rows are read back two bytes at a time by other lanes of this wave: LDS serves a wave's requests in
            // order, the compiler must not move the differently-typed reads across the writes either
            asm volatile("" ::: "memory");
            float4_t sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[ks].v, qf[ks].v, sa, 0, 0, 0);
                sb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kb[ks].v, qf[ks].v, sb, 0, 0, 0);
            }
            float sv[8];
            float rmax = NEG;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sv[i] = (t0 + 4 * g + i) < climit ? sa[i] * scale : NEG;
                sv[4 + i] = (t0 + 16 + 4 * g + i) < climit ? sb[i] * scale : NEG;
                rmax = fmaxf(rmax, fmaxf(sv[i], sv[4 + i]));
            }
            rmax = fmaxf(rmax, __shfl_xor(rmax, 16));
            rmax = fmaxf(rmax, __shfl_xor(rmax, 32));
            const float mnew = fmaxf(mrun, rmax);
            const float alpha = __expf(mrun - mnew);                       // 1 while the row has seen nothing
            mrun = mnew;
            // a probability enters the second product as the sum of two bf16 numbers (16 mantissa bits), so that the
            // result stays as close to the fp32 softmax of the definition as the per-row kernel's
            float psum = 0.f;
            Frag pf, pl;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float pr = sv[i] > 0.5f * NEG ? __expf(sv[i] - mnew) : 0.f;
                pf.h[i] = f2bf(pr);
                pl.h[i] = f2bf(pr - bf2f(pf.h[i]));
                psum += pr;
            }
            lrun = lrun * alpha + psum;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                o[dt] *= alpha;
                Frag vt;
                const unsigned char *col = vs + 2 * (16 * dt + r);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    vt.h[i] = *reinterpret_cast<const bf16_t *>(col + (4 * g + i) * PA_VROW);
                    vt.h[4 + i] = *reinterpret_cast<const bf16_t *>(col + (16 + 4 * g + i) * PA_VROW);
                }
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt.v, pf.v, o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt.v, pl.v, o[dt], 0, 0, 0);
            }
            asm volatile("" ::: "memory");
        }
    }
    // ---- merge the token groups of a head: (m, l, O^T) through the wave's LDS region
    lrun += __shfl_xor(lrun, 16);
    lrun += __shfl_xor(lrun, 32);
    __syncthreads();                                                       // (waves may still be reading V rows)
    {
        float4_t *po = reinterpret_cast<float4_t *>(vs);                   // [8 d-tiles][64 lanes]
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) po[dt * 64 + lane] = o[dt];
        float *pml = reinterpret_cast<float *>(vs + 8192);                 // [16 rows][2]
        if (g == 0) {
            pml[2 * r] = mrun;
            pml[2 * r + 1] = lrun;
        }
    }
    __syncthreads();
    if (tg != 0) return;
    float M = NEG;
#pragma unroll
    for (int k = 0; k < NTG; ++k) {
        const float *pml = reinterpret_cast<const float *>(pa_smem + (size_t)(k * G + h) * PA_WAVE_LDS + 8192);
        M = fmaxf(M, pml[2 * r]);
    }
    float L = 0.f, wk[NTG];
#pragma unroll
    for (int k = 0; k < NTG; ++k) {
        const float *pml = reinterpret_cast<const float *>(pa_smem + (size_t)(k * G + h) * PA_WAVE_LDS + 8192);
        wk[k] = __expf(pml[2 * r] - M);
        L += pml[2 * r + 1] * wk[k];
    }
    const float inv = L > 0.f ? 1.0f / L : 0.f;                            // a row without context gives zeros
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
        float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < NTG; ++k) {
            const float4_t *po = reinterpret_cast<const float4_t *>(pa_smem + (size_t)(k * G + h) * PA_WAVE_LDS);
            acc += po[dt * 64 + lane] * wk[k];
        }
        if (rvalid)
            *reinterpret_cast<uint2 *>(out + act_index(m, hq * 128 + 16 * dt + 4 * g, ldo, opks)) =
                make_uint2(pack2(acc[0] * inv, acc[1] * inv), pack2(acc[2] * inv, acc[3] * inv));
    }
}

static inline int attn_nsplit(int max_context) { return max_context <= ATT_SPLIT ? 1 : (max_context + ATT_SPLIT - 1) / ATT_SPLIT; }

extern "C" size_t qtts_attn_workspace_bytes(int N, int Hq, int max_context) {
    int ns = attn_nsplit(max_context);
    if (ns == 1) return 256;
    return (size_t)N * Hq * ns * (128 + 2) * sizeof(float) + 256;
}

// opks > 0: `out` in fragment order (act_index) for the prompt GEMM that reads it
static int paged_attention_impl(const void *q, int ldq, const void *k_cache, const void *v_cache,
                                const int32_t *block_tables, int max_blocks, const int32_t *row_seq,
                                const int32_t *context_lens, int max_context, float scale, void *out, int ldo,
                                int N, int Hq, int Hkv, void *workspace, qtts_stream_t stream, int opks) {
    QCHECK(q && k_cache && v_cache && block_tables && context_lens && out, "paged_attention: null pointer");
    QCHECK(N > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "paged_attention: bad head counts %d/%d", Hq, Hkv);
    QCHECK(ldq % 8 == 0 && ldq >= Hq * 128 && ldo >= Hq * 128, "paged_attention: bad strides");
    QCHECK(max_blocks > 0 && max_context > 0 && max_context <= max_blocks * QTTS_KV_BLOCK,
           "paged_attention: max_context %d exceeds %d blocks", max_context, max_blocks);
    const int G = Hq / Hkv;
    hipStream_t s = (hipStream_t)stream;
    if (row_seq && N >= g_mfma_attn_rows && (G == 1 || G == 2)) {
        // several rows per sequence (a prompt): 16 rows per workgroup on MFMA.  Below about a hundred rows the
        // per-row kernel is faster (a 48-token prompt: 6.6 against 10 us per layer; 200 tokens: 25 against 14 us)
        const dim3 grid(Hkv, (N + 15) / 16);
        const size_t shmem = (size_t)8 * PA_WAVE_LDS;                  // eight waves, whatever G is
        const void *kern = G == 1 ? reinterpret_cast<const void *>(&prefill_attn_mfma_kernel<1>)
                                  : reinterpret_cast<const void *>(&prefill_attn_mfma_kernel<2>);
        if (!allow_dynamic_lds(kern, shmem)) return QTTS_ESTATE;
        if (G == 1)
            hipLaunchKernelGGL((prefill_attn_mfma_kernel<1>), grid, dim3(512), shmem, s, (const bf16_t *)q, ldq,
                               (const bf16_t *)k_cache, (const bf16_t *)v_cache, block_tables, max_blocks, row_seq,
                               context_lens, scale, (bf16_t *)out, ldo, N, Hkv, opks);
        else
            hipLaunchKernelGGL((prefill_attn_mfma_kernel<2>), grid, dim3(512), shmem, s, (const bf16_t *)q, ldq,
                               (const bf16_t *)k_cache, (const bf16_t *)v_cache, block_tables, max_blocks, row_seq,
                               context_lens, scale, (bf16_t *)out, ldo, N, Hkv, opks);
        QLAUNCH();
        return QTTS_OK;
    }
    const int ns = attn_nsplit(max_context);
    QCHECK(ns == 1 || workspace, "paged_attention: workspace required for contexts > %d", ATT_SPLIT);
    QCHECK(N <= 65535, "paged_attention: at most 65535 rows per launch (got %d)", N);
    float *part_o = (float *)workspace;
    float *part_ml = part_o ? part_o + (size_t)N * Hq * ns * 128 : nullptr;
    dim3 grid(Hkv, N, ns);
#define QA(G_) case G_: hipLaunchKernelGGL((paged_attn_kernel<G_>), grid, dim3(256), 0, s, (const bf16_t *)q, ldq, \
        (const bf16_t *)k_cache, (const bf16_t *)v_cache, block_tables, max_blocks, row_seq, context_lens, scale, \
        (bf16_t *)out, ldo, Hkv, part_o, part_ml, ns, opks); break;
    switch (G) {
        QA(1) QA(2) QA(4)
        default: QFAIL(QTTS_EINVAL, "paged_attention: unsupported q/kv head ratio %d", G);
    }
#undef QA
    QLAUNCH();
    if (ns > 1) {
        hipLaunchKernelGGL(attn_combine_kernel, dim3(N, Hq), dim3(128), 0, s, part_o, part_ml, context_lens,
                           (bf16_t *)out, ldo, Hq, ns, ATT_SPLIT, opks);
        QLAUNCH();
    }
    return QTTS_OK;
}

extern "C" int qtts_paged_attention(const void *q, int ldq, const void *k_cache, const void *v_cache,
                                    const int32_t *block_tables, int max_blocks, const int32_t *row_seq,
                                    const int32_t *context_lens, int max_context, float scale, void *out, int ldo,
                                    int N, int Hq, int Hkv, void *workspace, qtts_stream_t stream) {
    return paged_attention_impl(q, ldq, k_cache, v_cache, block_tables, max_blocks, row_seq, context_lens, max_context,
                                scale, out, ldo, N, Hq, Hkv, workspace, stream, 0);
}


// ------------------------------------------------------------------ decode attention with fused q/k norm, RoPE, KV store
// One query row per sequence (decode).  Reads the raw q/k/v rows the qkv GEMM produced, applies the
// per-head RMSNorm and rotary embedding in registers (same rounding points as qknorm_rope_store_kernel),
// stores the new token's K/V into its cache slot and attends over context_lens[row] cached tokens, the
// newest one taken from registers.  grid (rows, Hkv, nsplit) with 128-token splits; 16 lanes per token,
// a lane owns 8 head dims (its rotate_half partner is lane ^ 8); a 16-lane group scores its tokens
// first and exponentiates against the group maximum afterwards (no per-token rescale of the accumulator).
// Rows longer than one split leave (max, sum, weighted V) partials in the workspace; the workgroup that
// arrives last at the row's counter merges them, so no second launch is needed.  The counters start at
// zero and the merging workgroup puts its counter back to zero.
#define ATT2_SPLIT 128
struct AttnFusedArgs {
    const bf16_t *qkv; int ld;
    const bf16_t *qw, *kw; float eps;
    const bf16_t *cosT, *sinT; int max_pos;
    const int64_t *positions;
    const int32_t *slots;
    bf16_t *kc, *vc;
    const int32_t *block_tables; int max_blocks;
    const int32_t *context_lens;
    float scale;
    bf16_t *out; int ldo;
    int Hq, Hkv;
    float *part_o, *part_ml; int nsplit;
    int *counters;
    const int64_t *row_ids; int id_rows;   // non-null: row r of the launch reads qkv row row_ids[r] (a table of rows)
    int opks;                              // > 0: `out` in fragment order (act_index) with this many k-steps
};

// wv: the lane's 8 norm weights; cv, sv: its 8 cos / sin values (dims d and d + 64 share them)
__device__ __forceinline__ void norm_rope8(const float (&x)[8], const Frag &wv, const Frag &cv, const Frag &sv, float eps,
                                           int l16, float (&o)[8]) {
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += x[j] * x[j];
    ss = row16_sum(ss);
    const float rstd = 1.0f / sqrtf(ss / 128.0f + eps);
    const bool hi = l16 >= 8;                         // dims 64..127
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float n = rbf(rbf(x[j] * rstd) * bf2f(wv.h[j]));
        const float partner = row16_xor8(n);          // the element 64 dims away
        const float c = bf2f(cv.h[j]), sn = bf2f(sv.h[j]);
        o[j] = rbf(rbf(n * c) + rbf((hi ? partner : -partner) * sn));
    }
}

// the same with the weight / cos / sin rows given as pointers
__device__ __forceinline__ void norm_rope8(const float (&x)[8], const bf16_t *__restrict__ w, float eps, int l16,
                                           const bf16_t *__restrict__ cosr, const bf16_t *__restrict__ sinr, float (&o)[8]) {
    const int d0 = 8 * l16;
    Frag wv, cv, sv;
    wv.u = *reinterpret_cast<const uint4 *>(w + d0);
    cv.u = *reinterpret_cast<const uint4 *>(cosr + (d0 & 63));
    sv.u = *reinterpret_cast<const uint4 *>(sinr + (d0 & 63));
    norm_rope8(x, wv, cv, sv, eps, l16, o);
}

__device__ __forceinline__ float ld_agent(const float *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(float *p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The first arguments are what the first loads need (16 dwords: they arrive in SGPRs with the wave, the struct
// behind them is fetched from the argument buffer); `p` repeats them for the host's convenience.
// IDS: the qkv rows come from a table, row r of the launch is table row p.row_ids[r] (one more dependent request
// in front of the row loads; everything else is unchanged).
// UNIFORM: every row r is sequence r with its tokens in KV block r, all at the same context length, which arrives
// as a scalar argument (in a_max_blocks' place; position = context - 1, slot = r * 256 + context - 1): a step of
// the code predictor's burst in the frame engine.  Nothing but the row ids (IDS) is device data then, so the cached
// K/V are requested with the wave's first instructions instead of one metadata round trip later.
template <int G, int NG, int TPG = ATT2_SPLIT / NG, bool IDS = false, bool UNIFORM = false>
__global__ __launch_bounds__(NG * 16) void attn_decode_fused_kernel(const int32_t *__restrict__ a_ctx,
                                                                    const int32_t *__restrict__ a_tables,
                                                                    const int64_t *__restrict__ a_pos,
                                                                    const int32_t *__restrict__ a_slots,
                                                                    const bf16_t *__restrict__ a_qkv, int a_max_blocks,
                                                                    int a_ld, int a_Hq, int a_Hkv, int a_max_pos,
                                                                    int a_nsplit, const AttnFusedArgs p) {
    constexpr int SPLIT = NG * TPG;                    // tokens per workgroup; TPG = tokens a 16-lane group walks
    constexpr int NT = NG * 16;
    __shared__ float sm_acc[NG][G][128];
    __shared__ float sm_m[NG][G];
    __shared__ float sm_l[NG][G];
    __shared__ __attribute__((aligned(16))) float sm_q[G + 1][128];
    __shared__ int sm_last;
    static_assert(NG >= G + 1, "one sixteen-lane group per head row to normalise");
    const int row = blockIdx.x, kvh = blockIdx.y, split = blockIdx.z;
    const int t0 = split * SPLIT;
    // A workgroup's tokens lie in one 256-token KV block (SPLIT divides 256): its table entry does not depend on the
    // context length, so both are requested together instead of one round trip after the other.
    static_assert(QTTS_KV_BLOCK % SPLIT == 0, "a workgroup's token range must not straddle KV blocks");
    // the row's metadata: four independent requests in one batch, no branch between them (a null slot array reads
    // the context length in its place and is sorted out afterwards)
    const int blk_all = UNIFORM ? row : a_tables[(size_t)row * a_max_blocks + (t0 >> 8)];
    // a context beyond what this launch's grid covers (the host's max_context) is cut to it: every split the
    // row counts on exists, so the arrival counter always completes and `out` is always written
    const int ctx = UNIFORM ? a_max_blocks : min(a_ctx[row], a_nsplit * SPLIT);
    const int slot_raw = UNIFORM ? row * QTTS_KV_BLOCK + a_max_blocks - 1 : (a_slots ? a_slots : a_ctx)[row];
    long pos = UNIFORM ? a_max_blocks - 1 : a_pos[row];
    long qrow = IDS ? p.row_ids[row] : row;
    // One wait for the whole batch, the struct's pointers included: without this the compiler moves each request
    // down to its first use, behind the early exits, and the wave pays one round trip after the other.
    asm volatile("" ::"s"(blk_all), "s"(ctx), "s"(slot_raw), "s"(pos), "s"(p.kc), "s"(p.vc), "s"(p.cosT), "s"(p.sinT),
                 "s"(p.qw), "s"(p.kw), "s"(p.out), "s"(qrow));
    if (IDS) qrow = qrow < 0 ? 0 : (qrow >= p.id_rows ? p.id_rows - 1 : qrow);
    const int slot = (UNIFORM || a_slots) ? slot_raw : -1;
    const int Hq = a_Hq, Hkv = a_Hkv;
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    // The G query heads and the new token's K are normalised and rotated once per workgroup, by the first G + 1
    // sixteen-lane groups (one head row each), and handed to the others through LDS: every group doing all of
    // them itself is G + 1 times the arithmetic on each SIMD, between the arrival of the row and the first score.
    // The row and the norm weights depend on no metadata and are requested with it (the other groups request the K
    // row as well and drop it: no branch around the loads).
    const bf16_t *rowp = a_qkv + (size_t)qrow * a_ld;
    const bool isq = grp < G;
    Frag xr, wv, vnew;
    xr.u = *reinterpret_cast<const uint4 *>(rowp + (size_t)(isq ? kvh * G + grp : Hq + kvh) * 128 + 8 * l16);
    wv.u = *reinterpret_cast<const uint4 *>((isq ? p.qw : p.kw) + 8 * l16);
    vnew.u = *reinterpret_cast<const uint4 *>(rowp + (size_t)(Hq + Hkv + kvh) * 128 + 8 * l16);
    if (ctx <= 0) {
        if (split == 0)
            for (int e = threadIdx.x; e < G * 128; e += NT) p.out[act_index(row, kvh * G * 128 + e, p.ldo, p.opks)] = 0;
        return;
    }
    if (t0 >= ctx) return;
    const int nact = (ctx + SPLIT - 1) / SPLIT;   // splits of this row that hold tokens
    pos = pos < 0 ? 0 : (pos >= a_max_pos ? a_max_pos - 1 : pos);      // metadata is device data: never index past the table
    Frag cv, sv;
    cv.u = *reinterpret_cast<const uint4 *>(p.cosT + pos * 64 + ((8 * l16) & 63));
    sv.u = *reinterpret_cast<const uint4 *>(p.sinT + pos * 64 + ((8 * l16) & 63));

    // Cached K/V are requested before the norm/rope arithmetic, behind what that arithmetic waits for, and without
    // a branch: a load under a condition is followed by the wait for it inside the conditional block (the merge
    // with the "no token" value), i.e. one round trip per token instead of all requests in flight together.  A
    // lane without a cached token (the new token, or past the context) reads this row's raw k/v instead: finite
    // values whose weight is zero, and for the new token V as it is wanted.
    Frag kk[TPG], vv[TPG];
    const bf16_t *krow = rowp + (size_t)(Hq + kvh) * 128 + 8 * l16;
    const bf16_t *vrow = rowp + (size_t)(Hq + Hkv + kvh) * 128 + 8 * l16;
#pragma unroll
    for (int i = 0; i < TPG; ++i) {
        const int t = t0 + i * NG + grp;
        const bool cached = t < ctx - 1;
        const size_t off = (((size_t)blk_all * QTTS_KV_BLOCK + (t & 255)) * Hkv + kvh) * 128 + 8 * l16;
        kk[i].u = *reinterpret_cast<const uint4 *>(cached ? p.kc + off : krow);
        vv[i].u = *reinterpret_cast<const uint4 *>(cached ? p.vc + off : vrow);
    }

    Frag knew;
    if (grp <= G) {
        float x[8], o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = bf2f(xr.h[j]);
        norm_rope8(x, wv, cv, sv, p.eps, l16, o);
        *reinterpret_cast<float4 *>(&sm_q[grp][8 * l16]) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4 *>(&sm_q[grp][8 * l16 + 4]) = make_float4(o[4], o[5], o[6], o[7]);
        // the new token (index ctx-1) goes to its cache slot: K normalised + rotated (exactly representable in
        // bf16), V as is
        if (!isq && slot >= 0 && (ctx - 1) >= t0 && (ctx - 1) < t0 + SPLIT) {
#pragma unroll
            for (int j = 0; j < 8; ++j) knew.h[j] = f2bf(o[j]);
            const size_t off = ((size_t)slot * Hkv + kvh) * 128 + 8 * l16;
            *reinterpret_cast<uint4 *>(p.kc + off) = knew.u;
            *reinterpret_cast<uint4 *>(p.vc + off) = vnew.u;
        }
    }
    __syncthreads();
    float qf[G][8];
#pragma unroll
    for (int h = 0; h < G; ++h) {
        const float4 a = *reinterpret_cast<const float4 *>(&sm_q[h][8 * l16]);
        const float4 b = *reinterpret_cast<const float4 *>(&sm_q[h][8 * l16 + 4]);
        qf[h][0] = a.x; qf[h][1] = a.y; qf[h][2] = a.z; qf[h][3] = a.w;
        qf[h][4] = b.x; qf[h][5] = b.y; qf[h][6] = b.z; qf[h][7] = b.w;
    }
    {
        const float4 a = *reinterpret_cast<const float4 *>(&sm_q[G][8 * l16]);
        const float4 b = *reinterpret_cast<const float4 *>(&sm_q[G][8 * l16 + 4]);
        knew.h[0] = f2bf(a.x); knew.h[1] = f2bf(a.y); knew.h[2] = f2bf(a.z); knew.h[3] = f2bf(a.w);
        knew.h[4] = f2bf(b.x); knew.h[5] = f2bf(b.y); knew.h[6] = f2bf(b.z); knew.h[7] = f2bf(b.w);
    }

    // scores of the group's tokens, then one maximum, then the exponentials
    const int rounds = min(TPG, (ctx - t0 + NG - 1) / NG);      // uniform: rounds that hold a token at all
    float sc[TPG][G];
#pragma unroll
    for (int i = 0; i < TPG; ++i) {
#pragma unroll
        for (int h = 0; h < G; ++h) sc[i][h] = -INFINITY;
        if (i >= rounds) continue;
        const int t = t0 + i * NG + grp;
        if (t == ctx - 1) kk[i].u = knew.u;            // its V is the raw row already loaded in its place
#pragma unroll
        for (int h = 0; h < G; ++h) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += qf[h][j] * bf2f(kk[i].h[j]);
            s = row16_sum(s) * p.scale;
            sc[i][h] = t < ctx ? s : -INFINITY;
        }
    }
    float mx[G], ls[G], acc[G][8];
#pragma unroll
    for (int h = 0; h < G; ++h) {
        mx[h] = sc[0][h];
#pragma unroll
        for (int i = 1; i < TPG; ++i) mx[h] = fmaxf(mx[h], sc[i][h]);
        ls[h] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[h][j] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < TPG; ++i) {
        if (i >= rounds) continue;
        float vf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) vf[j] = bf2f(vv[i].h[j]);
#pragma unroll
        for (int h = 0; h < G; ++h) {
            // a masked score is -inf: its weight is 0 unless the whole group is empty (-inf - -inf), hence the test
            const float pr = sc[i][h] == -INFINITY ? 0.f : __expf(sc[i][h] - mx[h]);
            ls[h] += pr;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[h][j] += pr * vf[j];
        }
    }
#pragma unroll
    for (int h = 0; h < G; ++h) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sm_acc[grp][h][8 * l16 + j] = acc[h][j];
        if (l16 == 0) {
            sm_m[grp][h] = mx[h];
            sm_l[grp][h] = ls[h];
        }
    }
    __syncthreads();
    const bool direct = nact == 1;
    for (int e = threadIdx.x; e < G * 128; e += NT) {
        const int h = e >> 7, d = e & 127;
        // every group's maximum, sum and accumulator element are read from LDS in batches before they are used (a
        // loop that reads and accumulates in turn waits out one LDS latency per group: up to 32 of them here)
        float mv[NG];
#pragma unroll
        for (int gI = 0; gI < NG; ++gI) mv[gI] = sm_m[gI][h];
        float M = -INFINITY;
#pragma unroll
        for (int gI = 0; gI < NG; ++gI) M = fmaxf(M, mv[gI]);
        float L = 0.f, o = 0.f;
        constexpr int MB = NG < 16 ? NG : 16;
#pragma unroll
        for (int g0 = 0; g0 < NG; g0 += MB) {
            float lv[MB], av[MB];
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                lv[j] = sm_l[g0 + j][h];
                av[j] = sm_acc[g0 + j][h][d];
            }
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                const float wgt = __expf(mv[g0 + j] - M);  // an empty group: exp(-inf) = 0
                L += lv[j] * wgt;
                o += av[j] * wgt;
            }
        }
        const int hq = kvh * G + h;
        if (direct) {
            p.out[act_index(row, hq * 128 + d, p.ldo, p.opks)] = f2bf(o / L);
        } else {
            const size_t pi = ((size_t)row * Hq + hq) * a_nsplit + split;
            st_agent(p.part_o + pi * 128 + d, o);
            if (d == 0) {
                st_agent(p.part_ml + pi * 2 + 0, M);
                st_agent(p.part_ml + pi * 2 + 1, L);
            }
        }
    }
    if (direct) return;
    // The last of the row's workgroups to arrive merges the partials.  Partials and counter are accessed
    // with device-scope atomics only (they go to the level all XCDs share), the barrier waits for this
    // workgroup's stores to be acknowledged before the counter moves, so no whole-cache write-back or
    // invalidate is needed.  This is an argument about gfx950, not about the HIP memory model (which would ask
    // for a release on the counter and an acquire on the merging side, i.e. exactly that write-back / invalidate,
    // measured at about 19 us per launch): agent-scope atomic stores and loads bypass the per-XCD L2 (sc1), stores
    // are counted by vmcnt until acknowledged, and the counter add below is issued after the wait and the barrier.
    // Another target needs this looked at again, hence the guard.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "attn_decode_fused_kernel: the relaxed partial/counter protocol is argued for gfx950 only"
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int old = __hip_atomic_fetch_add(p.counters + row * Hkv + kvh, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sm_last = old == nact - 1;
    }
    __syncthreads();
    if (!sm_last) return;
    for (int e = threadIdx.x; e < G * 128; e += NT) {
        const int h = e >> 7, d = e & 127;
        const int hq = kvh * G + h;
        const size_t p0 = ((size_t)row * Hq + hq) * a_nsplit;
        float M = -INFINITY, L = 0.f, o = 0.f;
        constexpr int MERGE_MAX = 8;                    // splits merged with all their partials requested at once
        if (nact <= MERGE_MAX) {
            // Every partial of the row in ONE batch of requests (a split past the row's last one re-reads that last
            // one and is left out of the sums): these loads go to the level all XCDs share, about a microsecond
            // each way, and a loop over the splits pays that once per split and again for the second pass.
            float pm[MERGE_MAX], pl[MERGE_MAX], po[MERGE_MAX];
#pragma unroll
            for (int s = 0; s < MERGE_MAX; ++s) {
                const int si = s < nact ? s : nact - 1;
                pm[s] = ld_agent(p.part_ml + (p0 + si) * 2);
                pl[s] = ld_agent(p.part_ml + (p0 + si) * 2 + 1);
                po[s] = ld_agent(p.part_o + (p0 + si) * 128 + d);
            }
#pragma unroll
            for (int s = 0; s < MERGE_MAX; ++s) M = fmaxf(M, pm[s]);          // (a repeated value changes nothing)
#pragma unroll
            for (int s = 0; s < MERGE_MAX; ++s) {
                const float wgt = __expf(pm[s] - M);
                L = s < nact ? L + pl[s] * wgt : L;
                o = s < nact ? o + po[s] * wgt : o;
            }
        } else {
            for (int s = 0; s < nact; ++s) M = fmaxf(M, ld_agent(p.part_ml + (p0 + s) * 2));
            for (int s = 0; s < nact; ++s) {
                const float wgt = __expf(ld_agent(p.part_ml + (p0 + s) * 2) - M);
                L += ld_agent(p.part_ml + (p0 + s) * 2 + 1) * wgt;
                o += ld_agent(p.part_o + (p0 + s) * 128 + d) * wgt;
            }
        }
        p.out[act_index(row, hq * 128 + d, p.ldo, p.opks)] = f2bf(o / L);
    }
    if (threadIdx.x == 0) __hip_atomic_store(p.counters + row * Hkv + kvh, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static inline int attn2_nsplit(int max_context) { return (max_context + ATT2_SPLIT - 1) / ATT2_SPLIT; }
static inline size_t attn2_partial_bytes(int B, int Hq, int max_context) {
    return ((size_t)B * Hq * attn2_nsplit(max_context) * (128 + 2) * sizeof(float) + 255) & ~(size_t)255;
}

static int attention_decode_fused_impl(const void *qkv, int ldqkv, const void *q_norm_w, const void *k_norm_w,
                                       float eps, const void *rope_cos, const void *rope_sin, int max_pos,
                                       const int64_t *positions, const int32_t *slot_mapping, void *k_cache,
                                       void *v_cache, const int32_t *block_tables, int max_blocks,
                                       const int32_t *context_lens, int max_context, float scale, void *out,
                                       int ldo, int B, int Hq, int Hkv, void *partials, int *counters,
                                       hipStream_t s, const int64_t *row_ids = nullptr, int id_rows = 0, int opks = 0) {
    // uniform rows (no metadata arrays at all): row r = sequence r in KV block r, every row at context max_context
    const bool uniform = !positions && !slot_mapping && !block_tables && !context_lens;
    QCHECK(qkv && q_norm_w && k_norm_w && rope_cos && rope_sin && k_cache && v_cache && out &&
               (uniform || (positions && block_tables && context_lens)), "attention_decode_fused: null pointer");
    QCHECK(!uniform || (max_context <= 32 && max_context <= QTTS_KV_BLOCK && Hq / Hkv <= 2),
           "attention_decode_fused: rows without metadata need max_context <= 32 and at most 2 query heads per KV head");
    if (uniform) max_blocks = 1;
    QCHECK(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && ldqkv % 8 == 0 && ldqkv >= (Hq + 2 * Hkv) * 128 &&
               ldo >= Hq * 128, "attention_decode_fused: bad shape");
    QCHECK(max_pos > 0 && max_blocks > 0 && max_context > 0 && max_context <= max_blocks * QTTS_KV_BLOCK,
           "attention_decode_fused: bad context bounds");
    // 128-token workgroups, except where 256-token ones (eight tokens per 16-lane group) pay: contexts of 129..256,
    // which then finish without the cross-workgroup merge (talker layer stack 1217 -> 1197 us at 200 tokens), and
    // launches whose 128-token workgroups would outnumber the 256 CUs (1459 -> 1392 us at 900 tokens; in between the
    // narrower split is 2-3 % faster)
    const int ns128 = (max_context + ATT2_SPLIT - 1) / ATT2_SPLIT;
    bool wide = Hq / Hkv <= 2 && ns128 > 1 && (ns128 == 2 || (long)B * Hkv * ns128 > 256);
    if (g_attn_wide && Hq / Hkv <= 2 && ns128 > 1) wide = g_attn_wide == 2;       // (tuning build: 1 = 128-token workgroups, 2 = 256-token)
    const int split = wide ? 2 * ATT2_SPLIT : ATT2_SPLIT;
    const int ns = (max_context + split - 1) / split;
    QCHECK(ns == 1 || (partials && counters), "attention_decode_fused: workspace required for contexts > %d", split);
    AttnFusedArgs a;
    a.qkv = (const bf16_t *)qkv; a.ld = ldqkv; a.qw = (const bf16_t *)q_norm_w; a.kw = (const bf16_t *)k_norm_w; a.eps = eps;
    a.cosT = (const bf16_t *)rope_cos; a.sinT = (const bf16_t *)rope_sin; a.max_pos = max_pos; a.positions = positions; a.slots = slot_mapping;
    a.kc = (bf16_t *)k_cache; a.vc = (bf16_t *)v_cache; a.block_tables = block_tables; a.max_blocks = max_blocks;
    a.context_lens = context_lens; a.scale = scale; a.out = (bf16_t *)out; a.ldo = ldo; a.Hq = Hq; a.Hkv = Hkv;
    a.part_o = (float *)partials; a.part_ml = a.part_o ? a.part_o + (size_t)B * Hq * ns * 128 : nullptr; a.nsplit = ns;
    a.counters = counters;
    a.row_ids = row_ids; a.id_rows = id_rows;
    QCHECK(opks == 0 || opks * 32 == Hq * 128, "attention_decode_fused: fragment-order output rows have Hq*128 columns (whole 16-row tiles)");
    a.opks = opks;
    dim3 grid(B, Hkv, ns);
    if (uniform) {
        // the context length travels in the max_blocks argument's place
#define QUNI(G_, IDS_) QTTS_LAUNCH_MAYBE_TIMED((attn_decode_fused_kernel<G_, 8, 4, IDS_, true>), grid, dim3(128), 0, s, a.context_lens, \
                                          a.block_tables, a.positions, a.slots, a.qkv, max_context, a.ld, a.Hq, a.Hkv, a.max_pos, a.nsplit, a)
        QCHECK(!row_ids || id_rows > 0, "attention_decode_fused: row ids without a table size");
        if (Hq / Hkv == 1) { if (row_ids) QUNI(1, true); else QUNI(1, false); }
        else { if (row_ids) QUNI(2, true); else QUNI(2, false); }
#undef QUNI
        QLAUNCH();
        return QTTS_OK;
    }
    if (row_ids) {
        // rows gathered from a table: built for the one-workgroup-per-(row, kv head) variants
        QCHECK(id_rows > 0 && (max_context <= 32 || Hq / Hkv == 4), "attention_decode_fused: gathered rows need max_context <= 32 (got %d)", max_context);
        switch (Hq / Hkv) {
            case 1: if (max_context <= 32) QTTS_LAUNCH_MAYBE_TIMED((attn_decode_fused_kernel<1, 8, 4, true>), grid, dim3(128), 0, s, a.context_lens, a.block_tables, a.positions, a.slots, a.qkv, a.max_blocks, a.ld, a.Hq, a.Hkv, a.max_pos, a.nsplit, a);
                    else QTTS_LAUNCH_MAYBE_TIMED((attn_decode_fused_kernel<1, 16, ATT2_SPLIT / 16, true>), grid, dim3(256), 0, s, a.context_lens, a.block_tables, a.positions, a.slots, a.qkv, a.max_blocks, a.ld, a.Hq, a.Hkv, a.max_pos, a.nsplit, a); break;
            case 2: if (max_context <= 32) QTTS_LAUNCH_MAYBE_TIMED((attn_decode_fused_kernel<2, 8, 4, true>), grid, dim3(128), 0, s, a.context_lens, a.block_tables, a.positions, a.slots, a.qkv, a.max_blocks, a.ld, a.Hq, a.Hkv, a.max_pos, a.nsplit, a);
                    else QTTS_LAUNCH_MAYBE_TIMED((attn_decode_fused_kernel<2, 16, ATT2_SPLIT / 16, true>), grid, dim3(256), 0, s, a.context_lens, a.block_tables, a.positions, a.slots, a.qkv, a.max_blocks, a.ld, a.Hq, a.Hkv, a.max_pos, a.nsplit, a); break;
            case 4: QTTS_LAUNCH_MAYBE_TIMED((attn_decode_fused_kernel<4, 16, ATT2_SPLIT / 16, true>), grid, dim3(256), 0, s, a.context_lens, a.block_tables, a.positions, a.slots, a.qkv, a.max_blocks, a.ld, a.Hq, a.Hkv, a.max_pos, a.nsplit, a); break;
            default: QFAIL(QTTS_EINVAL, "attention_decode_fused: unsupported q/kv head ratio %d", Hq / Hkv);
        }
        QLAUNCH();
        return QTTS_OK;
    }
    switch (Hq / Hkv) {
        // short contexts (the code predictor's 17 tokens) leave most of a 32-group workgroup idle: eight
        // 16-lane groups (two waves) of four tokens cover 32 tokens (burst 1669 us with sixteen groups of two tokens, 1664 us
        // with eight of four, 1712 us with four of eight; sixteen groups of eight tokens, 16 K/V requests per lane: 1788 us)
        case 1: if (max_context <= 32) QTTS_LAUNCH_MAYBE_TIMED((attn_decode_fused_kernel<1, 8, 4>), grid, dim3(128), 0, s, a.context_lens, a.block_tables, a.positions, a.slots, a.qkv, a.max_blocks, a.ld, a.Hq, a.Hkv, a.max_pos, a.nsplit, a);
                else if (wide) QTTS_LAUNCH_MAYBE_TIMED((attn_decode_fused_kernel<1, 32, 8>), grid, dim3(512), 0, s, a.context_lens, a.block_tables, a.positions, a.slots, a.qkv, a.max_blocks, a.ld, a.Hq, a.Hkv, a.max_pos, a.nsplit, a);
                else QTTS_LAUNCH_MAYBE_TIMED((attn_decode_fused_kernel<1, 32>), grid, dim3(512), 0, s, a.context_lens, a.block_tables, a.positions, a.slots, a.qkv, a.max_blocks, a.ld, a.Hq, a.Hkv, a.max_pos, a.nsplit, a);
                break;
        case 2: if (max_context <= 32) QTTS_LAUNCH_MAYBE_TIMED((attn_decode_fused_kernel<2, 8, 4>), grid, dim3(128), 0, s, a.context_lens, a.block_tables, a.positions, a.slots, a.qkv, a.max_blocks, a.ld, a.Hq, a.Hkv, a.max_pos, a.nsplit, a);
                else if (wide) QTTS_LAUNCH_MAYBE_TIMED((attn_decode_fused_kernel<2, 32, 8>), grid, dim3(512), 0, s, a.context_lens, a.block_tables, a.positions, a.slots, a.qkv, a.max_blocks, a.ld, a.Hq, a.Hkv, a.max_pos, a.nsplit, a);
                else QTTS_LAUNCH_MAYBE_TIMED((attn_decode_fused_kernel<2, 32>), grid, dim3(512), 0, s, a.context_lens, a.block_tables, a.positions, a.slots, a.qkv, a.max_blocks, a.ld, a.Hq, a.Hkv, a.max_pos, a.nsplit, a);
                break;
        case 4: QTTS_LAUNCH_MAYBE_TIMED((attn_decode_fused_kernel<4, 16>), grid, dim3(256), 0, s, a.context_lens, a.block_tables, a.positions, a.slots, a.qkv, a.max_blocks, a.ld, a.Hq, a.Hkv, a.max_pos, a.nsplit, a); break;
        default: QFAIL(QTTS_EINVAL, "attention_decode_fused: unsupported q/kv head ratio %d", Hq / Hkv);
    }
    QLAUNCH();
    return QTTS_OK;
}

// ---- the 2-position prompt of a fresh sequence (the code predictor's burst starts with one per sequence): rows
// (2i, 2i+1) are positions (p, p+1) of sequence i, contexts 1 and 2.  q/k norm + RoPE + KV store + attention of
// both rows in one launch, one (sequence, kv head) per workgroup, sixteen lanes per query head.  Row 2i attends
// to itself only (weight exactly 1: its output is its V row); row 2i+1 to both, the first row's K taken from
// registers, not from the cache it was just written to.
template <int G>
__global__ __launch_bounds__(64) void attn_pair_kernel(const bf16_t *__restrict__ qkv, int ld,
                                                       const int64_t *__restrict__ positions,
                                                       const int32_t *__restrict__ slots, const bf16_t *__restrict__ qw,
                                                       const bf16_t *__restrict__ kw, const bf16_t *__restrict__ cosT,
                                                       const bf16_t *__restrict__ sinT, int max_pos, float eps,
                                                       bf16_t *__restrict__ kc, bf16_t *__restrict__ vc, float scale,
                                                       bf16_t *__restrict__ out, int ldo, int Hq, int Hkv, int opks) {
    const int seq = blockIdx.x, kvh = blockIdx.y;
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    if (grp >= G) return;
    // every request of the launch goes out before the first use
    long pos[2];
    int slot[2];
    Frag kraw[2], vb[2], fq, qwv, kwv;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int row = 2 * seq + rr;
        const bf16_t *rowp = qkv + (size_t)row * ld;
        pos[rr] = positions[row];
        slot[rr] = (slots ? slots : reinterpret_cast<const int32_t *>(positions))[row];
        kraw[rr].u = *reinterpret_cast<const uint4 *>(rowp + (size_t)(Hq + kvh) * 128 + 8 * l16);
        vb[rr].u = *reinterpret_cast<const uint4 *>(rowp + (size_t)(Hq + Hkv + kvh) * 128 + 8 * l16);
    }
    fq.u = *reinterpret_cast<const uint4 *>(qkv + (size_t)(2 * seq + 1) * ld + (size_t)(kvh * G + grp) * 128 + 8 * l16);
    qwv.u = *reinterpret_cast<const uint4 *>(qw + 8 * l16);
    kwv.u = *reinterpret_cast<const uint4 *>(kw + 8 * l16);
    Frag cv[2], sv[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const long p = pos[rr] < 0 ? 0 : (pos[rr] >= max_pos ? max_pos - 1 : pos[rr]);
        cv[rr].u = *reinterpret_cast<const uint4 *>(cosT + p * 64 + ((8 * l16) & 63));
        sv[rr].u = *reinterpret_cast<const uint4 *>(sinT + p * 64 + ((8 * l16) & 63));
    }
    Frag kb[2];
    float x[8], o8[8];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = bf2f(kraw[rr].h[j]);
        norm_rope8(x, kwv, cv[rr], sv[rr], eps, l16, o8);
#pragma unroll
        for (int j = 0; j < 8; ++j) kb[rr].h[j] = f2bf(o8[j]);
        const int sl = slots ? slot[rr] : -1;
        if (sl >= 0 && grp == 0) {
            const size_t off = ((size_t)sl * Hkv + kvh) * 128 + 8 * l16;
            *reinterpret_cast<uint4 *>(kc + off) = kb[rr].u;
            *reinterpret_cast<uint4 *>(vc + off) = vb[rr].u;
        }
    }
    float q1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = bf2f(fq.h[j]);
    norm_rope8(x, qwv, cv[1], sv[1], eps, l16, q1);
    const int hq = kvh * G + grp;
    *reinterpret_cast<uint4 *>(out + act_index(2 * seq, hq * 128 + 8 * l16, ldo, opks)) = vb[0].u;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s0 += q1[j] * bf2f(kb[0].h[j]);
        s1 += q1[j] * bf2f(kb[1].h[j]);
    }
    s0 = row16_sum(s0) * scale;
    s1 = row16_sum(s1) * scale;
    const float mx = fmaxf(s0, s1);
    const float p0 = __expf(s0 - mx), p1 = __expf(s1 - mx);
    const float L = p0 + p1;
    Frag o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float acc = p0 * bf2f(vb[0].h[j]);
        acc += p1 * bf2f(vb[1].h[j]);
        o.h[j] = f2bf(acc / L);
    }
    *reinterpret_cast<uint4 *>(out + act_index(2 * seq + 1, hq * 128 + 8 * l16, ldo, opks)) = o.u;
}

static int attention_pairs_impl(const void *qkv, int ldqkv, const void *q_norm_w, const void *k_norm_w, float eps,
                                const void *rope_cos, const void *rope_sin, int max_pos, const int64_t *positions,
                                const int32_t *slot_mapping, void *k_cache, void *v_cache, float scale, void *out, int ldo,
                                int N, int Hq, int Hkv, hipStream_t s, int opks = 0) {
    QCHECK(qkv && q_norm_w && k_norm_w && rope_cos && rope_sin && positions && k_cache && v_cache && out,
           "attention_pairs: null pointer");
    QCHECK(opks == 0 || (N <= 16 && opks * 32 == Hq * 128), "attention_pairs: fragment-order output is one 16-row tile of Hq*128 columns");
    QCHECK(N > 0 && N % 2 == 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && ldqkv % 8 == 0 && ldqkv >= (Hq + 2 * Hkv) * 128 &&
               ldo >= Hq * 128 && max_pos > 0, "attention_pairs: bad shape");
    dim3 grid(N / 2, Hkv);
#define QPAIR(G_) hipLaunchKernelGGL((attn_pair_kernel<G_>), grid, dim3(64), 0, s, (const bf16_t *)qkv, ldqkv, positions, \
                                     slot_mapping, (const bf16_t *)q_norm_w, (const bf16_t *)k_norm_w, (const bf16_t *)rope_cos, \
                                     (const bf16_t *)rope_sin, max_pos, eps, (bf16_t *)k_cache, (bf16_t *)v_cache, scale, \
                                     (bf16_t *)out, ldo, Hq, Hkv, opks)
    switch (Hq / Hkv) {
        case 1: QPAIR(1); break;
        case 2: QPAIR(2); break;
        case 4: QPAIR(4); break;
        default: QFAIL(QTTS_EINVAL, "attention_pairs: unsupported q/kv head ratio %d", Hq / Hkv);
    }
#undef QPAIR
    QLAUNCH();
    return QTTS_OK;
}

extern "C" size_t qtts_attn_decode_workspace_bytes(int B, int Hq, int Hkv, int max_context) {
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || max_context <= 0) return 0;
    return attn2_partial_bytes(B, Hq, max_context) + (((size_t)B * Hkv * sizeof(int) + 255) & ~(size_t)255);
}

extern "C" int qtts_attention_decode_fused(const void *qkv, int ldqkv, const void *q_norm_w, const void *k_norm_w,
                                           float eps, const void *rope_cos, const void *rope_sin, int max_pos,
                                           const int64_t *positions, const int32_t *slot_mapping, void *k_cache,
                                           void *v_cache, const int32_t *block_tables, int max_blocks,
                                           const int32_t *context_lens, int max_context, float scale, void *out,
                                           int ldo, int B, int Hq, int Hkv, void *workspace, qtts_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    int *counters = nullptr;
    if (workspace && B > 0 && Hq > 0 && Hkv > 0 && max_context > 0) {
        // the arrival counters sit behind the partials and must be zero when the kernel starts
        counters = (int *)((char *)workspace + attn2_partial_bytes(B, Hq, max_context));
        if (max_context > ATT2_SPLIT) {
            hipError_t e = hipMemsetAsync(counters, 0, (size_t)B * Hkv * sizeof(int), s);
            if (e != hipSuccess) QFAIL((int)e, "attention_decode_fused: memset failed: %s", hipGetErrorString(e));
        }
    }
    return attention_decode_fused_impl(qkv, ldqkv, q_norm_w, k_norm_w, eps, rope_cos, rope_sin, max_pos, positions,
                                       slot_mapping, k_cache, v_cache, block_tables, max_blocks, context_lens,
                                       max_context, scale, out, ldo, B, Hq, Hkv, workspace, counters, s);
}

// ---- the same entry point for callers that cannot allocate: the partials and counters live in a static device
// array instead of a caller-provided workspace.  Each GPU holds its own copy of the array at its own address; host
// code only ever uses the address hipGetSymbolAddress returns for the current device (the array's name is a
// host-side stand-in, not a device address).
#define QTTS_STATIC_ROWS_X_HEADS 256                 // B * Hq at most (e.g. 16 sequences x 16 query heads)
#define QTTS_STATIC_MAX_CONTEXT 1024
#define QTTS_STATIC_PARTIAL_BYTES ((size_t)QTTS_STATIC_ROWS_X_HEADS * (QTTS_STATIC_MAX_CONTEXT / ATT2_SPLIT) * (128 + 2) * sizeof(float))
#define QTTS_STATIC_COUNTER_BYTES ((size_t)QTTS_STATIC_ROWS_X_HEADS * sizeof(int))       // B * Hkv <= B * Hq
static __device__ __align__(256) unsigned char scratch[QTTS_STATIC_PARTIAL_BYTES + QTTS_STATIC_COUNTER_BYTES];

#define QTTS_MAX_DEVICES 64
static void *g_scratch_ptr[QTTS_MAX_DEVICES];        // per device: the address of that device's `scratch`
static std::mutex g_scratch_mutex;

// The current device's scratch array (looked up once per device, with that device current), or null + error text.
static unsigned char *static_scratch_for_current_device() {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= QTTS_MAX_DEVICES) {
        snprintf(g_err, sizeof(g_err), "static scratch: no usable current device (%s)", hipGetErrorString(e));
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void *ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) {
            snprintf(g_err, sizeof(g_err), "static scratch: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
            return nullptr;
        }
        g_scratch_ptr[dev] = ptr;
    }
    return (unsigned char *)g_scratch_ptr[dev];
}

extern "C" int qtts_attention_decode_fused_static(const void *qkv, int ldqkv, const void *q_norm_w, const void *k_norm_w,
                                                  float eps, const void *rope_cos, const void *rope_sin, int max_pos,
                                                  const int64_t *positions, const int32_t *slot_mapping, void *k_cache,
                                                  void *v_cache, const int32_t *block_tables, int max_blocks,
                                                  const int32_t *context_lens, int max_context, float scale, void *out,
                                                  int ldo, int B, int Hq, int Hkv, qtts_stream_t stream) {
    QCHECK(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "attention_decode_fused_static: bad shape");
    QCHECK((long)B * Hq <= QTTS_STATIC_ROWS_X_HEADS,
           "attention_decode_fused_static: B * Hq = %ld exceeds the static scratch's %d (use qtts_attention_decode_fused with a workspace)",
           (long)B * Hq, QTTS_STATIC_ROWS_X_HEADS);
    QCHECK(max_context > 0 && max_context <= QTTS_STATIC_MAX_CONTEXT,
           "attention_decode_fused_static: max_context %d exceeds the static scratch's %d", max_context, QTTS_STATIC_MAX_CONTEXT);
    unsigned char *ptr = static_scratch_for_current_device();
    if (!ptr) return QTTS_ESTATE;
    hipStream_t s = (hipStream_t)stream;
    // the array keeps the previous call's contents: partials are written before they are read, the arrival counters
    // must start at zero
    int *counters = (int *)(ptr + QTTS_STATIC_PARTIAL_BYTES);
    if (max_context > ATT2_SPLIT) {
        hipError_t e = hipMemsetAsync(counters, 0, (size_t)B * Hkv * sizeof(int), s);
        if (e != hipSuccess) QFAIL((int)e, "attention_decode_fused_static: memset failed: %s", hipGetErrorString(e));
    }
    return attention_decode_fused_impl(qkv, ldqkv, q_norm_w, k_norm_w, eps, rope_cos, rope_sin, max_pos, positions,
                                       slot_mapping, k_cache, v_cache, block_tables, max_blocks, context_lens,
                                       max_context, scale, out, ldo, B, Hq, Hkv, ptr, counters, s);
}

// ------------------------------------------------------------------ sampling
__global__ __launch_bounds__(256) void argmax_kernel(const bf16_t *__restrict__ logits, int ld, int64_t *__restrict__ out, int ldout, int V) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int row = blockIdx.x;
    const bf16_t *p = logits + (size_t)row * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < V; i += 256) {
        float v = bf2f(p[i]);
        if (v > best || (v == best && i < bi)) {
            best = v;
            bi = i;
        }
    }
    if (bi == 0x7fffffff && threadIdx.x < V) bi = threadIdx.x;   // all -inf / NaN rows
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o);
        int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        sv[threadIdx.x >> 6] = best;
        si[threadIdx.x >> 6] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) {
                best = sv[w];
                bi = si[w];
            }
        out[(size_t)row * ldout] = bi;
    }
}

extern "C" int qtts_argmax(const void *logits, int ld, int64_t *out, int M, int V, qtts_stream_t stream) {
    QCHECK(logits && out && M > 0 && V > 0 && ld >= V, "argmax: bad arguments");
    hipLaunchKernelGGL(argmax_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)logits, ld, out, 1, V);
    QLAUNCH();
    return QTTS_OK;
}

// splitmix64-based counter RNG: uniform in [0,1)
__device__ __forceinline__ float rng_uniform(uint64_t seed, uint64_t offset, uint64_t row) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (offset * 0x100000001B3ull + row + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// Order-preserving map float -> uint32 (larger float <-> larger key) and back.
__device__ __forceinline__ uint32_t fkey(float v) {
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
// Inclusive prefix sums over the 256 threads of a workgroup (thread order); sm: 4 words of LDS scratch.
template <typename T>
__device__ __forceinline__ T block_scan256(T v, T *sm) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    T x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    if (lane == 63) sm[w] = x;
    __syncthreads();
    T off = 0;
    for (int i = 0; i < w; ++i) off += sm[i];
    __syncthreads();
    return x + off;
}

// One workgroup per row; V <= 4096 values staged in LDS.  Greedy rows: argmax, lowest index on ties.  Sampling rows
// (P/layers/sampler.py:15-24: logits / T, everything below the k-th largest value removed, softmax, one draw):
// the k-th largest value by a 4-pass radix select over the order-preserving keys (a 256-bin histogram per byte),
// then the inverse CDF in index order with a chunk of consecutive indices per thread and a workgroup prefix sum.
// Argument order: what the first requests need comes first (the leading 16 dwords arrive in SGPRs with the wave).
__global__ __launch_bounds__(256) void sample_topk_kernel(const bf16_t *__restrict__ logits, int ld, int V,
                                                          const float *__restrict__ temps,
                                                          const uint8_t *__restrict__ greedy,
                                                          const uint64_t *__restrict__ offset_dev, int top_k, int ldout,
                                                          int64_t *__restrict__ out, uint64_t seed, uint64_t offset) {
    __shared__ float val[4096];
    __shared__ int hist[256];
    __shared__ float wv[4];
    __shared__ int wi[4];
    __shared__ int sel[2];
    __shared__ float fs[4];
    __shared__ int picked;
    const int row = blockIdx.x;
    const bf16_t *lr = logits + (size_t)row * ld;
    const bool vec = ((V | ld) & 7) == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0;
    // Every request of the prologue goes out before the first wait: the row (16 bytes per request, at most two per
    // thread for V <= 4096), the row's greedy flag and temperature, the RNG offset.  Optional arrays that are absent
    // read a valid word in their place (sorted out below), so no request sits behind a branch.
    const int nch = V >> 3;
    Frag f[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
        f[c].u = *reinterpret_cast<const uint4 *>(
            vec ? lr + 8 * min((int)threadIdx.x + 256 * c, nch - 1)
                : reinterpret_cast<const bf16_t *>(reinterpret_cast<uintptr_t>(logits) & ~(uintptr_t)15));   // unused
    const uint8_t gflag = (greedy ? greedy + row : reinterpret_cast<const uint8_t *>(temps))[0];
    const float Traw = temps[row];
    const uint64_t odev = *(offset_dev ? offset_dev : reinterpret_cast<const uint64_t *>(out));
    asm volatile("" ::"v"((int)gflag), "s"(Traw), "s"(odev));
    const bool is_greedy = greedy && gflag;
    const float T = is_greedy ? 1.0f : Traw;
    if (offset_dev) offset += odev;
    // maximum (and its lowest index) while staging
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if (vec) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ch = threadIdx.x + 256 * c;
            if (ch < nch) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = bf2f(f[c].h[j]) / T;
                    val[8 * ch + j] = v;
                    if (v > best) {       // ascending index per thread: the first maximum is kept
                        best = v;
                        bi = 8 * ch + j;
                    }
                }
            }
        }
    } else {
        for (int i = threadIdx.x; i < V; i += 256) {
            const float v = bf2f(lr[i]) / T;
            val[i] = v;
            if (v > best) {               // ascending i per thread: the first maximum is kept
                best = v;
                bi = i;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        wv[threadIdx.x >> 6] = best;
        wi[threadIdx.x >> 6] = bi;
    }
    if (threadIdx.x == 0) picked = -1;
    __syncthreads();
    best = wv[0];
    bi = wi[0];
    for (int w = 1; w < 4; ++w)
        if (wv[w] > best || (wv[w] == best && wi[w] < bi)) {
            best = wv[w];
            bi = wi[w];
        }
    if (is_greedy || top_k <= 1) {
        if (threadIdx.x == 0) out[(size_t)row * ldout] = bi;
        return;
    }
    const float maxv = best;
    // ---- k-th largest value
    float kth;
    if (vec) {
        // logits / T is monotone in the bf16 logit (T > 0), so the k-th largest is found on the 16-bit keys of the raw
        // logits, which every thread still holds in registers: two 8-bit passes instead of four over the 32-bit
        // keys of the staged values, no LDS reads in a pass, and three barriers per pass (the histogram of the next
        // pass is cleared while this one is scanned).
        __shared__ int hist2[2][256];
        uint32_t k16[2][8];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint32_t u = f[c].h[j];
                if (u == 0x8000u) u = 0;                              // -0 and +0 are one value
                k16[c][j] = (u & 0x8000u) ? (~u & 0xffffu) : (u | 0x8000u);
            }
        hist2[0][threadIdx.x] = 0;
        __syncthreads();
        uint32_t prefix = 0, mask = 0;
        int krem = top_k;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int shift = 8 - 8 * pass;
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if ((int)threadIdx.x + 256 * c < nch) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if ((k16[c][j] & mask) == prefix) atomicAdd(&hist2[pass][(k16[c][j] >> shift) & 255], 1);
                }
            __syncthreads();
            const int cnt = hist2[pass][255 - threadIdx.x];           // thread t looks at bin 255 - t: counts from the top
            if (pass == 0) hist2[1][threadIdx.x] = 0;
            int x = cnt;
            const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int y = __shfl_up(x, o);
                if (lane >= o) x += y;
            }
            if (lane == 63) wi[w] = x;
            __syncthreads();
            int off = 0;
            for (int i = 0; i < w; ++i) off += wi[i];
            const int incl = x + off;
            if (incl >= krem && incl - cnt < krem) {                  // exactly one thread
                sel[0] = 255 - threadIdx.x;
                sel[1] = krem - (incl - cnt);
            }
            __syncthreads();
            prefix |= (uint32_t)sel[0] << shift;
            mask |= 0xffu << shift;
            krem = sel[1];
        }
        const uint32_t kb = (prefix & 0x8000u) ? (prefix & 0x7fffu) : (~prefix & 0xffffu);
        kth = bf2f((bf16_t)kb) / T;
    } else {
    uint32_t prefix = 0, mask = 0;
    int krem = top_k;
#pragma unroll 1
    for (int shift = 24; shift >= 0; shift -= 8) {
        hist[threadIdx.x] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < V; i += 256) {
            const uint32_t key = fkey(val[i]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1);
        }
        __syncthreads();
        const int c = hist[255 - threadIdx.x];                   // thread t looks at bin 255 - t: counts from the top
        const int incl = block_scan256<int>(c, wi);
        if (incl >= krem && incl - c < krem) {                    // exactly one thread
            sel[0] = 255 - threadIdx.x;
            sel[1] = krem - (incl - c);
        }
        __syncthreads();
        prefix |= (uint32_t)sel[0] << shift;
        mask |= 0xffu << shift;
        krem = sel[1];
        __syncthreads();
    }
    kth = fkey_inv(prefix);
    }
    // ---- softmax weights over values >= kth (ties at the threshold all stay, as `logits < kth` removes only smaller
    // ones), a chunk of consecutive indices per thread
    const int C = (V + 255) >> 8;
    const int i0 = threadIdx.x * C, i1 = min(i0 + C, V);
    float part = 0.f;
    for (int i = i0; i < i1; ++i) {
        const float v = val[i];
        const float p = (v >= kth) ? __expf(v - maxv) : 0.f;
        val[i] = p;
        part += p;
    }
    const float incl = block_scan256<float>(part, fs);
    if (threadIdx.x == 255) wv[0] = incl;
    __syncthreads();
    const float total = wv[0];
    const float u = rng_uniform(seed, offset, row) * total;
    const float excl = incl - part;
    if (part > 0.f && u >= excl && u < incl) {
        float c = excl;
        int pick = -1, lastnz = i0;
        for (int i = i0; i < i1; ++i) {
            const float p = val[i];
            if (p > 0.f) {
                lastnz = i;
                c += p;
                if (c > u) {
                    pick = i;
                    break;
                }
            }
        }
        picked = pick >= 0 ? pick : lastnz;
    }
    __syncthreads();
    if (picked < 0) {
        // u fell on the total through rounding: the last index that has weight
        int last = -1;
        for (int i = i0; i < i1; ++i)
            if (val[i] > 0.f) last = i;
        for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o));
        if ((threadIdx.x & 63) == 0) wi[threadIdx.x >> 6] = last;
        __syncthreads();
        if (threadIdx.x == 0) picked = max(max(wi[0], wi[1]), max(wi[2], wi[3]));
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(size_t)row * ldout] = picked;
}

extern "C" int qtts_pick_token(const void *logits, int ld, const float *temperatures, const uint8_t *greedy_flags,
                               int top_k, uint64_t seed, uint64_t offset, const uint64_t *offset_dev, int64_t *out,
                               int ldout, int M, int V, qtts_stream_t stream) {
    QCHECK(logits && temperatures && out && M > 0 && V > 0 && ld >= V && ldout >= 1, "pick_token: bad arguments");
    QCHECK(V <= 4096, "pick_token: V=%d > 4096", V);
    QCHECK(top_k >= 1 && top_k <= V, "pick_token: top_k=%d out of range", top_k);
    hipLaunchKernelGGL(sample_topk_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)logits, ld, V,
                       temperatures, greedy_flags, offset_dev, top_k, ldout, out, seed, offset);
    QLAUNCH();
    return QTTS_OK;
}

extern "C" int qtts_sample_topk(const void *logits, int ld, const float *temperatures, int top_k, uint64_t seed,
                                uint64_t offset, int64_t *out, int M, int V, qtts_stream_t stream) {
    return qtts_pick_token(logits, ld, temperatures, nullptr, top_k, seed, offset, nullptr, out, 1, M, V, stream);
}

// ------------------------------------------------------------------ gathers / frame glue
__global__ __launch_bounds__(256) void embedding_gather_kernel(const bf16_t *__restrict__ table,
                                                               const int64_t *__restrict__ ids,
                                                               bf16_t *__restrict__ out, int ldo, int H, int vocab) {
    const int row = blockIdx.x;
    long id = ids[row];
    if (id < 0) id = 0;
    if (id >= vocab) id = vocab - 1;
    for (int i = threadIdx.x * 8; i < H; i += 256 * 8)
        *reinterpret_cast<uint4 *>(out + (size_t)row * ldo + i) = *reinterpret_cast<const uint4 *>(table + (size_t)id * H + i);
}

extern "C" int qtts_embedding_gather(const void *table, const int64_t *ids, void *out, int ldo, int M, int H, int vocab,
                                     qtts_stream_t stream) {
    QCHECK(table && ids && out && M > 0 && H > 0 && H % 8 == 0 && ldo % 8 == 0 && ldo >= H && vocab > 0,
           "embedding_gather: bad arguments");
    hipLaunchKernelGGL(embedding_gather_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)table, ids,
                       (bf16_t *)out, ldo, H, vocab);
    QLAUNCH();
    return QTTS_OK;
}

__global__ __launch_bounds__(256) void predictor_input_kernel(const bf16_t *__restrict__ hidden,
                                                              const bf16_t *__restrict__ emb,
                                                              const int64_t *__restrict__ tok,
                                                              bf16_t *__restrict__ out, int H, int vocab) {
    const int b = blockIdx.x;
    long id = tok[b];
    if (id < 0) id = 0;
    if (id >= vocab) id = vocab - 1;
    for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
        *reinterpret_cast<uint4 *>(out + ((size_t)b * 2 + 0) * H + i) = *reinterpret_cast<const uint4 *>(hidden + (size_t)b * H + i);
        *reinterpret_cast<uint4 *>(out + ((size_t)b * 2 + 1) * H + i) = *reinterpret_cast<const uint4 *>(emb + (size_t)id * H + i);
    }
}

extern "C" int qtts_predictor_input(const void *hidden, const void *talker_emb, const int64_t *token0, void *pred_in,
                                    int B, int H, int vocab, qtts_stream_t stream) {
    QCHECK(hidden && talker_emb && token0 && pred_in && B > 0 && H > 0 && H % 8 == 0 && vocab > 0,
           "predictor_input: bad arguments");
    hipLaunchKernelGGL(predictor_input_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)hidden,
                       (const bf16_t *)talker_emb, token0, (bf16_t *)pred_in, H, vocab);
    QLAUNCH();
    return QTTS_OK;
}

// Sum order follows torch's CPU reduction over the 16 stacked rows (sequential, row 0 first).
// token0[b] is the talker's id, pred_tokens[c*ldp + b] the predictor's id for codebook c+1.
__global__ __launch_bounds__(256) void frame_glue_kernel(const bf16_t *__restrict__ temb,
                                                         const bf16_t *const *__restrict__ ptabs,
                                                         const int64_t *__restrict__ token0,
                                                         const int64_t *__restrict__ pred_tokens, int ldp,
                                                         const float *__restrict__ text_add, long ld_text,
                                                         const int32_t *__restrict__ frame_idx,
                                                         bf16_t *__restrict__ next, int64_t *__restrict__ codes,
                                                         int H, int tv, int pv) {
    const int b = blockIdx.x;
    __shared__ long ids[16];
    __shared__ const bf16_t *rows[16];
    // ids, table pointers and the frame counter are requested together; the 16 rows' addresses go through LDS, so
    // the row loads that follow the barrier depend on nothing else in memory
    const int fidx = frame_idx ? frame_idx[b] : 0;
    if (threadIdx.x < 16) {
        long id = threadIdx.x == 0 ? token0[b] : pred_tokens[(size_t)(threadIdx.x - 1) * ldp + b];
        const bf16_t *tab = threadIdx.x == 0 ? temb : ptabs[threadIdx.x - 1];
        if (codes) codes[b * 16 + threadIdx.x] = id;
        const int lim = threadIdx.x == 0 ? tv : pv;
        id = id < 0 ? 0 : (id >= lim ? lim - 1 : id);
        ids[threadIdx.x] = id;
        rows[threadIdx.x] = tab + (size_t)id * H;
    }
    __syncthreads();
    const float *ta = text_add + (size_t)b * ld_text + (size_t)fidx * H;
    // eight columns per thread, 16 bytes per table row: a hidden size of 2048 is one pass with all 18 requests of a
    // thread in flight together (H % 8 == 0 and 16-byte aligned rows are checked on the host)
    for (int d = 8 * threadIdx.x; d < H; d += 8 * 256) {
        Frag v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c].u = *reinterpret_cast<const uint4 *>(rows[c] + d);
        const float4_t t0 = *reinterpret_cast<const float4_t *>(ta + d), t1 = *reinterpret_cast<const float4_t *>(ta + d + 4);
        float s[8];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const uint32_t w4[4] = {v[c].u.x, v[c].u.y, v[c].u.z, v[c].u.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float lo, hi;
                unpack2(w4[j], lo, hi);
                s[2 * j] = c == 0 ? lo : s[2 * j] + lo;            // table 0 first, then 1..15, the text row last
                s[2 * j + 1] = c == 0 ? hi : s[2 * j + 1] + hi;
            }
        }
        const float tx[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
        uint32_t o4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o4[j] = pack2(s[2 * j] + tx[2 * j], s[2 * j + 1] + tx[2 * j + 1]);
        *reinterpret_cast<uint4 *>(next + (size_t)b * H + d) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
    }
}

extern "C" int qtts_frame_glue(const void *talker_emb, const void *const *pred_emb_tables, const int64_t *token0,
                               const int64_t *pred_tokens, int ldp, const float *text_add, long ld_text,
                               const int32_t *frame_idx, void *next, int64_t *codes, int B, int H, int talker_vocab,
                               int pred_vocab, qtts_stream_t stream) {
    QCHECK(talker_emb && pred_emb_tables && token0 && pred_tokens && text_add && next && B > 0 && H > 0 && ldp >= B,
           "frame_glue: bad arguments");
    QCHECK(H % 8 == 0 && ld_text % 4 == 0 && ((uintptr_t)talker_emb | (uintptr_t)text_add | (uintptr_t)next) % 16 == 0,
           "frame_glue: hidden size must be a multiple of 8, rows 16-byte aligned");
    hipLaunchKernelGGL(frame_glue_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)talker_emb,
                       (const bf16_t *const *)pred_emb_tables, token0, pred_tokens, ldp, text_add, ld_text, frame_idx,
                       (bf16_t *)next, codes, H, talker_vocab, pred_vocab);
    QLAUNCH();
    return QTTS_OK;
}

// Advances the per-stream decode metadata by one token on the device, so that a captured
// frame graph can be replayed without a host upload: for active rows (context_lens > 0)
// position += 1, context += 1, slot = table[(context-1) / 256] * 256 + (context-1) % 256,
// frame_idx += 1.  A row whose context has reached max_context, or its own limit context_limits[b] (the tokens its
// KV blocks were reserved for), is frozen: nothing moves, so the row keeps rewriting its own last slot and can
// never reach a block-table entry that is not its own.
__global__ void advance_decode_kernel(int64_t *__restrict__ positions, int32_t *__restrict__ slots,
                                      int32_t *__restrict__ ctx, const int32_t *__restrict__ tables, int max_blocks,
                                      int32_t *__restrict__ frame_idx, uint64_t *__restrict__ rng_counter,
                                      const int32_t *__restrict__ limits, int max_context, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0 && rng_counter) *rng_counter += 64;      // 64 distinct RNG offsets per frame
    if (b >= B) return;
    const int c = ctx[b];
    const int lim = limits ? min(limits[b], max_context) : max_context;
    if (c <= 0 || c >= lim) return;
    const int n = c + 1;
    ctx[b] = n;
    positions[b] += 1;
    slots[b] = tables[(size_t)b * max_blocks + ((n - 1) >> 8)] * QTTS_KV_BLOCK + ((n - 1) & 255);
    if (frame_idx) frame_idx[b] += 1;
}

extern "C" int qtts_advance_decode(int64_t *positions, int32_t *slot_mapping, int32_t *context_lens,
                                   const int32_t *block_tables, int max_blocks, int32_t *frame_idx,
                                   uint64_t *rng_counter, const int32_t *context_limits, int max_context, int B,
                                   qtts_stream_t stream) {
    QCHECK(positions && slot_mapping && context_lens && block_tables && B > 0 && max_blocks > 0 &&
               max_context <= max_blocks * QTTS_KV_BLOCK, "advance_decode: bad arguments");
    hipLaunchKernelGGL(advance_decode_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, positions,
                       slot_mapping, context_lens, block_tables, max_blocks, frame_idx, rng_counter, context_limits,
                       max_context, B);
    QLAUNCH();
    return QTTS_OK;
}

// ------------------------------------------------------------------ decoder stack
struct LayerW {
    const void *in_norm = nullptr, *wqkv = nullptr, *q_norm = nullptr, *k_norm = nullptr, *wo = nullptr,
               *post_norm = nullptr, *wgu = nullptr, *wdown = nullptr;
    // fp8 weights: scale per packed row of the four GEMM weights (all null: bf16 weights)
    const float *s_qkv = nullptr, *s_o = nullptr, *s_gu = nullptr, *s_down = nullptr;
};

// The decoder stack's GEMMs with bf16 or fp8 weights (wscale != null).
// How a prompt GEMM's activations are laid out (all zero: row-major everywhere)
struct ActLayout {
    int xpacked = 0, opks = 0, nnpks = 0;
    int rpks = 0, opks0 = 0;            // plain epilogue: residual rows / output in fragment order
    float *ssq_out = nullptr;           // plain epilogue: the stored rows' sums of squares per 16 columns
    const float *ssq_in = nullptr;      // with a norm weight: the rows are normalised from these partials (prologue 3)
};

static int stack_linear(const void *x, int ldx, const void *w, const float *wscale, const void *residual, int ldr,
                        const void *norm_w, float eps, void *out, int ldo, int M, int N, int K, float *scratch,
                        size_t scratch_bytes, qtts_stream_t stream, const void *next_norm_w = nullptr,
                        void *next_xn = nullptr, int ldnn = 0, const int64_t *x_ids = nullptr,
                        const int64_t *res_ids = nullptr, int id_rows = 0, ActLayout lay = ActLayout()) {
    LinArgs a;
    a.xpacked = lay.xpacked; a.opks = lay.opks; a.nnpks = lay.nnpks;
    a.next_nw = (const bf16_t *)next_norm_w; a.next_xn = (bf16_t *)next_xn; a.ldnn = ldnn;
    a.res_ids = res_ids; a.res_id_rows = res_ids ? id_rows : 0;
    a.x = (const bf16_t *)x; a.ldx = ldx; a.ids = x_ids; a.id_rows = x_ids ? id_rows : 0; a.wp = (const uint4 *)w; a.wscale = wscale;
    a.bias = nullptr; a.res = (const bf16_t *)residual; a.ldr = ldr; a.nw = (const bf16_t *)norm_w; a.eps = eps;
    a.xnorm_out = nullptr; a.ldxn = 0; a.out = (bf16_t *)out; a.ldo = ldo; a.M = M; a.N = N; a.K = K; a.tiles_per_wg = 1;
    a.rpks = lay.rpks; a.opks0 = lay.opks0; a.ssq_out = lay.ssq_out; a.ssq_in = lay.ssq_in;
    return launch_linear2(a, norm_w ? (lay.ssq_in ? 3 : 1) : 0, (hipStream_t)stream, false, scratch, scratch_bytes);
}

static int stack_swiglu(const void *x, int ldx, const void *wgu, const float *wscale, const void *norm_w, float eps,
                        void *out, int ldo, int M, int I, int K, qtts_stream_t stream, ActLayout lay = ActLayout()) {
    LinArgs a;
    a.xpacked = lay.xpacked; a.opks = lay.opks; a.nnpks = lay.nnpks;
    a.x = (const bf16_t *)x; a.ldx = ldx; a.ids = nullptr; a.id_rows = 0; a.wp = (const uint4 *)wgu; a.wscale = wscale;
    a.bias = nullptr; a.res = nullptr; a.ldr = 0; a.nw = (const bf16_t *)norm_w; a.eps = eps; a.xnorm_out = nullptr;
    a.ldxn = 0; a.out = (bf16_t *)out; a.ldo = ldo; a.M = M; a.N = 2 * I; a.K = K; a.tiles_per_wg = 1;
    a.ssq_in = lay.ssq_in;
    return launch_linear2(a, norm_w ? (lay.ssq_in ? 3 : 1) : 0, (hipStream_t)stream, true);
}

struct qtts_model {
    int hidden, n_layers, n_heads, n_kv_heads, head_dim, inter;
    float eps;
    std::vector<LayerW> layers;
    const void *final_norm = nullptr;
    const void *cosT = nullptr, *sinT = nullptr;
    int max_pos = 0;
    char *kv = nullptr;
    int num_blocks = 0;
    char *ws = nullptr;
    size_t ws_bytes = 0;
    int ws_rows = 0, ws_ctx = 0;
    // measurement hook (qtts_model_set_layer_events): events recorded around one launch of every layer
    int ev_which = 0;
    std::vector<hipEvent_t> ev_begin, ev_end;
};

// Hands the layer's begin / end event to the decode GEMM launcher for launch number `which` of the layer body when
// the hook is set: the launch then goes through hipExtLaunchKernelGGL, which binds the two events to the kernel's
// own start and end (the times a profiler reports for it).
struct LayerEventScope {
    bool on;
    LayerEventScope(const qtts_model *m, int layer, int which, qtts_stream_t)
        : on(m->ev_which == which && which != 0) {
        if (on) g_kernel_events = {m->ev_begin[layer], m->ev_end[layer]};
    }
    ~LayerEventScope() { if (on) g_kernel_events = {nullptr, nullptr}; }    // (a launch that took another kernel)
};

extern "C" qtts_model_t *qtts_model_create(int hidden, int n_layers, int n_heads, int n_kv_heads, int head_dim,
                                           int intermediate, float rms_eps) {
    if (hidden <= 0 || n_layers <= 0 || n_heads <= 0 || n_kv_heads <= 0 || intermediate <= 0 ||
        head_dim != QTTS_HEAD_DIM || n_heads % n_kv_heads != 0 || hidden % 32 != 0 || intermediate % 32 != 0) {
        snprintf(g_err, sizeof(g_err), "model_create: unsupported shape (head_dim must be 128, sizes multiples of 32)");
        return nullptr;
    }
    qtts_model *m = new qtts_model();
    m->hidden = hidden;
    m->n_layers = n_layers;
    m->n_heads = n_heads;
    m->n_kv_heads = n_kv_heads;
    m->head_dim = head_dim;
    m->inter = intermediate;
    m->eps = rms_eps;
    m->layers.resize(n_layers);
    return m;
}

extern "C" void qtts_model_destroy(qtts_model_t *m) { delete m; }

extern "C" int qtts_model_set_layer(qtts_model_t *m, int layer, const void *input_norm_w, const void *wqkv_packed,
                                    const void *q_norm_w, const void *k_norm_w, const void *wo_packed,
                                    const void *post_norm_w, const void *w_gate_up_packed, const void *w_down_packed) {
    QCHECK(m && layer >= 0 && layer < m->n_layers, "model_set_layer: bad layer index");
    QCHECK(input_norm_w && wqkv_packed && q_norm_w && k_norm_w && wo_packed && post_norm_w && w_gate_up_packed &&
               w_down_packed, "model_set_layer: null weight");
    LayerW &L = m->layers[layer];
    L.in_norm = input_norm_w;
    L.wqkv = wqkv_packed;
    L.q_norm = q_norm_w;
    L.k_norm = k_norm_w;
    L.wo = wo_packed;
    L.post_norm = post_norm_w;
    L.wgu = w_gate_up_packed;
    L.wdown = w_down_packed;
    L.s_qkv = L.s_o = L.s_gu = L.s_down = nullptr;
    return QTTS_OK;
}

extern "C" int qtts_model_set_layer_fp8(qtts_model_t *m, int layer, const void *input_norm_w, const void *wqkv_fp8,
                                        const float *wqkv_scales, const void *q_norm_w, const void *k_norm_w,
                                        const void *wo_fp8, const float *wo_scales, const void *post_norm_w,
                                        const void *w_gate_up_fp8, const float *w_gate_up_scales, const void *w_down_fp8,
                                        const float *w_down_scales) {
    QCHECK(wqkv_scales && wo_scales && w_gate_up_scales && w_down_scales, "model_set_layer_fp8: null scales");
    QCHECK(m && m->hidden % 512 == 0 && m->inter % 512 == 0, "model_set_layer_fp8: hidden and intermediate size must be multiples of 512");
    int rc = qtts_model_set_layer(m, layer, input_norm_w, wqkv_fp8, q_norm_w, k_norm_w, wo_fp8, post_norm_w, w_gate_up_fp8,
                                  w_down_fp8);
    if (rc) return rc;
    LayerW &L = m->layers[layer];
    L.s_qkv = wqkv_scales; L.s_o = wo_scales; L.s_gu = w_gate_up_scales; L.s_down = w_down_scales;
    return QTTS_OK;
}

extern "C" int qtts_model_set_layer_events(qtts_model_t *m, int which, void *const *begin, void *const *end) {
    QCHECK(m, "model_set_layer_events: null model");
    QCHECK(which >= 0 && which <= QTTS_LAUNCH_HEAD, "model_set_layer_events: unknown launch %d", which);
    QCHECK(which == 0 || (begin && end), "model_set_layer_events: null event arrays");
    m->ev_which = 0;
    m->ev_begin.clear();
    m->ev_end.clear();
    if (which) {
        for (int l = 0; l < m->n_layers; ++l) {
            QCHECK(begin[l] && end[l], "model_set_layer_events: null event for layer %d", l);
            m->ev_begin.push_back((hipEvent_t)begin[l]);
            m->ev_end.push_back((hipEvent_t)end[l]);
        }
        m->ev_which = which;
    }
    return QTTS_OK;
}

extern "C" int qtts_model_set_final_norm(qtts_model_t *m, const void *w) {
    QCHECK(m && w, "model_set_final_norm: null");
    m->final_norm = w;
    return QTTS_OK;
}

extern "C" int qtts_model_set_rope(qtts_model_t *m, const void *cos, const void *sin, int max_pos) {
    QCHECK(m && cos && sin && max_pos > 0, "model_set_rope: bad arguments");
    m->cosT = cos;
    m->sinT = sin;
    m->max_pos = max_pos;
    return QTTS_OK;
}

extern "C" int qtts_model_set_kv_cache(qtts_model_t *m, void *kv, int num_blocks) {
    QCHECK(m && kv && num_blocks > 0, "model_set_kv_cache: bad arguments");
    m->kv = (char *)kv;
    m->num_blocks = num_blocks;
    return QTTS_OK;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
#define QTTS_MAX_DECODE_ROWS 512   // sequences per decode step: the reference's max_num_seqs (P/config.py:45)

struct WsLayout {
    size_t qkv, attn, act, xa, xb, xn, ssq, parts, parts_bytes, att_ws, att_cnt, total;
};

static WsLayout ws_layout(const qtts_model *m, int rows, int max_context) {
    WsLayout L;
    size_t off = 0;
    const size_t qkvw = (size_t)(m->n_heads + 2 * m->n_kv_heads) * 128;
    L.qkv = off; off += align256((size_t)rows * qkvw * 2);
    const size_t rows16 = ((size_t)rows + 15) & ~(size_t)15;      // fragment-order buffers hold whole 16-row tiles
    L.attn = off; off += align256(rows16 * m->n_heads * 128 * 2);
    L.act = off; off += align256(rows16 * m->inter * 2);
    L.xa = off; off += align256(rows16 * m->hidden * 2);              // (residual stream: fragment order inside a prompt's layers)
    L.xb = off; off += align256(rows16 * m->hidden * 2);
    L.xn = off; off += align256(rows16 * m->hidden * 2);              // normalised rows of a prompt (prefill GEMMs)
    L.ssq = off; off += align256(rows16 * (m->hidden / 16) * sizeof(float));   // residual rows' sums of squares per 16 columns
    L.parts_bytes = linear_rows_scratch_bytes(rows, m->hidden, m->inter);   // K parts of a prompt's down projection
    L.parts = off; off += align256(L.parts_bytes);
    // prefill and decode attention share the partials region; the decode kernel's arrival counters follow it.
    // Partials exist per (row, head, 128-token split), 520 bytes each: sized for the rows that can need them, not
    // for every row of the largest prompt batch.  Decode: at most QTTS_MAX_DECODE_ROWS sequences per step.
    // Prompts: with 1 or 2 query heads per KV head only launches below the MFMA kernel's row threshold use the
    // per-row kernel (and its partials).
    const int dec_rows = std::min(rows, QTTS_MAX_DECODE_ROWS);
#ifdef QTTS_TUNING
    const int pre_rows = rows;                          // the row threshold is a run-time variable in this build
#else
    const int pre_rows = m->n_heads / m->n_kv_heads <= 2 ? std::min(rows, g_mfma_attn_rows) : rows;
#endif
    const size_t pre = qtts_attn_workspace_bytes(pre_rows, m->n_heads, max_context);
    const size_t dec = attn2_partial_bytes(dec_rows, m->n_heads, max_context);
    L.att_ws = off; off += align256(pre > dec ? pre : dec);
    L.att_cnt = off; off += align256((size_t)dec_rows * m->n_kv_heads * sizeof(int));
    L.total = off;
    return L;
}

extern "C" size_t qtts_model_workspace_bytes(const qtts_model_t *m, int max_rows, int max_context) {
    if (!m || max_rows <= 0 || max_context <= 0) return 0;
    return ws_layout(m, max_rows, max_context).total;
}

extern "C" int qtts_model_set_workspace(qtts_model_t *m, void *ws, size_t bytes, int max_rows, int max_context) {
    QCHECK(m && ws && max_rows > 0 && max_context > 0, "model_set_workspace: bad arguments");
    QCHECK(((uintptr_t)ws & 255) == 0, "model_set_workspace: workspace must be 256-byte aligned");
    QCHECK(bytes >= ws_layout(m, max_rows, max_context).total, "model_set_workspace: %zu bytes < required %zu", bytes,
           ws_layout(m, max_rows, max_context).total);
    // the decode attention's arrival counters must start at zero (the kernel leaves them at zero)
    const WsLayout W = ws_layout(m, max_rows, max_context);
    hipError_t e = hipMemset((char *)ws + W.att_cnt, 0, W.total - W.att_cnt);
    if (e != hipSuccess) QFAIL((int)e, "model_set_workspace: memset failed: %s", hipGetErrorString(e));
    m->ws = (char *)ws;
    m->ws_bytes = bytes;
    m->ws_rows = max_rows;
    m->ws_ctx = max_context;
    return QTTS_OK;
}

static int model_forward_impl(qtts_model_t *m, const void *x, const int64_t *positions, const int32_t *slot_mapping,
                              const int32_t *row_seq, const int32_t *context_lens, const int32_t *block_tables,
                              int max_blocks, int max_context, int N, void *hidden_out, const void *head_w, int vocab,
                              void *logits_out, qtts_stream_t stream, const int64_t *x_ids = nullptr, int x_id_rows = 0,
                              const void *qkv0_rows = nullptr, bool pairs = false) {
    QCHECK(m, "model_forward: null model");
    // x_ids: x is a table of x_id_rows rows and row r of the step is x[x_ids[r]] (read in place by layer 0's qkv GEMM
    // and o_proj residual); qkv0_rows: a table of the same rows after layer 0's input norm and qkv GEMM, which then
    // is not launched
    QCHECK(!x_ids || (x_id_rows > 0 && !row_seq && (N <= 16 || qkv0_rows)), "model_forward: gathered input rows are for decode steps (beyond 16 rows: with a layer-0 qkv table)");
    QCHECK(!qkv0_rows || x_ids, "model_forward: a layer-0 qkv table needs row ids");
    if (!m->final_norm || !m->cosT || !m->kv || !m->ws) QFAIL(QTTS_ESTATE, "model_forward: model not fully configured");
    for (const LayerW &L : m->layers)
        if (!L.wqkv) QFAIL(QTTS_ESTATE, "model_forward: layer weights missing");
    // uniform decode rows: no metadata arrays (row r = sequence r in KV block r at context max_context), see
    // attention_decode_fused_impl
    const bool uniform = !pairs && !row_seq && !positions && !slot_mapping && !context_lens && !block_tables;
    QCHECK(x && hidden_out && (uniform || (positions && (pairs || (context_lens && block_tables)))), "model_forward: null pointer");
    QCHECK(!uniform || N <= m->num_blocks, "model_forward: rows without metadata need a KV block each (N=%d)", N);
    // pairs: rows (2i, 2i+1) are the two positions of a fresh sequence i; the head (required) runs on the second rows
    QCHECK(!pairs || (N % 2 == 0 && N <= 16 && !row_seq && !x_ids && head_w), "model_forward: bad 2-position prompt arguments");
    QCHECK(N > 0 && N <= m->ws_rows, "model_forward: N=%d exceeds workspace rows %d", N, m->ws_rows);
    QCHECK(row_seq || N <= QTTS_MAX_DECODE_ROWS, "model_forward: a decode step takes at most %d rows (got %d)", QTTS_MAX_DECODE_ROWS, N);
    QCHECK(max_context > 0 && max_context <= m->ws_ctx, "model_forward: max_context=%d exceeds workspace %d", max_context, m->ws_ctx);
    QCHECK(pairs || uniform || (max_blocks > 0 && max_context <= max_blocks * QTTS_KV_BLOCK), "model_forward: max_context exceeds block table");
    const WsLayout W = ws_layout(m, m->ws_rows, m->ws_ctx);
    bf16_t *qkv = (bf16_t *)(m->ws + W.qkv);
    bf16_t *attn = (bf16_t *)(m->ws + W.attn);
    bf16_t *act = (bf16_t *)(m->ws + W.act);
    bf16_t *xa = (bf16_t *)(m->ws + W.xa);
    bf16_t *xb = (bf16_t *)(m->ws + W.xb);
    bf16_t *xn = (bf16_t *)(m->ws + W.xn);
    void *att_ws = m->ws + W.att_ws;
    const int H = m->hidden, Hq = m->n_heads, Hkv = m->n_kv_heads;
    // More than one 16-row tile (prefill): the rows are normalised by their own launch and the GEMMs run without
    // the norm prologue, whose registers would leave room for one workgroup per CU only; up to 16 rows the norm
    // rides in the GEMM prologue.
    const bool split_norm = N > 16;
    const int qkvw = (Hq + 2 * Hkv) * 128;
    // A prompt whose four GEMMs per layer all run as the row-stationary prompt kernel hands the activations between
    // the kernels of a layer (normalised rows, attention output, SwiGLU output) over in fragment order: that kernel
    // then reads one contiguous KiB per request.  The residual stream, qkv and everything returned stay row-major.
    const bool frag = split_norm && !pairs &&
                      linear_rows_covers(N, qkvw, H, false, false, false, 0) &&
                      linear_rows_covers(N, H, Hq * 128, false, false, false, 0) &&
                      linear_rows_covers(N, 2 * m->inter, H, true, false, false, 0) &&
                      linear_rows_covers(N, H, m->inter, false, false, true, W.parts_bytes);
    const int pkH = frag ? H / 32 : 0, pkA = frag ? Hq * 128 / 32 : 0, pkI = frag ? m->inter / 32 : 0;
    ActLayout lay_in;                                  // a GEMM reading fragment-order rows
    lay_in.xpacked = frag;
    // Such a prompt also keeps its residual stream in fragment order from the first o_proj on, and the GEMMs that
    // finish residual rows (o_proj, a single-pass down_proj) leave every row's sum of squares per 16 columns behind:
    // the GEMM that consumes the rows normalised (gate/up, the next layer's qkv, the head) then computes the row
    // scale from those partials and normalises in registers, and the RMSNorm launch between the two is gone.
    // Up to 64 rows at a hidden size of 2048 (beyond that the four-row-tile workgroups, whose registers leave no room
    // for the normalisation, stream the weights faster than two-row-tile ones save launches: a 200-token prompt 2.89
    // against 3.10 ms) and up to 256 rows at 1024.
    const bool n3 = frag && g_rows_norm && N <= g_rows_norm_max && linear_rows_norm_covers(H) && N <= (H > 1024 ? 64 : 256);
    float *ssq = (float *)(m->ws + W.ssq);
    const float scale = 1.0f / sqrtf((float)m->head_dim);
    const size_t layer_kv = (size_t)m->num_blocks * QTTS_KV_BLOCK * Hkv * 128 * 2;   // bytes per layer per K or V
    const bf16_t *cur = (const bf16_t *)x;
    bool cur_frag = false, cur_ssq = false;            // `cur` in fragment order / its sums of squares are in `ssq`
    int rc;
    // A step of more than 16 rows whose input rows come from a table (x_ids; with layer 0's qkv rows from their table):
    // the prompt GEMM kernels take no row ids, so the rows are gathered once, into the buffer of the normalised rows
    // (free until the first post-attention norm, by when layer 0's o_proj has read them as its residual rows), and
    // the ids are used by layer 0's attention only.
    const int64_t *res_ids0 = x_ids;
    if (x_ids && split_norm) {
        hipLaunchKernelGGL(embedding_gather_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)x, x_ids, xn, H, H, x_id_rows);
        cur = xn;
        res_ids0 = nullptr;
    }
    const bool decode = row_seq == nullptr;           // one row per sequence: fused norm/RoPE/store/attention
    // Decode steps of 9..16 rows: every lane of a GEMM's activation requests carries a row of its own (no lanes to
    // share the prologue with, as up to 8 rows), so the two activations a layer produces for a plain prologue -
    // attention output for o_proj, SwiGLU output for down_proj - are handed over in fragment order: one contiguous
    // KiB per request instead of 64 bytes from each of 16 rows.
    // (The 2-position prompts of up to 8 fresh sequences are such a step too, and so is a prompt of 9..16 tokens.)
    const bool dfrag = N > 8 && N <= 16 && m->ws_rows >= 16;
    ActLayout lay_dx, lay_dgu;
    lay_dx.xpacked = dfrag;
    lay_dgu.opks = dfrag ? m->inter / 32 : 0;
    bool xn_ready = false, final_ready = false;       // a norm already applied by the producing launch
    for (int l = 0; l < m->n_layers; ++l) {
        const LayerW &L = m->layers[l];
        char *kc = m->kv + (size_t)l * layer_kv;
        char *vc = m->kv + ((size_t)m->n_layers + l) * layer_kv;
        if (split_norm && l == 0 && qkv0_rows) {
            // layer 0's normalised, projected rows come from the table
        } else if (split_norm) {
            if (!xn_ready && cur_ssq) {
                // the rows are normalised inside the GEMM
                ActLayout lay = lay_in;
                lay.ssq_in = ssq;
                if ((rc = stack_linear(cur, H, L.wqkv, L.s_qkv, nullptr, 0, L.in_norm, m->eps, qkv, qkvw, N, qkvw, H, nullptr, 0, stream,
                                       nullptr, nullptr, 0, nullptr, nullptr, 0, lay)))
                    return rc;
            } else {
                if (!xn_ready && (rc = rmsnorm_impl(cur, H, L.in_norm, m->eps, xn, H, N, H, pkH, stream, cur_frag ? pkH : 0))) return rc;
                if ((rc = stack_linear(xn, H, L.wqkv, L.s_qkv, nullptr, 0, nullptr, 0.f, qkv, qkvw, N, qkvw, H, nullptr, 0, stream,
                                       nullptr, nullptr, 0, nullptr, nullptr, 0, lay_in)))
                    return rc;
            }
            xn_ready = false;
        } else if (l == 0 && qkv0_rows) {
            // layer 0's normalised, projected rows come from the table
        } else {
            LayerEventScope ev(m, l, QTTS_LAUNCH_QKV, stream);
            if ((rc = stack_linear(cur, H, L.wqkv, L.s_qkv, nullptr, 0, L.in_norm, m->eps, qkv, qkvw, N, qkvw, H, nullptr, 0,
                                   stream, nullptr, nullptr, 0, l == 0 ? x_ids : nullptr, nullptr, x_id_rows))) return rc;
        }
        const bool tab0 = l == 0 && qkv0_rows;
        if (pairs) {
            if ((rc = attention_pairs_impl(qkv, qkvw, L.q_norm, L.k_norm, m->eps, m->cosT, m->sinT, m->max_pos, positions,
                                           slot_mapping, kc, vc, scale, attn, Hq * 128, N, Hq, Hkv, (hipStream_t)stream,
                                           dfrag ? Hq * 128 / 32 : 0)))
                return rc;
        } else if (decode) {
            LayerEventScope ev(m, l, QTTS_LAUNCH_ATTENTION, stream);
            if ((rc = attention_decode_fused_impl(tab0 ? qkv0_rows : qkv, qkvw, L.q_norm, L.k_norm, m->eps, m->cosT, m->sinT,
                                                  m->max_pos, positions, slot_mapping, kc, vc, block_tables, max_blocks,
                                                  context_lens, max_context, scale, attn, Hq * 128, N, Hq, Hkv, att_ws,
                                                  (int *)(m->ws + W.att_cnt), (hipStream_t)stream,
                                                  tab0 ? x_ids : nullptr, x_id_rows, dfrag ? Hq * 128 / 32 : pkA))) return rc;
        } else {
            if ((rc = qtts_qknorm_rope_store(qkv, qkvw, L.q_norm, L.k_norm, m->eps, m->cosT, m->sinT, m->max_pos,
                                             positions, slot_mapping, kc, vc, N, Hq, Hkv, stream))) return rc;
            if ((rc = paged_attention_impl(qkv, qkvw, kc, vc, block_tables, max_blocks, row_seq, context_lens,
                                           max_context, scale, attn, Hq * 128, N, Hq, Hkv, att_ws, stream,
                                           dfrag ? Hq * 128 / 32 : pkA))) return rc;
        }
        {
            LayerEventScope ev(m, l, QTTS_LAUNCH_O_PROJ, stream);
            ActLayout lay = split_norm ? lay_in : lay_dx;
            if (n3) {
                lay.rpks = cur_frag ? pkH : 0;
                lay.opks0 = pkH;
                lay.ssq_out = ssq;
            }
            if ((rc = stack_linear(attn, Hq * 128, L.wo, L.s_o, cur, H, nullptr, 0.f, xa, H, N, H, Hq * 128, nullptr, 0, stream,
                                   nullptr, nullptr, 0, nullptr, l == 0 ? res_ids0 : nullptr, x_id_rows, lay)))
                return rc;
        }
        if (split_norm) {
            ActLayout lay_gu = lay_in;
            lay_gu.opks = pkI;
            if (n3) {
                lay_gu.ssq_in = ssq;
                if ((rc = stack_swiglu(xa, H, L.wgu, L.s_gu, L.post_norm, m->eps, act, m->inter, N, m->inter, H, stream, lay_gu))) return rc;
            } else {
                if ((rc = rmsnorm_impl(xa, H, L.post_norm, m->eps, xn, H, N, H, pkH, stream))) return rc;
                if ((rc = stack_swiglu(xn, H, L.wgu, L.s_gu, nullptr, 0.f, act, m->inter, N, m->inter, H, stream, lay_gu))) return rc;
            }
        } else {
            LayerEventScope ev(m, l, QTTS_LAUNCH_GATE_UP, stream);
            if ((rc = stack_swiglu(xa, H, L.wgu, L.s_gu, L.post_norm, m->eps, act, m->inter, N, m->inter, H, stream, lay_dgu))) return rc;
        }
        // a prompt's K-split down_proj also normalises the finished rows for whoever reads them next: the next
        // layer's input norm, or the final norm (the returned hidden state)
        const bool last = l + 1 == m->n_layers;
        const bool fuse_next = split_norm && linear_rows_is_ksplit(N, H, m->inter, W.parts_bytes);
        LayerEventScope ev_down(m, l, QTTS_LAUNCH_DOWN, stream);
        ActLayout lay_down;
        lay_down.xpacked = split_norm ? frag : dfrag;
        lay_down.nnpks = fuse_next && !last ? pkH : 0;   // the next layer's normalised rows; the returned hidden state stays row-major
        if (n3) {
            lay_down.rpks = lay_down.opks0 = pkH;
            lay_down.ssq_out = fuse_next ? nullptr : ssq;
        }
        if ((rc = stack_linear(act, m->inter, L.wdown, L.s_down, xa, H, nullptr, fuse_next ? m->eps : 0.f, xb, H, N, H, m->inter,
                               (float *)(m->ws + W.parts), W.parts_bytes, stream,
                               fuse_next ? (last ? m->final_norm : m->layers[l + 1].in_norm) : nullptr,
                               fuse_next ? (last ? (bf16_t *)hidden_out : xn) : nullptr, H, nullptr, nullptr, 0, lay_down))) return rc;
        if (fuse_next) (last ? final_ready : xn_ready) = true;
        // Fixed roles: xa = post-attention residual stream, xb = layer output.  The next layer reads xb
        // (qkv GEMM, o_proj residual) before its down_proj overwrites it, all stream-ordered.
        cur = xb;
        cur_frag = n3;
        cur_ssq = n3 && !fuse_next;
    }
    if (head_w) {
        if (split_norm) {
            if (!final_ready && cur_ssq && vocab % 16 == 0) {
                // the final norm inside the head GEMM; its first column block writes the normalised rows (the hidden state)
                LinArgs a;
                a.xpacked = 1; a.ssq_in = ssq;
                a.x = cur; a.ldx = H; a.ids = nullptr; a.id_rows = 0; a.wp = (const uint4 *)head_w; a.bias = nullptr; a.res = nullptr;
                a.ldr = 0; a.nw = (const bf16_t *)m->final_norm; a.eps = m->eps; a.xnorm_out = (bf16_t *)hidden_out; a.ldxn = H;
                a.out = (bf16_t *)logits_out; a.ldo = vocab; a.M = N; a.N = vocab; a.K = H; a.tiles_per_wg = 1;
                return launch_linear2(a, 3, (hipStream_t)stream);
            }
            if (!final_ready && (rc = rmsnorm_impl(cur, H, m->final_norm, m->eps, hidden_out, H, N, H, 0, stream, cur_frag ? pkH : 0))) return rc;
            return qtts_linear_fused(hidden_out, H, nullptr, 0, head_w, nullptr, nullptr, 0, nullptr, 0.f, nullptr, 0, 0,
                                     logits_out, vocab, N, vocab, H, stream);
        }
        if (pairs)      // the second row of every pair: a strided view; hidden_out gets those N/2 rows
            return qtts_linear_fused(cur + H, 2 * H, nullptr, 0, head_w, nullptr, nullptr, 0, m->final_norm, m->eps, hidden_out, H,
                                     0, logits_out, vocab, N / 2, vocab, H, stream);
        // final RMSNorm folded into the head GEMM's prologue; the normalised rows are the returned hidden state
        LayerEventScope ev(m, 0, QTTS_LAUNCH_HEAD, stream);
        if ((rc = qtts_linear_fused(cur, H, nullptr, 0, head_w, nullptr, nullptr, 0, m->final_norm, m->eps, hidden_out, H, 0,
                                    logits_out, vocab, N, vocab, H, stream))) return rc;
        return QTTS_OK;
    }
    if (!final_ready && (rc = rmsnorm_impl(cur, H, m->final_norm, m->eps, hidden_out, H, N, H, 0, stream, cur_frag ? pkH : 0))) return rc;
    return QTTS_OK;
}

extern "C" int qtts_model_forward(qtts_model_t *m, const void *x, const int64_t *positions, const int32_t *slot_mapping,
                                  const int32_t *row_seq, const int32_t *context_lens, const int32_t *block_tables,
                                  int max_blocks, int max_context, int N, void *hidden_out, qtts_stream_t stream) {
    return model_forward_impl(m, x, positions, slot_mapping, row_seq, context_lens, block_tables, max_blocks, max_context,
                              N, hidden_out, nullptr, 0, nullptr, stream);
}

extern "C" int qtts_model_forward_logits(qtts_model_t *m, const void *x, const int64_t *positions,
                                         const int32_t *slot_mapping, const int32_t *row_seq,
                                         const int32_t *context_lens, const int32_t *block_tables, int max_blocks,
                                         int max_context, int N, void *hidden_out, const void *head_w_packed, int vocab,
                                         void *logits_out, qtts_stream_t stream) {
    QCHECK(head_w_packed && logits_out && vocab > 0 && vocab % 16 == 0, "model_forward_logits: bad head arguments");
    return model_forward_impl(m, x, positions, slot_mapping, row_seq, context_lens, block_tables, max_blocks, max_context,
                              N, hidden_out, head_w_packed, vocab, logits_out, stream);
}

extern "C" int qtts_model_forward_logits_gathered(qtts_model_t *m, const void *x_table, const void *qkv0_table,
                                                  const int64_t *ids, int table_rows, const int64_t *positions,
                                                  const int32_t *slot_mapping, const int32_t *context_lens,
                                                  const int32_t *block_tables, int max_blocks, int max_context, int N,
                                                  void *hidden_out, const void *head_w_packed, int vocab,
                                                  void *logits_out, qtts_stream_t stream) {
    QCHECK(head_w_packed && logits_out && vocab > 0 && vocab % 16 == 0, "model_forward_logits_gathered: bad head arguments");
    QCHECK(x_table && ids && table_rows > 0, "model_forward_logits_gathered: null table or ids");
    return model_forward_impl(m, x_table, positions, slot_mapping, nullptr, context_lens, block_tables, max_blocks, max_context,
                              N, hidden_out, head_w_packed, vocab, logits_out, stream, ids, table_rows, qkv0_table);
}

extern "C" int qtts_model_forward_logits_pairs(qtts_model_t *m, const void *x, const int64_t *positions,
                                               const int32_t *slot_mapping, int N, void *hidden_out,
                                               const void *head_w_packed, int vocab, void *logits_out,
                                               qtts_stream_t stream) {
    QCHECK(head_w_packed && logits_out && vocab > 0 && vocab % 16 == 0, "model_forward_logits_pairs: bad head arguments");
    return model_forward_impl(m, x, positions, slot_mapping, nullptr, nullptr, nullptr, 0, 2, N, hidden_out, head_w_packed,
                              vocab, logits_out, stream, nullptr, 0, nullptr, true);
}

#ifdef QTTS_TUNING
#include "qtts_tuning.h"
// ------------------------------------------------------------------ grid barrier probe (tuning experiments only)
// `iters` device-wide barriers inside one resident kernel.  mode 0: every workgroup adds to one counter and polls
// it.  mode 1: one arrival counter, the last arriver publishes the round number to per-group flags (256 B apart)
// and a workgroup polls its group's flag only.  mode 2: arrival through per-group counters, the group's last
// arriver adds to the top counter, flags as in mode 1.  Every wait is bounded: a workgroup that gives up raises
// the error word and all leave.
#define GB_STRIDE 64      // ints: one 256 B line per counter / flag
__device__ __forceinline__ int gb_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gb_st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int gb_add(int *p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(512) void grid_barrier_probe_kernel(int *mem, int iters, int mode, int gsize, int sleep) {
    const int nwg = gridDim.x, wg = blockIdx.x;
    const int ngroups = (nwg + gsize - 1) / gsize, grp = wg / gsize;
    const int gmembers = min(gsize, nwg - grp * gsize);
    int *err = mem, *top = mem + GB_STRIDE, *flags = mem + 2 * GB_STRIDE, *gcnt = mem + (2 + 64) * GB_STRIDE;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const int round = it + 1;
            const int *watch;
            int want;
            if (mode == 0) {
                gb_add(top, 1);
                watch = top;
                want = round * nwg;
            } else {
                bool last;
                if (mode == 1) {
                    last = gb_add(top, 1) == round * nwg - 1;
                } else {
                    last = false;
                    if (gb_add(gcnt + grp * GB_STRIDE, 1) == round * gmembers - 1)
                        last = gb_add(top, 1) == round * ngroups - 1;
                }
                if (last)
                    for (int g = 0; g < ngroups; ++g) gb_st(flags + g * GB_STRIDE, round);
                watch = flags + grp * GB_STRIDE;
                want = round;
            }
            int spins = 0;
            while (gb_ld(watch) < want) {
                if (++spins > 2000000) {
                    gb_st(err, 1);
                    break;
                }
                if (sleep) __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        if (gb_ld(err)) return;
    }
}

extern "C" int qtts_debug_grid_barrier(int *mem, int iters, int nwg, int threads, int mode, int gsize, int sleep,
                                       qtts_stream_t stream) {
    // mem: (2 + 64 + 64) * 256 bytes
    QCHECK(mem && iters > 0 && nwg > 0 && nwg <= 256 && threads > 0 && threads <= 512 && gsize > 0 &&
               (nwg + gsize - 1) / gsize <= 64, "grid barrier probe: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(mem, 0, (size_t)(2 + 64 + 64) * GB_STRIDE * sizeof(int), s);
    if (e != hipSuccess) QFAIL((int)e, "grid barrier probe: memset failed");
    hipLaunchKernelGGL(grid_barrier_probe_kernel, dim3(nwg), dim3(threads), 0, s, mem, iters, mode, gsize, sleep);
    QLAUNCH();
    return QTTS_OK;
}

// A prompt GEMM in any of its forms by itself (timing experiments): x row-major or in fragment order, rows normalised
// from ssq_in (with norm_w) or taken as they are, plain epilogue with the residual rows / the output in fragment order
// and the sums of squares left in ssq_out, or the SwiGLU epilogue.
extern "C" int qtts_debug_linear_rows(const void *x, int ldx, int xpacked, const float *ssq_in, const void *norm_w, float eps,
                                      const void *w_packed, const void *residual, int ldr, int rpks, void *out, int ldo,
                                      int opks0, float *ssq_out, int M, int N, int K, int swiglu, qtts_stream_t stream) {
    LinArgs a;
    a.x = (const bf16_t *)x; a.ldx = ldx; a.xpacked = xpacked; a.ids = nullptr; a.id_rows = 0; a.wp = (const uint4 *)w_packed;
    a.bias = nullptr; a.res = (const bf16_t *)residual; a.ldr = ldr; a.rpks = rpks; a.nw = (const bf16_t *)norm_w; a.eps = eps;
    a.xnorm_out = nullptr; a.ldxn = 0; a.out = (bf16_t *)out; a.ldo = ldo; a.opks0 = opks0; a.ssq_out = ssq_out; a.ssq_in = ssq_in;
    a.M = M; a.N = N; a.K = K; a.tiles_per_wg = 1;
    if (swiglu) a.opks = opks0, a.opks0 = 0;
    return launch_linear2(a, norm_w ? (ssq_in ? 3 : 1) : 0, (hipStream_t)stream, swiglu != 0);
}

#include "qtts_resident.inc"
#endif  // QTTS_TUNING

// ------------------------------------------------------------------ hipGraph helpers
struct qtts_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

extern "C" int qtts_graph_begin(qtts_stream_t stream) {
    hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) QFAIL((int)e, "graph_begin: %s", hipGetErrorString(e));
    return QTTS_OK;
}

extern "C" qtts_graph_t *qtts_graph_end(qtts_stream_t stream) {
    qtts_graph *g = new qtts_graph();
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g->graph);
    if (e != hipSuccess || !g->graph) {
        snprintf(g_err, sizeof(g_err), "graph_end: %s", hipGetErrorString(e));
        delete g;
        return nullptr;
    }
    e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "graph_instantiate: %s", hipGetErrorString(e));
        (void)hipGraphDestroy(g->graph);
        delete g;
        return nullptr;
    }
    return g;
}

extern "C" int qtts_graph_launch(qtts_graph_t *g, qtts_stream_t stream) {
    QCHECK(g && g->exec, "graph_launch: null graph");
    hipError_t e = hipGraphLaunch(g->exec, (hipStream_t)stream);
    if (e != hipSuccess) QFAIL((int)e, "graph_launch: %s", hipGetErrorString(e));
    return QTTS_OK;
}

extern "C" void qtts_graph_destroy(qtts_graph_t *g) {
    if (!g) return;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}
