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
#include "qtts_common.h"

thread_local char g_err[512] = "";
thread_local KernelEvents g_kernel_events = {nullptr, nullptr};

// Kernels that ask for more dynamic LDS than the default limit: the limit is raised once per (kernel, device).
// Returns false (with the error text set) when the device refuses.
bool allow_dynamic_lds(const void *kernel, size_t bytes) {
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

// Dispatch thresholds of the -DQTTS_TUNING build: process-wide variables with setters (the shipped library holds
// them as constants, qtts_common.h).
#ifdef QTTS_TUNING
int g_tune_ksw = 0;   // 0 = automatic
int g_tune_t = 0;
int g_dbg_linear = 0;
int g_rows_min = 16;              // most rows that still take single-tile workgroups instead of the prompt GEMM
int g_share_prologue = 1;         // norm prologue shared between the lane halves for up to 8 rows (0: tuning)
int g_sub_min_bytes = 32 << 10;   // smallest weight slice per workgroup when a tile is split over workgroups
int g_mfma_attn_rows = 112;       // rows from which qtts_paged_attention takes the MFMA kernel for prompts
int g_rows_tm = 0;                // row tiles per workgroup of the prompt GEMM (0 = chosen per shape)
int g_attn_wide = 0;              // decode attention: 0 = per shape, 1 = 128-token workgroups, 2 = 256-token ones
int g_rows_tm1 = 2;               // prompt GEMMs that may take one row tile per workgroup (0 none, 1 narrow plain outputs, 2 any)
int g_rows_norm = 1;              // prompt GEMMs normalise their rows themselves (0: a norm launch in front, for A/B runs)
int g_rows_norm_max = 1 << 30;    // most rows for which they do
long long *g_trace = nullptr;     // timestamp buffer of traced GEMM launches
int g_trace_launch = 0;
extern "C" void qtts_tune_attn_wide(int mode) { g_attn_wide = mode; }
extern "C" void qtts_tune_rows_tm1(int on) { g_rows_tm1 = on; }
extern "C" void qtts_tune_rows_norm(int on, int max_rows) { g_rows_norm = on; g_rows_norm_max = max_rows; }
extern "C" void qtts_tune_rows_tm(int tm) { g_rows_tm = tm; }
extern "C" void qtts_tune_linear(int ksw, int t) { g_tune_ksw = ksw; g_tune_t = t; }
extern "C" void qtts_tune_subtiles(int min_bytes) { g_sub_min_bytes = min_bytes; }
extern "C" void qtts_tune_rows(int min_rows) { g_rows_min = min_rows; }
extern "C" void qtts_tune_shared_prologue(int on) { g_share_prologue = on; }
extern "C" void qtts_tune_prompt_attention(int min_rows) { g_mfma_attn_rows = min_rows; }
extern "C" void qtts_debug_linear(int flags) { g_dbg_linear = flags; }
extern "C" void qtts_debug_linear_trace(long long *buf) { g_trace = buf; g_trace_launch = 0; }
#endif

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

// ------------------------------------------------------------------ linear (skinny GEMM on MFMA): kernels
// 16 fp8 (e4m3, OCP) values -> two bf16 fragments, exactly (every e4m3 value is a bf16 value).  Bytes 0..7 belong
// to the even k-step of the pair, bytes 8..15 to the odd one.
__device__ __forceinline__ uint32_t fp8x2_to_bf16x2(uint32_t w, bool hi) {
    const float2_t f = hi ? __builtin_amdgcn_cvt_pk_f32_fp8(w, true) : __builtin_amdgcn_cvt_pk_f32_fp8(w, false);
    return (__float_as_uint(f[0]) >> 16) | (__float_as_uint(f[1]) & 0xffff0000u);
}
__device__ __forceinline__ void fp8x16_to_bf16(const uint4 q, Frag &even, Frag &odd) {
    even.u = make_uint4(fp8x2_to_bf16x2(q.x, false), fp8x2_to_bf16x2(q.x, true), fp8x2_to_bf16x2(q.y, false), fp8x2_to_bf16x2(q.y, true));
    odd.u = make_uint4(fp8x2_to_bf16x2(q.z, false), fp8x2_to_bf16x2(q.z, true), fp8x2_to_bf16x2(q.w, false), fp8x2_to_bf16x2(q.w, true));
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
    if (N <= 256 * 8) {
        // One pass (as in rmsnorm_kernel): the finished row stays in registers across the reduction and the norm
        // weights are requested with everything else; same arithmetic, same order of additions.
        const int n = threadIdx.x * 8;
        const bool on = n < N;
        const int nc = on ? n : 0;
        float q[8];
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const float4_t sum = sum_k_parts(part, (size_t)m * N + nc + 4 * hlf, (size_t)mpad * N, kparts);
#pragma unroll
            for (int j = 0; j < 4; ++j) q[4 * hlf + j] = sum[j];
        }
        Frag rf, bfv, o;
        rf.u = res ? *reinterpret_cast<const uint4 *>(res + act_index(m, nc, ldr, rpks)) : make_uint4(0, 0, 0, 0);
        bfv.u = bias ? *reinterpret_cast<const uint4 *>(bias + nc) : make_uint4(0, 0, 0, 0);
        const uint4 wv = *reinterpret_cast<const uint4 *>(nw + nc);
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = q[j];
            if (bias) v += bf2f(bfv.h[j]);
            v = rbf(v);
            if (res) v = v + bf2f(rf.h[j]);
            o.h[j] = f2bf(v);
            const float r = bf2f(o.h[j]);
            if (on) ss += r * r;
        }
        if (on) *reinterpret_cast<uint4 *>(out + act_index(m, n, ldo, opks0)) = o.u;
        for (int of = 32; of > 0; of >>= 1) ss += __shfl_xor(ss, of);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
        __syncthreads();
        const float tot = red[0] + red[1] + red[2] + red[3];
        const float rstd = 1.0f / sqrtf(tot / (float)N + eps);
        if (on) *reinterpret_cast<uint4 *>(xn + act_index(m, n, ldxn, xnpks)) = norm_frag(o.u, rstd, wv);
        return;
    }
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
bool linear_rows_norm_covers(int K) {
    const int ksw = K / 256, parts = K / 16;
    return K % 256 == 0 && (ksw == 4 || ksw == 8) && parts % 8 == 0 && ((parts / 8) & (parts / 8 - 1)) == 0 && parts / 8 <= 16;
}

// Bytes of fp32 scratch the K-split form of a prompt GEMM needs (0 when the GEMM runs in one part).
size_t linear_rows_scratch_bytes(int M, int N, int K) {
    const int KS = K / 32;
    if (M <= 16 || KS % 8 != 0 || KS / 8 <= 12) return 0;
    return (size_t)(KS / 64) * M * N * sizeof(float);
}

// True when launch_linear_rows runs this shape as K parts (and can therefore take the fused norm of the result).
bool linear_rows_is_ksplit(int M, int N, int K, size_t scratch_bytes) {
    const size_t need = linear_rows_scratch_bytes(M, N, K);
    return M > g_rows_min && need > 0 && scratch_bytes >= need && (K / 32) % 64 == 0 && N % 8 == 0 && !(g_dbg_linear & (2 | 32));   // bit 5: tests switch the fusion off
}

// Prompt GEMM (more than one 16-row tile, plain activations).  Returns false when the shape is not covered and the
// caller should take the single-tile path.
// Whether launch_linear_rows takes this shape (the decision a caller needs before it lays activations out in fragment
// order, which only that kernel reads).
bool linear_rows_covers(int M, int N, int K, bool silu_pairs, bool has_bias, bool has_scratch, size_t scratch_bytes) {
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

int launch_linear2(LinArgs a, int pro, hipStream_t s, bool silu_pairs, float *scratch, size_t scratch_bytes) {
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
    if (H <= 256 * 8) {
        // One pass: a thread's 8 columns and their weights are requested together and stay in registers across the
        // reduction (the general form below reads the row a second time and asks for the weights only then, one
        // round trip later).  Same additions in the same order.
        const int i = threadIdx.x * 8;
        const bool on = i < H;
        Frag f;
        f.u = *reinterpret_cast<const uint4 *>(x + act_index(row, on ? i : 0, ldx, xpks));
        const uint4 wv = *reinterpret_cast<const uint4 *>(w + (on ? i : 0));
        float ss = 0.f;
        if (on) {
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
        if (on) *reinterpret_cast<uint4 *>(y + act_index(row, i, ldy, ypks)) = norm_frag(f.u, rstd, wv);
        return;
    }
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
int rmsnorm_impl(const void *x, int ldx, const void *w, float eps, void *y, int ldy, int M, int H, int ypks,
                 qtts_stream_t stream, int xpks) {
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

#ifdef QTTS_TUNING
#include "qtts_tuning.h"
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

#endif  // QTTS_TUNING
