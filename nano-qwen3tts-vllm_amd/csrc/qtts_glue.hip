// libqtts_hip - token picks (arg-max, top-k sampling), embedding gathers, the per-frame glue between talker and
// predictor, the device-side advance of the decode metadata.  C ABI in include/qtts_hip.h.
#include "qtts_common.h"

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
