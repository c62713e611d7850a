// libqtts_hip - attention over the paged KV cache: per-head q/k RMSNorm + RoPE + KV store, prompt attention (per row and
// on MFMA), the fused decode launch, the burst's 2-position prompts.  C ABI in include/qtts_hip.h.
#include "qtts_common.h"

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
int paged_attention_impl(const void *q, int ldq, const void *k_cache, const void *v_cache,
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
#ifdef QTTS_TUNING
    long long *trace;                      // 8 timestamps per workgroup (the traced launch), or null
#endif
};

#ifdef QTTS_TUNING
// slot s of this workgroup's timestamps; `dep` is a value the stamp must not be taken before
#define QTTS_ATRACE(slot, dep) do { asm volatile("" ::"v"(dep) : "memory"); if (p.trace && threadIdx.x == 0) \
    p.trace[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define QTTS_ATRACE(slot, dep) do { } while (0)
#endif

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
    QTTS_ATRACE(0, t0);
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
    QTTS_ATRACE(1, ctx);
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
    __builtin_amdgcn_sched_barrier(0);      // (the requests above are not to be scheduled behind the ones below)
    // Order of the requests (a wave's requests return in the order they were made): the row, the norm weights and the
    // cos / sin row first - the q / k normalisation below needs only those and is done while the cached rows are
    // still on their way -, then every K row, then every V row: the scores are complete when the K half has
    // arrived, the weighted sum follows the V rows in.
    Frag kk[TPG], vv[TPG];
    const bf16_t *krow = rowp + (size_t)(Hq + kvh) * 128 + 8 * l16;
    const bf16_t *vrow = rowp + (size_t)(Hq + Hkv + kvh) * 128 + 8 * l16;
#pragma unroll
    for (int i = 0; i < TPG; ++i) {
        const int t = t0 + i * NG + grp;
        const size_t off = (((size_t)blk_all * QTTS_KV_BLOCK + (t & 255)) * Hkv + kvh) * 128 + 8 * l16;
        kk[i].u = *reinterpret_cast<const uint4 *>(t < ctx - 1 ? p.kc + off : krow);
    }
#pragma unroll
    for (int i = 0; i < TPG; ++i) {
        const int t = t0 + i * NG + grp;
        const size_t off = (((size_t)blk_all * QTTS_KV_BLOCK + (t & 255)) * Hkv + kvh) * 128 + 8 * l16;
        vv[i].u = *reinterpret_cast<const uint4 *>(t < ctx - 1 ? p.vc + off : vrow);
    }
    // (a use outside the branch below: without it the compiler moves the cos / sin and weight requests into that
    // branch, behind the K / V requests, and the normalisation waits for every cached row)
    asm volatile("" ::"v"(xr.u.x), "v"(xr.u.y), "v"(xr.u.z), "v"(xr.u.w), "v"(wv.u.x), "v"(wv.u.y), "v"(wv.u.z), "v"(wv.u.w),
                 "v"(cv.u.x), "v"(cv.u.y), "v"(cv.u.z), "v"(cv.u.w), "v"(sv.u.x), "v"(sv.u.y), "v"(sv.u.z), "v"(sv.u.w),
                 "v"(vnew.u.x), "v"(vnew.u.y), "v"(vnew.u.z), "v"(vnew.u.w));

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
    QTTS_ATRACE(2, ctx);
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
    QTTS_ATRACE(3, sc[0][0]);
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
    QTTS_ATRACE(4, ctx);
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
    QTTS_ATRACE(5, ctx);
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
    QTTS_ATRACE(6, ctx);
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
    QTTS_ATRACE(7, ctx);
}

static inline int attn2_nsplit(int max_context) { return (max_context + ATT2_SPLIT - 1) / ATT2_SPLIT; }
size_t attn2_partial_bytes(int B, int Hq, int max_context) {
    return ((size_t)B * Hq * attn2_nsplit(max_context) * (128 + 2) * sizeof(float) + 255) & ~(size_t)255;
}

int attention_decode_fused_impl(const void *qkv, int ldqkv, const void *q_norm_w, const void *k_norm_w,
                                       float eps, const void *rope_cos, const void *rope_sin, int max_pos,
                                       const int64_t *positions, const int32_t *slot_mapping, void *k_cache,
                                       void *v_cache, const int32_t *block_tables, int max_blocks,
                                       const int32_t *context_lens, int max_context, float scale, void *out,
                                       int ldo, int B, int Hq, int Hkv, void *partials, int *counters,
                                       hipStream_t s, const int64_t *row_ids, int id_rows, int opks) {
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
#ifdef QTTS_TUNING
    a.trace = g_trace;                      // (armed by qtts_debug_linear_trace: 8 words per workgroup of this launch)
#endif
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
    // (positions and slots of both rows are wanted here, with the first batch: left alone, the compiler requests each
    // where it is first used and waits for it there, one round trip after the other)
    asm volatile("" ::"v"((int)pos[0]), "v"((int)(pos[0] >> 32)), "v"((int)pos[1]), "v"((int)(pos[1] >> 32)), "v"(slot[0]), "v"(slot[1]));
    Frag cv[2], sv[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const long p = pos[rr] < 0 ? 0 : (pos[rr] >= max_pos ? max_pos - 1 : pos[rr]);
        cv[rr].u = *reinterpret_cast<const uint4 *>(cosT + p * 64 + ((8 * l16) & 63));
        sv[rr].u = *reinterpret_cast<const uint4 *>(sinT + p * 64 + ((8 * l16) & 63));
    }
#define QPIN4(f) "v"((f).u.x), "v"((f).u.y), "v"((f).u.z), "v"((f).u.w)
    // (whole 16-byte requests, all of them up here: the compiler otherwise narrows a request to the words used first
    // and fetches the rest where it is used, behind a wait for everything)
    asm volatile("" ::QPIN4(kraw[0]), QPIN4(kraw[1]), QPIN4(vb[0]), QPIN4(vb[1]), QPIN4(fq), QPIN4(qwv), QPIN4(kwv));
    // (all four cos / sin requests in one batch: left alone, the compiler asks for the second row's where they are
    // first used, one round trip later)
    asm volatile("" ::"v"(cv[0].u.x), "v"(cv[0].u.y), "v"(cv[0].u.z), "v"(cv[0].u.w), "v"(sv[0].u.x), "v"(sv[0].u.y), "v"(sv[0].u.z),
                 "v"(sv[0].u.w), "v"(cv[1].u.x), "v"(cv[1].u.y), "v"(cv[1].u.z), "v"(cv[1].u.w), "v"(sv[1].u.x), "v"(sv[1].u.y),
                 "v"(sv[1].u.z), "v"(sv[1].u.w));
#undef QPIN4
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

int attention_pairs_impl(const void *qkv, int ldqkv, const void *q_norm_w, const void *k_norm_w, float eps,
                                const void *rope_cos, const void *rope_sin, int max_pos, const int64_t *positions,
                                const int32_t *slot_mapping, void *k_cache, void *v_cache, float scale, void *out, int ldo,
                                int N, int Hq, int Hkv, hipStream_t s, int opks) {
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
