// Probe, not product: what a chain of dependent decode-GEMM launches would gain if a launch could start before its
// predecessor has ended.  The kernel is a cut-down single-tile decode GEMM (8 rows, K = 1024, N = 4096: 256
// workgroups of four waves, each wave 8 k-steps of one 16-row weight tile, v_mfma_f32_16x16x32_bf16, cross-wave sum
// through LDS).  In `overlap` mode a workgroup requests its weight fragments first, then waits (bounded) for the
// predecessor's completion word, then reads the activations with agent-scope loads; every launch publishes its
// output with agent-scope stores and a completion word (one per launch, set by the last workgroup to arrive at a
// counter; or one per workgroup, all of them watched by every consumer workgroup).  tools/pdl_probe.py runs a chain of launches
// the plain way (one queue, kernel boundaries) and the overlapped way (alternating on two queues) and compares.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
union Frag { uint4 u; bf16x8_t v; uint64_t q[2]; u32x4_t n; };

#define SPIN_LIMIT 400000          // x ~0.1 us: a wait that is never satisfied ends after tens of milliseconds

struct Sync {
    const int *flag_in;            // predecessor's completion word (null: none)
    int *flag_out, *counter;       // this launch's completion word and arrival counter
    int epoch;                     // value that means "done" in this replay
    int *timeouts;                 // incremented by a workgroup whose wait ran out
};

__device__ __forceinline__ uint32_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

template <bool OVERLAP>
__global__ __launch_bounds__(256) void probe_gemm(const uint16_t *x, int ldx, const uint4 *wp, uint16_t *out, int ldo,
                                                  int N, const Sync sy) {
    __shared__ float4_t red[4 * 64];
    __shared__ int ok;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r = lane & 15;
    const int t = blockIdx.x, ks0 = wave * 8, KS = 32;
    Frag a[8], xb[8];
    const uint4 *wptr = wp + ((size_t)t * KS + ks0) * 64 + lane;
    const uint16_t *xp = x + (size_t)(r & 7) * ldx + 32 * ks0 + 8 * g;
    if (!OVERLAP) {
#pragma unroll
        for (int i = 0; i < 8; ++i) xb[i].u = *reinterpret_cast<const uint4 *>(xp + 32 * i);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i].n = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(wptr + i * 64));
    if (OVERLAP) {
        if (sy.counter) {
            // one completion word per launch (set by the predecessor's last workgroup)
            if (threadIdx.x == 0) {
                int seen = 1;
                if (sy.flag_in) {
                    seen = 0;
                    for (int spin = 0; spin < SPIN_LIMIT; ++spin) {
                        if (__hip_atomic_load(sy.flag_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == sy.epoch) { seen = 1; break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (!seen) atomicAdd(sy.timeouts, 1);
                }
                ok = seen;
            }
            __syncthreads();
        } else if (sy.flag_in) {
            // one word per workgroup of the predecessor (gridDim.x == blockDim.x == 256 here): thread j watches word j
            int seen = 0;
            for (int spin = 0; spin < SPIN_LIMIT / 8; ++spin) {
                seen = __hip_atomic_load(sy.flag_in + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == sy.epoch;
                if (__syncthreads_and(seen)) { seen = 1; break; }
                seen = 0;
            }
            if (!seen && threadIdx.x == 0) atomicAdd(sy.timeouts, 1);
        }
        // the predecessor's rows: agent-scope loads (this XCD's L2 may still hold the previous occupant of the lines)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint64_t *q = reinterpret_cast<const uint64_t *>(xp + 32 * i);
            xb[i].q[0] = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            xb[i].q[1] = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    float4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].v, xb[i].v, acc, 0, 0, 0);
    red[wave * 64 + lane] = acc;
    __syncthreads();
    if (wave == 0) {
        float4_t s = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
        if (r < 8) {
            // keep the numbers bounded along the chain
            const uint64_t v = (uint64_t)(f2bf(s[0] * 0.6f) | (f2bf(s[1] * 0.6f) << 16)) |
                               ((uint64_t)(f2bf(s[2] * 0.6f) | (f2bf(s[3] * 0.6f) << 16)) << 32);
            uint64_t *dst = reinterpret_cast<uint64_t *>(out + (size_t)r * ldo + 16 * t + 4 * g);
            if (OVERLAP) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *dst = v;
        }
    }
    if (OVERLAP) {
        // publish: this workgroup's stores acknowledged, then the arrival counter; the last arriver sets the word
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (sy.counter) {
                const int old = __hip_atomic_fetch_add(sy.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == (int)gridDim.x - 1) {
                    __hip_atomic_store(sy.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(sy.flag_out, sy.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                __hip_atomic_store(sy.flag_out + blockIdx.x, sy.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

extern "C" int pdl_probe_launch(const void *x, int ldx, const void *wp, void *out, int ldo, int N, int overlap,
                                const int *flag_in, int *flag_out, int *counter, int epoch, int *timeouts, void *stream) {
    if (!x || !wp || !out || N <= 0 || N % 16 != 0 || ldx < 1024 || ldo < N) return -1;
    if (overlap && (!flag_out || !timeouts || (!counter && N != 4096))) return -1;   // counter null: a word per workgroup
    Sync sy{flag_in, flag_out, counter, epoch, timeouts};
    dim3 grid(N / 16), block(256);
    if (overlap)
        hipLaunchKernelGGL(probe_gemm<true>, grid, block, 0, (hipStream_t)stream, (const uint16_t *)x, ldx, (const uint4 *)wp,
                           (uint16_t *)out, ldo, N, sy);
    else
        hipLaunchKernelGGL(probe_gemm<false>, grid, block, 0, (hipStream_t)stream, (const uint16_t *)x, ldx, (const uint4 *)wp,
                           (uint16_t *)out, ldo, N, sy);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
