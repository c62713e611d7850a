// How fast can one CU pull bytes that are already in its XCD's L2 into registers?  Every workgroup reads the same
// region (so it is L2-resident after the first pass) with 16-byte loads per lane, 1 KiB per wave-instruction, U loads
// in flight per wave; no arithmetic beyond an XOR that keeps the loads alive.
//   hipcc --offload-arch=gfx950 -O3 -o l2_ingest_probe l2_ingest_probe.hip && ./l2_ingest_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// SPREAD 0: every workgroup walks (almost) the same pieces at the same time, as the workgroups of a GEMM that share
// operands do; SPREAD 1: every wave starts at a pseudo-random piece of the region.
template <int U, bool NT, int SPREAD>
__global__ __launch_bounds__(1024) void ingest(const uint4 *src, size_t region_vec, int iters, uint4 *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // wave w of a workgroup walks the region in 1 KiB pieces, starting at a piece of its own
    size_t piece = SPREAD ? ((size_t)(blockIdx.x * 16 + wave) * 2654435761u >> 7) % (region_vec / 64)
                          : ((size_t)wave * 131 + (size_t)blockIdx.x * 17) % (region_vec / 64);
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint4 *p = src + ((piece + (size_t)u * nw) % (region_vec / 64)) * 64 + lane;
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            if (NT) {
                const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
                v[u] = make_uint4(t.x, t.y, t.z, t.w);
            } else {
                v[u] = *p;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
        piece = (piece + (size_t)U * nw) % (region_vec / 64);
    }
    if (acc.x == 0x12345678u) sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;     // never true: keeps the loads
}

template <int U, bool NT, int SPREAD = 0>
static void run(const char *name, const uint4 *src, size_t region_bytes, int waves, uint4 *sink) {
    const int iters = 256 / U * 4;                       // 1024 KiB per wave
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((ingest<U, NT, SPREAD>), dim3(256), dim3(64 * waves), 0, 0, src, region_bytes / 16, iters, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double bytes_cu = (double)waves * iters * U * 1024.0;
    printf("%-10s region %6.1f MB  %2d waves/CU  %2d loads in flight/wave: %7.1f us  %6.1f GB/s per CU  %5.2f TB/s chip\n", name,
           region_bytes / 1e6, waves, U, ms * 1e3, bytes_cu / (ms * 1e-3) / 1e9, bytes_cu * 256 / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t cap = 512ull << 20;
    uint4 *src, *sink;
    CK(hipMalloc(&src, cap)); CK(hipMalloc(&sink, 256 * 1024 * 16));
    CK(hipMemset(src, 1, cap));
    for (size_t region : {(size_t)1 << 20, (size_t)2 << 20, (size_t)8 << 20, (size_t)24 << 20}) {
        for (int waves : {4, 8}) {
            run<8, false, 0>("same", src, region, waves, sink);
            run<8, false, 1>("spread", src, region, waves, sink);
            run<16, false, 1>("spread", src, region, waves, sink);
        }
        run<8, true, 0>("same nt", src, region, 8, sink);
        run<8, true, 1>("spread nt", src, region, 8, sink);
    }
    return 0;
}
