// Per-CU ingest rate from L2 by cache-policy bits of the load (raw buffer loads, 16 bytes per lane, 1 KiB per
// wave-instruction, 8 in flight per wave, 8 waves per CU, every workgroup reading the same region).
//   aux bits (gfx94x/gfx950): 1 = sc0, 2 = nt, 16 = sc1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int AUX>
__global__ __launch_bounds__(512) void ingest(const void *src, unsigned region_bytes, int iters, u32x4 *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(src), 0, region_bytes, 0x00020000);
    const unsigned npieces = region_bytes >> 10;
    unsigned piece = (wave * 131u + blockIdx.x * 17u) % npieces;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(((piece + u * nw) % npieces) * 1024u + lane * 16u), 0, AUX);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
        piece = (piece + 8u * nw) % npieces;
    }
    if (acc.x == 0x12345678u) sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int AUX>
static void run(const void *src, unsigned region_bytes, u32x4 *sink) {
    const int iters = 128, waves = 8;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((ingest<AUX>), dim3(256), dim3(64 * waves), 0, 0, src, region_bytes, iters, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double bytes_cu = (double)waves * iters * 8 * 1024.0;
    printf("aux %2d (%s%s%s) region %5.1f MB: %7.1f us  %6.1f GB/s per CU  %5.2f TB/s chip\n", AUX, AUX & 1 ? "sc0 " : "", AUX & 2 ? "nt " : "",
           AUX & 16 ? "sc1" : "", region_bytes / 1e6, ms * 1e3, bytes_cu / (ms * 1e-3) / 1e9, bytes_cu * 256 / (ms * 1e-3) / 1e12);
}

int main() {
    void *src; u32x4 *sink;
    CK(hipMalloc(&src, 64u << 20)); CK(hipMalloc(&sink, 256 * 512 * 16));
    CK(hipMemset(src, 1, 64u << 20));
    for (unsigned region : {1u << 20, 2u << 20, 8u << 20, 48u << 20}) {
        run<0>(src, region, sink);
        run<1>(src, region, sink);
        run<2>(src, region, sink);
        run<3>(src, region, sink);
        run<16>(src, region, sink);
        run<17>(src, region, sink);
    }
    return 0;
}
