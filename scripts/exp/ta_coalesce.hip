// Experiment: how many vector-L1 (TCP) cycles does one dwordx4 wave-load cost as a function of how its 64 lane
// addresses fall into cache lines?  Every wave issues ITER dependent-free loads from an L1-resident region; the lane
// -> address pattern is the variable.  Build: hipcc --offload-arch=gfx950 -O3 -o ta_coalesce ta_coalesce.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// pattern p: byte offset of lane l inside the region for iteration it (region = `span` bytes, power of two)
__device__ __forceinline__ unsigned lane_offset(int p, int l, int it, unsigned span) {
    unsigned o;
    switch (p) {
    case 0: o = 16u * l + 1024u * it; break;                                  // fully contiguous 1 KB per wave-load
    case 1: o = (l >> 1) * 640u + (l & 1) * 16u + 48u * it; break;            // pairs share a row (the correlate kernel's shape), rows 640 B apart
    case 2: o = (l >> 2) * 640u + (l & 3) * 16u + 64u * it; break;            // quads: 64 contiguous, 64-aligned bytes
    case 3: o = l * 640u + 16u * it; break;                                   // every lane its own row
    case 4: o = (l >> 1) * 640u + (l & 1) * 16u + 48u * it + 4u; break;       // pairs, dword-aligned but not 16-aligned
    case 5: o = (l >> 2) * 640u + (l & 3) * 16u + 64u * it + 4u; break;       // quads, dword-aligned but not 16-aligned
    case 6: o = (l >> 3) * 640u + (l & 7) * 16u + 128u * it; break;           // octets: one full 128 B line
    default: o = (l >> 4) * 640u + (l & 15) * 16u + 256u * it; break;         // 16 lanes contiguous (256 B)
    }
    return o & (span - 1u) & ~3u;
}

template <int P>
__global__ __launch_bounds__(256) void probe(const unsigned char *buf, unsigned span, int iters, unsigned *sink, unsigned long long *cycles) {
    const int l = threadIdx.x & 63;
    const unsigned char *base = buf + (size_t)(blockIdx.x % 64) * span; // a few regions so that blocks do not all share lines
    unsigned acc = 0;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; it += 8) {
        uint4 w[8];
#pragma unroll
        for (int u = 0; u < 8; u++) w[u] = *reinterpret_cast<const uint4 *>(__builtin_assume_aligned(base + lane_offset(P, l, it + u, span), 4));
#pragma unroll
        for (int u = 0; u < 8; u++) acc += w[u].x + w[u].y + w[u].z + w[u].w;
    }
    const unsigned long long t1 = clock64();
    if (acc == 0x12345678u) sink[0] = acc;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    const unsigned span = 32768; // bytes per region: L1/L2 resident
    const int iters = 4096, blocks = 256 * 8;
    unsigned char *buf; unsigned *sink; unsigned long long *cyc;
    CHECK(hipMalloc(&buf, (size_t)64 * span + 4096));
    CHECK(hipMemset(buf, 1, (size_t)64 * span + 4096));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * blocks));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const char *names[] = {"contiguous 1 KB", "pairs (32 B) per row", "quads (64 B aligned) per row", "one lane per row",
                           "pairs, +4 B", "quads, +4 B", "octets (128 B line)", "16 lanes (256 B)"};
#define RUN(P)                                                                                                        \
    do {                                                                                                              \
        hipLaunchKernelGGL(probe<P>, dim3(blocks), dim3(256), 0, 0, buf, span, 64, sink, cyc);                        \
        CHECK(hipDeviceSynchronize());                                                                                \
        CHECK(hipEventRecord(a));                                                                                     \
        hipLaunchKernelGGL(probe<P>, dim3(blocks), dim3(256), 0, 0, buf, span, iters, sink, cyc);                     \
        CHECK(hipEventRecord(b));                                                                                     \
        CHECK(hipEventSynchronize(b));                                                                                \
        float ms;                                                                                                     \
        CHECK(hipEventElapsedTime(&ms, a, b));                                                                        \
        const double loads = (double)blocks * 4 * iters;                                                              \
        /* 256 CUs; clock from the event time: report wave-loads per us per CU and ns per wave-load per CU */         \
        printf("%-32s %8.3f ms  %7.2f ns per wave-load per CU  (%.1f clk at 2.4 GHz)  %.2f TB/s\n", names[P], ms,     \
               ms * 1e6 / (loads / 256.0), ms * 1e6 / (loads / 256.0) * 2.4, loads * 1024 / (ms * 1e-3) / 1e12);      \
    } while (0)
    RUN(0); RUN(1); RUN(2); RUN(3); RUN(4); RUN(5); RUN(6); RUN(7);
    return 0;
}
