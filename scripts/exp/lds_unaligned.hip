// build: hipcc -O2 --offload-arch=gfx950 -o scripts/exp/lds_unaligned scripts/exp/lds_unaligned.hip
// experiment: does ds_read_b128 honour 4-byte-aligned (not 16-byte-aligned) LDS addresses on gfx950, and what does it cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(uint32_t *out, int shift_bytes, int iters, unsigned long long *cycles) {
    __shared__ __attribute__((aligned(16))) uint32_t buf[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t addr = (uint32_t)(size_t)buf + (lane >> 1) * 2 * 112 + (lane & 1) * 16 + shift_bytes;
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        u32x4 x;
        asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(addr + (uint32_t)((i & 3) * 224 * 32)) : "memory");
        acc += x;
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x < 64) {
        u32x4 x;
        asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(addr) : "memory");
        out[lane * 4 + 0] = x.x; out[lane * 4 + 1] = x.y; out[lane * 4 + 2] = x.z; out[lane * 4 + 3] = x.w;
        if (lane == 0) { cycles[0] = t1 - t0; out[256] = acc.x + acc.y + acc.z + acc.w; }
    }
}
int main() {
    uint32_t *out; unsigned long long *cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    for (int sh = 0; sh <= 12; sh += 4) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, sh, 4096, cyc);
        uint32_t h[256]; unsigned long long c;
        hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        int ok = 1;
        for (int lane = 0; lane < 64; lane++) {
            const uint32_t first = ((lane >> 1) * 2 * 112 + (lane & 1) * 16 + sh) / 4;
            for (int j = 0; j < 4; j++) ok &= h[lane * 4 + j] == first + j;
        }
        printf("shift %2d B: %s, %.1f clk per ds_read_b128 (1 wave, dependent)\n", sh, ok ? "correct" : "WRONG", (double)c / 4096);
    }
    return 0;
}
