// build: hipcc -O2 --offload-arch=gfx950 -o scripts/exp/lds_unaligned2 scripts/exp/lds_unaligned2.hip
// experiment (round 2): does ds_read2_b32 honour a BYTE-unaligned address on gfx950 (returning the dwords that start at that
// byte), and what does it cost?  If it did at full rate, the region correlate would need no byte funnel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define BYTES 40960
__global__ __launch_bounds__(1024) void probe(uint32_t *out, int shift, int iters, unsigned long long *cycles, int *bad) {
    __shared__ __attribute__((aligned(16))) unsigned char buf[BYTES];
    for (int i = threadIdx.x; i < BYTES; i += blockDim.x) buf[i] = (unsigned char)((i * 7 + (i >> 8)) & 0xff);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int row = lane & 31, half = lane >> 5;
    const uint32_t lane_off = (uint32_t)(size_t)buf + (row < 26 ? row * 100 + 16 * half : 0);
    uint32_t acc = 0;
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        const uint32_t origin = (uint32_t)(((i * 37) % 300) * 100 + ((i * 12) % 60)) + (uint32_t)shift;
        u32x2 p, q;
        asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %2 offset0:2 offset1:3\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(p), "=&v"(q) : "v"(lane_off + origin) : "memory");
        acc += p.x + p.y + q.x + q.y;
        if (i < 4) { // check against the bytes
            const uint32_t a = lane_off + origin - (uint32_t)(size_t)buf;
            uint32_t e[4];
            for (int j = 0; j < 4; j++) {
                e[j] = 0;
                for (int k = 0; k < 4; k++) e[j] |= (uint32_t)buf[a + 4 * j + k] << (8 * k);
            }
            if (e[0] != p.x || e[1] != p.y || e[2] != q.x || e[3] != q.y) atomicAdd(bad, 1);
        }
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    uint32_t *out; unsigned long long *cyc; int *bad;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8); hipMalloc(&bad, 4);
    for (int shift = 0; shift < 4; shift++) {
        hipMemset(bad, 0, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(probe, dim3(256), dim3(1024), 0, 0, out, shift, 4000, cyc, bad);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe, dim3(256), dim3(1024), 0, 0, out, shift, 4000, cyc, bad);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        int hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
        printf("byte shift %d: %s (%d mismatching lanes), %.2f clk per (wave, 16-byte gather) at CU level\n", shift, hb ? "NOT the bytes at that address" : "exact bytes", hb,
               ms * 1e-3 * 2.4e9 / (4000.0 * 16));
    }
    return 0;
}
