// build: hipcc -O3 --offload-arch=gfx950 -o scripts/exp/sdwa_min scripts/exp/sdwa_min.hip
// experiment (round 4): byte-wise minimum of two dwords as four v_min_u32_sdwa, against the scalar definition; v_mul_i32_i24 and
// v_mad_u32_u24 as written in ym_k_raster.hpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#ifndef NOPS
#define NOPS ""
#endif
__device__ __forceinline__ uint32_t bmin(uint32_t x, uint32_t y) {
    asm("v_min_u32_sdwa %0, %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0\n\t" NOPS
        "v_min_u32_sdwa %0, %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_1\n\t" NOPS
        "v_min_u32_sdwa %0, %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:BYTE_2\n\t" NOPS
        "v_min_u32_sdwa %0, %0, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:BYTE_3"
        : "+v"(x) : "v"(y));
    return x;
}
// the form ym_k_raster.hpp uses: two independent dwords interleaved, so that no instruction follows the one that wrote its register
__device__ __forceinline__ void bmin2(uint32_t &x0, uint32_t &x1, uint32_t y0, uint32_t y1) {
#define BM(k) "v_min_u32_sdwa %0, %0, %2 dst_sel:BYTE_" #k " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #k " src1_sel:BYTE_" #k "\n\t" \
              "v_min_u32_sdwa %1, %1, %3 dst_sel:BYTE_" #k " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #k " src1_sel:BYTE_" #k "\n\t"
    asm(BM(0) BM(1) BM(2) BM(3) : "+v"(x0), "+v"(x1) : "v"(y0), "v"(y1));
#undef BM
}
__global__ void k(const uint32_t *a, const uint32_t *b, uint32_t *out, int n, int *bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t x = a[i], y = b[i];
    const uint32_t r = bmin(x, y);
    uint32_t e = 0;
    for (int k = 0; k < 4; k++) { const uint32_t p = (x >> (8 * k)) & 255u, q = (y >> (8 * k)) & 255u; e |= (p < q ? p : q) << (8 * k); }
    if (r != e) { if (atomicAdd(&bad[0], 1) < 6) printf("x %08x y %08x got %08x want %08x\n", x, y, r, e); }
    {
        uint32_t p0 = x, p1 = y ^ 0x5a5a5a5au;
        bmin2(p0, p1, y, x);
        uint32_t e0 = 0, e1 = 0;
        for (int k = 0; k < 4; k++) {
            const uint32_t a0 = (x >> (8 * k)) & 255u, b0 = (y >> (8 * k)) & 255u, a1 = ((y ^ 0x5a5a5a5au) >> (8 * k)) & 255u;
            e0 |= (a0 < b0 ? a0 : b0) << (8 * k); e1 |= (a1 < a0 ? a1 : a0) << (8 * k);
        }
        if (p0 != e0 || p1 != e1) atomicAdd(&bad[3], 1);
    }
    int dy = (int)(x % 41u) - 20, d2;
    asm("v_mul_i32_i24 %0, %1, %1" : "=v"(d2) : "v"(dy));
    if (d2 != dy * dy) atomicAdd(&bad[1], 1);
    int ly = (int)(y % 104u), w = (int)(x % 6u), rw;
    asm("v_mad_u32_u24 %0, %1, 6, %2" : "=v"(rw) : "v"(ly), "v"(w));
    if (rw != ly * 6 + w) atomicAdd(&bad[2], 1);
    out[i] = r;
}
int main() {
    const int n = 1 << 20;
    uint32_t *a, *b, *o; int *bad;
    hipMallocManaged(&a, n * 4); hipMallocManaged(&b, n * 4); hipMallocManaged(&o, n * 4); hipMallocManaged(&bad, 16);
    uint32_t s = 12345;
    for (int i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; a[i] = s; s = s * 1664525u + 1013904223u; b[i] = s; }
    bad[0] = bad[1] = bad[2] = bad[3] = 0;
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, a, b, o, n, bad);
    hipDeviceSynchronize();
    printf("byte-wise minimum (one chain, NOPS = \"%s\") wrong in %d of %d, interleaved pair in %d, v_mul_i32_i24 in %d, v_mad_u32_u24 in %d\n", NOPS, bad[0], n, bad[3], bad[1], bad[2]);
    return 0;
}
