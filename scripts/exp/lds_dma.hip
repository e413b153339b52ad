// build: hipcc -O2 --offload-arch=gfx950 -o scripts/exp/lds_dma scripts/exp/lds_dma.hip
// experiment (round 3): global_load_lds_dword / dwordx3 / dwordx4 on gfx950 -- where does lane i's data land (LDS base +
// i * bytes per lane?), and how many bytes per second does a CU take in per width when every wave streams rows of a big
// buffer into a 24 KB LDS image (the gather correlate's staging pattern: 7 waves per block, 3 blocks per CU)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int BYTES>
__device__ __forceinline__ void dma(const void *src, void *lds) {
    const __attribute__((address_space(1))) void *g = (const __attribute__((address_space(1))) void *)src;
    __attribute__((address_space(3))) void *l = (__attribute__((address_space(3))) void *)lds;
    if constexpr (BYTES == 4) __builtin_amdgcn_global_load_lds(g, l, 4, 0, 0);
    else if constexpr (BYTES == 12) __builtin_amdgcn_global_load_lds(g, l, 12, 0, 0);
    else __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
}
template <int BYTES>
__global__ void where(const uint32_t *src, uint32_t *out) {
    __shared__ __attribute__((aligned(16))) uint32_t buf[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) buf[i] = 0xdeadbeefu;
    __syncthreads();
    // lane i reads BYTES bytes from src + 64 * i dwords (distinct rows) -> where do they land?
    dma<BYTES>(src + 64 * threadIdx.x, buf);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = buf[i];
}
template <int BYTES>
__global__ __launch_bounds__(448) void stream(const uint8_t *src, size_t src_bytes, int regions, uint32_t *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int image = 24576, per_instr = 64 * BYTES, n_instr = image / per_instr;
    uint32_t acc = 0;
    for (int r = 0; r < regions; r++) {
        const size_t base = ((size_t)(blockIdx.x * regions + r) * 40960) % (src_bytes - 65536);
        unsigned char *buf = lds + (r & 1) * image;
        for (int t = wave; t < n_instr; t += nw) dma<BYTES>(src + base + (size_t)t * per_instr + lane * BYTES, buf + t * per_instr);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += *reinterpret_cast<uint32_t *>(buf + 4 * threadIdx.x);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    uint32_t *src, *out;
    const size_t n = (size_t)1 << 28; // 1 GiB
    hipMalloc(&src, n * 4); hipMalloc(&out, 4096 * 1024 * 4);
    std::vector<uint32_t> h(64 * 64);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)i;
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<uint32_t> o(1024);
    auto show = [&](const char *name, int bytes) {
        hipMemcpy(o.data(), out, 4096, hipMemcpyDeviceToHost);
        printf("%s: LDS dwords 0..11 = ", name);
        for (int i = 0; i < 12; i++) printf("%x ", o[i]);
        // lane 1's first dword is src[64]: where is it?
        int at = -1;
        for (int i = 0; i < 1024; i++) if (o[i] == 64) { at = i; break; }
        printf(" | lane 1's first dword lands at LDS dword %d (lane * %d bytes would be %d)\n", at, bytes, bytes / 4);
    };
    hipLaunchKernelGGL(where<4>, dim3(1), dim3(64), 0, 0, src, out); hipDeviceSynchronize(); show("dword  ", 4);
    hipLaunchKernelGGL(where<12>, dim3(1), dim3(64), 0, 0, src, out); hipDeviceSynchronize(); show("dwordx3", 12);
    hipLaunchKernelGGL(where<16>, dim3(1), dim3(64), 0, 0, src, out); hipDeviceSynchronize(); show("dwordx4", 16);
    hipMemset(src, 1, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int regions = 200, blocks = 256 * 3;
    auto timeit = [&](auto kern, const char *name) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(448), 49152, 0, (const uint8_t *)src, n * 4, regions, out);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(448), 49152, 0, (const uint8_t *)src, n * 4, regions, out);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)blocks * regions * 24576;
        printf("%s: %.1f us for %.2f GB = %.2f TB/s chip, %.1f GB/s per CU (3 blocks of 7 waves per CU, wait + barrier per 24 KB image)\n", name, ms * 1e3,
               bytes * 1e-9, bytes / ms * 1e-9, bytes / ms * 1e-6 / 256);
    };
    timeit(stream<4>, "dword  ");
    timeit(stream<12>, "dwordx3");
    timeit(stream<16>, "dwordx4");
    return 0;
}
