// build: hipcc -O2 --offload-arch=gfx950 -o scripts/exp/lds_gather scripts/exp/lds_gather.hip
// experiment (round 2): THROUGHPUT of the LDS gather a region-staged correlate kernel would make.
// Every lane (row = lane >> 1, half = lane & 1; 52 lanes active like a 26 x 26 lattice) reads the 16 bytes that start at
// row * pitch + 16 * half + a wave-uniform, 4-byte-aligned origin.  Variants of the instruction mix and of the pitch;
// NW waves per block, one block per CU; clocks per (wave, gather) at CU level = block time / (NW * gathers per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define REGION_BYTES 40960

template <int V>
__device__ __forceinline__ void gather(uint32_t addr, uint32_t (&x)[4]) {
    if (V == 0) { // one ds_read_b128 (address 16-byte aligned by the caller)
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
        x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    } else if (V == 1) { // one ds_read_b128 at a 4-byte-aligned address
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
        x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    } else if (V == 2) { // two ds_read2_b32
        u32x2 p, q;
        asm volatile("ds_read2_b32 %0, %2 offset0:0 offset1:1\n ds_read2_b32 %1, %2 offset0:2 offset1:3" : "=&v"(p), "=&v"(q) : "v"(addr) : "memory");
        x[0] = p.x; x[1] = p.y; x[2] = q.x; x[3] = q.y;
    } else if (V == 3) { // four ds_read_b32
        asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:4\n ds_read_b32 %2, %4 offset:8\n ds_read_b32 %3, %4 offset:12"
                     : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]) : "v"(addr) : "memory");
    } else if (V == 4) { // two ds_read_b64 at 4-byte-aligned addresses
        u32x2 p, q;
        asm volatile("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:8" : "=&v"(p), "=&v"(q) : "v"(addr) : "memory");
        x[0] = p.x; x[1] = p.y; x[2] = q.x; x[3] = q.y;
    } else if (V == 5) { // two aligned ds_read_b128 (what round 1's staged kernel does: 32 bytes per lane)
        u32x4 v, w;
        asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16" : "=&v"(v), "=&v"(w) : "v"(addr) : "memory");
        x[0] = v.x + w.x; x[1] = v.y + w.y; x[2] = v.z + w.z; x[3] = v.w + w.w;
    }
}

template <int V>
__global__ __launch_bounds__(1024) void probe(uint32_t *out, int pitch, int iters, int lanemap, unsigned long long *cycles) {
    __shared__ __attribute__((aligned(16))) unsigned char buf[REGION_BYTES];
    for (int i = threadIdx.x; i < REGION_BYTES / 4; i += blockDim.x) reinterpret_cast<uint32_t *>(buf)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int row, half;
    if (lanemap == 0) { row = lane >> 1; half = lane & 1; }
    else { row = lane & 31; half = lane >> 5; }
    const bool on = row < 26;
    const uint32_t lane_off = (uint32_t)(size_t)buf + (on ? row * pitch + 16 * half : 0);
    uint32_t acc[4] = {0, 0, 0, 0};
    // 64 wave-uniform origins per wave, one per lane, picked up with v_readlane (the address arithmetic must not be what is timed)
    const int span_rows = (REGION_BYTES / pitch) - 26 - 1, span_x = pitch - 32 - 4;
    uint32_t seed = (12345u + 977u * wave + 31u * blockIdx.x + 7919u * lane) * 2654435761u;
    seed = seed * 1664525u + 1013904223u;
    uint32_t oy = (seed >> 8) % (uint32_t)span_rows, ox = ((seed >> 20) % (uint32_t)span_x) & ~3u;
    if (V == 0 || V == 5) ox = 0; // 16-byte aligned (needs pitch % 16 == 0)
    const uint32_t origin = oy * pitch + ox;
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        uint32_t x[8][4];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t ub = __builtin_amdgcn_readlane(origin, (i * 8 + u) & 63);
            gather<V>(lane_off + ub, x[u]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] += x[u][j];
    }
    __syncthreads();
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int V>
static void run(const char *name, int pitch, int nw, int lanemap, int blocks) {
    uint32_t *out; unsigned long long *cyc;
    hipMalloc(&out, (size_t)blocks * 1024 * 4); hipMalloc(&cyc, (size_t)blocks * 8);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(64 * nw), 0, 0, out, pitch, iters, lanemap, cyc);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(64 * nw), 0, 0, out, pitch, iters, lanemap, cyc);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long *h = (unsigned long long *)malloc((size_t)blocks * 8);
    hipMemcpy(h, cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < blocks; i++) mean += (double)h[i];
    mean /= blocks;
    // clock64 counts at 100 MHz on gfx9 (s_memtime would be core clocks): report both raw and a 2.4 GHz estimate
    printf("%-28s pitch %3d nw %2d map %d blocks %4d: %6.2f ticks %6.2f clk\n", name, pitch, nw,
           lanemap, blocks, mean / ((double)iters * 8 * nw), ms * 1e-3 * 2.4e9 / ((double)iters * 8 * nw * ((blocks + 255) / 256)));
    free(h); hipFree(out); hipFree(cyc);
}

int main(int argc, char **argv) {
    if (argc > 1) { // extra pitches for the lane map the kernel uses: lds_gather 132 148 164 ...
        for (int i = 1; i < argc; i++) run<2>("2 x read2_b32", atoi(argv[i]), 16, 1, 256);
        return 0;
    }
    const int pitches[] = {100, 116, 104, 112};
    for (int nw : {8, 16}) {
        for (int p : pitches) {
            if (p % 16 == 0) { run<0>("b128 aligned", p, nw, 0, 256); run<5>("2 x b128 aligned", p, nw, 0, 256); }
            run<1>("b128 at 4-byte alignment", p, nw, 0, 256);
            run<2>("2 x read2_b32", p, nw, 0, 256);
            run<3>("4 x read_b32", p, nw, 0, 256);
            run<4>("2 x read_b64 (4-byte)", p, nw, 0, 256);
            run<2>("2 x read2_b32", p, nw, 1, 256);
        }
    }
    return 0;
}
