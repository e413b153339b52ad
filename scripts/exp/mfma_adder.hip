// build: hipcc -O2 --offload-arch=gfx950 -o scripts/exp/mfma_adder scripts/exp/mfma_adder.hip
// experiment (round 5): the matrix core as a byte-widening ADDER for the region correlate's gather.
// The gather adds, per patch, 26 rows x 26 bytes (<= 100 each) of an LDS image into 676 sums; a lane reads 16 bytes (13 used) at a
// 4-byte-aligned address and the vector ALU spends ~11 instructions per patch and lane on the byte funnel, the widening and the adds.
// v_mfma_i32_32x32x32_i8 computes D[32][32] += A[32][32] x B[32][32] on int8 with int32 sums.  Give it
//   A[m][k] = the raw bytes as they come out of LDS: lane l holds row m = l & 31 and the 16 bytes k = 16 (l >> 5) .. + 15 -- i.e. lanes
//             0 .. 31 the first 16-byte chunk of the patch's 32 rows, lanes 32 .. 63 the second chunk (13 bytes further): EXACTLY what the
//             kernel's two ds_read2_b32 per lane return today, no instruction in between;
//   B[k][n] = a 0 / 1 SELECTOR that depends on the patch's byte misalignment r alone: chunk 0's byte r + n -> column n (n < 13), chunk 1's
//             byte r1 + n - 13 -> column n (13 <= n < 26), r1 = (r + 1) & 3: the funnel and the widening are the multiplication by B;
// and D accumulates hypothesis (row m, column n) of the wave's angle over every patch, in int32: no packed 16-bit sums, no flush.
// Part 1 derives / checks the operand layout with exact integer data (random bytes, all four misalignments) against a scalar loop;
// part 2 measures the loop: two ds_read2_b32 + one MFMA per patch against the kernel's 45 vector instructions per four patches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define PITCH 100
#define ROWS 106
#define IMG (PITCH * ROWS)

// the selector B for misalignment r, as lane l's four dwords: column n = l & 31, chunk g = l >> 5; byte x of the chunk is k = 16 g + x
__device__ __forceinline__ i32x4 selector(int lane, int r) {
    const int n = lane & 31, g = lane >> 5;
    int x = -1;
    if (g == 0 && n < 13) x = r + n;                        // chunk 0 holds the bytes r .. r + 12 of hypotheses 0 .. 12
    if (g == 1 && n >= 13 && n < 26) x = ((r + 1) & 3) + (n - 13); // chunk 1 starts 13 bytes further: misalignment (r + 13) & 3
    i32x4 b = {0, 0, 0, 0};
    if (x >= 0) {
        const int v = 1 << (8 * (x & 3));
        if ((x >> 2) == 0) b.x = v; else if ((x >> 2) == 1) b.y = v; else if ((x >> 2) == 2) b.z = v; else b.w = v;
    }
    return b;
}

// sums of `np` patches (origins: byte offsets into an LDS image) for one wave: D rows = 32 lattice rows, columns = 26 hypotheses
__global__ __launch_bounds__(64) void k_check(const uint8_t *image, const uint16_t *origins, int np, int *out /* [32][32] */) {
    __shared__ __attribute__((aligned(16))) uint8_t img[IMG + 64];
    for (int i = threadIdx.x; i < IMG + 64; i += 64) img[i] = i < IMG ? image[i] : 0;
    __syncthreads();
    const int lane = threadIdx.x, row = lane & 31, g = lane >> 5;
    i32x16 acc;
    for (int j = 0; j < 16; j++) acc[j] = 0;
    for (int p = 0; p < np; p++) {
        const uint32_t e = origins[p], r = e & 3u;
        // chunk 0 at (e & ~3), chunk 1 at ((e + 13) & ~3) = 12 + ((e + 1) & ~3); rows beyond 25 read row 0 (their sums are not used)
        const uint32_t ad = (uint32_t)(size_t)img + (row < 26 ? row : 0) * PITCH + (g ? 12u + ((e + 1u) & ~3u) : (e & ~3u));
        u32x2 p0, p1;
        asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %2 offset0:2 offset1:3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(p0), "=&v"(p1) : "v"(ad) : "memory");
        const i32x4 a = {(int)p0.x, (int)p0.y, (int)p1.x, (int)p1.y};
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, selector(lane, (int)r), acc, 0, 0, 0);
    }
    // C/D layout of the 32x32 shapes: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    for (int reg = 0; reg < 16; reg++) out[((reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[reg];
}

// rate: every wave gathers `np` patches `iters` times, the way the kernel would: the origins in a register pair (lane i = entries
// 4 i .. 4 i + 3, read with v_readlane), sorted by misalignment so that a run of quads shares its selector (runs: [r4[r], r4[r + 1]) in
// quads); 8 waves per block, blocks per CU by the launch.   MODE 0: reads + MFMA, 1: reads only, 2: MFMA only
template <int MODE>
__global__ __launch_bounds__(512, 6) void k_rate(const uint2 *origins4, const int *r4, int iters, int *sink, unsigned long long *cycles) {
    __shared__ __attribute__((aligned(16))) uint8_t img[4 * IMG + 2700];
    for (int i = threadIdx.x; i < 4 * IMG + 2700; i += 512) img[i] = (uint8_t)(i * 7 % 101);
    __syncthreads();
    const int lane = threadIdx.x & 63, row = lane & 31, g = lane >> 5;
    const uint32_t base = (uint32_t)(size_t)img + (row < 26 ? row : 0) * PITCH + (g ? 12u : 0u);
    const uint2 cev = origins4[lane];
    i32x16 acc;
    for (int j = 0; j < 16; j++) acc[j] = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const i32x4 sel = selector(lane, r);
            const int c1 = __builtin_amdgcn_readfirstlane(r4[r + 1]);
            for (int c = __builtin_amdgcn_readfirstlane(r4[r]); c < c1; c++) {
                const uint32_t ex = __builtin_amdgcn_readlane(cev.x, c), ey = __builtin_amdgcn_readlane(cev.y, c);
                const uint32_t e[4] = {ex & 0xffffu, ex >> 16, ey & 0xffffu, ey >> 16};
                u32x2 q[8];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    // chunk 1 of misalignment 3 starts a dword later (r + 13 = 16): a per-lane constant of this run
                    const uint32_t ad = base + (e[u] & ~3u) + ((r == 3 && g) ? 4u : 0u);
                    if (MODE != 2) asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %2 offset0:2 offset1:3" : "=&v"(q[2 * u]), "=&v"(q[2 * u + 1]) : "v"(ad) : "memory");
                    else { q[2 * u].x = ad; q[2 * u].y = e[u]; q[2 * u + 1].x = ad + 1; q[2 * u + 1].y = e[u] + 1; }
                }
                if (MODE != 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) : : "memory");
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const i32x4 a = {(int)q[2 * u].x, (int)q[2 * u].y, (int)q[2 * u + 1].x, (int)q[2 * u + 1].y};
                    if (MODE != 1) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, sel, acc, 0, 0, 0);
                    else acc[0] += a.x ^ a.y ^ a.z ^ a.w;
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int j = 0; j < 16; j++) s ^= acc[j];
    if (s == 0x7fffffff) sink[0] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    // ---- part 1: exactness
    std::vector<uint8_t> image(IMG);
    srand(12345);
    for (auto &v : image) v = (uint8_t)(rand() % 101);
    const int np = 200;
    std::vector<uint16_t> org(np);
    for (int p = 0; p < np; p++) org[p] = (uint16_t)((rand() % 80) * PITCH + rand() % 60); // origin row < 80, byte < 60: every misalignment
    std::vector<int> want(32 * 32, 0);
    for (int p = 0; p < np; p++)
        for (int m = 0; m < 26; m++)
            for (int n = 0; n < 26; n++) want[m * 32 + n] += image[org[p] + m * PITCH + n];
    uint8_t *d_img; uint16_t *d_org; int *d_out;
    CHECK(hipMalloc(&d_img, IMG)); CHECK(hipMalloc(&d_org, np * 2)); CHECK(hipMalloc(&d_out, 32 * 32 * 4));
    CHECK(hipMemcpy(d_img, image.data(), IMG, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_org, org.data(), np * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, d_img, d_org, np, d_out);
    std::vector<int> got(32 * 32);
    CHECK(hipMemcpy(got.data(), d_out, 32 * 32 * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int m = 0; m < 26; m++)
        for (int n = 0; n < 26; n++)
            if (got[m * 32 + n] != want[m * 32 + n]) { if (bad < 8) printf("  (%d, %d): got %d want %d\n", m, n, got[m * 32 + n], want[m * 32 + n]); bad++; }
    int spill = 0;
    for (int m = 0; m < 32; m++) for (int n = 26; n < 32; n++) spill += got[m * 32 + n] != 0;
    printf("part 1: %d patches, 676 sums: %d differ; %d non-zero sums in the unused columns 26 .. 31\n", np, bad, spill);
    // ---- part 2: rate
    unsigned long long *d_cyc; int *d_sink;
    CHECK(hipMalloc(&d_cyc, 8192 * 8)); CHECK(hipMalloc(&d_sink, 64));
    const int iters = 400, nq4 = 32; // 128 patches per wave and pass: 32 quads, sorted by misalignment, 8 quads each
    std::vector<uint32_t> o4(128);
    std::vector<int> r4 = {0, 8, 16, 24, 32};
    for (int q = 0; q < nq4; q++)
        for (int u = 0; u < 4; u++) {
            const uint32_t r = (uint32_t)(q / 8), e = (uint32_t)(((q * 13 + u * 7) % 78) * PITCH + ((q * 5 + u * 11) % 15) * 4) + r;
            o4[2 * q + (u >> 1)] = (u & 1) ? (o4[2 * q + (u >> 1)] | e << 16) : e;
        }
    uint2 *d_o4; int *d_r4;
    CHECK(hipMalloc(&d_o4, 64 * 8)); CHECK(hipMalloc(&d_r4, 5 * 4));
    CHECK(hipMemset(d_o4, 0, 64 * 8));
    CHECK(hipMemcpy(d_o4, o4.data(), nq4 * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_r4, r4.data(), 5 * 4, hipMemcpyHostToDevice));
    for (int mode = 0; mode < 3; mode++) {
        for (int bpc = 1; bpc <= 3; bpc++) {
            const int blocks = 256 * bpc;
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(512), 0, 0, d_o4, d_r4, iters, d_sink, d_cyc);
                else if (mode == 1) hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(512), 0, 0, d_o4, d_r4, iters, d_sink, d_cyc);
                else hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(512), 0, 0, d_o4, d_r4, iters, d_sink, d_cyc);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            }
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> cyc(blocks);
            CHECK(hipMemcpy(cyc.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost));
            double avg = 0; for (auto c : cyc) avg += (double)c; avg /= blocks;
            const double patches_per_cu = (double)bpc * 8 * iters * nq4 * 4;
            printf("part 2: %s, %d block(s) of 8 waves per CU: %.3f ms = %.2f CU clocks of 2.4 GHz per patch; in-kernel clock %.2f GHz -> %.2f shader clocks per patch\n",
                   mode == 0 ? "reads + MFMA" : mode == 1 ? "reads only  " : "MFMA only   ", bpc, ms, ms * 1e-3 * 2.4e9 / patches_per_cu,
                   avg / (ms * 1e-3) * 1e-9, avg / (patches_per_cu / bpc));
        }
    }
    return bad != 0;
}
