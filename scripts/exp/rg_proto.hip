// build: hipcc -O3 --offload-arch=gfx950 -o scripts/exp/rg_proto scripts/exp/rg_proto.hip
// experiment (round 4): what does the GATHER LOOP of correlate_region_kernel (ym_k_region.hpp) reach on its own -- the same
// instructions, the same LDS image and lane map, entry lists in LDS, but no staging, no region walk, no scoring -- and what
// do variants of it reach?  Every block = 8 waves; a wave gathers Q patches per round from a resident 42 KB image, R rounds,
// optionally a barrier per round.  Reported: CU clocks per patch (all waves of the CU together), i.e. the time the whole
// kernel would need for its 93 M patches per launch of 4096 items if nothing but this loop ran.
//   variant 0: the kernel's loop (rg_gather4: four patches per trip, pairs share a funnel)
//   variant 1: misalignment-indexed accumulator sets (no funnel in the loop), software-pipelined by the compiler
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <type_traits>

#define RG_W 64
#define RG_H 80
#define RG_PITCH 100
#define RG_ROWS (RG_H + 26)
#define RG_CLS (RG_PITCH * RG_ROWS)
#define RG_G 13
#define RG_ZERO (4 * RG_CLS)
#define RG_LDS_BYTES (RG_ZERO + 26 * RG_PITCH + 32)

typedef unsigned int rg_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void rg_funnel_pair(const rg_u32x2 &pa, const rg_u32x2 &qa, const rg_u32x2 &pb, const rg_u32x2 &qb, uint32_t rr, uint32_t (&x)[4]) {
    const uint32_t s0 = pa.x + pb.x, s1 = pa.y + pb.y, s2 = qa.x + qb.x, s3 = qa.y + qb.y;
    x[0] = __builtin_amdgcn_alignbyte(s1, s0, rr);
    x[1] = __builtin_amdgcn_alignbyte(s2, s1, rr);
    x[2] = __builtin_amdgcn_alignbyte(s3, s2, rr);
    x[3] = __builtin_amdgcn_alignbyte(0u, s3, rr);
}
__device__ __forceinline__ void rg_gather4(uint32_t (&acc)[8], uint32_t lane_off, uint2 ee) {
    const uint32_t ad0 = lane_off + (ee.x & 0xffffu), ad1 = lane_off + (ee.x >> 16), ad2 = lane_off + (ee.y & 0xffffu), ad3 = lane_off + (ee.y >> 16);
    rg_u32x2 p0, q0, p1, q1, p2, q2, p3, q3;
    asm volatile("ds_read2_b32 %0, %8 offset1:1\n\tds_read2_b32 %1, %8 offset0:2 offset1:3\n\t"
                 "ds_read2_b32 %2, %9 offset1:1\n\tds_read2_b32 %3, %9 offset0:2 offset1:3\n\t"
                 "ds_read2_b32 %4, %10 offset1:1\n\tds_read2_b32 %5, %10 offset0:2 offset1:3\n\t"
                 "ds_read2_b32 %6, %11 offset1:1\n\tds_read2_b32 %7, %11 offset0:2 offset1:3\n\t"
                 "s_waitcnt lgkmcnt(4)"
                 : "=&v"(p0), "=&v"(q0), "=&v"(p1), "=&v"(q1), "=&v"(p2), "=&v"(q2), "=&v"(p3), "=&v"(q3)
                 : "v"(ad0 & ~3u), "v"(ad1 & ~3u), "v"(ad2 & ~3u), "v"(ad3 & ~3u)
                 : "memory");
    uint32_t x[2][4];
    rg_funnel_pair(p0, q0, p1, q1, ad0 & 3u, x[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p2), "+v"(q2), "+v"(p3), "+v"(q3) : : "memory");
    rg_funnel_pair(p2, q2, p3, q3, ad2 & 3u, x[1]);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        acc[2 * j] = acc[2 * j] + (x[0][j] & 0x00FF00FFu) + (x[1][j] & 0x00FF00FFu);
        if (j < 3) acc[2 * j + 1] = acc[2 * j + 1] + (x[0][j] >> 8) + (x[1][j] >> 8);
    }
}
__device__ __forceinline__ void rg_odd(uint32_t (&acc)[8]) {
#pragma unroll
    for (int j = 0; j < 4; j++) acc[2 * j + 1] -= (acc[2 * j] >> 16) << 8;
}

// variant 2: one v_perm_b32 per 16-bit pair of hypotheses does the byte funnel AND the widening (selector byte 0x0c = zero):
// even = (B[rr], 0, B[rr + 2], 0), odd = (B[rr + 1], 0, B[rr + 3], 0) of the eight bytes s[j + 1] : s[j]; no v_alignbyte, no v_and,
// no v_lshrrev, clean 16-bit fields in the odd accumulators too (no rg_odd at the end)
__device__ __forceinline__ void rg_perm_pair(const rg_u32x2 &pa, const rg_u32x2 &qa, const rg_u32x2 &pb, const rg_u32x2 &qb, uint32_t rr,
                                             uint32_t (&e)[4], uint32_t (&o)[4]) {
    const uint32_t s0 = pa.x + pb.x, s1 = pa.y + pb.y, s2 = qa.x + qb.x, s3 = qa.y + qb.y;
    const uint32_t selE = rr * 0x00010001u + 0x0c020c00u, selO = selE + 0x00010001u;
    e[0] = __builtin_amdgcn_perm(s1, s0, selE); o[0] = __builtin_amdgcn_perm(s1, s0, selO);
    e[1] = __builtin_amdgcn_perm(s2, s1, selE); o[1] = __builtin_amdgcn_perm(s2, s1, selO);
    e[2] = __builtin_amdgcn_perm(s3, s2, selE); o[2] = __builtin_amdgcn_perm(s3, s2, selO);
    e[3] = __builtin_amdgcn_perm(0u, s3, selE);
}
__device__ __forceinline__ void rg_gather4_perm(uint32_t (&acc)[8], uint32_t lane_off, uint2 ee) {
    const uint32_t ad0 = lane_off + (ee.x & 0xffffu), ad1 = lane_off + (ee.x >> 16), ad2 = lane_off + (ee.y & 0xffffu), ad3 = lane_off + (ee.y >> 16);
    rg_u32x2 p0, q0, p1, q1, p2, q2, p3, q3;
    asm volatile("ds_read2_b32 %0, %8 offset1:1\n\tds_read2_b32 %1, %8 offset0:2 offset1:3\n\t"
                 "ds_read2_b32 %2, %9 offset1:1\n\tds_read2_b32 %3, %9 offset0:2 offset1:3\n\t"
                 "ds_read2_b32 %4, %10 offset1:1\n\tds_read2_b32 %5, %10 offset0:2 offset1:3\n\t"
                 "ds_read2_b32 %6, %11 offset1:1\n\tds_read2_b32 %7, %11 offset0:2 offset1:3\n\t"
                 "s_waitcnt lgkmcnt(4)"
                 : "=&v"(p0), "=&v"(q0), "=&v"(p1), "=&v"(q1), "=&v"(p2), "=&v"(q2), "=&v"(p3), "=&v"(q3)
                 : "v"(ad0 & ~3u), "v"(ad1 & ~3u), "v"(ad2 & ~3u), "v"(ad3 & ~3u)
                 : "memory");
    uint32_t e[2][4], o[2][4];
    rg_perm_pair(p0, q0, p1, q1, ad0 & 3u, e[0], o[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p2), "+v"(q2), "+v"(p3), "+v"(q3) : : "memory");
    rg_perm_pair(p2, q2, p3, q3, ad2 & 3u, e[1], o[1]);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        acc[2 * j] = acc[2 * j] + e[0][j] + e[1][j];
        if (j < 3) acc[2 * j + 1] = acc[2 * j + 1] + o[0][j] + o[1][j];
    }
}

struct Args {
    const uint16_t *entries; // [waves in the grid][Q]: LDS offsets of patch origins, sorted by misalignment in runs of even length
    const uint16_t *runs;    // [waves in the grid][4]: first PAIR of the runs with misalignment 1, 2, 3 and the pair count
    uint32_t *out;           // [waves in the grid][64 lanes][13]
    int Q, R, barrier;
    int lds_pad;             // (dynamic LDS bytes are what limits the blocks per CU)
};

// the image every block gathers from: bytes <= 100, the same in every block
__device__ __forceinline__ void fill_image(unsigned char *region, int tid, int nt) {
    for (int i = tid; i < RG_LDS_BYTES / 4; i += nt) {
        uint32_t v = 0u;
        if (i < RG_ZERO / 4)
            for (int b = 0; b < 4; b++) v |= ((uint32_t)(((i * 4 + b) * 2654435761u) >> 24) % 101u) << (8 * b);
        reinterpret_cast<uint32_t *>(region)[i] = v;
    }
}

template <int NW, int VAR>
__global__ __launch_bounds__(64 * NW) void proto(Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    unsigned char *region = dyn;
    uint2 (*elist)[64] = reinterpret_cast<uint2 (*)[64]>(dyn + ((RG_LDS_BYTES + 15) & ~15));
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gw = blockIdx.x * NW + wave;
    fill_image(region, tid, 64 * NW);
    const int Q = a.Q;
    const uint2 *src = reinterpret_cast<const uint2 *>(a.entries + (size_t)gw * Q);
    if (4 * lane < Q) elist[wave][lane] = src[lane];
    __syncthreads();
    const int row = lane & 31, half = lane >> 5;
    const bool job = row < 26;
    const uint32_t lds0 = (uint32_t)(size_t)region;
    if (VAR == 2) {
        const uint32_t lane_off = lds0 + (uint32_t)((job ? row : 0) * RG_PITCH + half * RG_G);
        uint32_t acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = 0u;
        for (int r = 0; r < a.R; r++) {
            const uint2 *el = elist[wave];
            const int n4 = Q >> 2;
            uint2 e0 = el[0], e1 = el[min(1, n4 - 1)];
            int c = 0;
            for (; c + 1 < n4; c += 2) {
                rg_gather4_perm(acc, lane_off, e0);
                e0 = el[min(c + 2, n4 - 1)];
                rg_gather4_perm(acc, lane_off, e1);
                e1 = el[min(c + 3, n4 - 1)];
            }
            if (c < n4) rg_gather4_perm(acc, lane_off, e0);
            if (a.barrier) __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < RG_G; j++) a.out[((size_t)gw * 64 + lane) * RG_G + j] = (acc[2 * (j >> 2) + (j & 1)] >> (16 * ((j >> 1) & 1))) & 0xffffu;
    } else if (VAR == 0) {
        const uint32_t lane_off = lds0 + (uint32_t)((job ? row : 0) * RG_PITCH + half * RG_G);
        uint32_t acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = 0u;
        for (int r = 0; r < a.R; r++) {
            const uint2 *el = elist[wave];
            const int n4 = Q >> 2;
            uint2 e0 = el[0], e1 = el[min(1, n4 - 1)];
            int c = 0;
            for (; c + 1 < n4; c += 2) {
                rg_gather4(acc, lane_off, e0);
                e0 = el[min(c + 2, n4 - 1)];
                rg_gather4(acc, lane_off, e1);
                e1 = el[min(c + 3, n4 - 1)];
            }
            if (c < n4) rg_gather4(acc, lane_off, e0);
            if (a.barrier) __syncthreads();
        }
        rg_odd(acc);
#pragma unroll
        for (int j = 0; j < RG_G; j++) a.out[((size_t)gw * 64 + lane) * RG_G + j] = (acc[2 * (j >> 2) + (j & 1)] >> (16 * ((j >> 1) & 1))) & 0xffffu;
    } else {
        // Misalignment-indexed sets: set s holds the patches whose origin has (offset & 3) == s, accumulated at their ALIGNED
        // byte positions p = 0..15 (E[j]: positions 4j | 4j + 2 as 16-bit lanes, O[j]: running sum of dword >> 8).  No funnel
        // in the loop; the positions are shifted into hypotheses when the sums leave the registers.  The second half of a
        // lattice row starts 13 bytes on: its lanes' misalignment is (s + 1) & 3, and when s == 3 their aligned address is
        // one dword further -- the lane base of the s == 3 run differs for them.
        const uint32_t laneA = (uint32_t)((job ? row : 0) * RG_PITCH + half * 12), laneB = laneA + (uint32_t)(half * 4); // (offsets into the image)
        uint32_t E[4][4], O[4][4];
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int j = 0; j < 4; j++) E[s][j] = O[s][j] = 0u;
        const uint16_t *rn = a.runs + (size_t)gw * 4;
        const int b1 = __builtin_amdgcn_readfirstlane((int)rn[0]), b2 = __builtin_amdgcn_readfirstlane((int)rn[1]);
        const int b3 = __builtin_amdgcn_readfirstlane((int)rn[2]), np = __builtin_amdgcn_readfirstlane((int)rn[3]);
        const uint32_t *el32 = reinterpret_cast<const uint32_t *>(elist[wave]); // one pair per dword
        for (int r = 0; r < a.R; r++) {
            auto load_pair = [&](int p, uint32_t (&d)[8]) __attribute__((always_inline)) {
                const uint32_t e = el32[p];
                const uint32_t base = p >= b3 ? laneB : laneA;
                const uint32_t a0 = base + (e & 0xfffcu), a1 = base + ((e >> 16) & 0xfffcu);
                const uint32_t *p0 = reinterpret_cast<const uint32_t *>(region + a0);
                const uint32_t *p1 = reinterpret_cast<const uint32_t *>(region + a1);
#pragma unroll
                for (int j = 0; j < 4; j++) { d[j] = p0[j]; d[4 + j] = p1[j]; }
            };
            auto accum = [&](auto sc, const uint32_t (&d)[8]) __attribute__((always_inline)) {
                constexpr int s = decltype(sc)::value;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t x = d[j] + d[4 + j];
                    E[s][j] += x & 0x00FF00FFu;
                    O[s][j] += x >> 8;
                }
            };
            auto run = [&](auto s, int lo, int hi) __attribute__((always_inline)) { // pairs [lo, hi) into set s, the next pair's reads in flight
                if (lo >= hi) return;
                uint32_t d0[8], d1[8];
                load_pair(lo, d0);
                int p = lo;
                for (; p + 2 <= hi - 1; p += 2) {
                    load_pair(p + 1, d1);
                    accum(s, d0);
                    load_pair(p + 2, d0);
                    accum(s, d1);
                }
                if (p + 1 < hi) { load_pair(p + 1, d1); accum(s, d0); accum(s, d1); }
                else accum(s, d0);
            };
            run(std::integral_constant<int, 0>(), 0, b1); run(std::integral_constant<int, 1>(), b1, b2);
            run(std::integral_constant<int, 2>(), b2, b3); run(std::integral_constant<int, 3>(), b3, np);
            if (a.barrier) __syncthreads();
        }
        // positions -> hypotheses: lane misalignment m = (s + half) & 3; hypothesis h of the lane is position h + m of set s
        uint32_t tot[RG_G];
#pragma unroll
        for (int j = 0; j < RG_G; j++) tot[j] = 0u;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            uint32_t pos[16];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t e = E[s][j], o = O[s][j] - ((e >> 16) << 8); // o = pos(4j + 1) | pos(4j + 3) << 16
                pos[4 * j] = e & 0xffffu; pos[4 * j + 2] = e >> 16; pos[4 * j + 1] = o & 0xffffu; pos[4 * j + 3] = o >> 16;
            }
            const int m = (s + half) & 3;
#pragma unroll
            for (int j = 0; j < RG_G; j++) {
                uint32_t v = pos[j];
                if (m == 1) v = pos[j + 1];
                if (m == 2) v = pos[j + 2];
                if (m == 3) v = pos[j + 3];
                tot[j] += v;
            }
        }
#pragma unroll
        for (int j = 0; j < RG_G; j++) a.out[((size_t)gw * 64 + lane) * RG_G + j] = tot[j] & 0xffffu;
    }
}

int main(int argc, char **argv) {
    const int Q = argc > 1 ? atoi(argv[1]) : 128, R = argc > 2 ? atoi(argv[2]) : 200;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, NW = 8;
    const int blocks = cus * 12;
    const int nwaves = blocks * NW;
    // entries: random patch origins inside the image, sorted by misalignment, every run of even length, Q a multiple of 4
    std::vector<uint16_t> ent((size_t)nwaves * Q), runs((size_t)nwaves * 4);
    srand(11);
    for (int w = 0; w < nwaves; w++) {
        std::vector<uint16_t> by[4];
        int left = Q;
        for (int s = 0; s < 4; s++) {
            int n = s == 3 ? left : ((rand() % (Q / 2 + 1)) & ~1);
            if (n > left) n = left;
            left -= n;
            for (int i = 0; i < n; i++) {
                const int cls = rand() & 3, er = rand() % RG_H, ex = ((rand() % (RG_W - 4)) & ~3) + s;
                by[s].push_back((uint16_t)(cls * RG_CLS + er * RG_PITCH + ex));
            }
        }
        int at = 0, pairs = 0;
        for (int s = 0; s < 4; s++) {
            if (s > 0) runs[(size_t)w * 4 + s - 1] = (uint16_t)pairs;
            for (uint16_t e : by[s]) ent[(size_t)w * Q + at++] = e;
            pairs += (int)by[s].size() / 2;
        }
        runs[(size_t)w * 4 + 3] = (uint16_t)pairs;
    }
    uint16_t *d_ent, *d_runs; uint32_t *d_out[3];
    hipMalloc(&d_ent, ent.size() * 2); hipMalloc(&d_runs, runs.size() * 2);
    hipMemcpy(d_ent, ent.data(), ent.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(d_runs, runs.data(), runs.size() * 2, hipMemcpyHostToDevice);
    const size_t out_n = (size_t)nwaves * 64 * RG_G;
    for (int v = 0; v < 3; v++) { hipMalloc(&d_out[v], out_n * 4); hipMemset(d_out[v], 0, out_n * 4); }
    std::vector<uint32_t> h[3];
    printf("# %s, %d CUs, %d blocks of %d waves, Q = %d patches per wave and round, R = %d rounds\n", prop.gcnArchName, cus, blocks, NW, Q, R);
    printf("%-48s %10s %12s %14s\n", "variant", "blocks/CU", "us", "CU clk/patch");
    const size_t base_lds = ((RG_LDS_BYTES + 15) & ~15) + NW * 64 * 8;
    for (int var = 0; var < 3; var++)
        for (int per_cu : {3, 2})
            for (int barrier : {0, 1}) {
                Args a{d_ent, d_runs, d_out[var], Q, R, barrier, 0};
                const size_t lds = std::max(base_lds, (size_t)(160 * 1024 / per_cu - 1024)) ;
                auto k = var == 0 ? proto<8, 0> : var == 1 ? proto<8, 1> : proto<8, 2>;
                hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * NW), lds, 0, a);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * NW), lds, 0, a);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                const double patches = (double)nwaves * Q * R;
                printf("%-48s %10d %12.1f %14.2f\n", var == 0 ? (barrier ? "kernel's loop, barrier per round" : "kernel's loop") :
                       var == 2 ? (barrier ? "perm funnel + widen, barrier per round" : "perm funnel + widen") :
                       (barrier ? "misalignment sets, barrier per round" : "misalignment sets"), per_cu, ms * 1e3, ms * 1e-3 * 2.4e9 * cus / patches);
                hipEventDestroy(e0); hipEventDestroy(e1);
            }
    for (int var = 0; var < 3; var++) { // one round: the 16-bit sums hold Q patches
        Args a{d_ent, d_runs, d_out[var], Q, 1, 0, 0};
        auto k = var == 0 ? proto<8, 0> : var == 1 ? proto<8, 1> : proto<8, 2>;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * NW), base_lds, 0, a);
    }
    hipDeviceSynchronize();
    for (int v = 0; v < 3; v++) { h[v].resize(out_n); hipMemcpy(h[v].data(), d_out[v], out_n * 4, hipMemcpyDeviceToHost); }
    size_t bad = 0;
    for (size_t i = 0; i < out_n; i++) {
        const int lane = (int)((i / RG_G) % 64);
        if ((lane & 31) >= 26) continue;
        // (R rounds of the same patches: the 16-bit sums wrap the same way in both variants only while they do not overflow --
        //  compare modulo 2^16)
        if ((h[0][i] & 0xffffu) != (h[1][i] & 0xffffu) || (h[0][i] & 0xffffu) != (h[2][i] & 0xffffu)) { if (bad < 5) printf("  differ at %zu: %u vs %u vs %u\n", i, h[0][i], h[1][i], h[2][i]); bad++; }
    }
    printf("# sums of the three variants differ in %zu of %zu places\n", bad, out_n);
    return 0;
}
