// build: hipcc -O2 --offload-arch=gfx950 -o scripts/exp/valu_issue scripts/exp/valu_issue.hip
// experiment (round 3): how many wave64 VALU instructions can one gfx950 CU issue per clock, by opcode and by waves per
// SIMD?  (DESIGN.md priced the region correlate against 1.0 per CU and clock; MI355X_MICROARCH.md says four SIMD-32,
// i.e. 2.0 for v_fma_f32.)  Every kernel is a loop of 64 independent instructions of ONE opcode over eight destination
// registers; all 256 CUs run it, with 1, 2, 4 or 8 waves per SIMD.  Reported: wave-instructions per CU and shader clock
// (wall time of the launch by HIP events x the clock the chip held, s_memtime / s_memrealtime inside the kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY64(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)

#define DEFINE_KERNEL(NAME, INSTR)                                                                                         \
    __global__ __launch_bounds__(1024) void k_##NAME(uint32_t *out, int iters, unsigned long long *ticks) {              \
        uint32_t r0 = threadIdx.x, r1 = r0 * 3u, r2 = r0 * 5u, r3 = r0 * 7u, r4 = r0 * 11u, r5 = r0 * 13u, r6 = r0 * 17u, \
                 r7 = r0 * 19u;                                                                                            \
        uint32_t a = 0x00ff00ffu + blockIdx.x, b = 0x0c030c01u, c = (threadIdx.x & 3u);                                   \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                        \
        const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();                                                    \
        for (int i = 0; i < iters; i++) {                                                                                  \
            asm volatile(BODY64(INSTR)                                                                                     \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)                 \
                         : "v"(a), "v"(b), "v"(c));                                                                        \
        }                                                                                                                  \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                        \
        const unsigned long long w1 = __builtin_amdgcn_s_memrealtime();                                                    \
        if ((threadIdx.x & 63) == 0) {                                                                                     \
            const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;                                         \
            ticks[2 * w] = t1 - t0;                                                                                        \
            ticks[2 * w + 1] = w1 - w0;                                                                                    \
        }                                                                                                                  \
        out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                        \
    }

// operand 8 = a, 9 = b, 10 = c
#define I_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n\t"
#define I_ALIGNBYTE(i) "v_alignbyte_b32 %" #i ", %" #i ", %8, %10\n\t"
#define I_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n\t"
#define I_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n\t"
#define I_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n\t"
#define I_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
#define I_MOV(i) "v_mov_b32 %" #i ", %8\n\t"
#define I_PKADDU16(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n\t"
#define I_PKMADU16(i) "v_pk_mad_u16 %" #i ", %8, %9, %" #i "\n\t"
#define I_MADU24(i) "v_mad_u32_u24 %" #i ", %8, %9, %" #i "\n\t"
#define I_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %8\n\t"
#define I_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n\t"
#define I_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 8, 8\n\t"
#define I_ADDSDWA(i) "v_add_u32_sdwa %" #i ", %8, %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
#define I_ADDU16SDWA(i) "v_add_u16_sdwa %" #i ", %8, %" #i " dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:WORD_1\n\t"
#define I_DOT4(i) "v_dot4_u32_u8 %" #i ", %8, %9, %" #i "\n\t"
#define I_SAD(i) "v_sad_u8 %" #i ", %8, %9, %" #i "\n\t"
#define I_PKFMAF32(i) "v_fma_f32 %" #i ", %8, %9, %" #i "\n\t"
#define I_MOVDPP(i) "v_mov_b32_dpp %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_LSHRREV(i) "v_lshrrev_b32 %" #i ", 8, %" #i "\n\t"
#define I_ADDF64(i) "v_add_u32 %" #i ", %" #i ", %9\n\t"

DEFINE_KERNEL(perm, I_PERM)
DEFINE_KERNEL(alignbyte, I_ALIGNBYTE)
DEFINE_KERNEL(add3, I_ADD3)
DEFINE_KERNEL(and, I_AND)
DEFINE_KERNEL(add, I_ADD)
DEFINE_KERNEL(fma_f32, I_FMA)
DEFINE_KERNEL(mov, I_MOV)
DEFINE_KERNEL(pk_add_u16, I_PKADDU16)
DEFINE_KERNEL(pk_mad_u16, I_PKMADU16)
DEFINE_KERNEL(mad_u32_u24, I_MADU24)
DEFINE_KERNEL(lshl_add, I_LSHLADD)
DEFINE_KERNEL(and_or, I_ANDOR)
DEFINE_KERNEL(bfe, I_BFE)
DEFINE_KERNEL(add_u32_sdwa, I_ADDSDWA)
DEFINE_KERNEL(add_u16_sdwa, I_ADDU16SDWA)
DEFINE_KERNEL(dot4_u32_u8, I_DOT4)
DEFINE_KERNEL(sad_u8, I_SAD)
DEFINE_KERNEL(mov_dpp, I_MOVDPP)
DEFINE_KERNEL(lshrrev, I_LSHRREV)

// the region correlate's inner mix per four patches, as one stream: 8 raw adds, 8 alignbyte, 8 and,
// 6 perm, 7 add3 and 11 address / bookkeeping adds = 48 instructions
#define I_MIX(i) "v_add_u32 %" #i ", %" #i ", %8\n\tv_alignbyte_b32 %" #i ", %" #i ", %8, %10\n\tv_and_b32 %" #i ", %" #i ", %8\n\t" \
                 "v_perm_b32 %" #i ", %" #i ", %8, %9\n\tv_add3_u32 %" #i ", %" #i ", %8, %9\n\tv_add_u32 %" #i ", %" #i ", %9\n\t"
#define BODY48(X) R8(X)
__global__ __launch_bounds__(1024) void k_mix(uint32_t *out, int iters, unsigned long long *ticks) {
    uint32_t r0 = threadIdx.x, r1 = r0 * 3u, r2 = r0 * 5u, r3 = r0 * 7u, r4 = r0 * 11u, r5 = r0 * 13u, r6 = r0 * 17u, r7 = r0 * 19u;
    uint32_t a = 0x00ff00ffu + blockIdx.x, b = 0x0c030c01u, c = (threadIdx.x & 3u);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        asm volatile(BODY48(I_MIX) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b), "v"(c));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long w1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        ticks[2 * w] = t1 - t0;
        ticks[2 * w + 1] = w1 - w0;
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
}

// scalar-ALU streams: 64 independent s_add_u32 / s_and_b32 / s_bfe_u32 on eight SGPRs; and a stream that alternates one
// scalar and one vector instruction (do they issue side by side?)
#define DEFINE_SKERNEL(NAME, INSTR, NV)                                                                                    \
    __global__ __launch_bounds__(1024) void k_##NAME(uint32_t *out, int iters, unsigned long long *ticks) {              \
        uint32_t r0 = blockIdx.x, r1 = r0 * 3u, r2 = r0 * 5u, r3 = r0 * 7u, r4 = r0 * 11u, r5 = r0 * 13u, r6 = r0 * 17u,   \
                 r7 = r0 * 19u;                                                                                            \
        uint32_t v0 = threadIdx.x, v1 = v0 * 3u, v2 = v0 * 5u, v3 = v0 * 7u, v4 = v0 * 11u, v5 = v0 * 13u, v6 = v0 * 17u, v7 = v0 * 19u; \
        uint32_t a = 0x00ff00ffu + blockIdx.x;                                                                             \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                        \
        const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();                                                    \
        for (int i = 0; i < iters; i++) {                                                                                  \
            asm volatile(BODY64(INSTR)                                                                                     \
                         : "+s"(r0), "+s"(r1), "+s"(r2), "+s"(r3), "+s"(r4), "+s"(r5), "+s"(r6), "+s"(r7), "+v"(v0), "+v"(v1), \
                           "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)                                      \
                         : "s"(a) : "scc");                                                                                \
        }                                                                                                                  \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                        \
        const unsigned long long w1 = __builtin_amdgcn_s_memrealtime();                                                    \
        if ((threadIdx.x & 63) == 0) {                                                                                     \
            const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;                                         \
            ticks[2 * w] = t1 - t0;                                                                                        \
            ticks[2 * w + 1] = w1 - w0;                                                                                    \
        }                                                                                                                  \
        out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7; \
    }
#define S_ADD(i) "s_add_u32 %" #i ", %" #i ", %16\n\t"
#define S_AND(i) "s_and_b32 %" #i ", %" #i ", %16\n\t"
#define S_BFE(i) "s_bfe_u32 %" #i ", %" #i ", 0x80008\n\t"
#define S_V_PAIR(i) "s_add_u32 %" #i ", %" #i ", %16\n\tv_add_u32 %1" #i ", %1" #i ", %16\n\t"
DEFINE_SKERNEL(s_add, S_ADD, 0)
DEFINE_SKERNEL(s_and, S_AND, 0)
DEFINE_SKERNEL(s_bfe, S_BFE, 0)

typedef void (*kern_t)(uint32_t *, int, unsigned long long *);
struct Entry { const char *name; kern_t k; int per_iter; };

int main(int argc, char **argv) {
    const Entry table[] = {
        {"v_perm_b32", k_perm, 64}, {"v_alignbyte_b32", k_alignbyte, 64}, {"v_add3_u32", k_add3, 64}, {"v_and_b32", k_and, 64},
        {"v_add_u32", k_add, 64}, {"v_fma_f32", k_fma_f32, 64}, {"v_mov_b32", k_mov, 64}, {"v_pk_add_u16", k_pk_add_u16, 64},
        {"v_pk_mad_u16", k_pk_mad_u16, 64}, {"v_mad_u32_u24", k_mad_u32_u24, 64}, {"v_lshl_add_u32", k_lshl_add, 64},
        {"v_and_or_b32", k_and_or, 64}, {"v_bfe_u32", k_bfe, 64}, {"v_add_u32_sdwa(byte)", k_add_u32_sdwa, 64},
        {"v_add_u16_sdwa(byte->word1,preserve)", k_add_u16_sdwa, 64}, {"v_dot4_u32_u8", k_dot4_u32_u8, 64}, {"v_sad_u8", k_sad_u8, 64},
        {"v_mov_b32_dpp", k_mov_dpp, 64}, {"v_lshrrev_b32", k_lshrrev, 64}, {"region-correlate mix (6 ops)", k_mix, 48},
        {"s_add_u32 (scalar)", k_s_add, 64}, {"s_and_b32 (scalar)", k_s_and, 64}, {"s_bfe_u32 (scalar)", k_s_bfe, 64},
    };
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint32_t *out; unsigned long long *ticks;
    const size_t max_threads = (size_t)cus * 2048;
    hipMalloc(&out, max_threads * 4); hipMalloc(&ticks, max_threads / 64 * 16);
    std::vector<unsigned long long> h(max_threads / 64 * 2);
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    printf("# %s, %d CUs; loop of %d x (instructions per iteration); wave-instructions per CU and shader clock\n", prop.gcnArchName, cus, iters);
    printf("%-40s %10s %10s %10s %10s   clock held (GHz, 4 waves/SIMD)\n", "opcode", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD");
    for (const Entry &e : table) {
        printf("%-40s", e.name);
        double ghz = 0;
        for (int wps : {1, 2, 4, 8}) {
            const int threads = wps >= 4 ? 1024 : 256 * wps, blocks = cus * (wps == 8 ? 2 : 1);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, out, iters, ticks);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, out, iters, ticks);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            hipEventDestroy(e0); hipEventDestroy(e1);
            const size_t nw = (size_t)blocks * threads / 64;
            hipMemcpy(h.data(), ticks, nw * 16, hipMemcpyDeviceToHost);
            double sum_t = 0, sum_w = 0;
            for (size_t i = 0; i < nw; i++) { sum_t += (double)h[2 * i]; sum_w += (double)h[2 * i + 1]; }
            const double clock_hz = sum_t / sum_w * 1e8; // shader clocks per 100 MHz tick, averaged over the waves
            // all instructions of the launch / (wall time of the launch in shader clocks x CUs)
            printf(" %10.3f", (double)nw * e.per_iter * iters / (ms * 1e-3 * clock_hz * cus));
            if (wps == 4) ghz = clock_hz * 1e-9;
        }
        printf("   %.2f\n", ghz);
    }
    return 0;
}
