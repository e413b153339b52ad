// build: hipcc -O2 --offload-arch=gfx950 -o scripts/exp/qsad_probe scripts/exp/qsad_probe.hip
// experiment (round 4): can the quad-SAD instructions widen-and-accumulate four grid bytes per lane in ONE instruction?
//   v_mqsad_pk_u16_u8 D(4 x u16), S0(64 bit), S1(32 bit), S2(4 x u16):
//       D[i] = S2[i] + sum over the bytes j of S1 that are NOT zero of | S0.byte[i + j] - S1.byte[j] |      (i = 0..3)
//   With S1 = 0xff << 8 r only byte r of the mask counts and |x - 255| = 255 - x for every byte x, so
//       D[i] = S2[i] + 255 - S0.byte[i + r]
//   = four CONSECUTIVE bytes at ANY byte offset r of a register pair, widened to 16 bits and accumulated: the byte funnel,
//   the even / odd split and the add of the region correlate's inner loop (9 instructions per patch) in one.
// Part 1 checks those semantics on the hardware (wrap-around of the 16-bit sums, destination = accumulator, mask from an
// SGPR); part 2 measures the issue rate per CU and clock like valu_issue.hip; part 3 the region correlate's real
// instruction mix with independent registers (the round-3 mix kernel chained every instruction to its predecessor).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;

__global__ void k_semantics(const u64 *s0, const uint32_t *s1, const u64 *s2, u64 *d_mq, u64 *d_q, u64 *d_inplace, uint32_t *d_msad, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 a = s0[i], c = s2[i], r;
    uint32_t m = s1[i];
    asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %3" : "=&v"(r) : "v"(a), "v"(m), "v"(c));
    d_mq[i] = r;
    asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %3" : "=&v"(r) : "v"(a), "v"(m), "v"(c));
    d_q[i] = r;
    u64 acc = c;
    asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(m)); // destination = accumulator
    d_inplace[i] = acc;
    uint32_t x;
    asm volatile("v_msad_u8 %0, %1, %2, %3" : "=v"(x) : "v"((uint32_t)a), "v"(m), "v"((uint32_t)c));
    d_msad[i] = x;
}

static u64 model_mqsad(u64 s0, uint32_t s1, u64 s2, bool masked) {
    u64 out = 0;
    for (int i = 0; i < 4; i++) {
        uint32_t sum = (uint32_t)((s2 >> (16 * i)) & 0xffff);
        for (int j = 0; j < 4; j++) {
            const int mb = (s1 >> (8 * j)) & 0xff, xb = (int)((s0 >> (8 * (i + j))) & 0xff);
            if (masked && mb == 0) continue;
            sum += (uint32_t)abs(xb - mb);
        }
        out |= (u64)(sum & 0xffff) << (16 * i);
    }
    return out;
}

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY64(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)
#define TIMED_HEAD                                                         \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();            \
    const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
#define TIMED_TAIL                                                         \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();            \
    const unsigned long long w1 = __builtin_amdgcn_s_memrealtime();        \
    if ((threadIdx.x & 63) == 0) {                                         \
        const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; \
        ticks[2 * w] = t1 - t0;                                            \
        ticks[2 * w + 1] = w1 - w0;                                        \
    }

// 64-bit destination streams: eight independent accumulator pairs
#define DEFINE_K64(NAME, INSTR)                                                                                            \
    __global__ __launch_bounds__(1024) void k_##NAME(uint32_t *out, int iters, unsigned long long *ticks) {              \
        u64 r0 = threadIdx.x, r1 = r0 * 3u, r2 = r0 * 5u, r3 = r0 * 7u, r4 = r0 * 11u, r5 = r0 * 13u, r6 = r0 * 17u, r7 = r0 * 19u; \
        u64 a = 0x0102030405060708ull + blockIdx.x;                                                                       \
        uint32_t b = 0x0000ff00u;                                                                                          \
        uint32_t sb = __builtin_amdgcn_readfirstlane(b);                                                                   \
        TIMED_HEAD                                                                                                         \
        for (int i = 0; i < iters; i++) {                                                                                  \
            asm volatile(BODY64(INSTR)                                                                                     \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)                 \
                         : "v"(a), "v"(b), "s"(sb));                                                                       \
        }                                                                                                                  \
        TIMED_TAIL                                                                                                         \
        out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7);          \
    }
#define I_MQSAD(i) "v_mqsad_pk_u16_u8 %" #i ", %8, %9, %" #i "\n\t"
#define I_MQSAD_S(i) "v_mqsad_pk_u16_u8 %" #i ", %8, %10, %" #i "\n\t"
#define I_QSAD(i) "v_qsad_pk_u16_u8 %" #i ", %8, %9, %" #i "\n\t"
#define I_LSHL64(i) "v_lshlrev_b64 %" #i ", 1, %" #i "\n\t"
DEFINE_K64(mqsad, I_MQSAD)
DEFINE_K64(mqsad_sgpr_mask, I_MQSAD_S)
DEFINE_K64(qsad, I_QSAD)
DEFINE_K64(lshl64, I_LSHL64)

// ping-pong form: destination differs from every source (what the compiler's earlyclobber constraint asks for)
__global__ __launch_bounds__(1024) void k_mqsad_pingpong(uint32_t *out, int iters, unsigned long long *ticks) {
    u64 r0 = threadIdx.x, r1 = r0 * 3u, r2 = r0 * 5u, r3 = r0 * 7u, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    u64 a = 0x0102030405060708ull + blockIdx.x;
    uint32_t b = 0x0000ff00u;
    TIMED_HEAD
    for (int i = 0; i < iters; i++) {
#define PP "v_mqsad_pk_u16_u8 %4, %8, %9, %0\n\tv_mqsad_pk_u16_u8 %5, %8, %9, %1\n\tv_mqsad_pk_u16_u8 %6, %8, %9, %2\n\tv_mqsad_pk_u16_u8 %7, %8, %9, %3\n\t" \
           "v_mqsad_pk_u16_u8 %0, %8, %9, %4\n\tv_mqsad_pk_u16_u8 %1, %8, %9, %5\n\tv_mqsad_pk_u16_u8 %2, %8, %9, %6\n\tv_mqsad_pk_u16_u8 %3, %8, %9, %7\n\t"
        asm volatile(PP PP PP PP PP PP PP PP
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
                     : "v"(a), "v"(b));
    }
    TIMED_TAIL
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(r0 ^ r1 ^ r2 ^ r3 ^ q0 ^ q1 ^ q2 ^ q3);
}

// 32-bit streams
#define DEFINE_K32(NAME, INSTR, PER)                                                                                       \
    __global__ __launch_bounds__(1024) void k_##NAME(uint32_t *out, int iters, unsigned long long *ticks) {              \
        uint32_t r0 = threadIdx.x, r1 = r0 * 3u, r2 = r0 * 5u, r3 = r0 * 7u, r4 = r0 * 11u, r5 = r0 * 13u, r6 = r0 * 17u, r7 = r0 * 19u; \
        uint32_t a = 0x00ff00ffu + blockIdx.x, b = 0x0000ff00u, c = (threadIdx.x & 3u);                                   \
        TIMED_HEAD                                                                                                         \
        for (int i = 0; i < iters; i++) {                                                                                  \
            asm volatile(INSTR                                                                                             \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)                 \
                         : "v"(a), "v"(b), "v"(c));                                                                        \
        }                                                                                                                  \
        TIMED_TAIL                                                                                                         \
        out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                        \
    }
#define I_MSAD(i) "v_msad_u8 %" #i ", %8, %9, %" #i "\n\t"
DEFINE_K32(msad, BODY64(I_MSAD), 64)
// The region correlate's six opcodes with NO instruction reading its predecessor's result: each opcode writes its own
// register and reads the shared inputs (the round-3 mix chained them through one register).
#define MIX6 "v_add_u32 %0, %0, %8\n\tv_alignbyte_b32 %1, %1, %8, %10\n\tv_and_b32 %2, %2, %8\n\t"  \
             "v_lshrrev_b32 %3, 8, %3\n\tv_add3_u32 %4, %4, %8, %9\n\tv_add_u32 %5, %5, %9\n\t"
DEFINE_K32(mix_indep, MIX6 MIX6 MIX6 MIX6 MIX6 MIX6 MIX6 MIX6, 48)
// the inner loop of rg_gather4 as the compiler emits it, loads removed (registers rotate so nothing depends on the
// instruction before it): per four patches 8 add (raw pair sums), 8 alignbyte, 8 and, 7 lshrrev, 8 add3 (even) -- 7 (odd),
// + 12 address / unpack instructions (and, add, lshr)
#define BODY_RG                                                                                                            \
    "v_and_b32 %0, %8, %0\n\tv_lshrrev_b32 %1, 16, %8\n\tv_and_b32 %2, %9, %2\n\tv_lshrrev_b32 %3, 16, %9\n\t"             \
    "v_add_u32 %0, %0, %10\n\tv_add_u32 %1, %1, %10\n\tv_add_u32 %2, %2, %10\n\tv_add_u32 %3, %3, %10\n\t"                 \
    "v_and_b32 %4, -4, %0\n\tv_and_b32 %5, -4, %1\n\tv_and_b32 %6, -4, %2\n\tv_and_b32 %7, -4, %3\n\t"                     \
    "v_add_u32 %0, %4, %5\n\tv_add_u32 %1, %5, %6\n\tv_add_u32 %2, %6, %7\n\tv_add_u32 %3, %7, %4\n\t"                     \
    "v_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8\n\t"                     \
    "v_alignbyte_b32 %0, %1, %0, %10\n\tv_alignbyte_b32 %1, %2, %1, %10\n\tv_alignbyte_b32 %2, %3, %2, %10\n\tv_alignbyte_b32 %3, %8, %3, %10\n\t" \
    "v_alignbyte_b32 %4, %5, %4, %10\n\tv_alignbyte_b32 %5, %6, %5, %10\n\tv_alignbyte_b32 %6, %7, %6, %10\n\tv_alignbyte_b32 %7, %8, %7, %10\n\t" \
    "v_and_b32 %0, %8, %0\n\tv_and_b32 %1, %8, %1\n\tv_and_b32 %2, %8, %2\n\tv_and_b32 %3, %8, %3\n\t"                     \
    "v_and_b32 %4, %8, %4\n\tv_and_b32 %5, %8, %5\n\tv_and_b32 %6, %8, %6\n\tv_and_b32 %7, %8, %7\n\t"                     \
    "v_lshrrev_b32 %0, 8, %0\n\tv_lshrrev_b32 %1, 8, %1\n\tv_lshrrev_b32 %2, 8, %2\n\tv_lshrrev_b32 %3, 8, %3\n\t"         \
    "v_lshrrev_b32 %4, 8, %4\n\tv_lshrrev_b32 %5, 8, %5\n\tv_lshrrev_b32 %6, 8, %6\n\t"                                    \
    "v_add3_u32 %0, %0, %4, %8\n\tv_add3_u32 %1, %1, %5, %8\n\tv_add3_u32 %2, %2, %6, %8\n\tv_add3_u32 %3, %3, %7, %8\n\t" \
    "v_add3_u32 %4, %0, %4, %9\n\tv_add3_u32 %5, %1, %5, %9\n\tv_add3_u32 %6, %2, %6, %9\n\t"
DEFINE_K32(rg_body, BODY_RG BODY_RG, 2 * 50)
// the same loop after the funnel and the widening became one v_perm_b32 per pair of hypotheses (ym_k_region.hpp rg_perm_pair), as
// hipcc emits it (llvm-objdump of libyagmatch.so), loads removed: per four patches 4 add (addresses: entry in an SGPR + lane
// offset), 4 and (-4), 2 and (3), 8 add (raw pair sums), 4 mad_u32_u24 (the two selectors of each pair), 14 perm, 7 add3,
// 2 readfirstlane (the next quad's entries) + 1 mov = 46
#define BODY_RGP                                                                                                           \
    "v_add_u32 %0, %8, %0\n\tv_add_u32 %1, %8, %1\n\tv_add_u32 %2, %9, %2\n\tv_add_u32 %3, %9, %3\n\t"                 \
    "v_and_b32 %4, -4, %0\n\tv_and_b32 %5, -4, %1\n\tv_and_b32 %6, -4, %2\n\tv_and_b32 %7, -4, %3\n\t"                 \
    "v_and_b32 %0, 3, %0\n\tv_and_b32 %2, 3, %2\n\t"                                                                    \
    "v_add_u32 %1, %4, %5\n\tv_add_u32 %3, %5, %6\n\tv_add_u32 %4, %6, %7\n\tv_add_u32 %5, %7, %8\n\t"                 \
    "v_add_u32 %6, %8, %9\n\tv_add_u32 %7, %9, %10\n\tv_add_u32 %1, %1, %10\n\tv_add_u32 %3, %3, %10\n\t"              \
    "v_mad_u32_u24 %0, %0, %8, %9\n\tv_mad_u32_u24 %2, %2, %8, %9\n\tv_mad_u32_u24 %4, %4, %8, %10\n\tv_mad_u32_u24 %6, %6, %8, %10\n\t" \
    "v_perm_b32 %1, %3, %1, %0\n\tv_perm_b32 %3, %5, %3, %0\n\tv_perm_b32 %5, %7, %5, %0\n\tv_perm_b32 %7, %8, %7, %0\n\t" \
    "v_perm_b32 %1, %3, %1, %2\n\tv_perm_b32 %3, %5, %3, %2\n\tv_perm_b32 %5, %7, %5, %2\n\t"                           \
    "v_perm_b32 %1, %3, %1, %4\n\tv_perm_b32 %3, %5, %3, %4\n\tv_perm_b32 %5, %7, %5, %4\n\tv_perm_b32 %7, %9, %7, %4\n\t" \
    "v_perm_b32 %1, %3, %1, %6\n\tv_perm_b32 %3, %5, %3, %6\n\tv_perm_b32 %5, %7, %5, %6\n\t"                           \
    "v_add3_u32 %0, %0, %1, %8\n\tv_add3_u32 %2, %2, %3, %8\n\tv_add3_u32 %4, %4, %5, %8\n\tv_add3_u32 %6, %6, %7, %8\n\t" \
    "v_add3_u32 %1, %1, %0, %9\n\tv_add3_u32 %3, %3, %2, %9\n\tv_add3_u32 %5, %5, %4, %9\n\t"                           \
    "v_readfirstlane_b32 s20, %7\n\tv_readfirstlane_b32 s21, %6\n\tv_mov_b32 %7, s20\n\t"
#define DEFINE_K32_SGPR(NAME, BODY, N)                                                                                      \
    __global__ __launch_bounds__(1024) void k_##NAME(uint32_t *out, int iters, unsigned long long *ticks) {                \
        uint32_t r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
        const uint32_t a = blockIdx.x + 3, b = 0x00ff00ffu, c = threadIdx.x & 3;                                           \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();                 \
        for (int i = 0; i < iters; i++)                                                                                    \
            asm volatile(BODY : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)             \
                         : "v"(a), "v"(b), "v"(c) : "s20", "s21");                                                         \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                                \
        if ((threadIdx.x & 63) == 0) {                                                                                     \
            const size_t w = (size_t)(blockIdx.x * blockDim.x + threadIdx.x) / 64;                                         \
            ticks[2 * w] = t1 - t0; ticks[2 * w + 1] = w1 - w0;                                                            \
        }                                                                                                                  \
    }
DEFINE_K32_SGPR(rgp_body, BODY_RGP BODY_RGP, 2 * 46)

typedef void (*kern_t)(uint32_t *, int, unsigned long long *);
struct Entry { const char *name; kern_t k; int per_iter; };

int main(int argc, char **argv) {
    // ---- part 1: semantics
    {
        const int n = 1 << 16;
        std::vector<u64> s0(n), s2(n), mq(n), q(n), ip(n);
        std::vector<uint32_t> s1(n), ms(n);
        srand(7);
        auto r64 = []() { u64 v = 0; for (int i = 0; i < 8; i++) v = v << 8 | (u64)(rand() & 0xff); return v; };
        for (int i = 0; i < n; i++) {
            s0[i] = r64(); s2[i] = r64();
            const int kind = i & 7;
            s1[i] = kind < 4 ? 0xffu << (8 * kind) : kind == 4 ? 0u : kind == 5 ? 0x00ff00ffu : (uint32_t)r64();
            if (i % 11 == 0) s2[i] = 0xfff0fff0fff0fff0ull; // the 16-bit sums wrap?
        }
        u64 *d0, *d2, *dmq, *dq, *dip; uint32_t *d1, *dms;
        hipMalloc(&d0, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&dmq, n * 8); hipMalloc(&dq, n * 8); hipMalloc(&dip, n * 8);
        hipMalloc(&d1, n * 4); hipMalloc(&dms, n * 4);
        hipMemcpy(d0, s0.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(d2, s2.data(), n * 8, hipMemcpyHostToDevice);
        hipMemcpy(d1, s1.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_semantics, dim3(n / 256), dim3(256), 0, 0, d0, d1, d2, dmq, dq, dip, dms, n);
        hipMemcpy(mq.data(), dmq, n * 8, hipMemcpyDeviceToHost); hipMemcpy(q.data(), dq, n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(ip.data(), dip, n * 8, hipMemcpyDeviceToHost); hipMemcpy(ms.data(), dms, n * 4, hipMemcpyDeviceToHost);
        int bad_mq = 0, bad_q = 0, bad_ip = 0, shown = 0;
        for (int i = 0; i < n; i++) {
            const u64 e_mq = model_mqsad(s0[i], s1[i], s2[i], true), e_q = model_mqsad(s0[i], s1[i], s2[i], false);
            if (mq[i] != e_mq) { bad_mq++; if (shown++ < 6) printf("  mqsad s0 %016llx s1 %08x s2 %016llx -> %016llx, model %016llx\n", s0[i], s1[i], s2[i], mq[i], e_mq); }
            if (q[i] != e_q) { bad_q++; if (shown++ < 12) printf("  qsad  s0 %016llx s1 %08x s2 %016llx -> %016llx, model %016llx\n", s0[i], s1[i], s2[i], q[i], e_q); }
            if (ip[i] != mq[i]) bad_ip++;
        }
        printf("# semantics over %d random inputs: v_mqsad_pk_u16_u8 differs from the wrap-around model in %d, v_qsad_pk_u16_u8 in %d;"
               " destination = accumulator differs from the separate destination in %d\n", n, bad_mq, bad_q, bad_ip);
        // the use: mask 0xff << 8r  ->  D[i] = S2[i] + 255 - byte[i + r]
        int bad_use = 0;
        for (int i = 0; i < n; i++) {
            if ((i & 7) >= 4) continue;
            const int r = i & 7;
            for (int j = 0; j < 4; j++) {
                const uint32_t want = (uint32_t)(((s2[i] >> (16 * j)) & 0xffff) + 255 - ((s0[i] >> (8 * (j + r))) & 0xff)) & 0xffff;
                if (((mq[i] >> (16 * j)) & 0xffff) != want) bad_use++;
            }
        }
        printf("# mask 0xff << 8r: D[i] == S2[i] + 255 - S0.byte[i + r] (mod 2^16) violated %d times\n", bad_use);
    }
    // ---- part 2 / 3: issue rates
    const Entry table[] = {
        {"v_mqsad_pk_u16_u8 (dst = acc)", k_mqsad, 64}, {"v_mqsad_pk_u16_u8 (mask in SGPR)", k_mqsad_sgpr_mask, 64},
        {"v_mqsad_pk_u16_u8 (ping-pong dst)", k_mqsad_pingpong, 64}, {"v_qsad_pk_u16_u8", k_qsad, 64},
        {"v_lshlrev_b64", k_lshl64, 64}, {"v_msad_u8", k_msad, 64},
        {"region mix, independent registers (6 ops)", k_mix_indep, 48},
        {"rg_gather4 body without loads (50 ops)", k_rg_body, 100},
        {"rg_gather4 body, v_perm form (46 ops)", k_rgp_body, 92},
    };
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint32_t *out; unsigned long long *ticks;
    const size_t max_threads = (size_t)cus * 2048;
    hipMalloc(&out, max_threads * 4); hipMalloc(&ticks, max_threads / 64 * 16);
    std::vector<unsigned long long> h(max_threads / 64 * 2);
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    printf("# %s, %d CUs; wave-instructions per CU and shader clock\n", prop.gcnArchName, cus);
    printf("%-44s %10s %10s %10s %10s   clock held (GHz, 4 waves/SIMD)\n", "opcode", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD");
    for (const Entry &e : table) {
        printf("%-44s", e.name);
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
            const double clock_hz = sum_t / sum_w * 1e8;
            printf(" %10.3f", (double)nw * e.per_iter * iters / (ms * 1e-3 * clock_hz * cus));
            if (wps == 4) ghz = clock_hz * 1e-9;
        }
        printf("   %.2f\n", ghz);
    }
    return 0;
}
