// build: hipcc -O2 --offload-arch=gfx950 -o scripts/exp/fetch_calib scripts/exp/fetch_calib.hip
// experiment (round 5): what does rocprofv3's FETCH_SIZE count for the access pattern of correlate_region_kernel's staging loads?
// bench.py's `roofline.traffic` multiplies FETCH_SIZE by two (MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a
// wide coalesced streaming read ... other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern").  Three kernels over a buffer far larger than the 256 MiB Infinity Cache, each touching a KNOWN number of bytes once:
//   k_stream   16 B per lane, fully coalesced (the guide's case): bytes = n
//   k_rows     the staging pattern (WIN form): a block reads a box of 192 contiguous bytes x 212 rows (16 B per lane, 12 lanes
//              per row) out of rows `pitch` bytes apart, boxes side by side: requested bytes = 192 per row, 128-byte lines touched
//              = 2 or 3 per row (the box starts at a multiple of 64 bytes: half of the boxes straddle three lines)
//   k_rows_half the same with every other row only (one row parity per thread group, as the kernel's class pairs are loaded)
// Run each under `rocprofv3 --pmc FETCH_SIZE` (and WRITE_SIZE in its own pass); the program prints the byte counts to compare with.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_stream(const uint4 *src, uint4 *sink, size_t n16) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = src[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc; // (never: keeps the loads)
}

// grid (boxes_x, boxes_y); 512 threads: thread t owns segment t % 12 of rows t / 12, t / 12 + 42, ... (504 of 512 threads copy)
__global__ void k_rows(const uint8_t *src, uint4 *sink, int pitch, int rows, int row_step) {
    const int t = threadIdx.x, seg = t % 12, r0 = t / 12;
    if (r0 >= 42) return;
    const uint8_t *base = src + ((size_t)blockIdx.y * rows * row_step) * pitch + (size_t)blockIdx.x * 192;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int r = r0; r < rows; r += 42) {
        const uint4 v = *reinterpret_cast<const uint4 *>(base + (size_t)r * row_step * pitch + 16 * seg);
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}

int main(int argc, char **argv) {
    const int which = argc > 1 ? atoi(argv[1]) : 0;
    const int pitch = 1152;                       // the bench window's row pitch (17 tiles x 64 + 64)
    const size_t bytes = (size_t)3 << 30;         // 3 GiB: twelve Infinity Caches
    uint8_t *buf; uint4 *sink;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, bytes));
    CHECK(hipDeviceSynchronize());
    const int reps = 3;
    if (which == 0) {
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_stream, dim3(256 * 8), dim3(256), 0, 0, reinterpret_cast<const uint4 *>(buf), sink, bytes / 16);
        CHECK(hipDeviceSynchronize());
        printf("k_stream: %d launches, per launch requested bytes %zu = lines x 128\n", reps, bytes);
    } else {
        const int rows = 212, row_step = which == 2 ? 2 : 1;
        const int boxes_x = pitch / 192;          // 6 boxes side by side (the last 0 bytes of the pitch unused)
        const int boxes_y = (int)(bytes / ((size_t)pitch * rows * row_step));
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_rows, dim3(boxes_x, boxes_y), dim3(512), 0, 0, buf, sink, pitch, rows, row_step);
        CHECK(hipDeviceSynchronize());
        size_t lines = 0; // 128-byte lines a row of a box touches: box bx of row r starts at byte r * pitch + 192 bx (pitch = 9 x 128)
        for (int bx = 0; bx < boxes_x; bx++) { const int lo = 192 * bx, hi = lo + 191; lines += (size_t)(hi / 128 - lo / 128 + 1); }
        const size_t rows_total = (size_t)boxes_y * rows;
        printf("k_rows (row step %d): %d launches, per launch requested bytes %zu, bytes in the 128-byte lines touched per box row %zu (lines are shared by neighbouring boxes: distinct line bytes %zu)\n",
               row_step, reps, rows_total * boxes_x * 192, rows_total * lines * 128, rows_total * (size_t)pitch);
    }
    return 0;
}
