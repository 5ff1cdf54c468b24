// Does the ADDRESS PATTERN of a GEMM epilogue's fp32-row traffic cost HBM bandwidth?  The reduced modes' output projection
// and fc2 read and write the residual stream x [rows][768] fp32 in 128 x 128 (or 128 x 256) tiles -- 16 rows x 64 B per
// wave instruction, 512 B (1 KB) per row and tile, rows 3 KB apart -- and reach ~4 TB/s where the LayerNorm kernel (whole
// rows per wave) reaches 6.5.  This probe copies x -> y (read + write, 620 MB) with nothing else going on, in four patterns:
//   rows      one wave per row, 3 KB contiguous
//   tile128   the 128 x 128 tile walk of gemm_p3_kernel<4,128,...,OUT_F32> (NATURAL column order), XCD-aware tile order off
//   tile256   128 x 256 tiles
//   planes    x kept as [768/32][rows][32] fp32 (128-byte rows per plane): the same 128 x 128 tile is 4 planes x 16 KB contiguous
// Diagnostic only.  hipcc -O3 --offload-arch=gfx950 tools/hbm_pattern_probe.hip -o tools/hbm_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int N = 768;

__global__ __launch_bounds__(256) void copy_rows(const float *x, float *y, int M)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const f32x4 *s = reinterpret_cast<const f32x4 *>(x + (size_t)row * N);
        f32x4 *d = reinterpret_cast<f32x4 *>(y + (size_t)row * N);
        f32x4 v[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) v[i] = s[lane + 64 * i];
#pragma unroll
        for (int i = 0; i < 3; ++i) d[lane + 64 * i] = v[i];
    }
}

// BN columns per tile, 128 rows, 4 waves of 32 rows; all loads first (as the residual-in-accumulators prologue), then all stores
template <int BN, bool PLANES>
__global__ __launch_bounds__(256) void copy_tiles(const float *x, float *y, int M)
{
    constexpr int JT = BN / 16, NT = N / BN;
    const int tile = blockIdx.x, m0 = (tile / NT) * 128, n0 = (tile % NT) * BN;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, q = lane >> 4;
    f32x4 v[2][JT];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = min(m0 + 32 * wave + 16 * i + l15, M - 1);
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            const int col = n0 + 16 * j + 4 * q;
            const size_t off = PLANES ? ((size_t)(col >> 5) * M + row) * 32 + (col & 31) : (size_t)row * N + col;
            v[i][j] = *reinterpret_cast<const f32x4 *>(x + off);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = m0 + 32 * wave + 16 * i + l15;
        if (row >= M) continue;
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            const int col = n0 + 16 * j + 4 * q;
            const size_t off = PLANES ? ((size_t)(col >> 5) * M + row) * 32 + (col & 31) : (size_t)row * N + col;
            *reinterpret_cast<f32x4 *>(y + off) = v[i][j];
        }
    }
}

int main()
{
    const int M = 100864;
    float *x, *y;
    CK(hipMalloc(&x, (size_t)M * N * 4));
    CK(hipMalloc(&y, (size_t)M * N * 4));
    CK(hipMemset(x, 0x11, (size_t)M * N * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char *names[4] = {"rows (one wave per 3 KB row)  ", "tile128 (16 rows x 64 B / inst)", "tile256                        ", "planes [24][rows][32] fp32     "};
    const int mt = (M + 127) / 128;
    double best[4] = {1e9, 1e9, 1e9, 1e9}, sum[4] = {0, 0, 0, 0};
    const int ROUNDS = 5, REPS = 10;
    for (int r = -1; r < ROUNDS; ++r)
        for (int k = 0; k < 4; ++k) {
            CK(hipEventRecord(e0));
            for (int rep = 0; rep < REPS; ++rep) {
                if (k == 0) hipLaunchKernelGGL(copy_rows, dim3(256 * 8), dim3(256), 0, 0, x, y, M);
                if (k == 1) hipLaunchKernelGGL((copy_tiles<128, false>), dim3(mt * 6), dim3(256), 0, 0, x, y, M);
                if (k == 2) hipLaunchKernelGGL((copy_tiles<256, false>), dim3(mt * 3), dim3(256), 0, 0, x, y, M);
                if (k == 3) hipLaunchKernelGGL((copy_tiles<128, true>), dim3(mt * 6), dim3(256), 0, 0, x, y, M);
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= REPS;
            if (r >= 0) { sum[k] += ms; if (ms < best[k]) best[k] = ms; }
        }
    const double bytes = 2.0 * M * N * 4;
    printf("copy of x [%d][%d] fp32 (read + write %.0f MB), %d rounds x %d launches, patterns interleaved\n", M, N, bytes / 1e6, ROUNDS, REPS);
    for (int k = 0; k < 4; ++k)
        printf("%s  mean %.4f ms  min %.4f ms   %.2f TB/s\n", names[k], sum[k] / ROUNDS, best[k], bytes / (sum[k] / ROUNDS * 1e-3) / 1e12);
    return 0;
}
