/* mx_lab.hip -- throw-away ablations of the block-scaled fp8 GEMM kernel (csrc/gemm_mx.hip), all variants timed in
 * interleaved rounds inside one process on random operands (the method of tools/p3_lab.hip).  Not part of the library.
 *   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I include -I vit-with-opencl_amd/csrc -I tools \
 *         tools/mx_lab.hip vit-with-opencl_amd/csrc/kernelHandler.hip -o tools/mx_lab
 *   tools/mx_lab [M N K [resid]]      (default: the fc1 shape 100864 3072 768; "resid" = the fc2 / out-projection epilogue) */
#include "mx_lab_kernel.inc"
#include <cstdio>
#include <vector>

namespace {
__global__ void fill_random(float *x, size_t n, unsigned seed, float scale)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    x[i] = ((int)(h >> 8) - (1 << 23)) * (scale / (1 << 23));
}
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NW, int BN, int EPI, int OUTK, int LAB>
void launch_variant(hipStream_t st, MxParams p)
{
    constexpr int LDS = 2 * (BN * 128 + 4 * BN);
    static bool set = false;
    if (!set) { CK(hipFuncSetAttribute((const void *)gemm_mx_kernel<NW, BN, EPI, OUTK, LAB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); set = true; }
    p.mtiles = (p.row_end - p.row_begin + 32 * NW - 1) / (32 * NW);
    p.ntiles = p.N / BN;
    hipLaunchKernelGGL((gemm_mx_kernel<NW, BN, EPI, OUTK, LAB>), dim3(p.mtiles * p.ntiles), dim3(64 * NW), LDS, st, p);
}

/* round 4: the two co-resident workgroups of a CU half a tile apart (gemm_common.h lab_stagger_start) */
unsigned *g_slots;
template <int EPI, int OUTK, int CYCLES, bool BY_SLOT>
void launch_stagger(hipStream_t st, MxParams p)
{
    p.lab_lo = 256; p.lab_hi = 512; p.lab_cycles = CYCLES; p.lab_slots = g_slots;
    if (BY_SLOT)
        CK(hipMemsetAsync(g_slots, 0, 4096 * sizeof(unsigned), st));
    launch_variant<4, 256, EPI, OUTK, BY_SLOT ? 1024 : 512>(st, p);
}
}

int main(int argc, char **argv)
{
    int M = 100864, N = 3072, K = 768;
    if (argc >= 4) { M = atoi(argv[1]); N = atoi(argv[2]); K = atoi(argv[3]); }
    const bool resid = argc >= 5;
    CK(hipSetDevice(0));
    hipStream_t st; CK(hipStreamCreate(&st));
    float *x, *w, *bias, *xres; char *xv, *xs, *wv, *ws, *cv, *cs;
    CK(hipMalloc(&x, (size_t)M * K * 4)); CK(hipMalloc(&w, (size_t)N * K * 4)); CK(hipMalloc(&bias, (size_t)N * 4));
    CK(hipMalloc(&xres, (size_t)M * N * 4));
    CK(hipMalloc(&xv, (size_t)M * K)); CK(hipMalloc(&xs, (size_t)M * K / 32 + 256));
    CK(hipMalloc(&wv, (size_t)N * K)); CK(hipMalloc(&ws, (size_t)N * K / 32 + 256));
    CK(hipMalloc(&cv, (size_t)M * N * 4)); CK(hipMalloc(&cs, (size_t)M * N / 32 + 256));   /* cv also takes the fp32-rows variants: 4 bytes per value */
    fill_random<<<(unsigned)(((size_t)M * K + 255) / 256), 256, 0, st>>>(x, (size_t)M * K, 1u, 1.0f);
    fill_random<<<(unsigned)(((size_t)N * K + 255) / 256), 256, 0, st>>>(w, (size_t)N * K, 2u, 0.04f);
    fill_random<<<(N + 255) / 256, 256, 0, st>>>(bias, (size_t)N, 3u, 0.1f);
    fill_random<<<(unsigned)(((size_t)M * N + 255) / 256), 256, 0, st>>>(xres, (size_t)M * N, 4u, 1.0f);
    if (vh_launch_quantize_mx_rows(st, x, xv, xs, M, K) || vh_launch_quantize_mx_rows(st, w, wv, ws, N, K)) { printf("quantise failed\n"); return 1; }
    CK(hipStreamSynchronize(st));
    MxParams p = {};
    p.A = xv; p.As = xs; p.W = wv; p.Ws = ws; p.bias = bias; p.C = cv; p.Cs = cs;
    p.row_begin = 0; p.row_end = M; p.a_rows = M; p.N = N; p.K = K;

    struct V { const char *name; void (*fn)(hipStream_t, MxParams); };
    std::vector<V> vs = {
        {"product: 128x256 (fc1: GELU, MX out) ", launch_variant<4, 256, EPI_GELU, OUT_MX, 0>},
        {"256x256 tiles, one workgroup per CU  ", launch_variant<8, 256, EPI_GELU, OUT_MX, 0>},
        {"128x128 tiles, three workgroups/CU   ", launch_variant<4, 128, EPI_GELU, OUT_MX, 0>},
        {"no W DMA                             ", launch_variant<4, 256, EPI_GELU, OUT_MX, 2>},
        {"no A loads                           ", launch_variant<4, 256, EPI_GELU, OUT_MX, 4>},
        {"no A scale-byte loads                ", launch_variant<4, 256, EPI_GELU, OUT_MX, 8>},
        {"A scales: one dword per 4 K steps    ", launch_variant<4, 256, EPI_GELU, OUT_MX, 2048>},
        {"no DMA, no A loads                   ", launch_variant<4, 256, EPI_GELU, OUT_MX, 6>},
        {"no reads, no DMA, no A loads         ", launch_variant<4, 256, EPI_GELU, OUT_MX, 7>},
        {"no epilogue stores                   ", launch_variant<4, 256, EPI_GELU, OUT_MX, 64>},
        {"no stores, no reads/DMA/A loads      ", launch_variant<4, 256, EPI_GELU, OUT_MX, 64 + 7>},
        {"no GELU, no quantisation: fp32 out   ", launch_variant<4, 256, EPI_NONE, OUT_F32, 0>},
        {"no GELU, MX out                      ", launch_variant<4, 256, EPI_NONE, OUT_MX, 0>},
        {"no GELU, fp32, no stores/movement    ", launch_variant<4, 256, EPI_NONE, OUT_F32, 64 + 7>},
    };
#define STAG(E, O, C) {"[256,512) late by " #C "             ", launch_stagger<E, O, C, false>}, {"2nd on CU late by " #C "             ", launch_stagger<E, O, C, true>}
    CK(hipMalloc(&g_slots, 4096 * sizeof(unsigned)));
    if (argc >= 5 && argv[4][0] == 's')   /* "stagger" */
        vs = {
            {"product: 128x256 (fc1: GELU, MX out) ", launch_variant<4, 256, EPI_GELU, OUT_MX, 0>},
            STAG(EPI_GELU, OUT_MX, 2000), STAG(EPI_GELU, OUT_MX, 4000), STAG(EPI_GELU, OUT_MX, 6000), STAG(EPI_GELU, OUT_MX, 9000),
            STAG(EPI_GELU, OUT_MX, 12000),
            {"no GELU, fp16 planes out (QKV epi.)  ", launch_variant<4, 256, EPI_NONE, OUT_PLANES_H, 0>},
            STAG(EPI_NONE, OUT_PLANES_H, 3000), STAG(EPI_NONE, OUT_PLANES_H, 6000),
        };
    else
    if (resid) {
        p.R = xres; p.C = xres; p.Cs = nullptr;
        vs = {
            {"product: 128x256 (+ residual, fp32)  ", launch_variant<4, 256, EPI_RESID, OUT_F32, 0>},
            {"256x256 tiles                        ", launch_variant<8, 256, EPI_RESID, OUT_F32, 0>},
            {"128x128 tiles, three workgroups/CU   ", launch_variant<4, 128, EPI_RESID, OUT_F32, 0>},
            {"no W DMA                             ", launch_variant<4, 256, EPI_RESID, OUT_F32, 2>},
            {"no A loads                           ", launch_variant<4, 256, EPI_RESID, OUT_F32, 4>},
            {"no A scale-byte loads                ", launch_variant<4, 256, EPI_RESID, OUT_F32, 8>},
            {"A scales: one dword per 4 K steps    ", launch_variant<4, 256, EPI_RESID, OUT_F32, 2048>},
            {"no DMA, no A loads                   ", launch_variant<4, 256, EPI_RESID, OUT_F32, 6>},
            {"no reads, no DMA, no A loads         ", launch_variant<4, 256, EPI_RESID, OUT_F32, 7>},
            {"no epilogue stores                   ", launch_variant<4, 256, EPI_RESID, OUT_F32, 64>},
            {"no residual                          ", launch_variant<4, 256, EPI_NONE, OUT_F32, 0>},
            {"no residual, no stores, no movement  ", launch_variant<4, 256, EPI_NONE, OUT_F32, 64 + 7>},
        };
    }
    const int ROUNDS = 4, REPS = 10;
    std::vector<double> best(vs.size(), 1e30), sum(vs.size(), 0.0);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto &v : vs) { v.fn(st, p); }
    CK(hipStreamSynchronize(st));
    for (int r = 0; r < ROUNDS; ++r)
        for (size_t i = 0; i < vs.size(); ++i) {
            CK(hipEventRecord(e0, st));
            for (int k = 0; k < REPS; ++k) vs[i].fn(st, p);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= REPS;
            sum[i] += ms; if (ms < best[i]) best[i] = ms;
        }
    const double flop = 2.0 * M * N * K;
    printf("M=%d N=%d K=%d, block-scaled e4m3, %d rounds x %d launches, random operands\n", M, N, K, ROUNDS, REPS);
    for (size_t i = 0; i < vs.size(); ++i)
        printf("%s  mean %.3f ms  min %.3f ms  %.0f TFLOP/s\n", vs[i].name, sum[i] / ROUNDS, best[i], flop / (sum[i] / ROUNDS * 1e-3) / 1e12);
    return 0;
}
