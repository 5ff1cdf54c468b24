/* p3_lab.hip -- throw-away ablations of the pre-split GEMM kernel (csrc/gemm_p3.hip), all variants timed
 * in interleaved rounds inside one process on random operands (guide rules 24/25).  Not part of the library.
 *   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I include -I vit-with-opencl_amd/csrc \
 *         tools/p3_lab.hip vit-with-opencl_amd/csrc/kernelHandler.hip -o tools/p3_lab
 *   tools/p3_lab [M N K]      (default: the fc1 shape 98304 3072 768) */
#include "../vit-with-opencl_amd/csrc/gemm_p3.hip"
#include <cstdio>
#include <vector>

/* defined in gemm_mfma.hip, which the lab does not link: the patch-embedding launcher is not exercised here */
int vh_cls_rows(hipStream_t, const float *, const float *, float *, int, int, int) { return 0; }

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

template <int NW, int BN, int EPI, int OUTK, int LAB, int NPL = 3>
void launch_variant(hipStream_t st, P3Params p)
{
    constexpr int LDS = 2 * (NPL == 3 ? 1 : P1_KG) * NPL * BN * 64;
    static bool set = false;
    if (!set) { CK(hipFuncSetAttribute((const void *)gemm_p3_kernel<NW, BN, EPI, OUTK, NPL, LAB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); set = true; }
    p.mtiles = (p.row_end - p.row_begin + 32 * NW - 1) / (32 * NW);
    p.ntiles = p.N / BN;
    hipLaunchKernelGGL((gemm_p3_kernel<NW, BN, EPI, OUTK, NPL, LAB>), dim3(p.mtiles * p.ntiles), dim3(64 * NW), LDS, st, p);
}
}

int main(int argc, char **argv)
{
    int M = 98304, N = 3072, K = 768, parts = 3;
    if (argc >= 4) { M = atoi(argv[1]); N = atoi(argv[2]); K = atoi(argv[3]); }
    if (argc >= 5) parts = atoi(argv[4]);
    CK(hipSetDevice(0));
    hipStream_t st; CK(hipStreamCreate(&st));
    float *x, *w, *bias; char *x3, *w3, *c3;
    CK(hipMalloc(&x, (size_t)M * K * 4)); CK(hipMalloc(&w, (size_t)N * K * 4)); CK(hipMalloc(&bias, (size_t)N * 4));
    CK(hipMalloc(&x3, (size_t)M * K * 6)); CK(hipMalloc(&w3, (size_t)N * K * 6)); CK(hipMalloc(&c3, (size_t)M * N * 6));
    fill_random<<<(unsigned)(((size_t)M * K + 255) / 256), 256, 0, st>>>(x, (size_t)M * K, 1u, 1.0f);
    fill_random<<<(unsigned)(((size_t)N * K + 255) / 256), 256, 0, st>>>(w, (size_t)N * K, 2u, 0.04f);
    fill_random<<<(N + 255) / 256, 256, 0, st>>>(bias, (size_t)N, 3u, 0.1f);
    if (vh_launch_split_rows(st, x, x3, M, K, parts) || vh_launch_split_rows(st, w, w3, N, K, parts)) { printf("split failed\n"); return 1; }
    CK(hipStreamSynchronize(st));
    P3Params p = {};
    p.A = x3; p.W = w3; p.bias = bias; p.C = c3; p.row_begin = 0; p.row_end = M; p.a_rows = M; p.N = N; p.K = K;

    struct V { const char *name; void (*fn)(hipStream_t, P3Params); };
    std::vector<V> vs = {
        {"product (fc1: GELU, planes out)", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 0>},
        {"no W fragment reads            ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 1>},
        {"no W DMA                       ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 2>},
        {"no A loads                     ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 4>},
        {"no reads, no DMA, no A loads   ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 7>},
        {"... and no barrier             ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 15>},
        {"product, no GELU, fp32 out     ", launch_variant<8, 256, EPI_NONE, OUT_F32, 0>},
        {"A rows of tile 0 everywhere    ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 16>},
        {"W rows of tile 0 everywhere    ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 32>},
        {"A and W of tile 0 everywhere   ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 48>},
        {"no epilogue stores             ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 64>},
        {"no stores, no reads/DMA/A loads", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 64 + 7>},
    };
    if (parts == 1)
        vs = {
            {"bf16 product (fc1: GELU, planes) ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 0, 1>},
            {"no W fragment reads              ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 1, 1>},
            {"no W DMA                         ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 2, 1>},
            {"no A loads                       ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 4, 1>},
            {"no reads, no DMA, no A loads     ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 7, 1>},
            {"... and no barrier               ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 15, 1>},
            {"product, no GELU, fp32 out       ", launch_variant<8, 256, EPI_NONE, OUT_F32, 0, 1>},
            {"A and W of tile 0 everywhere     ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 48, 1>},
            {"small tile 128x128               ", launch_variant<4, 128, EPI_GELU, OUT_PLANES, 0, 1>},
            {"128x256 tile, 2 workgroups/CU    ", launch_variant<4, 256, EPI_GELU, OUT_PLANES, 0, 1>},
            {"128x256, no stores               ", launch_variant<4, 256, EPI_GELU, OUT_PLANES, 64, 1>},
            {"128x256, no A loads              ", launch_variant<4, 256, EPI_GELU, OUT_PLANES, 4, 1>},
            {"128x256, no DMA                  ", launch_variant<4, 256, EPI_GELU, OUT_PLANES, 2, 1>},
            {"128x256, no W reads              ", launch_variant<4, 256, EPI_GELU, OUT_PLANES, 1, 1>},






            {"no epilogue stores               ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 64, 1>},
            {"no stores, no reads/DMA/A loads  ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 64 + 7, 1>},
            {"no stores/reads/DMA/A/barrier    ", launch_variant<8, 256, EPI_NONE, OUT_PLANES, 64 + 15, 1>},
        };
    if (argc >= 6 && parts == 1) {   /* "resid": the residual epilogue (out-projection / fc2 shapes), fp32 rows in place */
        float *xres;
        CK(hipMalloc(&xres, (size_t)M * N * 4));
        fill_random<<<(unsigned)(((size_t)M * N + 255) / 256), 256, 0, st>>>(xres, (size_t)M * N, 4u, 1.0f);
        p.R = xres; p.C = xres;
        vs = {
            {"128x128, residual in accumulators", launch_variant<4, 128, EPI_RESID, OUT_F32, 0, 1>},
            {"256x256, residual in accumulators", launch_variant<8, 256, EPI_RESID, OUT_F32, 0, 1>},
            {"128x256, residual in accumulators", launch_variant<4, 256, EPI_RESID, OUT_F32, 0, 1>},
            {"128x128, permuted columns        ", launch_variant<4, 128, EPI_RESID, OUT_F32, 256, 1>},
            {"256x256, permuted columns        ", launch_variant<8, 256, EPI_RESID, OUT_F32, 256, 1>},
            {"128x128, residual in the epilogue", launch_variant<4, 128, EPI_RESID, OUT_F32, 128 + 256, 1>},
            {"256x256, residual in the epilogue", launch_variant<8, 256, EPI_RESID, OUT_F32, 128 + 256, 1>},
            {"128x128, no residual read        ", launch_variant<4, 128, EPI_NONE, OUT_F32, 0, 1>},
            {"128x128, no stores               ", launch_variant<4, 128, EPI_RESID, OUT_F32, 64, 1>},
            {"128x128, no resid read, no stores", launch_variant<4, 128, EPI_NONE, OUT_F32, 64, 1>},
            {"128x128, no A loads              ", launch_variant<4, 128, EPI_RESID, OUT_F32, 4, 1>},
            {"128x128, no W DMA                ", launch_variant<4, 128, EPI_RESID, OUT_F32, 2, 1>},
            {"128x128, no A/DMA/W reads        ", launch_variant<4, 128, EPI_RESID, OUT_F32, 7, 1>},
            {"128x128, none of those, no stores", launch_variant<4, 128, EPI_NONE, OUT_F32, 64 + 7, 1>},
            {"256x256, no residual read        ", launch_variant<8, 256, EPI_NONE, OUT_F32, 0, 1>},
            {"256x256, no stores               ", launch_variant<8, 256, EPI_RESID, OUT_F32, 64, 1>},
        };
    }
    const int ROUNDS = 4, REPS = 10;
    std::vector<double> best(vs.size(), 1e30), sum(vs.size(), 0.0);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto &v : vs) { v.fn(st, p); }          /* warm-up, attributes */
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
    printf("M=%d N=%d K=%d, %d rounds x %d launches, random operands\n", M, N, K, ROUNDS, REPS);
    for (size_t i = 0; i < vs.size(); ++i)
        printf("%s  mean %.3f ms  min %.3f ms  %.1f TFLOP/s of products (%.0f bf16 MFMA TFLOP/s)\n", vs[i].name,
               sum[i] / ROUNDS, best[i], flop / (sum[i] / ROUNDS * 1e-3) / 1e12, (parts == 3 ? 6 : 1) * flop / (sum[i] / ROUNDS * 1e-3) / 1e12);
    return 0;
}
