/* p3_lab.hip -- throw-away ablations of the pre-split GEMM kernel (csrc/gemm_p3.hip), all variants timed
 * in interleaved rounds inside one process on random operands (guide rules 24/25).  Not part of the library.
 *   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I include -I vit-with-opencl_amd/csrc -I tools \
 *         tools/p3_lab.hip vit-with-opencl_amd/csrc/kernelHandler.hip -o tools/p3_lab
 *   tools/p3_lab [M N K]      (default: the fc1 shape 98304 3072 768) */
#include "p3_lab_kernel.inc"
#include <cmath>
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

/* ---- the A/B the round-2 review asked for: the three-part fc1 loop on v_mfma_f32_32x32x16_bf16 --------------------
 * Same tile (256 x 256, 8 waves of 32 x 256), same planes, same LDS image and DMA, same six products per block in the
 * same order, A straight from global memory one K step ahead.  What changes: one 32-row A block per wave instead of
 * two 16-row fragments, W taken in units of (32 columns, 16 k) -- three ds_read_b128 and SIX 32x32x16 MFMAs per unit
 * where the product kernel has three reads and TWELVE 16x16x32 MFMAs per fragment: half the MFMA instructions and
 * half the operand-register reads per FLOP at the same LDS traffic.  A K step of 32 is two sub-steps of 16: lane
 * (x = l & 31, h = l >> 5) contracts the row's 16-byte chunk h + 2s in sub-step s (the same assignment for A and W).
 * W rows are taken in the order sigma(x) = 8 ((x >> 2) & 1) + 16 (x >> 4) + 4 ((x >> 3) & 1) + (x & 3), so that a lane
 * ends up with columns 8h + 16 (r >> 3) + (r & 7) of its row: two runs of eight consecutive columns (16-byte
 * stores of bf16 parts), and the 16 lanes of a ds_read_b128 group still read 16 consecutive LDS rows. */
template <int EPI, int OUTK, int LAB>
__global__ __launch_bounds__(512, 2) void gemm_p3_m32_kernel(const P3Params p)
{
    constexpr int NW = 8, BN = 256, NPL = 3, BM = 32 * NW, JB = BN / 32;
    constexpr int RING = 2, U = JB;                 /* W units (sub-step s, column-block PAIR jp) per K step: two accumulator chains per unit */
    constexpr int GROUP = NPL * BN * 64, STAGE = GROUP;
    constexpr int PW = NPL * BN / 16 / NW;
    typedef bf16x8 frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tile = xcd_tile(blockIdx.x, p.mtiles * p.ntiles);
    const int m0 = p.row_begin + (tile / p.ntiles) * BM;
    const int n0 = (tile % p.ntiles) * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = lane & 31, h = lane >> 5;

    const unsigned aoff = (unsigned)min(m0 + 32 * wave + x, p.row_end - 1) * 64u + 16u * h;   /* + 32 s */
    const size_t a_plane = (size_t)p.a_rows * 64, w_plane = (size_t)p.N * 64;
    const unsigned wlane = (unsigned)(lane >> 2) * 64u + 16u * ((lane & 3) ^ swz64(lane >> 4));
    const gchar_t wtile = (gchar_t)p.W + (size_t)n0 * 64;
    const int sig = 8 * ((x >> 2) & 1) + 16 * (x >> 4) + 4 * ((x >> 3) & 1) + (x & 3);
    unsigned woff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
        woff[s] = (unsigned)sig * 64u + 16u * ((h + 2 * s) ^ swz64(sig >> 2));

    f32x16 acc[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const float *bp = p.bias + n0 + 32 * j + 8 * h + 16 * g;
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(bp), b1 = *reinterpret_cast<const f32x4 *>(bp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[j][8 * g + e] = b0[e];
                acc[j][8 * g + 4 + e] = b1[e];
            }
        }

    frag_t a0[2][NPL], a1[2][NPL], w[RING][2 * NPL];
    auto load_a = [&](frag_t (&a)[2][NPL], int kt) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            gchar_t base = (gchar_t)p.A + (size_t)(kt * NPL + pl) * a_plane;
            asm volatile("" : "+s"(base));
#pragma unroll
            for (int s = 0; s < 2; ++s)
                a[s][pl] = __builtin_bit_cast(frag_t, *reinterpret_cast<gvec_t>(base + aoff + 32u * s));
        }
    };
    auto dma_w = [&](int stage, int kt) {
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int pc = wave * PW + i, gp = pc / (BN / 16), rb = pc - gp * (BN / 16);
            gchar_t src = wtile + ((size_t)(kt * NPL + gp) * w_plane + (size_t)rb * 1024);
            asm volatile("" : "+s"(src));
            __builtin_amdgcn_global_load_lds((gptr_t)(src + wlane), (lptr_t)(smem + stage * STAGE + pc * 1024), 16, 0, 0);
        }
    };
    auto read_w = [&](frag_t (&wf)[2 * NPL], const char *stage, int u) {   /* unit u = s * (JB / 2) + jp: column blocks 2 jp, 2 jp + 1 */
        const int s = u / (JB / 2), jp = u % (JB / 2);
#pragma unroll
        for (int o = 0; o < NPL; ++o) {
            const int pl = (NPL - o) % NPL;
#pragma unroll
            for (int b = 0; b < 2; ++b)
                wf[2 * pl + b] = __builtin_bit_cast(frag_t, *reinterpret_cast<const f32x4 *>(stage + woff[s] + (2 * jp + b) * 2048 + pl * (BN * 64)));
        }
    };
    auto mfma_unit = [&](const frag_t (&a)[2][NPL], const frag_t (&wf)[2 * NPL], int u) {
        const int s = u / (JB / 2), jp = u % (JB / 2);
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc[2 * jp + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[2 * term_w<3>(t) + b], a[s][term_a<3>(t)], acc[2 * jp + b], 0, 0, 0);
    };
    auto interleave = [&]() {
#pragma unroll
        for (int r = 0; r < NPL; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    const int nk = p.K / 32;
    auto step = [&](const frag_t (&au)[2][NPL], frag_t (&al)[2][NPL], int kt) {
        const char *cur = smem + (kt & 1) * STAGE, *nxt = smem + ((kt + 1) & 1) * STAGE;
        const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
        load_a(al, more1 ? kt + 1 : kt);
#pragma unroll
        for (int u = 0; u <= U - RING; ++u) {
            read_w(w[(u + RING - 1) % RING], cur, u + RING - 1);
            mfma_unit(au, w[u % RING], u);
            interleave();
        }
        __syncthreads();
        if (more2)
            dma_w(kt & 1, kt + 2);
#pragma unroll
        for (int u = U - RING + 1; u < U; ++u) {
            if (more1)
                read_w(w[(u + RING - 1) % RING], nxt, u + RING - 1 - U);
            mfma_unit(au, w[u % RING], u);
            interleave();
        }
    };

    dma_w(0, 0);
    load_a(a0, 0);
    __syncthreads();
    if (nk > 1)
        dma_w(1, 1);
    read_w(w[0], smem, 0);
    for (int kt = 0; kt < nk; kt += 2) {
        step(a0, a1, kt);
        step(a1, a0, kt + 1);
    }

    const int row = m0 + 32 * wave + x;
    if (row >= p.row_end)
        return;
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f32x4 lo, hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lo[e] = acc[j][8 * g + e];
                hi[e] = acc[j][8 * g + 4 + e];
            }
            const int col = n0 + 32 * j + 8 * h + 16 * g;
            if (EPI == EPI_GELU) {
                const f32x2 g0 = gelu_exact2(f32x2{lo[0], lo[1]}), g1 = gelu_exact2(f32x2{lo[2], lo[3]});
                const f32x2 g2 = gelu_exact2(f32x2{hi[0], hi[1]}), g3 = gelu_exact2(f32x2{hi[2], hi[3]});
                lo = f32x4{g0[0], g0[1], g1[0], g1[1]};
                hi = f32x4{g2[0], g2[1], g3[0], g3[1]};
            }
            if (LAB & 64) {
                asm volatile("" ::"v"(lo), "v"(hi));
            } else if (OUTK == OUT_PLANES) {
                frag_t part[3];
                split8(lo, hi, part[0], part[1], part[2]);
                char *dst = static_cast<char *>(p.C) + ((size_t)(col >> 5) * NPL * p.a_rows + row) * 64 + 2 * (col & 31);
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    *reinterpret_cast<f32x4 *>(dst + pl * a_plane) = __builtin_bit_cast(f32x4, part[pl]);
            } else {
                float *cp = static_cast<float *>(p.C) + (size_t)row * p.N + col;
                *reinterpret_cast<f32x4 *>(cp) = lo;
                *reinterpret_cast<f32x4 *>(cp + 4) = hi;
            }
        }
}

template <int EPI, int OUTK, int LAB>
void launch_m32(hipStream_t st, P3Params p)
{
    constexpr int LDS = 2 * 3 * 256 * 64;
    static bool set = false;
    if (!set) { CK(hipFuncSetAttribute((const void *)gemm_p3_m32_kernel<EPI, OUTK, LAB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); set = true; }
    p.mtiles = (p.row_end - p.row_begin + 255) / 256;
    p.ntiles = p.N / 256;
    hipLaunchKernelGGL((gemm_p3_m32_kernel<EPI, OUTK, LAB>), dim3(p.mtiles * p.ntiles), dim3(512), LDS, st, p);
}

template <int NW, int BN, int EPI, int OUTK, int LAB, int NPL = 3>
void launch_variant(hipStream_t st, P3Params p)
{
    constexpr int LDS = ((LAB & 262144) ? 3 : 2) * (NPL == 3 ? 1 : P1_KG) * NPL * BN * 64;
    static bool set = false;
    if (!set) { CK(hipFuncSetAttribute((const void *)gemm_p3_kernel<NW, BN, EPI, OUTK, NPL, LAB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); set = true; }
    p.mtiles = (p.row_end - p.row_begin + 32 * NW - 1) / (32 * NW);
    p.ntiles = p.N / BN;
    hipLaunchKernelGGL((gemm_p3_kernel<NW, BN, EPI, OUTK, NPL, LAB>), dim3(p.mtiles * p.ntiles), dim3(64 * NW), LDS, st, p);
}

/* round 4: the two co-resident 128x256 workgroups of a CU half a tile apart (gemm_common.h lab_stagger_start).
 * BY_SLOT: the second workgroup to ARRIVE on a CU is the late one (counters zeroed before every launch); otherwise the
 * workgroups [256, 512) of the grid are. */
unsigned *g_slots;
template <int EPI, int OUTK, int CYCLES, bool BY_SLOT>
void launch_stagger(hipStream_t st, P3Params p)
{
    p.lab_lo = 256; p.lab_hi = 512; p.lab_cycles = CYCLES; p.lab_slots = g_slots;
    if (BY_SLOT)
        CK(hipMemsetAsync(g_slots, 0, 4096 * sizeof(unsigned), st));
    launch_variant<4, 256, EPI, OUTK, BY_SLOT ? 1024 : 512, 1>(st, p);
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
        {"128x128 tiles (fc1: GELU, planes) ", launch_variant<4, 128, EPI_GELU, OUT_PLANES, 0>},
        {"256x256, no GELU, planes out (QKV)", launch_variant<8, 256, EPI_NONE, OUT_PLANES, 0>},
        {"128x128, no GELU, planes out (QKV)", launch_variant<4, 128, EPI_NONE, OUT_PLANES, 0>},
        {"32x32x16 MFMA (fc1: GELU, planes)", launch_m32<EPI_GELU, OUT_PLANES, 0>},
        {"32x32x16 MFMA, no stores         ", launch_m32<EPI_GELU, OUT_PLANES, 64>},
        {"32x32x16 MFMA, no GELU, fp32 out ", launch_m32<EPI_NONE, OUT_F32, 0>},
    };
    if (parts == 3) {   /* the 32x32x16 form against the product kernel, fp32 rows out: same products, another instruction shape */
        float *c_a, *c_b;
        CK(hipMalloc(&c_a, (size_t)M * N * 4)); CK(hipMalloc(&c_b, (size_t)M * N * 4));
        P3Params q = p;
        q.C = c_a; launch_variant<8, 256, EPI_NONE, OUT_F32, 0>(st, q);
        q.C = c_b; launch_m32<EPI_NONE, OUT_F32, 0>(st, q);
        CK(hipStreamSynchronize(st));
        std::vector<float> ha((size_t)4096 * N), hb((size_t)4096 * N);
        double worst = 0.0, mag = 0.0; size_t diff = 0;
        for (size_t r0 : {(size_t)0, (size_t)M / 2, (size_t)M - 4096}) {
            CK(hipMemcpy(ha.data(), c_a + r0 * N, ha.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hb.data(), c_b + r0 * N, hb.size() * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < ha.size(); ++i) {
                const double d = fabs((double)ha[i] - hb[i]);
                if (d > worst) worst = d;
                if (fabs(ha[i]) > mag) mag = fabs(ha[i]);
                diff += ha[i] != hb[i];
            }
        }
        printf("32x32x16 form vs product kernel (fp32 rows, 3 x 4096 rows): max |difference| %.3e at magnitudes up to %.2f, %zu of %zu values differ in bits\n",
               worst, mag, diff, (size_t)3 * 4096 * N);
        CK(hipFree(c_a)); CK(hipFree(c_b));
    }
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
#define STAG(E, O, C) {"128x256, [256,512) late by " #C, launch_stagger<E, O, C, false>}, {"128x256, 2nd on CU late by " #C, launch_stagger<E, O, C, true>}
    if (argc >= 6 && parts == 1 && argv[5][0] == 's') {   /* "stagger": fc1 (GELU, planes out) */
        CK(hipMalloc(&g_slots, 4096 * sizeof(unsigned)));
        vs = {
            {"256x256, one workgroup per CU    ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 0, 1>},
            {"128x256, two per CU, in step     ", launch_variant<4, 256, EPI_GELU, OUT_PLANES, 0, 1>},
            STAG(EPI_GELU, OUT_PLANES, 3000), STAG(EPI_GELU, OUT_PLANES, 6000), STAG(EPI_GELU, OUT_PLANES, 9000),
            STAG(EPI_GELU, OUT_PLANES, 12000), STAG(EPI_GELU, OUT_PLANES, 16000), STAG(EPI_GELU, OUT_PLANES, 22000),
            {"128x256, no stores, in step      ", launch_variant<4, 256, EPI_GELU, OUT_PLANES, 64, 1>},
        };
    } else if (argc >= 6 && argv[5][0] == 'p' && parts == 3) {   /* "prio": static wave priorities (round 4) */
        vs = {
            {"product (fc1: GELU, planes out)   ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 0>},
            {"waves 4-7 at priority 1 in K loop ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 4096>},
            {"product again                     ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 0>},
            {"all waves at 2 in K loop, 0 after ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 8192>},
            {"all waves at 1 in K loop, 0 after ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 8192 + 16384>},
            {"all waves at 3 in K loop, 0 after ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 8192 + 32768>},
            {"product a third time              ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 0>},
            {"128x128 tiles                     ", launch_variant<4, 128, EPI_GELU, OUT_PLANES, 0>},
            {"128x128, K loop at 2, epilogue 0  ", launch_variant<4, 128, EPI_GELU, OUT_PLANES, 8192>},
            {"QKV shape epilogue: 256x256       ", launch_variant<8, 256, EPI_NONE, OUT_PLANES, 0>},
            {"QKV: all waves at 2 in K loop     ", launch_variant<8, 256, EPI_NONE, OUT_PLANES, 8192>},
        };
    } else if (argc >= 6 && argv[5][0] == 'h') {   /* "half-step": waves NW/2.. half a K step behind their SIMD partners */
        if (parts == 1)
            vs = {
                {"256x256 bf16 product (fc1)        ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 0, 1>},
                {"  waves 4-7 half a step behind    ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 262144, 1>},
                {"256x256 again                     ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 0, 1>},
                {"QKV epilogue (fp16 planes) 256x256", launch_variant<8, 256, EPI_NONE, OUT_PLANES_H, 0, 1>},
                {"  waves 4-7 half a step behind    ", launch_variant<8, 256, EPI_NONE, OUT_PLANES_H, 262144, 1>},
                {"no stores, 256x256                ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 64, 1>},
                {"  waves 4-7 half a step behind    ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 64 + 262144, 1>},
            };
        else
            vs = {
                {"product (fc1: GELU, planes out)   ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 0>},
                {"  waves 4-7 half a step behind    ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 262144>},
                {"product again                     ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 0>},
                {"QKV epilogue: 256x256             ", launch_variant<8, 256, EPI_NONE, OUT_PLANES, 0>},
                {"  waves 4-7 half a step behind    ", launch_variant<8, 256, EPI_NONE, OUT_PLANES, 262144>},
            };
    } else if (argc >= 6 && argv[5][0] == 'w' && parts == 1) {   /* "waits": LDS reads and counted waits by hand */
        vs = {
            {"256x256 bf16 product (fc1)        ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 0, 1>},
            {"  LDS reads + counted waits in asm", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 131072, 1>},
            {"256x256 again                     ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 0, 1>},
            {"128x256, two workgroups per CU    ", launch_variant<4, 256, EPI_GELU, OUT_PLANES, 0, 1>},
            {"  LDS reads + counted waits in asm", launch_variant<4, 256, EPI_GELU, OUT_PLANES, 131072, 1>},
            {"no stores, 256x256                ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 64, 1>},
            {"  LDS reads + counted waits in asm", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 64 + 131072, 1>},
            {"QKV epilogue (fp16 planes) 256x256", launch_variant<8, 256, EPI_NONE, OUT_PLANES_H, 0, 1>},
            {"  LDS reads + counted waits in asm", launch_variant<8, 256, EPI_NONE, OUT_PLANES_H, 131072, 1>},
        };
    } else if (argc >= 6 && argv[5][0] == 'p' && parts == 1) {
        vs = {
            {"256x256 bf16 product (fc1)        ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 0, 1>},
            {"waves 4-7 at priority 1 in K loop ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 4096, 1>},
            {"all waves at 2 in K loop, 0 after ", launch_variant<8, 256, EPI_GELU, OUT_PLANES, 8192, 1>},
            {"128x256, two workgroups per CU    ", launch_variant<4, 256, EPI_GELU, OUT_PLANES, 0, 1>},
            {"128x256, K loop at 2, epilogue 0  ", launch_variant<4, 256, EPI_GELU, OUT_PLANES, 8192, 1>},
            {"128x256, waves 2-3 at 1 in K loop ", launch_variant<4, 256, EPI_GELU, OUT_PLANES, 4096, 1>},
        };
    } else if (argc >= 6 && parts == 1 && argv[5][0] == 'q') {   /* "qkv-stagger": no GELU, fp16 planes out */
        CK(hipMalloc(&g_slots, 4096 * sizeof(unsigned)));
        vs = {
            {"256x256, one workgroup per CU    ", launch_variant<8, 256, EPI_NONE, OUT_PLANES_H, 0, 1>},
            {"128x256, two per CU, in step     ", launch_variant<4, 256, EPI_NONE, OUT_PLANES_H, 0, 1>},
            STAG(EPI_NONE, OUT_PLANES_H, 3000), STAG(EPI_NONE, OUT_PLANES_H, 6000), STAG(EPI_NONE, OUT_PLANES_H, 9000),
            STAG(EPI_NONE, OUT_PLANES_H, 12000), STAG(EPI_NONE, OUT_PLANES_H, 16000),
        };
    } else
    if (argc >= 6 && parts == 1 && argv[5][0] != 'w' && argv[5][0] != 'h') {   /* "resid": the residual epilogue (out-projection / fc2 shapes), fp32 rows in place */
        float *xres;
        CK(hipMalloc(&xres, (size_t)M * N * 4));
        fill_random<<<(unsigned)(((size_t)M * N + 255) / 256), 256, 0, st>>>(xres, (size_t)M * N, 4u, 1.0f);
        p.R = xres; p.C = xres;
        vs = {
            {"128x128, residual in accumulators", launch_variant<4, 128, EPI_RESID, OUT_F32, 0, 1>},
            {"128x128, <= 128 registers (4/CU) ", launch_variant<4, 128, EPI_RESID, OUT_F32, 65536, 1>},
            {"128x128 again                    ", launch_variant<4, 128, EPI_RESID, OUT_F32, 0, 1>},
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
    if (argc >= 6 && parts == 3 && argv[5][0] != 'p' && argv[5][0] != 'h') {   /* "resid", three parts: the fp32 path's out-projection / fc2 (residual added to the finished sum) */
        float *xres;
        CK(hipMalloc(&xres, (size_t)M * N * 4));
        fill_random<<<(unsigned)(((size_t)M * N + 255) / 256), 256, 0, st>>>(xres, (size_t)M * N, 4u, 1.0f);
        p.R = xres; p.C = xres;
        vs = {
            {"128x128 tiles, + residual        ", launch_variant<4, 128, EPI_RESID, OUT_F32, 0>},
            {"256x256 tiles, + residual        ", launch_variant<8, 256, EPI_RESID, OUT_F32, 0>},
            {"128x128, no residual read        ", launch_variant<4, 128, EPI_NONE, OUT_F32, 0>},
            {"128x128, no stores               ", launch_variant<4, 128, EPI_RESID, OUT_F32, 64>},
            {"128x128, no resid read, no stores", launch_variant<4, 128, EPI_NONE, OUT_F32, 64>},
            {"128x128, no A loads              ", launch_variant<4, 128, EPI_RESID, OUT_F32, 4>},
            {"128x128, no W DMA                ", launch_variant<4, 128, EPI_RESID, OUT_F32, 2>},
            {"128x128, no A/DMA/W reads        ", launch_variant<4, 128, EPI_RESID, OUT_F32, 7>},
            {"128x128, none of those, no stores", launch_variant<4, 128, EPI_NONE, OUT_F32, 64 + 7>},
            {"256x256, no residual read        ", launch_variant<8, 256, EPI_NONE, OUT_F32, 0>},
            {"256x256, no stores               ", launch_variant<8, 256, EPI_RESID, OUT_F32, 64>},
        };
    }
    if (argc >= 6 && argv[5][0] == 'h') {   /* the staggered variant must write the same bytes as the product kernel */
        const size_t nbytes = (size_t)M * N * 2 * parts, probe = nbytes < ((size_t)64 << 20) ? nbytes : ((size_t)64 << 20);
        std::vector<unsigned char> ha(2 * probe), hb(2 * probe);
        CK(hipMemsetAsync(c3, 0, nbytes, st));
        vs[0].fn(st, p);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(ha.data(), c3, probe, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ha.data() + probe, (char *)c3 + nbytes - probe, probe, hipMemcpyDeviceToHost));
        CK(hipMemsetAsync(c3, 0, nbytes, st));
        vs[1].fn(st, p);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(hb.data(), c3, probe, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb.data() + probe, (char *)c3 + nbytes - probe, probe, hipMemcpyDeviceToHost));
        size_t diff = 0, nz = 0;
        for (size_t i = 0; i < 2 * probe; ++i) { diff += ha[i] != hb[i]; nz += ha[i] != 0; }
        printf("half-step variant against the product kernel, first and last %zu MB of the output: %zu bytes differ (%zu non-zero bytes compared)\n",
               probe >> 20, diff, nz);
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
