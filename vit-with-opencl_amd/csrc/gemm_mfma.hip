/*
 * gemm_mfma.hip -- the dense projections of the ViT forward pass on the gfx950 matrix cores.
 *
 *   C[M][N] = A[M][K] . W[N][K]^T  (+ bias, + GELU | + residual | patch epilogue)
 *
 * Replaces the reference's 8x8-tile OpenCL GEMMs `linear_layer` (ll.cl:7-86)
 * and `QKV` (multihead.cl:3-63), the element-wise `encoderResidual`
 * (layer_norm.cl:55-65) and -- through the im2row A-loader and the token
 * epilogue -- `conv2d_kernel` + `postprocess` (conv2d.cl:1-80).
 *
 * This file: the GEMMs whose fp32 operands are split INSIDE the K loop, and the alternates.  The model's four
 * big projections run in gemm_p3.hip (both operands pre-split by their producers); here are
 *   fp32, default   exact 3-way bf16 split of both operands in registers, six v_mfma_f32_16x16x32_bf16 per
 *                   product block: vh_launch_linear on raw fp32 weights and the patch embedding (im2row on load);
 *                   with the weights pre-split into planes (vh_launch_linear_w3) only the activations are split
 *   fp32, native    v_mfma_f32_32x32x2_f32: the ragged-N classifier, and everything under VIT_HIP_GEMM_FP32=native
 *   fp16 pairs      vh_launch_linear_h2 (opt-in emulation mode: two fp16 parts, three products)
 * Two kernel templates share one staging scheme: gemm_mf16_kernel (the 16x16x32 shapes) and
 * gemm_f32_kernel (native fp32, ragged N).
 *
 * Staging (MI355X / CDNA4):
 *  - Block tiles of BM x BN (template: 128x128 ... 256x256) x one 128-byte K step, cut into
 *    WM x WN waves.  Both operands are K-contiguous ("NT" GEMM), so A and W use the same LDS
 *    image: 128-byte rows whose 16-byte chunks are XOR-swizzled (chunk c of row r at
 *    c ^ ((r >> 1) & 7)), which makes every ds_read_b128 fragment read conflict-free (the 16
 *    rows of a lane group hit 16 distinct 16-B slots of the 256-B bank row).
 *  - K-tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 (LDS-DMA): no staging VGPRs, no
 *    ds_write instructions.  An LDS-DMA wave-instruction writes 64 x 16 B linearly
 *    (wave-uniform base + lane*16), so rows cannot be padded; the swizzle is applied to the
 *    per-lane SOURCE address and, identically, to the reads.  Two LDS stages: step t issues the
 *    DMA of K-tile t+1 into the stage read in step t-1 right after the barrier, computes
 *    K-tile t, and __syncthreads() (vmcnt(0) + barrier) publishes K-tile t+1 -- a load has a
 *    whole step of MFMAs to land.
 *  - The contraction order inside a K step is permuted to suit the fragment reads (the same
 *    permutation for A and W), so only the summation order differs from the scalar loop.
 *  - The accumulator starts at the bias, like the scalar loop it replaces
 *    (`sum = bias[o]`, ViT_seq.c:301), and the residual is added to the finished sum
 *    (ViT_seq.c:350,362).
 *  - blockIdx -> tile map is XCD-aware: each of the 8 XCDs walks a contiguous range of tiles,
 *    N fastest, so the blocks resident on one XCD share A row panels and W column panels in
 *    that XCD's private 4 MiB L2.
 */
#include "kernelHandler.h"
#include "vit_kernels.h"
#include "fp32_split.h"
#include "gemm_common.h"

#include <cstdint>
#include <cstdlib>

namespace {

constexpr int BK = 32;   /* 32-bit words per LDS row: 32 fp32 or 64 bf16 K elements per K step */


enum { A_ROWS = 0, A_PATCH = 1 };
enum { EPI_NONE = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_PATCH = 3 };
enum { K_F32 = 0 };   /* element kind of GEMM operands / output */
#define SGB_M 1   /* scheduled split loop: SGB_M MFMAs, then SGB_V VALU instructions, repeated (measured best of 1:1 .. 3:6) */
#define SGB_V 2

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

/* Block-tile configuration: BM x BN outputs, WM x WN waves. */
template <int BM_, int BN_, int WM_, int WN_>
struct Tile {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
    static constexpr int NW = WM * WN;               /* waves per workgroup */
    static constexpr int NT = 64 * NW;               /* threads */
    static constexpr int IT = BM / WM / 32;          /* 32-row MFMA tiles per wave */
    static constexpr int JT = BN / WN / 32;          /* 32-col MFMA tiles per wave */
    static constexpr int CHA = BM / 8 / NW;          /* 8-row DMA pieces per wave, A */
    static constexpr int CHW = BN / 8 / NW;          /* 8-row DMA pieces per wave, W */
    static constexpr int STAGE_F = (BM + BN) * BK;   /* floats per LDS stage */
    static constexpr size_t LDS = sizeof(float) * 2 * STAGE_F;
    /* W as NPL pre-split 16-bit planes (3 x bf16 or 2 x fp16): BN rows x 64 B per plane instead of
     * BN x 128 B of fp32 */
    static constexpr int stage_f(int npl) { return npl ? BM * BK + npl * BN * 16 : STAGE_F; }
    static constexpr size_t lds(int npl) { return sizeof(float) * 2 * stage_f(npl); }
    static constexpr int chw(int npl) { return npl ? npl * BN / 16 / NW : CHW; }   /* W DMA pieces per wave */
    static constexpr int WG_PER_CU = (2 * LDS <= 160 * 1024) ? 2 : 1;
    static constexpr int MIN_WAVES_PER_SIMD = WG_PER_CU * NW / 4;
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "DMA pieces must divide evenly over waves");
    static_assert(IT >= 1 && JT >= 1 && LDS <= 160 * 1024, "tile does not fit");
};

struct GemmParams {
    const void *A, *W;        /* operands: fp32 or bf16, row-major [M][K] and [N][K] */
    const float *bias, *R, *pos;
    float w_scale, inv_w_scale; /* fp16 weight parts: the power of two the weights were multiplied by, and its inverse */
    void *C;                  /* output: fp32 or bf16 */
    int M, N, K;
    int mtiles, ntiles;
    /* patch-embed geometry (A_PATCH / EPI_PATCH only) */
    int img, patch, chans, grid, tokens;
};

/* The native fp32 matrix instruction (v_mfma_f32_32x32x2_f32: exact fp32, 1/16 of the bf16 rate): the
 * classifier head (ragged N), and every projection under VIT_HIP_GEMM_FP32=native. */
template <class T, int AMODE, int EPI, bool NGUARD>
__global__ __launch_bounds__(T::NT, T::MIN_WAVES_PER_SIMD) void gemm_f32_kernel(const GemmParams p)
{
    constexpr int BM = T::BM, BN = T::BN, IT = T::IT, JT = T::JT;
    constexpr int ES = 4;                /* operand element size */
    constexpr int KE = 128 / ES;         /* K elements per step (one 128-byte LDS row) */

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tile = xcd_tile(blockIdx.x, p.mtiles * p.ntiles);
    const int m0 = (tile / p.ntiles) * BM;
    const int n0 = (tile % p.ntiles) * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / T::WN, wn = wave % T::WN, lr = lane & 31, lh = lane >> 5;

    /* DMA piece q covers tile rows 8q..8q+7; this lane fills physical chunk (lane & 7)
     * of row 8q + (lane >> 3) with logical chunk phys ^ swizzle(row). */
    const char *a_src[T::CHA], *w_src[T::CHW];
    int a_k[T::CHA];
#pragma unroll
    for (int i = 0; i < T::CHA; ++i) {
        const int r = 8 * (wave * T::CHA + i) + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);   /* logical 16-byte chunk of the row's K step */
        a_k[i] = 4 * chunk;
        const int m = min(m0 + r, p.M - 1);
        if (AMODE == A_ROWS) {
            a_src[i] = static_cast<const char *>(p.A) + (size_t)m * p.K * ES + 16 * chunk;
        } else {
            const int np = p.grid * p.grid;
            const int b = m / np, pp = m - b * np;
            const int oh = pp / p.grid, ow = pp - oh * p.grid;
            a_src[i] = reinterpret_cast<const char *>(
                static_cast<const float *>(p.A) + ((size_t)b * p.chans * p.img + (size_t)oh * p.patch) * p.img +
                (size_t)ow * p.patch);
        }
    }
#pragma unroll
    for (int i = 0; i < T::CHW; ++i) {
        const int r = 8 * (wave * T::CHW + i) + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        int n = n0 + r;
        if (NGUARD)
            n = min(n, p.N - 1);
        w_src[i] = static_cast<const char *>(p.W) + (size_t)n * p.K * ES + 16 * chunk;
    }

    auto dma = [&](int stage, int kt) { /* K step kt: 128 bytes of every operand row */
        float *As = smem + stage * T::STAGE_F, *Ws = As + BM * BK;
#pragma unroll
        for (int i = 0; i < T::CHA; ++i) {
            const char *ap;
            if (AMODE == A_ROWS) {
                ap = a_src[i] + (size_t)kt * 128;
            } else {
                /* im2row on load: k = (ic, kh, kw); 4 consecutive kw are contiguous. */
                const int k = kt * BK + a_k[i], pp2 = p.patch * p.patch;
                const int ic = k / pp2, rem = k - ic * pp2;
                const int kh = rem / p.patch, kw = rem - kh * p.patch;
                ap = a_src[i] + (((size_t)ic * p.img + kh) * p.img + kw) * 4;
            }
            __builtin_amdgcn_global_load_lds((gptr_t)ap, (lptr_t)(As + (wave * T::CHA + i) * 8 * BK), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < T::CHW; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(w_src[i] + (size_t)kt * 128),
                                             (lptr_t)(Ws + (wave * T::CHW + i) * 8 * BK), 16, 0, 0);
    };

    /* Accumulators start at the bias (column = lane & 31 of each 32-wide tile), like
     * the scalar loop (`sum = bias[o]`, ViT_seq.c:301). */
    f32x16 acc[IT][JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        int col = n0 + wn * (32 * JT) + j * 32 + lr;
        if (NGUARD)
            col = min(col, p.N - 1);
        const float bv = p.bias[col];
#pragma unroll
        for (int i = 0; i < IT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = bv;
    }

    /* Fragment rows wm*32*IT + i*32 + lr (and the W analogue) all have
     * (row >> 1) & 7 == (lr >> 1) & 7, so the read swizzle is one per-lane constant.
     * One ds_read_b128 gives k = 8*kk + 4*lh + e in element e; MFMA step e contracts
     * the pair {e, 4 + e} of that 8-wide k group (same permutation for A and W). */
    const int swz = (lr >> 1) & 7;
    int koff[BK / 8];
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk)
        koff[kk] = 4 * ((2 * kk + lh) ^ swz);

    auto compute_kk = [&](const float *a_base, const float *w_base, int kk) {
        f32x4 a[IT], b[JT];
#pragma unroll
        for (int i = 0; i < IT; ++i)
            a[i] = *reinterpret_cast<const f32x4 *>(a_base + i * 32 * BK + koff[kk]);
#pragma unroll
        for (int j = 0; j < JT; ++j)
            b[j] = *reinterpret_cast<const f32x4 *>(w_base + j * 32 * BK + koff[kk]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < IT; ++i)
#pragma unroll
                for (int j = 0; j < JT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    };

    const int nk = p.K / KE;
    dma(0, 0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk)
            dma(cur ^ 1, kt + 1);
        const float *As = smem + cur * T::STAGE_F, *Ws = As + BM * BK;
        const float *a_base = As + (wm * 32 * IT + lr) * BK;
        const float *w_base = Ws + (wn * 32 * JT + lr) * BK;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk)
            compute_kk(a_base, w_base, kk);
        __syncthreads();
    }

    /* Epilogue.  C/D map of the 32x32 tile: col = lane & 31,
     * row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5). */
#pragma unroll
    for (int i = 0; i < IT; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 * IT + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row >= p.M)
                continue;
            size_t orow = (size_t)row;
            const float *posrow = nullptr;
            if (EPI == EPI_PATCH) {
                const int np = p.grid * p.grid;
                const int b = row / np, pp = row - b * np;
                orow = (size_t)b * p.tokens + 1 + pp;
                posrow = p.pos + (size_t)(1 + pp) * p.N;
            }
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                const int col = n0 + wn * 32 * JT + j * 32 + lr;
                if (NGUARD && col >= p.N)
                    continue;
                float v = acc[i][j][r];
                if (EPI == EPI_GELU)
                    v = gelu_exact(v);
                if (EPI == EPI_RESID)
                    v = p.R[orow * p.N + col] + v;
                if (EPI == EPI_PATCH)
                    v = v + posrow[col];
                static_cast<float *>(p.C)[orow * p.N + col] = v;
            }
        }
    }
}

/* Operand staging shared by the 16x16x32 kernel below: per-lane source addresses of the
 * 8-row LDS-DMA pieces (same LDS image as gemm_f32_kernel: 128-byte rows, chunk c of row r
 * at c ^ ((r >> 1) & 7)). */
template <class T, int AMODE, int ES, int NPL = 0>
struct Staging {
    static constexpr int NWP = T::chw(NPL);               /* W pieces per wave */
    static constexpr int STAGE = T::stage_f(NPL);
    const char *a_src[T::CHA], *w_src[NWP];
    int a_k[T::CHA];

    __device__ __forceinline__ void init(const GemmParams &p, int m0, int n0, int wave, int lane)
    {
#pragma unroll
        for (int i = 0; i < T::CHA; ++i) {
            const int r = 8 * (wave * T::CHA + i) + (lane >> 3);
            const int chunk = (lane & 7) ^ ((r >> 1) & 7);
            a_k[i] = 4 * chunk;
            const int m = min(m0 + r, p.M - 1);
            if (AMODE == A_ROWS) {
                a_src[i] = static_cast<const char *>(p.A) + (size_t)m * p.K * ES + 16 * chunk;
            } else {
                const int np = p.grid * p.grid;
                const int b = m / np, pp = m - b * np;
                const int oh = pp / p.grid, ow = pp - oh * p.grid;
                a_src[i] = reinterpret_cast<const char *>(
                    static_cast<const float *>(p.A) + ((size_t)b * p.chans * p.img + (size_t)oh * p.patch) * p.img +
                    (size_t)ow * p.patch);
            }
        }
        if constexpr (NPL != 0) {
            /* p.W = [K/32][NPL][N][32] 16-bit (K step, plane, row): the 64 bytes a row contributes to
             * one K step sit next to its neighbours', so a tile's W read of a step is NPL runs of
             * BN*64 contiguous bytes -- whole cache lines.  (Plane-major [3][N][K] reads half a
             * line per row and step and doubled the traffic beyond L2: measured.)
             * Piece pc = 16 rows x 64 B of one plane; lane fills
             * physical 16-byte chunk (lane & 3) of row (lane >> 2) with logical chunk
             * phys ^ ((row >> 2) & 3) (64-byte rows: the 16 rows of a fragment read then hit
             * 16 distinct 16-byte slots of the bank row). */
#pragma unroll
            for (int i = 0; i < NWP; ++i) {
                const int pc = wave * NWP + i, plane = pc / (T::BN / 16), rb = pc - plane * (T::BN / 16);
                const int r = 16 * rb + (lane >> 2);
                const int chunk = (lane & 3) ^ ((lane >> 4) & 3);
                w_src[i] = static_cast<const char *>(p.W) + ((size_t)plane * p.N + n0 + r) * 64 + 16 * chunk;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NWP; ++i) {
                const int r = 8 * (wave * NWP + i) + (lane >> 3);
                const int chunk = (lane & 7) ^ ((r >> 1) & 7);
                w_src[i] = static_cast<const char *>(p.W) + (size_t)(n0 + r) * p.K * ES + 16 * chunk;
            }
        }
    }

    __device__ __forceinline__ void dma(const GemmParams &p, float *smem, int stage, int kt, int wave) const
    {
        float *As = smem + stage * STAGE, *Ws = As + T::BM * BK;
#pragma unroll
        for (int i = 0; i < T::CHA; ++i) {
            const char *ap;
            if (AMODE == A_ROWS) {
                ap = a_src[i] + (size_t)kt * 128;
            } else {
                const int k = kt * BK + a_k[i], pp2 = p.patch * p.patch;
                const int ic = k / pp2, rem = k - ic * pp2;
                const int kh = rem / p.patch, kw = rem - kh * p.patch;
                ap = a_src[i] + (((size_t)ic * p.img + kh) * p.img + kw) * 4;
            }
            __builtin_amdgcn_global_load_lds((gptr_t)ap, (lptr_t)(As + (wave * T::CHA + i) * 8 * BK), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NWP; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(w_src[i] + (size_t)kt * (NPL ? (size_t)64 * NPL * p.N : (size_t)128)),
                                             (lptr_t)(Ws + (wave * NWP + i) * 8 * BK), 16, 0, 0);
    }
};

/* The same GEMM on v_mfma_f32_16x16x32_bf16 (SPLIT3 for fp32 operands, or bf16 operands).
 * Under an MFMA-dense load the chip holds a higher clock on the 16x16x32 shape than on
 * 32x32x16 (MI355X_MICROARCH.md, DVFS give-back item 7), so cycles per FLOP do not decide.
 * The product is formed transposed, D[n][m] = sum_k W[n][k] A[m][k] (W fragment as the
 * MFMA's A operand), so that a lane ends up with four CONSECUTIVE output columns of one
 * row: lane l holds out[m = i*16 + (l & 15)][n = j*16 + 4*(l >> 4) + r], r = 0..3, and the
 * bias, residual, position-embedding reads and the store are one 16-byte access each.
 * Operand fragment: lane l holds k = 8*(l >> 4) .. +7 of row (l & 15), natural k order. */
template <class T, int AMODE, int EPI, int INK, int OUTK, bool SCHED = false, int NPL = 0>
__global__ __launch_bounds__(T::NT, T::MIN_WAVES_PER_SIMD) void gemm_mf16_kernel(const GemmParams p)
{
    constexpr int BM = T::BM, BN = T::BN;
    constexpr int IT = BM / T::WM / 16, JT = BN / T::WN / 16;   /* 16x16 blocks per wave */
    constexpr int IC = IT < 4 ? IT : 4;                          /* A fragments split at a time */
    constexpr int ES = 4;
    constexpr int KE = 128 / ES;
        static_assert(IT % IC == 0, "row blocks per wave");

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tile = xcd_tile(blockIdx.x, p.mtiles * p.ntiles);
    const int m0 = (tile / p.ntiles) * BM;
    const int n0 = (tile % p.ntiles) * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / T::WN, wn = wave % T::WN, l15 = lane & 15, q = lane >> 4;

    static_assert(NPL == 0 || (SCHED && INK == K_F32 && IT < JT), "pre-split weight planes: scheduled fp32 loop, wave tile wider along N");
    constexpr int STG = T::stage_f(NPL);   /* floats per LDS stage */
    Staging<T, AMODE, ES, NPL> stg;
    stg.init(p, m0, n0, wave, lane);

    f32x4 acc[IT][JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        f32x4 bv = *reinterpret_cast<const f32x4 *>(p.bias + n0 + wn * 16 * JT + j * 16 + 4 * q);
        if (NPL == 2)   /* fp16 weight parts carry a power-of-two scale: exact, undone in the epilogue */
            bv = bv * p.w_scale;
#pragma unroll
        for (int i = 0; i < IT; ++i)
            acc[i][j] = bv;
    }

    /* rows i*16 + l15 (+ multiples of 16) all have (row >> 1) & 7 == l15 >> 1 */
    const int swz = l15 >> 1;
    const float *a_lane = smem + (wm * 16 * IT + l15) * BK;
    const float *w_lane = smem + BM * BK + (wn * 16 * JT + l15) * BK;


    auto compute = [&](int stage) {
        const float *ab = a_lane + stage * STG, *wb = w_lane + stage * STG;
        if constexpr (SCHED) {
            /* Fragment-grained pipeline inside the K step: group i issues the LDS reads of A
             * fragment i+2, splits fragment i+1 and runs the 6*JT MFMAs of fragment i, the
             * MFMA / VALU interleave pinned with sched_group_barrier (left alone, the compiler
             * emits the splits of several fragments, then their MFMAs, and the matrix pipe
             * idles during the former).  The W splits and the first A split stay exposed
             * (hiding them too -- a per-W-fragment ramp, or carrying the next step's W across
             * the barrier -- measured slower). */
            static_assert(IT >= 2, "pipeline depth");
            const int k0 = 4 * ((2 * q) ^ swz), k1 = 4 * ((2 * q + 1) ^ swz);
            bf16x8 w0[JT], w1[JT], w2[JT], c0, c1, c2;
            f32x4 ra[2][2];
            ra[0][0] = *reinterpret_cast<const f32x4 *>(ab + k0);
            ra[0][1] = *reinterpret_cast<const f32x4 *>(ab + k1);
            ra[1][0] = *reinterpret_cast<const f32x4 *>(ab + 16 * BK + k0);
            ra[1][1] = *reinterpret_cast<const f32x4 *>(ab + 16 * BK + k1);
#pragma unroll
            for (int j = 0; j < JT; ++j)
                split8(*reinterpret_cast<const f32x4 *>(wb + j * 16 * BK + k0),
                       *reinterpret_cast<const f32x4 *>(wb + j * 16 * BK + k1), w0[j], w1[j], w2[j]);
            split8(ra[0][0], ra[0][1], c0, c1, c2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < IT; ++i) {
                bf16x8 n0, n1, n2;
                if (i + 1 < IT)
                    split8(ra[(i + 1) & 1][0], ra[(i + 1) & 1][1], n0, n1, n2);
                if (i + 2 < IT) {
                    ra[i & 1][0] = *reinterpret_cast<const f32x4 *>(ab + (i + 2) * 16 * BK + k0);
                    ra[i & 1][1] = *reinterpret_cast<const f32x4 *>(ab + (i + 2) * 16 * BK + k1);
                }
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int j = 0; j < JT; ++j) { /* per accumulator: smallest terms first */
                        const bf16x8 wp = (t == 0 || t == 3 || t == 5) ? w0[j] : (t == 1) ? w2[j] : w1[j];
                        const bf16x8 ap = (t == 0) ? c2 : (t == 2 || t == 3) ? c1 : c0;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp, ap, acc[i][j], 0, 0, 0);
                    }
                if (i + 1 < IT) {
                    if (i + 2 < IT)
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    constexpr int VPM = JT >= 8 ? 1 : SGB_V;   /* ~44 split instructions over the group's MFMAs */
#pragma unroll
                    for (int r = 0; r < (6 * JT - 2) / SGB_M; ++r) {
                        __builtin_amdgcn_sched_group_barrier(0x008, SGB_M, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 6 * JT - (6 * JT - 2) / SGB_M * SGB_M, 0);
                    c0 = n0;
                    c1 = n1;
                    c2 = n2;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            static_assert(SCHED, "fp32 operands use the scheduled loop");
        }
    };

    const int nk = p.K / KE;
    if constexpr (NPL != 0 && (IT < JT)) {
        /* Wave tile wider along N than along M (e.g. 64 x 128): with the weights pre-split only the A
         * fragments cost VALU work, and a wave with FEWER A fragments repeats less of it (an A fragment
         * is split by every wave of its tile row).  All IT A fragments stay resident as parts; W is
         * taken in NC chunks of JC fragments:
         *   chunk 0:    group i = MFMAs(A_i, W chunk 0) || split A_{i+1}; its last group refills W in
         *               place with chunk 1 -- after chunk 0 every A read of tile t is done
         *   chunks >= 1: group i = MFMAs(A_i, W chunk c), no VALU of their own; the last group of each
         *               refills W in place with the next chunk
         *   barrier before the last chunk's last group (every LDS read of tile t is done by then;
         *   tile t+1 has landed), DMA of tile t+2; under the last chunk the next tile's A_0 is read
         *   and split into A_0's registers (dead after that chunk's group 0). */
        constexpr int JC = NPL == 3 ? 2 : 4, NC = JT / JC;
        static_assert(IT >= 2 && JT % JC == 0 && NC >= 2, "W chunks");
        typedef typename PartT<NPL>::type frag_t;
        constexpr int NT6 = NPL == 3 ? 6 : 3;
        const int k0 = 4 * ((2 * q) ^ swz), k1 = 4 * ((2 * q + 1) ^ swz);
        const int wpo = BM * BK + (wn * 16 * JT + l15) * 16 + 4 * (q ^ ((l15 >> 2) & 3));
        const int apo = (wm * 16 * IT + l15) * BK;
        frag_t w[JC][NPL], a[IT][NPL];
        f32x4 ra[2];
        auto read_w = [&](const float *base, int chunk, int j) {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
                w[j][pl] = __builtin_bit_cast(frag_t, *reinterpret_cast<const f32x4 *>(base + pl * BN * 16 + (chunk * JC + j) * 256));
        };
        auto read_a = [&](const float *ab, int i) {
            ra[0] = *reinterpret_cast<const f32x4 *>(ab + i * 16 * BK + k0);
            ra[1] = *reinterpret_cast<const f32x4 *>(ab + i * 16 * BK + k1);
        };
        auto mfma_group = [&](int i, int chunk) { /* per accumulator: smallest terms first */
#pragma unroll
            for (int t = 0; t < NT6; ++t)
#pragma unroll
                for (int j = 0; j < JC; ++j)
                    acc[i][chunk * JC + j] = mfma_part(w[j][term_w<NPL>(t)], a[i][term_a<NPL>(t)], acc[i][chunk * JC + j]);
        };
        auto last_group_refill = [&](int chunk, const float *wnext, int next_chunk, bool refill) {
#pragma unroll
            for (int j = 0; j < JC; ++j) {
                f32x4 cc = acc[IT - 1][chunk * JC + j];
#pragma unroll
                for (int t = 0; t < NT6; ++t)
                    cc = mfma_part(w[j][term_w<NPL>(t)], a[IT - 1][term_a<NPL>(t)], cc);
                acc[IT - 1][chunk * JC + j] = cc;
                if (refill)
                    read_w(wnext, next_chunk, j);   /* in place: this W fragment is dead from here on */
            }
        };
        auto interleave = [&]() {
#pragma unroll
            for (int r = 0; r < NT6 * JC - 2; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NPL == 3 ? 4 : 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        };

        stg.dma(p, smem, 0, 0, wave);
        __syncthreads();
        if (nk > 1)
            stg.dma(p, smem, 1, 1, wave);
#pragma unroll
        for (int j = 0; j < JC; ++j)
            read_w(smem + wpo, 0, j);
        read_a(smem + apo, 0);
        split_parts(ra[0], ra[1], a[0]);
        read_a(smem + apo, 1);
        __builtin_amdgcn_sched_barrier(0);

        for (int kt = 0; kt < nk; ++kt) {
            const float *cur = smem + (kt & 1) * STG, *nxt = smem + ((kt + 1) & 1) * STG;
            const bool more = kt + 1 < nk;
            /* chunk 0: the A splits */
#pragma unroll
            for (int i = 0; i < IT - 1; ++i) {
                split_parts(ra[0], ra[1], a[i + 1]);
                if (i + 2 < IT)
                    read_a(cur + apo, i + 2);
                mfma_group(i, 0);
                interleave();
                __builtin_amdgcn_sched_barrier(0);
            }
            last_group_refill(0, cur + wpo, 1, true);
            __builtin_amdgcn_sched_barrier(0);
            /* middle chunks */
#pragma unroll
            for (int c = 1; c < NC - 1; ++c) {
#pragma unroll
                for (int i = 0; i < IT - 1; ++i)
                    mfma_group(i, c);
                last_group_refill(c, cur + wpo, c + 1, true);
                __builtin_amdgcn_sched_barrier(0);
            }
            /* last chunk: W chunk NC-1 is in registers, so tile kt is read completely */
            __syncthreads();                  /* tile kt read by every wave; tile kt+1 has landed */
            if (kt + 2 < nk)
                stg.dma(p, smem, kt & 1, kt + 2, wave);
            if (more)
                read_a(nxt + apo, 0);
            mfma_group(0, NC - 1);
            __builtin_amdgcn_sched_barrier(0);
            if (more)
                split_parts(ra[0], ra[1], a[0]);          /* next tile's A_0: a[0] is dead */
#pragma unroll
            for (int i = 1; i < IT - 1; ++i)
                mfma_group(i, NC - 1);
            if (more) {
                interleave();
                read_a(nxt + apo, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            last_group_refill(NC - 1, nxt + wpo, 0, more);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        stg.dma(p, smem, 0, 0, wave);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk)
                stg.dma(p, smem, cur ^ 1, kt + 1, wave);
            compute(cur);
            __syncthreads();
        }
    }

#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int row = m0 + wm * 16 * IT + i * 16 + l15;
        if (row >= p.M)
            continue;
        size_t orow = (size_t)row;
        const float *posrow = nullptr;
        if (EPI == EPI_PATCH) {
            const int np = p.grid * p.grid;
            const int b = row / np, pp = row - b * np;
            orow = (size_t)b * p.tokens + 1 + pp;
            posrow = p.pos + (size_t)(1 + pp) * p.N;
        }
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            const int col = n0 + wn * 16 * JT + j * 16 + 4 * q;
            f32x4 v = acc[i][j];
            if (NPL == 2)
                v = v * p.inv_w_scale;
            if (EPI == EPI_GELU) {
                const f32x2 lo = gelu_exact2(f32x2{v[0], v[1]}), hi = gelu_exact2(f32x2{v[2], v[3]});
                v = f32x4{lo[0], lo[1], hi[0], hi[1]};
            }
            if (EPI == EPI_RESID)
                v = *reinterpret_cast<const f32x4 *>(p.R + orow * p.N + col) + v;
            if (EPI == EPI_PATCH)
                v = v + *reinterpret_cast<const f32x4 *>(posrow + col);
            *reinterpret_cast<f32x4 *>(static_cast<float *>(p.C) + orow * p.N + col) = v;
        }
    }
}

template <class T, int AMODE, int EPI, int INK, int OUTK, bool SCHED = false, int NPL = 0>
int launch_mf16(hipStream_t st, GemmParams p)
{
    VH_SET_LDS_ONCE((gemm_mf16_kernel<T, AMODE, EPI, INK, OUTK, SCHED, NPL>), T::lds(NPL));
    p.mtiles = (p.M + T::BM - 1) / T::BM;
    p.ntiles = p.N / T::BN;
    hipLaunchKernelGGL((gemm_mf16_kernel<T, AMODE, EPI, INK, OUTK, SCHED, NPL>), dim3(p.mtiles * p.ntiles),
                       dim3(T::NT), T::lds(NPL), st, p);
    VH_LAUNCH_CHECK("gemm_mf16_kernel");
    return 0;
}

/* Token 0 of every image: class token + pos_embed[0] (ViT_seq.c:90-93,114-117). */
__global__ void cls_rows_kernel(const float *cls, const float *pos, float *tokens, int n_images,
                                int tokens_per_image, int E)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_images * E)
        return;
    const int b = idx / E, e = idx - b * E;
    tokens[(size_t)b * tokens_per_image * E + e] = cls[e] + pos[e];
}

template <class T, int AMODE, int EPI, bool NGUARD>
int launch_tile(hipStream_t st, GemmParams p)
{
    VH_SET_LDS_ONCE((gemm_f32_kernel<T, AMODE, EPI, NGUARD>), T::LDS);
    p.mtiles = (p.M + T::BM - 1) / T::BM;
    p.ntiles = (p.N + T::BN - 1) / T::BN;
    hipLaunchKernelGGL((gemm_f32_kernel<T, AMODE, EPI, NGUARD>),
                       dim3(p.mtiles * p.ntiles), dim3(T::NT), T::LDS, st, p);
    VH_LAUNCH_CHECK("gemm_f32_kernel");
    return 0;
}

/* Tile configurations (every configuration computes the same k order, so results are identical). */
using Tile0 = Tile<128, 128, 2, 4>; /*  8 waves of 64x32, 2 workgroups per CU: ragged N and small shapes (native fp32) */
using Tile1 = Tile<128, 128, 2, 2>; /*  4 waves of 64x64, 2 workgroups per CU */
using Tile3 = Tile<256, 256, 2, 4>; /*  8 waves of 128x64 */
using Tile4 = Tile<256, 256, 4, 4>; /* 16 waves of 64x64: native fp32 */
using Tile8 = Tile<256, 256, 4, 2>; /*  8 waves of 64x128 (pre-split weights: fewer A splits per wave) */
using Tile9 = Tile<128, 128, 4, 1>; /*  4 waves of 32x128, 2 workgroups per CU */
using TileS = Tile<32, 64, 1, 2>;   /*  2 waves of 32x32: skinny problems (the classifier at batch <= 512) */

/* fp32 products: the exact 3-way bf16 split on the bf16 cores (default), or the native
 * fp32 MFMA (VIT_HIP_GEMM_FP32=native).  Both give fp32-level results (same measured
 * logit parity); the split is ~1.4x faster end to end. */
bool use_split3()
{
    static int v = -1;
    if (v < 0) {
        const char *env = getenv("VIT_HIP_GEMM_FP32");
        v = (env && env[0] == 'n') ? 0 : 1;
    }
    return v == 1;
}

bool aligned16(const GemmParams &p)
{
    return (((uintptr_t)p.C | (uintptr_t)p.bias | (uintptr_t)p.R | (uintptr_t)p.pos) & 15) == 0;
}

/* fp32 operands, both split inside the K loop (vh_launch_linear on raw fp32 weights, and the patch
 * embedding): 256x256 tiles where N allows and there is enough work, else 128x128. */
template <int AMODE, int EPI>
int launch(hipStream_t st, const GemmParams &p)
{
    const bool big = p.N % 256 == 0 && p.M >= 4096 && !(EPI == EPI_RESID && p.K < 2048);
    if (p.N % 128 != 0) {                    /* ragged N (the classifier): the guarded tiles on the native fp32 MFMA */
        /* a skinny problem (M = batch, N = 1000: 32 tiles of 128x128 on 256 CUs, 53 us) goes to 32x64 tiles,
         * one 32x32 block per wave -- 256 workgroups for 512 x 1000, same k order, same bits */
        const long tiles128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
        if (2 * tiles128 < vh_device_cus(vh_current_device()))
            return launch_tile<TileS, AMODE, EPI, true>(st, p);
        return launch_tile<Tile0, AMODE, EPI, true>(st, p);
    }
    if (use_split3() && aligned16(p))
        return big ? launch_mf16<Tile3, AMODE, EPI, K_F32, K_F32, true>(st, p)
                   : launch_mf16<Tile1, AMODE, EPI, K_F32, K_F32, true>(st, p);
    return big ? launch_tile<Tile4, AMODE, EPI, false>(st, p) : launch_tile<Tile0, AMODE, EPI, false>(st, p);
}

} // namespace

extern "C" int vh_launch_linear(vh_stream_t s, float *output, const float *weight,
                                const float *input, const float *bias, int rowA, int colA,
                                int colB, int doGelu, const float *residual)
{
    if (!output || !weight || !input || !bias)
        return vh_fail(1, "vh_launch_linear: null pointer argument");
    if (rowA <= 0 || colA <= 0 || colB <= 0)
        return vh_fail(1, "vh_launch_linear: non-positive dimension (%d,%d,%d)", rowA, colA, colB);
    if (colA % BK != 0)
        return vh_fail(1, "vh_launch_linear: colA=%d must be a multiple of %d", colA, BK);
    if (doGelu && residual)
        return vh_fail(1, "vh_launch_linear: GELU and residual together are not a model op");

    GemmParams p = {};
    p.A = input; p.W = weight; p.bias = bias; p.R = residual; p.C = output;
    p.M = rowA; p.N = colB; p.K = colA;
    hipStream_t st = (hipStream_t)s;
    if (doGelu)
        return launch<A_ROWS, EPI_GELU>(st, p);
    if (residual)
        return launch<A_ROWS, EPI_RESID>(st, p);
    return launch<A_ROWS, EPI_NONE>(st, p);
}

namespace {

__device__ __forceinline__ float block_max_256(float v, float *red)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        v = fmaxf(v, __shfl_xor(v, m));
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = v;
    __syncthreads();
    v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    return v;
}

/* max |x| over a tensor into *amax (non-negative floats order like their bit patterns). */
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ in, size_t n, float *__restrict__ amax)
{
    __shared__ float red[4];
    float mx = 0.0f;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i + 3 < n; i += (size_t)gridDim.x * 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(in + i);
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3))
        mx = fmaxf(mx, fabsf(in[n - 1 - threadIdx.x]));
    mx = block_max_256(mx, red);
    if (threadIdx.x == 0 && mx == mx) /* NaNs are not a maximum */
        atomicMax(reinterpret_cast<unsigned *>(amax), __builtin_bit_cast(unsigned, mx));
}

} // namespace

namespace {

/* fp32 [n] -> three bf16 planes [3][n]: x = p0 + p1 + p2 exactly (the SPLIT3 parts), done once
 * for the weights so that the GEMM's inner loop splits only the activations. */
__global__ void split3_planes_kernel(const float *__restrict__ in, __bf16 *__restrict__ out, int N, int K)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * K)
        return;
    const int n = (int)(i / K), k = (int)(i - (size_t)n * K);
    const float x = in[i];
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    /* [K/32][3][N][32]: K step, plane, row, element (see Staging) */
    const size_t o = (((size_t)(k >> 5) * 3) * N + n) * 32 + (k & 31), plane = (size_t)N * 32;
    out[o] = h;
    out[o + plane] = m;
    out[o + 2 * plane] = (__bf16)r2;
}

} // namespace

namespace {

/* Pre-split-weight GEMM on the tile the shape wants, with the last, partly filled scheduling round
 * of 256x256 tiles handed to 128x128 tiles instead: with one
 * workgroup per CU a grid of r.f rounds costs ceil(r.f) rounds; the remainder rows as quarter-size
 * tiles cost about f/2. */
template <int EPI, int NPL>
int launch_planes(hipStream_t st, GemmParams p, bool prefer_small)
{
    if (!(p.N % 256 == 0 && p.M >= 4096 && !prefer_small))
        return launch_mf16<Tile9, A_ROWS, EPI, K_F32, K_F32, true, NPL>(st, p);
    const int num_cus = vh_device_cus(vh_current_device());
    const int ntiles = p.N / 256, mtiles = (p.M + 255) / 256;
    const long tiles = (long)mtiles * ntiles, full = tiles / num_cus, rem = tiles % num_cus;
    const int rows_big = (int)(full * num_cus / ntiles) * 256;
    if (full < 1 || rem == 0 || 4 * rem > 3 * num_cus || rows_big <= 0 || rows_big >= p.M)
        return launch_mf16<Tile8, A_ROWS, EPI, K_F32, K_F32, true, NPL>(st, p);
    GemmParams big = p, rest = p;
    big.M = rows_big;
    rest.M = p.M - rows_big;
    rest.A = static_cast<const float *>(p.A) + (size_t)rows_big * p.K;
    rest.C = static_cast<float *>(p.C) + (size_t)rows_big * p.N;
    if (p.R)
        rest.R = p.R + (size_t)rows_big * p.N;
    const int rc = launch_mf16<Tile8, A_ROWS, EPI, K_F32, K_F32, true, NPL>(st, big);
    return rc ? rc : launch_mf16<Tile9, A_ROWS, EPI, K_F32, K_F32, true, NPL>(st, rest);
}

} // namespace

extern "C" int vh_launch_split3_planes(vh_stream_t s, const float *weight, void *planes, int rows, int cols)
{
    if (!weight || !planes || rows <= 0 || cols <= 0 || cols % BK != 0)
        return vh_fail(1, "vh_launch_split3_planes: bad argument (cols must be a multiple of %d)", BK);
    const size_t count = (size_t)rows * cols;
    hipLaunchKernelGGL(split3_planes_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)s, weight,
                       static_cast<__bf16 *>(planes), rows, cols);
    VH_LAUNCH_CHECK("split3_planes_kernel");
    return 0;
}

namespace {

/* fp32 [N][K] -> two fp16 planes of w*scale, [K/32][2][N][32] (K step, part, row, element). */
__global__ void split2h_planes_kernel(const float *__restrict__ in, _Float16 *__restrict__ out, int N, int K, float scale)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * K)
        return;
    const int n = (int)(i / K), k = (int)(i - (size_t)n * K);
    const float x = in[i] * scale;
    const _Float16 h = (_Float16)x;
    const size_t o = (((size_t)(k >> 5) * 2) * N + n) * 32 + (k & 31);
    out[o] = h;
    out[o + (size_t)N * 32] = (_Float16)(x - (float)h);
}

} // namespace

extern "C" int vh_launch_split2h_planes(vh_stream_t s, const float *weight, void *planes, int rows, int cols, float scale)
{
    if (!weight || !planes || rows <= 0 || cols <= 0 || cols % BK != 0 || !(scale > 0.0f))
        return vh_fail(1, "vh_launch_split2h_planes: bad argument (cols must be a multiple of %d, scale > 0)", BK);
    const size_t count = (size_t)rows * cols;
    hipLaunchKernelGGL(split2h_planes_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)s, weight,
                       static_cast<_Float16 *>(planes), rows, cols, scale);
    VH_LAUNCH_CHECK("split2h_planes_kernel");
    return 0;
}

extern "C" int vh_launch_linear_h2(vh_stream_t s, float *output, const void *weight_planes, float weight_scale,
                                   const float *input, const float *bias, int rowA, int colA, int colB, int doGelu,
                                   const float *residual)
{
    if (!output || !weight_planes || !input || !bias)
        return vh_fail(1, "vh_launch_linear_h2: null pointer argument");
    if (rowA <= 0 || colA <= 0 || colB <= 0 || colA % BK != 0 || colB % 128 != 0 || !(weight_scale > 0.0f))
        return vh_fail(1, "vh_launch_linear_h2: needs colA %% 32 == 0, colB %% 128 == 0, scale > 0 (%d,%d,%d)", rowA, colA, colB);
    if (doGelu && residual)
        return vh_fail(1, "vh_launch_linear_h2: GELU and residual together are not a model op");
    GemmParams p = {};
    p.A = input; p.W = weight_planes; p.bias = bias; p.R = residual; p.C = output;
    p.M = rowA; p.N = colB; p.K = colA;
    p.w_scale = weight_scale; p.inv_w_scale = 1.0f / weight_scale;
    if (!aligned16(p) || (((uintptr_t)weight_planes | (uintptr_t)input) & 15) != 0)
        return vh_fail(1, "vh_launch_linear_h2: pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)s;
    const bool prefer_small = residual && colA < 2048;
    if (doGelu)
        return launch_planes<EPI_GELU, 2>(st, p, prefer_small);
    if (residual)
        return launch_planes<EPI_RESID, 2>(st, p, prefer_small);
    return launch_planes<EPI_NONE, 2>(st, p, prefer_small);
}

extern "C" int vh_launch_linear_w3(vh_stream_t s, float *output, const void *weight_planes, const float *input,
                                   const float *bias, int rowA, int colA, int colB, int doGelu, const float *residual)
{
    if (!output || !weight_planes || !input || !bias)
        return vh_fail(1, "vh_launch_linear_w3: null pointer argument");
    if (rowA <= 0 || colA <= 0 || colB <= 0 || colA % BK != 0 || colB % 128 != 0)
        return vh_fail(1, "vh_launch_linear_w3: needs colA %% 32 == 0 and colB %% 128 == 0 (%d,%d,%d)", rowA, colA, colB);
    if (doGelu && residual)
        return vh_fail(1, "vh_launch_linear_w3: GELU and residual together are not a model op");
    GemmParams p = {};
    p.A = input; p.W = weight_planes; p.bias = bias; p.R = residual; p.C = output;
    p.M = rowA; p.N = colB; p.K = colA;
    if (!aligned16(p) || (((uintptr_t)weight_planes | (uintptr_t)input) & 15) != 0)
        return vh_fail(1, "vh_launch_linear_w3: pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)s;
    const bool prefer_small = residual && colA < 2048;   /* measured: the N = 768, K = 768 out-projection */
    if (doGelu)
        return launch_planes<EPI_GELU, 3>(st, p, prefer_small);
    if (residual)
        return launch_planes<EPI_RESID, 3>(st, p, prefer_small);
    return launch_planes<EPI_NONE, 3>(st, p, prefer_small);
}

extern "C" int vh_launch_absmax(vh_stream_t s, const float *input, size_t count, float *amax)
{
    if (!input || !amax || count == 0)
        return vh_fail(1, "vh_launch_absmax: bad argument");
    size_t blocks = (count / 4 + 255) / 256;
    if (blocks > 2048)
        blocks = 2048;
    if (blocks == 0)
        blocks = 1;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, input, count, amax);
    VH_LAUNCH_CHECK("absmax_kernel");
    return 0;
}

namespace {

/* Patch geometries the im2row-on-load GEMM cannot take (patch % 4 != 0 or C*P*P % 32 != 0;
 * ViT-H/14: 3*14*14 = 588): the patches are gathered once into rows of Kp = roundup(K, 32)
 * floats, zero-padded, and the weights are padded alike; then it is an ordinary rows GEMM. */
__global__ void im2row_pad_kernel(const float *__restrict__ images, float *__restrict__ rows, int n_rows, int chans,
                                  int img, int patch, int grid, int K, int Kp)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n_rows * Kp)
        return;
    const int m = (int)(idx / Kp), k = (int)(idx - (size_t)m * Kp);
    float v = 0.0f;
    if (k < K) {
        const int np = grid * grid, b = m / np, pp = m - b * np, oh = pp / grid, ow = pp - oh * grid;
        const int pp2 = patch * patch, ic = k / pp2, rem = k - ic * pp2, kh = rem / patch, kw = rem - kh * patch;
        v = images[(((size_t)b * chans + ic) * img + (size_t)oh * patch + kh) * img + (size_t)ow * patch + kw];
    }
    rows[idx] = v;
}

__global__ void pad_rows_kernel(const float *__restrict__ in, float *__restrict__ out, int n_rows, int K, int Kp)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n_rows * Kp)
        return;
    const int r = (int)(idx / Kp), k = (int)(idx - (size_t)r * Kp);
    out[idx] = k < K ? in[(size_t)r * K + k] : 0.0f;
}

bool patch_direct(int in_chans, int img_size, int patch_size)
{
    return patch_size % 4 == 0 && (in_chans * patch_size * patch_size) % BK == 0 && img_size % 4 == 0;
}

} // namespace

int vh_cls_rows(hipStream_t st, const float *cls_token, const float *pos_embed, float *tokens, int n_images,
                int tokens_per_image, int embed_dim)
{
    const int total = n_images * embed_dim;
    hipLaunchKernelGGL(cls_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, st, cls_token, pos_embed, tokens, n_images,
                       tokens_per_image, embed_dim);
    VH_LAUNCH_CHECK("cls_rows_kernel");
    return 0;
}

extern "C" size_t vh_patch_embed_workspace(int n_images, int in_chans, int img_size, int patch_size, int embed_dim)
{
    if (n_images <= 0 || in_chans <= 0 || img_size <= 0 || patch_size <= 0 || embed_dim <= 0 ||
        img_size % patch_size != 0 || patch_direct(in_chans, img_size, patch_size))
        return 0;
    const size_t grid = img_size / patch_size, K = (size_t)in_chans * patch_size * patch_size;
    const size_t Kp = (K + BK - 1) / BK * BK;
    return ((size_t)n_images * grid * grid + (size_t)embed_dim) * Kp * sizeof(float);
}

extern "C" int vh_launch_patch_embed_ws(vh_stream_t s, const float *images, const float *conv_w,
                                        const float *conv_b, const float *cls_token,
                                        const float *pos_embed, float *tokens, int n_images,
                                        int in_chans, int img_size, int patch_size, int embed_dim,
                                        void *workspace, size_t workspace_bytes)
{
    if (!images || !conv_w || !conv_b || !cls_token || !pos_embed || !tokens)
        return vh_fail(1, "vh_launch_patch_embed: null pointer argument");
    if (n_images <= 0 || in_chans <= 0 || img_size <= 0 || patch_size <= 0 || embed_dim <= 0 ||
        img_size % patch_size != 0)
        return vh_fail(1, "vh_launch_patch_embed: bad geometry");
    const int K = in_chans * patch_size * patch_size;
    const int grid = img_size / patch_size;
    const bool direct = patch_direct(in_chans, img_size, patch_size);
    const size_t need = vh_patch_embed_workspace(n_images, in_chans, img_size, patch_size, embed_dim);
    if (!direct && (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15)))
        return vh_fail(1, "vh_launch_patch_embed: patch=%d (K=%d) is not a multiple of 4 / %d: this geometry needs "
                          "vh_launch_patch_embed_ws with %zu bytes of workspace", patch_size, K, BK, need);

    GemmParams p = {};
    p.A = images; p.W = conv_w; p.bias = conv_b; p.pos = pos_embed; p.C = tokens;
    p.M = n_images * grid * grid; p.N = embed_dim; p.K = K;
    p.img = img_size; p.patch = patch_size; p.chans = in_chans; p.grid = grid;
    p.tokens = grid * grid + 1;
    hipStream_t st = (hipStream_t)s;

    if (int rc = vh_cls_rows(st, cls_token, pos_embed, tokens, n_images, p.tokens, embed_dim))
        return rc;
    if (direct)
        return launch<A_PATCH, EPI_PATCH>(st, p);

    const int Kp = (K + BK - 1) / BK * BK;
    float *rows = static_cast<float *>(workspace), *wpad = rows + (size_t)p.M * Kp;
    const size_t n1 = (size_t)p.M * Kp, n2 = (size_t)embed_dim * Kp;
    hipLaunchKernelGGL(im2row_pad_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, st, images, rows, p.M,
                       in_chans, img_size, patch_size, grid, K, Kp);
    VH_LAUNCH_CHECK("im2row_pad_kernel");
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, conv_w, wpad,
                       embed_dim, K, Kp);
    VH_LAUNCH_CHECK("pad_rows_kernel");
    p.A = rows; p.W = wpad; p.K = Kp;
    return launch<A_ROWS, EPI_PATCH>(st, p);
}

extern "C" int vh_launch_patch_embed(vh_stream_t s, const float *images, const float *conv_w,
                                     const float *conv_b, const float *cls_token,
                                     const float *pos_embed, float *tokens, int n_images,
                                     int in_chans, int img_size, int patch_size, int embed_dim)
{
    return vh_launch_patch_embed_ws(s, images, conv_w, conv_b, cls_token, pos_embed, tokens, n_images, in_chans,
                                    img_size, patch_size, embed_dim, nullptr, 0);
}
