/*
 * gemm_f32.hip -- the dense projections of the ViT forward pass on the gfx950
 * matrix cores, in exact fp32.
 *
 *   C[M][N] = A[M][K] . W[N][K]^T  (+ bias, + GELU | + residual | patch epilogue)
 *
 * Replaces the reference's 8x8-tile OpenCL GEMMs `linear_layer` (ll.cl:7-86)
 * and `QKV` (multihead.cl:3-63), the element-wise `encoderResidual`
 * (layer_norm.cl:55-65) and -- through the im2row A-loader and the token
 * epilogue -- `conv2d_kernel` + `postprocess` (conv2d.cl:1-80).
 *
 * Design (MI355X / CDNA4):
 *  - v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit-for-bit a k-ordered
 *    fmaf chain, 64 FLOP/clk/SIMD = 157.3 TFLOP/s chip peak.  There is no
 *    TF32-like shortcut on gfx950, so this is the fp32 roofline.
 *  - 128x128x32 block tile, 256 threads = 4 waves (2x2), each wave a 64x64
 *    output tile = 2x2 MFMA tiles (64 accumulator VGPRs).  Both operands are
 *    K-contiguous ("NT" GEMM), so A and W use the same LDS image: rows padded
 *    to 36 floats (144 B), which makes every ds_read_b128 fragment read
 *    conflict-free (16 rows -> 16 distinct 16-B slots of the 256-B bank row).
 *  - One ds_read_b128 per 32-row fragment yields four k-pairs: lane l holds
 *    k = 8*kk + 4*(l>>5) + e in register e, and MFMA step e contracts the
 *    pair {e, 4+e}; A and W use the same permutation, so only the summation
 *    order inside an 8-wide k group differs from the scalar loop.
 *  - Global->register->LDS double buffering: the next K-tile's 8 x 16-B loads
 *    per thread are issued before the 64 MFMAs of the current tile and written
 *    to the other LDS stage afterwards; one barrier per K-tile.  At fp32 MFMA
 *    rate (4096 cycles of MFMA per wave per K-tile) this hides HBM/L2 latency
 *    with 2 blocks (8 waves) per CU.
 *  - The accumulator starts at the bias, like the scalar loop it replaces
 *    (`sum = bias[o]`, ViT_seq.c:301), and the residual is added to the
 *    finished sum (ViT_seq.c:350,362).
 *  - blockIdx -> tile map is XCD-aware: each of the 8 XCDs walks a contiguous
 *    range of tiles, N fastest, so the blocks resident on one XCD share A row
 *    panels and W column panels in that XCD's private 4 MiB L2.
 */
#include "kernelHandler.h"
#include "vit_kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDT = BK + 4;          /* padded LDS row length in floats */
constexpr int TILE_F = BM * LDT;     /* floats per operand tile per stage */
constexpr int NTHREADS = 256;
constexpr size_t LDS_BYTES = sizeof(float) * 4 * TILE_F; /* 2 stages x (A,W) */

enum { A_ROWS = 0, A_PATCH = 1 };
enum { EPI_NONE = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_PATCH = 3 };

struct GemmParams {
    const float *A, *W, *bias, *R, *pos;
    float *C;
    int M, N, K;
    int mtiles, ntiles;
    /* patch-embed geometry (A_PATCH / EPI_PATCH only) */
    int img, patch, chans, grid, tokens;
};

/* Bijective XCD remap (blocks b and b+8 share an XCD; which one is not known
 * and not needed): XCD x gets a contiguous run of tiles. */
__device__ __forceinline__ int xcd_tile(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

__device__ __forceinline__ float gelu_exact(float x)
{
    /* ViT_seq.c:285 / ll.cl:4, same association order. */
    return 0.5f * x * (1.0f + erff(x / sqrtf(2.0f)));
}

template <int AMODE, int EPI, bool NGUARD>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_f32_kernel(const GemmParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tile = xcd_tile(blockIdx.x, p.mtiles * p.ntiles);
    const int m0 = (tile / p.ntiles) * BM;
    const int n0 = (tile % p.ntiles) * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
    const int ld_row = tid >> 3, ld_kc = (tid & 7) * 4; /* staging: rows ld_row+32i, floats ld_kc..+3 */

    /* Per-thread source rows for the four staged chunks of each operand. */
    const float *a_src[4], *w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = min(m0 + ld_row + 32 * i, p.M - 1);
        if (AMODE == A_ROWS) {
            a_src[i] = p.A + (size_t)m * p.K + ld_kc;
        } else {
            const int np = p.grid * p.grid;
            const int b = m / np, pp = m - b * np;
            const int oh = pp / p.grid, ow = pp - oh * p.grid;
            a_src[i] = p.A + ((size_t)b * p.chans * p.img + (size_t)oh * p.patch) * p.img +
                       (size_t)ow * p.patch;
        }
        int n = n0 + ld_row + 32 * i;
        if (NGUARD)
            n = min(n, p.N - 1);
        w_src[i] = p.W + (size_t)n * p.K + ld_kc;
    }

    f32x4 ra[4], rw[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (AMODE == A_ROWS) {
                ra[i] = *reinterpret_cast<const f32x4 *>(a_src[i] + k0);
            } else {
                /* im2row on load: k = (ic, kh, kw); 4 consecutive kw are contiguous. */
                const int k = k0 + ld_kc, pp2 = p.patch * p.patch;
                const int ic = k / pp2, rem = k - ic * pp2;
                const int kh = rem / p.patch, kw = rem - kh * p.patch;
                ra[i] = *reinterpret_cast<const f32x4 *>(
                    a_src[i] + ((size_t)ic * p.img + kh) * p.img + kw);
            }
            rw[i] = *reinterpret_cast<const f32x4 *>(w_src[i] + k0);
        }
    };
    auto lstore = [&](int stage) {
        float *As = smem + stage * 2 * TILE_F, *Ws = As + TILE_F;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4 *>(As + (ld_row + 32 * i) * LDT + ld_kc) = ra[i];
            *reinterpret_cast<f32x4 *>(Ws + (ld_row + 32 * i) * LDT + ld_kc) = rw[i];
        }
    };

    /* Accumulators start at the bias (column = lane & 31 of each 32-wide tile). */
    f32x16 acc[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int col = n0 + wn * 64 + j * 32 + lr;
        if (NGUARD)
            col = min(col, p.N - 1);
        const float bv = p.bias[col];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = bv;
    }

    const int nk = p.K / BK;
    gload(0);
    lstore(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk)
            gload((kt + 1) * BK);

        const float *As = smem + cur * 2 * TILE_F, *Ws = As + TILE_F;
        const float *a_base = As + (wm * 64 + lr) * LDT + lh * 4;
        const float *w_base = Ws + (wn * 64 + lr) * LDT + lh * 4;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(a_base + kk * 8);
            const f32x4 a1 = *reinterpret_cast<const f32x4 *>(a_base + 32 * LDT + kk * 8);
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(w_base + kk * 8);
            const f32x4 b1 = *reinterpret_cast<const f32x4 *>(w_base + 32 * LDT + kk * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc[1][1], 0, 0, 0);
            }
        }

        if (kt + 1 < nk)
            lstore(cur ^ 1);
        __syncthreads();
    }

    /* Epilogue.  C/D map of the 32x32 tile: col = lane & 31,
     * row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5). */
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
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
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn * 64 + j * 32 + lr;
                if (NGUARD && col >= p.N)
                    continue;
                float v = acc[i][j][r];
                if (EPI == EPI_GELU)
                    v = gelu_exact(v);
                if (EPI == EPI_RESID)
                    v = p.R[orow * p.N + col] + v;
                if (EPI == EPI_PATCH)
                    v = v + posrow[col];
                p.C[orow * p.N + col] = v;
            }
        }
    }
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

template <int AMODE, int EPI, bool NGUARD>
int launch(hipStream_t st, const GemmParams &p)
{
    static bool attr_set = false; /* per instantiation; benign race (idempotent) */
    if (!attr_set) {
        VH_TRY(hipFuncSetAttribute((const void *)gemm_f32_kernel<AMODE, EPI, NGUARD>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
        attr_set = true;
    }
    const int nwg = p.mtiles * p.ntiles;
    hipLaunchKernelGGL((gemm_f32_kernel<AMODE, EPI, NGUARD>), dim3(nwg), dim3(NTHREADS), LDS_BYTES,
                       st, p);
    VH_LAUNCH_CHECK("gemm_f32_kernel");
    return 0;
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
    p.mtiles = (rowA + BM - 1) / BM;
    p.ntiles = (colB + BN - 1) / BN;
    hipStream_t st = (hipStream_t)s;
    const bool nguard = (colB % BN) != 0;
    if (doGelu)
        return nguard ? launch<A_ROWS, EPI_GELU, true>(st, p) : launch<A_ROWS, EPI_GELU, false>(st, p);
    if (residual)
        return nguard ? launch<A_ROWS, EPI_RESID, true>(st, p) : launch<A_ROWS, EPI_RESID, false>(st, p);
    return nguard ? launch<A_ROWS, EPI_NONE, true>(st, p) : launch<A_ROWS, EPI_NONE, false>(st, p);
}

extern "C" int vh_launch_patch_embed(vh_stream_t s, const float *images, const float *conv_w,
                                     const float *conv_b, const float *cls_token,
                                     const float *pos_embed, float *tokens, int n_images,
                                     int in_chans, int img_size, int patch_size, int embed_dim)
{
    if (!images || !conv_w || !conv_b || !cls_token || !pos_embed || !tokens)
        return vh_fail(1, "vh_launch_patch_embed: null pointer argument");
    if (n_images <= 0 || in_chans <= 0 || img_size <= 0 || patch_size <= 0 || embed_dim <= 0 ||
        img_size % patch_size != 0)
        return vh_fail(1, "vh_launch_patch_embed: bad geometry");
    const int K = in_chans * patch_size * patch_size;
    if (patch_size % 4 != 0 || K % BK != 0 || img_size % 4 != 0)
        return vh_fail(1, "vh_launch_patch_embed: patch=%d (K=%d) needs patch%%4==0 and K%%%d==0",
                       patch_size, K, BK);
    const int grid = img_size / patch_size;

    GemmParams p = {};
    p.A = images; p.W = conv_w; p.bias = conv_b; p.pos = pos_embed; p.C = tokens;
    p.M = n_images * grid * grid; p.N = embed_dim; p.K = K;
    p.mtiles = (p.M + BM - 1) / BM;
    p.ntiles = (p.N + BN - 1) / BN;
    p.img = img_size; p.patch = patch_size; p.chans = in_chans; p.grid = grid;
    p.tokens = grid * grid + 1;
    hipStream_t st = (hipStream_t)s;

    const int total = n_images * embed_dim;
    hipLaunchKernelGGL(cls_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, st, cls_token,
                       pos_embed, tokens, n_images, p.tokens, embed_dim);
    VH_LAUNCH_CHECK("cls_rows_kernel");
    if (embed_dim % BN != 0)
        return launch<A_PATCH, EPI_PATCH, true>(st, p);
    return launch<A_PATCH, EPI_PATCH, false>(st, p);
}
