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
 *  - Global->register->LDS double buffering, issue-early / write-late: K-tile
 *    t+2's 8 x 16-B loads per thread are issued right after K-tile t+1 has been
 *    written to the free LDS stage at the START of step t, so a load has a whole
 *    step (4096-8192 cycles of MFMA) to land and the LDS writes overlap the
 *    MFMAs; one barrier per K-tile, 2 blocks (8 waves) per CU.
 *  - The accumulator starts at the bias, like the scalar loop it replaces
 *    (`sum = bias[o]`, ViT_seq.c:301), and the residual is added to the
 *    finished sum (ViT_seq.c:350,362).
 *  - blockIdx -> tile map is XCD-aware: each of the 8 XCDs walks a contiguous
 *    range of tiles, N fastest, so the blocks resident on one XCD share A row
 *    panels and W column panels in that XCD's private 4 MiB L2.
 */
#include "kernelHandler.h"
#include "vit_kernels.h"

#include <cstdlib>

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDT = BK + 4;          /* padded LDS row length in floats */
constexpr int TILE_F = BM * LDT;     /* floats per operand tile per stage */
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

/* Branch-free fp32 erf for the fc1 epilogue.  The scalar loop calls libm erff
 * (ViT_seq.c:285); a device libm erff is two divergent branches of ~50 VALU
 * instructions each, which made the GELU epilogue a third of the fc1 kernel.  Here:
 *   |x| <  0.921875 : x + x*P(x^2)
 *   otherwise       : sign(x) * (1 - exp(-t*Q(t))), t = min(|x|, 4)
 * with P, Q least-squares fits on Chebyshev nodes (tools/fit_erf.py), both evaluated
 * unconditionally and selected.  Measured there: max |this - erf| = 8.1e-8 and
 * max |this - glibc erff| = 6.0e-8 (1 ulp at 0.5..1), glibc itself being 4.5e-8 off. */
__device__ __forceinline__ float erf_f32(float x)
{
    const float t = fminf(fabsf(x), 4.0f);
    const float s = x * x;
    float p = 8.392696327e-05f;
    p = __builtin_fmaf(p, s, -8.148506167e-04f);
    p = __builtin_fmaf(p, s, 5.201591644e-03f);
    p = __builtin_fmaf(p, s, -2.685964666e-02f);
    p = __builtin_fmaf(p, s, 1.128370017e-01f);
    p = __builtin_fmaf(p, s, -3.761263490e-01f);
    p = __builtin_fmaf(p, s, 1.283791661e-01f);
    const float small = __builtin_fmaf(p, x, x);
    float q = -9.613538623e-07f;
    q = __builtin_fmaf(q, t, 3.291785833e-05f);
    q = __builtin_fmaf(q, t, -4.882355570e-04f);
    q = __builtin_fmaf(q, t, 4.262940958e-03f);
    q = __builtin_fmaf(q, t, -2.504872903e-02f);
    q = __builtin_fmaf(q, t, 1.077397019e-01f);
    q = __builtin_fmaf(q, t, 6.342266202e-01f);
    q = __builtin_fmaf(q, t, 1.128881097e+00f);
    const float large = copysignf(1.0f - __expf(-t * q), x);
    return fabsf(x) < 0.921875f ? small : large;
}

__device__ __forceinline__ float gelu_exact(float x)
{
    /* 0.5*x*(1+erf(x/sqrt(2))), ViT_seq.c:285 / ll.cl:4; the division by sqrt(2) is a
     * multiplication by its fp32 reciprocal (<= 1 ulp on the erf argument). */
    return 0.5f * x * (1.0f + erf_f32(x * 0.70710678118654752f));
}

/* NW = waves per workgroup: 4 (2x2 waves, 64x64 each) or 8 (2x4 waves, 64x32 each).
 * Two workgroups are resident per CU either way (LDS), i.e. 2 or 4 waves per SIMD. */
template <int AMODE, int EPI, bool NGUARD, int NW>
__global__ __launch_bounds__(64 * NW, NW / 2) void gemm_f32_kernel(const GemmParams p)
{
    constexpr int NT = 64 * NW;          /* threads */
    constexpr int WN = NW / 2;           /* waves along N */
    constexpr int WCOLS = BN / WN;       /* columns per wave: 64 or 32 */
    constexpr int JT = WCOLS / 32;       /* 32-wide MFMA column tiles per wave */
    constexpr int RP = NT / 8;           /* rows staged per pass (8 x 16-B chunks per row) */
    constexpr int CH = BM / RP;          /* staged chunks per thread per operand */

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tile = xcd_tile(blockIdx.x, p.mtiles * p.ntiles);
    const int m0 = (tile / p.ntiles) * BM;
    const int n0 = (tile % p.ntiles) * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, lr = lane & 31, lh = lane >> 5;
    const int ld_row = tid >> 3, ld_kc = (tid & 7) * 4; /* staging: rows ld_row+RP*i, floats ld_kc..+3 */

    /* Per-thread source rows for the staged chunks of each operand. */
    const float *a_src[CH], *w_src[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        int m = min(m0 + ld_row + RP * i, p.M - 1);
        if (AMODE == A_ROWS) {
            a_src[i] = p.A + (size_t)m * p.K + ld_kc;
        } else {
            const int np = p.grid * p.grid;
            const int b = m / np, pp = m - b * np;
            const int oh = pp / p.grid, ow = pp - oh * p.grid;
            a_src[i] = p.A + ((size_t)b * p.chans * p.img + (size_t)oh * p.patch) * p.img +
                       (size_t)ow * p.patch;
        }
        int n = n0 + ld_row + RP * i;
        if (NGUARD)
            n = min(n, p.N - 1);
        w_src[i] = p.W + (size_t)n * p.K + ld_kc;
    }

    f32x4 ra[CH], rw[CH];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
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
        for (int i = 0; i < CH; ++i) {
            *reinterpret_cast<f32x4 *>(As + (ld_row + RP * i) * LDT + ld_kc) = ra[i];
            *reinterpret_cast<f32x4 *>(Ws + (ld_row + RP * i) * LDT + ld_kc) = rw[i];
        }
    };

    /* Accumulators start at the bias (column = lane & 31 of each 32-wide tile). */
    f32x16 acc[2][JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        int col = n0 + wn * WCOLS + j * 32 + lr;
        if (NGUARD)
            col = min(col, p.N - 1);
        const float bv = p.bias[col];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = bv;
    }

    /* K loop.  Invariant at the top of step kt: LDS stage kt&1 holds K-tile kt
     * (visible to all waves), the staging registers hold (or are receiving) K-tile
     * kt+1.  The other stage was last read in step kt-1 and every wave has passed
     * the barrier since, so K-tile kt+1 is written there right away and the loads for
     * K-tile kt+2 are re-issued into the same registers: the ds_writes and the
     * global loads then sit among this step's MFMAs instead of in an MFMA-free
     * tail, and the only exposed latency per step is barrier + first fragment read. */
    auto compute_kk = [&](const float *a_base, const float *w_base, int kk) {
        f32x4 a[2], b[JT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
            a[i] = *reinterpret_cast<const f32x4 *>(a_base + i * 32 * LDT + kk * 8);
#pragma unroll
        for (int j = 0; j < JT; ++j)
            b[j] = *reinterpret_cast<const f32x4 *>(w_base + j * 32 * LDT + kk * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < JT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    };

    const int nk = p.K / BK;
    gload(0);
    lstore(0);
    if (nk > 1)
        gload(BK);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const float *As = smem + cur * 2 * TILE_F, *Ws = As + TILE_F;
        const float *a_base = As + (wm * 64 + lr) * LDT + lh * 4;
        const float *w_base = Ws + (wn * WCOLS + lr) * LDT + lh * 4;

        compute_kk(a_base, w_base, 0);
        if (kt + 1 < nk) {
            lstore(cur ^ 1);
            if (kt + 2 < nk)
                gload((kt + 2) * BK);
        }
#pragma unroll
        for (int kk = 1; kk < BK / 8; ++kk)
            compute_kk(a_base, w_base, kk);
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
            for (int j = 0; j < JT; ++j) {
                const int col = n0 + wn * WCOLS + j * 32 + lr;
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

template <int AMODE, int EPI, bool NGUARD, int NW>
int launch_nw(hipStream_t st, const GemmParams &p)
{
    static bool attr_set = false; /* per instantiation; benign race (idempotent) */
    if (!attr_set) {
        VH_TRY(hipFuncSetAttribute((const void *)gemm_f32_kernel<AMODE, EPI, NGUARD, NW>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
        attr_set = true;
    }
    const int nwg = p.mtiles * p.ntiles;
    hipLaunchKernelGGL((gemm_f32_kernel<AMODE, EPI, NGUARD, NW>), dim3(nwg), dim3(64 * NW), LDS_BYTES,
                       st, p);
    VH_LAUNCH_CHECK("gemm_f32_kernel");
    return 0;
}

/* Waves per workgroup: 8 unless VIT_HIP_GEMM_NW=4 (tuning knob; results are identical). */
int gemm_waves()
{
    static int nw = 0;
    if (nw == 0) {
        const char *env = getenv("VIT_HIP_GEMM_NW");
        nw = (env && env[0] == '4') ? 4 : 8;
    }
    return nw;
}

template <int AMODE, int EPI, bool NGUARD>
int launch(hipStream_t st, const GemmParams &p)
{
    return gemm_waves() == 4 ? launch_nw<AMODE, EPI, NGUARD, 4>(st, p)
                             : launch_nw<AMODE, EPI, NGUARD, 8>(st, p);
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
