/*
 * gemm_p3.hip -- the four big projections of an encoder layer with BOTH operands pre-split.
 *
 *   C[M][N] = A[M][K] . W[N][K]^T + bias  (+ GELU | + residual), fp32 results
 *
 * Replaces `linear_layer` (ll.cl:7-86), `QKV` (multihead.cl:3-63) and `encoderResidual`
 * (layer_norm.cl:55-65); CPU statement linear_layer_seq ViT_seq.c:295-309, gelu :283,
 * residual loops :348-351,360-363.
 *
 * The fp32 product a*w is formed on the bf16 matrix cores from the exact three-way split of
 * both operands (fp32_split.h: x = x0 + x1 + x2, six partial products of weight >= 2^-16, fp32
 * accumulation).  gemm_mfma.hip splits the activations inside its K loop -- 2.5 VALU
 * instructions per MFMA, which is what bounds that loop.  Here nothing is split in the loop:
 * the PRODUCER of every GEMM input (LayerNorm, the attention epilogue, the fc1 GELU epilogue)
 * writes its result as three bf16 planes, exactly as the weights are pre-split at context
 * creation, and the K loop is MFMAs, LDS reads and loads only.  Same six products per block, in
 * the same order, on the same k assignment as gemm_mfma.hip: results are bit-identical.
 *
 * Plane layout ("P3"), the same for activations [rows][K] and weights [N][K]:
 *     planes[K/32][3][rows][32] bf16  -- K step, part, row, element
 * so the 64 bytes a row contributes to one K step and one part sit next to its neighbours':
 *  - a 16-row MFMA operand fragment of A is ONE contiguous, aligned KiB (lane l: row l & 15, 16-byte
 *    chunk l >> 4).  A therefore goes HBM/L2 -> VGPR directly with perfectly coalesced
 *    global_load_dwordx4, double-buffered in registers one K step ahead: every wave owns its
 *    32 rows (8 waves x 32 = the 256-row tile), so nothing about A is shared and A never
 *    touches the LDS;
 *  - a tile's W read of one K step is 3 runs of BN*64 contiguous bytes, moved by LDS-DMA
 *    (global_load_lds_dwordx4) into a [part][BN][64 B] image whose 16-byte chunks are
 *    XOR-swizzled (chunk c of row r at c ^ f((r >> 2) & 3), f = {0,2,3,1}: the 16 lanes of a
 *    ds_read_b128 group hit 16 distinct slots of the 256-byte bank row).  Two LDS stages.
 * Wave tile 32 x BN: both A fragments (2 x 3 parts) stay in registers for the whole step, W is
 * taken in chunks of two fragments, double-buffered in registers.  One barrier per K step,
 * placed before the LAST chunk's MFMAs (that chunk is already in registers, so every LDS read
 * of stage t is done): behind it the DMA of step t+2 is issued into the freed stage and chunk 0
 * of step t+1 is fetched while the matrix pipe still has the last chunk's 24 MFMAs to run.
 *
 * W fragment rows are permuted so that a lane ends up with EIGHT consecutive output columns
 * (fragment pair (2s, 2s+1), lane group g: columns 32s + 8g .. +7): the plane-format epilogue
 * stores 16 bytes per lane and part, 1 KiB contiguous per wave-instruction; the fp32 epilogue
 * two adjacent 16-byte stores.
 */
#include "kernelHandler.h"
#include "vit_kernels.h"
#include "fp32_split.h"
#include "gemm_common.h"

#include <cstdint>
#include <cstdlib>

namespace {

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
typedef const __attribute__((address_space(1))) char *gchar_t;     /* global-memory bytes (never flat) */
typedef const __attribute__((address_space(1))) f32x4 *gvec_t;

enum { EPI_NONE = 0, EPI_GELU = 1, EPI_RESID = 2 };
enum { OUT_F32 = 0, OUT_P3 = 1 };

struct P3Params {
    const char *A;            /* activation planes [K/32][3][a_rows][32] bf16 */
    const char *W;            /* weight planes     [K/32][3][N][32] bf16 */
    const float *bias;        /* [N] */
    const float *R;           /* residual [a_rows][N] fp32 (EPI_RESID) */
    void *C;                  /* fp32 [a_rows][N], or planes [N/32][3][a_rows][32] */
    int row_begin, row_end;   /* rows of the activation matrix this launch covers */
    int N, K;
    int a_rows;               /* rows of the whole activation matrix = the planes' row count */
    int mtiles, ntiles;
};

/* 16-byte chunk swizzle of a 64-byte LDS row r: f((r >> 2) & 3), f = {0, 2, 3, 1} */
__device__ __forceinline__ int swz64(int r4) { return (0x78 >> (2 * (r4 & 3))) & 3; }

/* LAB (tools/p3_lab.hip only; 0 in the library): bit 0 skips the W fragment reads after the first step, bit 1
 * the W DMA after the prologue, bit 2 the A loads after the prologue, bit 3 the barrier, bit 4 makes every
 * workgroup load the A rows of tile 0, bit 5 the W rows of tile 0 (operands served by L2 alone) -- throw-away
 * ablations that price each data movement; their results are wrong by construction. */
template <int NW, int BN, int EPI, int OUTK, int LAB = 0>
__global__ __launch_bounds__(64 * NW, 2) void gemm_p3_kernel(const P3Params p)
{
    constexpr int BM = 32 * NW, JT = BN / 16;
    constexpr int STAGE = 3 * BN * 64;            /* bytes per LDS stage: [part][BN][64] */
    constexpr int PW = 3 * BN / 16 / NW;          /* 1-KiB DMA pieces per wave and stage */
    static_assert(JT % 2 == 0 && (3 * BN / 16) % NW == 0, "tile shape");
    typedef bf16x8 frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tile = xcd_tile(blockIdx.x, p.mtiles * p.ntiles);
    const int m0 = p.row_begin + (tile / p.ntiles) * BM;
    const int n0 = (tile % p.ntiles) * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q = lane >> 4;

    /* A fragment i of this wave: rows m0 + 32*wave + 16*i + l15, chunk q of the row's 64 bytes.
     * Rows past the end re-read the last row (finite duplicates, never stored). */
    unsigned aoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
        aoff[i] = (unsigned)min(((LAB & 16) ? 0 : m0) + 32 * wave + 16 * i + l15, p.row_end - 1) * 64u + 16u * q;
    const size_t a_plane = (size_t)p.a_rows * 64, w_plane = (size_t)p.N * 64;

    /* W DMA piece pc = 16 rows x 64 B of one part: lane fills physical chunk (lane & 3) of row
     * 16*rb + (lane >> 2) with logical chunk phys ^ f(row >> 2). */
    const unsigned wlane = (unsigned)(lane >> 2) * 64u + 16u * ((lane & 3) ^ swz64(lane >> 4));
    const gchar_t wtile = (gchar_t)p.W + (size_t)((LAB & 32) ? 0 : n0) * 64;

    /* W fragment j = 2s + b, MFMA row l15 = LDS row 32s + 8*(l15 >> 2) + 4b + (l15 & 3) */
    const int rl = 8 * (l15 >> 2) + (l15 & 3);
    const unsigned woff_e = (unsigned)rl * 64u + 16u * (q ^ swz64(2 * (l15 >> 2)));
    const unsigned woff_o = (unsigned)(rl + 4) * 64u + 16u * (q ^ swz64(2 * (l15 >> 2) + 1));

    f32x4 acc[2][JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(p.bias + n0 + 32 * (j >> 1) + 8 * q + 4 * (j & 1));
        acc[0][j] = bv;
        acc[1][j] = bv;
    }

    frag_t a0[2][3], a1[2][3], w[2][3];

    /* Uniform (SGPR) base + 32-bit per-lane offset, the base made opaque per step: otherwise the
     * compiler keeps one 64-bit per-lane induction pointer per load (24 VGPRs, spilled). */
    auto load_a = [&](frag_t (&a)[2][3], int kt) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            gchar_t base = (gchar_t)p.A + (size_t)(kt * 3 + pl) * a_plane;
            asm volatile("" : "+s"(base));
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[i][pl] = __builtin_bit_cast(frag_t, *reinterpret_cast<gvec_t>(base + aoff[i]));
        }
    };
    auto dma_w = [&](int stage, int kt) {
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int pc = wave * PW + i, plane = pc / (BN / 16), rb = pc - plane * (BN / 16);
            gchar_t src = wtile + ((size_t)(kt * 3 + plane) * w_plane + (size_t)rb * 1024);
            asm volatile("" : "+s"(src));
            __builtin_amdgcn_global_load_lds((gptr_t)(src + wlane), (lptr_t)(smem + stage * STAGE + pc * 1024), 16, 0, 0);
        }
    };
    auto read_w = [&](frag_t (&wf)[3], const char *stage, int j) {
#pragma unroll
        for (int o = 0; o < 3; ++o) {   /* in the order the products need them: part 0, 2, 1 */
            const int pl = (3 - o) % 3;
            wf[pl] = __builtin_bit_cast(frag_t, *reinterpret_cast<const f32x4 *>(
                                                    stage + ((j & 1) ? woff_o : woff_e) + (j >> 1) * 2048 + pl * (BN * 64)));
        }
    };
    auto mfma_frag = [&](const frag_t (&a)[2][3], const frag_t (&wf)[3], int j) {
#pragma unroll
        for (int t = 0; t < 6; ++t) /* per accumulator: smallest terms first */
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[i][j] = mfma_part(wf[term_w<3>(t)], a[i][term_a<3>(t)], acc[i][j]);
    };

    const int nk = p.K / 32;
    auto step = [&](const frag_t (&au)[2][3], frag_t (&al)[2][3], int kt) {
        const char *cur = smem + (kt & 1) * STAGE, *nxt = smem + ((kt + 1) & 1) * STAGE;
        const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
        if (more1 && !(LAB & 4))
            load_a(al, kt + 1);
#pragma unroll
        for (int j = 0; j < JT - 1; ++j) {
            if (!(LAB & 1) || kt == 0)
                read_w(w[(j + 1) & 1], cur, j + 1);
            mfma_frag(au, w[j & 1], j);
            /* fragment j+1 is fetched under fragment j's MFMAs (left alone, the compiler issues each read
             * right before its use and the matrix pipe waits out the LDS latency) */
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(LAB & 8))
            __syncthreads();   /* stage kt read by every wave (its last fragment is in registers); stage kt+1 has landed */
        if (more2 && !(LAB & 2))
            dma_w(kt & 1, kt + 2);
        if (more1 && (!(LAB & 1) || kt == 0))
            read_w(w[0], nxt, 0);
        mfma_frag(au, w[(JT - 1) & 1], JT - 1);
    };

    dma_w(0, 0);
    load_a(a0, 0);
    __syncthreads();
    if (nk > 1)
        dma_w(1, 1);
    read_w(w[0], smem, 0);
    for (int kt = 0; kt < nk; kt += 2) {   /* nk is even (launcher) */
        step(a0, a1, kt);
        if (LAB & 4)
            step(a0, a1, kt + 1);
        else
            step(a1, a0, kt + 1);
    }

    /* Epilogue: fragment pair (2s, 2s+1) of row block i = 8 consecutive columns of one row. */
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = m0 + 32 * wave + 16 * i + l15;
        if (row >= p.row_end)
            continue;
#pragma unroll
        for (int s = 0; s < JT / 2; ++s) {
            const int col = n0 + 32 * s + 8 * q;
            f32x4 lo = acc[i][2 * s], hi = acc[i][2 * s + 1];
            if (EPI == EPI_GELU) {
                const f32x2 g0 = gelu_exact2(f32x2{lo[0], lo[1]}), g1 = gelu_exact2(f32x2{lo[2], lo[3]});
                const f32x2 g2 = gelu_exact2(f32x2{hi[0], hi[1]}), g3 = gelu_exact2(f32x2{hi[2], hi[3]});
                lo = f32x4{g0[0], g0[1], g1[0], g1[1]};
                hi = f32x4{g2[0], g2[1], g3[0], g3[1]};
            }
            if (EPI == EPI_RESID) {
                const float *rp = p.R + (size_t)row * p.N + col;
                lo = *reinterpret_cast<const f32x4 *>(rp) + lo;
                hi = *reinterpret_cast<const f32x4 *>(rp + 4) + hi;
            }
            if (OUTK == OUT_P3) {
                frag_t part[3];
                split8(lo, hi, part[0], part[1], part[2]);
                char *dst = static_cast<char *>(p.C) + ((size_t)((n0 >> 5) + s) * 3 * p.a_rows + row) * 64 + 16 * q;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    *reinterpret_cast<f32x4 *>(dst + pl * a_plane) = __builtin_bit_cast(f32x4, part[pl]);
            } else {
                float *cp = static_cast<float *>(p.C) + (size_t)row * p.N + col;
                *reinterpret_cast<f32x4 *>(cp) = lo;
                *reinterpret_cast<f32x4 *>(cp + 4) = hi;
            }
        }
    }
}

template <int NW, int BN, int EPI, int OUTK>
int launch_p3_tile(hipStream_t st, P3Params p)
{
    constexpr int LDS = 2 * 3 * BN * 64;
    VH_SET_LDS_ONCE((gemm_p3_kernel<NW, BN, EPI, OUTK>), LDS);
    p.mtiles = (p.row_end - p.row_begin + 32 * NW - 1) / (32 * NW);
    p.ntiles = p.N / BN;
    hipLaunchKernelGGL((gemm_p3_kernel<NW, BN, EPI, OUTK>), dim3(p.mtiles * p.ntiles), dim3(64 * NW), LDS, st, p);
    VH_LAUNCH_CHECK("gemm_p3_kernel");
    return 0;
}

/* Tile choice (measured on ViT-B/16, profiles/r02_*): 256x256 tiles (8 waves of 32x256, one workgroup per
 * CU) where N allows and the grid is at least 2.5 scheduling rounds of them, with the last, partly filled
 * round handed to 128x128 tiles (4 waves of 32x128, two to three workgroups per CU): a grid of r.f rounds
 * costs ceil(r.f) rounds, the remainder rows as quarter-size tiles about f/2.  Smaller problems -- batch 64
 * has 0.6 to 2.3 rounds of big tiles per projection -- and the N = K = 768 output projection run on the
 * 128x128 tile alone: at equal work it is within 3 % of the big tile, and it quantises four times finer
 * (batch 64: 5495 against 4947 images/s).  Every tile computes the same k order: results do not depend on
 * the choice. */
template <int EPI, int OUTK>
int launch_p3(hipStream_t st, const P3Params &p, int small_only)
{
    const int rows = p.row_end - p.row_begin;
    const int num_cus = vh_device_cus(vh_current_device());
    const int ntiles = p.N / 256, mtiles = (rows + 255) / 256;
    const long tiles = (long)mtiles * ntiles;
    if (p.N % 256 != 0 || small_only || 2 * tiles < 5 * (long)num_cus)
        return launch_p3_tile<4, 128, EPI, OUTK>(st, p);
    const long full = tiles / num_cus, rem = tiles % num_cus;
    const int rows_big = (int)(full * num_cus / ntiles) * 256;
    if (rem == 0 || 4 * rem > 3 * num_cus || rows_big <= 0 || rows_big >= rows)
        return launch_p3_tile<8, 256, EPI, OUTK>(st, p);
    P3Params big = p, rest = p;
    big.row_end = p.row_begin + rows_big;
    rest.row_begin = big.row_end;
    const int rc = launch_p3_tile<8, 256, EPI, OUTK>(st, big);
    return rc ? rc : launch_p3_tile<4, 128, EPI, OUTK>(st, rest);
}

/* fp32 [rows][K] <-> planes [K/32][3][rows][32]: one thread per 8 consecutive elements */
__global__ void split3_rows_kernel(const float *__restrict__ in, char *__restrict__ planes, int rows, int K)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k8 = K >> 3;
    if (idx >= (size_t)rows * k8)
        return;
    const int row = (int)(idx / k8), c8 = (int)(idx - (size_t)row * k8);
    const float *src = in + (size_t)row * K + 8 * c8;
    bf16x8 part[3];
    split8(*reinterpret_cast<const f32x4 *>(src), *reinterpret_cast<const f32x4 *>(src + 4), part[0], part[1], part[2]);
    char *dst = planes + ((size_t)(c8 >> 2) * 3 * rows + row) * 64 + 16 * (c8 & 3);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
        *reinterpret_cast<f32x4 *>(dst + (size_t)pl * rows * 64) = __builtin_bit_cast(f32x4, part[pl]);
}

__global__ void merge3_rows_kernel(const char *__restrict__ planes, float *__restrict__ out, int rows, int K)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k8 = K >> 3;
    if (idx >= (size_t)rows * k8)
        return;
    const int row = (int)(idx / k8), c8 = (int)(idx - (size_t)row * k8);
    const char *src = planes + ((size_t)(c8 >> 2) * 3 * rows + row) * 64 + 16 * (c8 & 3);
    bf16x8 part[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
        part[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(src + (size_t)pl * rows * 64));
    float *dst = out + (size_t)row * K + 8 * c8;
#pragma unroll
    for (int e = 0; e < 8; ++e)   /* exact: the parts are disjoint slices of one 24-bit significand */
        dst[e] = ((float)part[0][e] + (float)part[1][e]) + (float)part[2][e];
}

} // namespace

extern "C" int vh_launch_split3_rows(vh_stream_t s, const float *input, void *planes, int rows, int cols)
{
    if (!input || !planes || rows <= 0 || cols <= 0 || cols % 32 != 0 || (((uintptr_t)input | (uintptr_t)planes) & 15))
        return vh_fail(1, "vh_launch_split3_rows: bad argument (cols %% 32 == 0, 16-byte aligned pointers)");
    const size_t threads = (size_t)rows * (cols / 8);
    hipLaunchKernelGGL(split3_rows_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)s, input,
                       static_cast<char *>(planes), rows, cols);
    VH_LAUNCH_CHECK("split3_rows_kernel");
    return 0;
}

extern "C" int vh_launch_merge3_rows(vh_stream_t s, const void *planes, float *output, int rows, int cols)
{
    if (!output || !planes || rows <= 0 || cols <= 0 || cols % 32 != 0 || (((uintptr_t)output | (uintptr_t)planes) & 15))
        return vh_fail(1, "vh_launch_merge3_rows: bad argument (cols %% 32 == 0, 16-byte aligned pointers)");
    const size_t threads = (size_t)rows * (cols / 8);
    hipLaunchKernelGGL(merge3_rows_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)s,
                       static_cast<const char *>(planes), output, rows, cols);
    VH_LAUNCH_CHECK("merge3_rows_kernel");
    return 0;
}

extern "C" int vh_launch_linear_p3(vh_stream_t s, void *output, int output_planes, const void *weight_planes,
                                   const void *input_planes, const float *bias, int rowA, int colA, int colB,
                                   int doGelu, const float *residual)
{
    if (!output || !weight_planes || !input_planes || !bias)
        return vh_fail(1, "vh_launch_linear_p3: null pointer argument");
    if (rowA <= 0 || colA <= 0 || colB <= 0 || colA % 64 != 0 || colB % 128 != 0)
        return vh_fail(1, "vh_launch_linear_p3: needs colA %% 64 == 0 and colB %% 128 == 0 (%d,%d,%d)", rowA, colA, colB);
    if ((doGelu && residual) || (residual && output_planes))
        return vh_fail(1, "vh_launch_linear_p3: unsupported epilogue combination");
    if ((((uintptr_t)output | (uintptr_t)weight_planes | (uintptr_t)input_planes | (uintptr_t)bias | (uintptr_t)residual) & 15) != 0)
        return vh_fail(1, "vh_launch_linear_p3: pointers must be 16-byte aligned");
    if ((size_t)rowA * 64 > 0xffffffffull)
        return vh_fail(1, "vh_launch_linear_p3: rowA=%d too large", rowA);
    P3Params p = {};
    p.A = static_cast<const char *>(input_planes);
    p.W = static_cast<const char *>(weight_planes);
    p.bias = bias; p.R = residual; p.C = output;
    p.row_begin = 0; p.row_end = rowA; p.a_rows = rowA;
    p.N = colB; p.K = colA;
    hipStream_t st = (hipStream_t)s;
    static int force_small = -1;
    if (force_small < 0) {
        const char *env = getenv("VIT_HIP_P3_TILE");   /* "128": only the 128x128 tile (measurements) */
        force_small = (env && env[0] == '1') ? 1 : 0;
    }
    const int small_only = force_small || (residual && colA < 2048);   /* the N = K = E output projection: measured */
    if (doGelu)
        return output_planes ? launch_p3<EPI_GELU, OUT_P3>(st, p, small_only) : launch_p3<EPI_GELU, OUT_F32>(st, p, small_only);
    if (residual)
        return launch_p3<EPI_RESID, OUT_F32>(st, p, small_only);
    return output_planes ? launch_p3<EPI_NONE, OUT_P3>(st, p, small_only) : launch_p3<EPI_NONE, OUT_F32>(st, p, small_only);
}
