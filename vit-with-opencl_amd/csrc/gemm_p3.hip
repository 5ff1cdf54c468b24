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
#include "norm_fold.h"

#include <cstdint>
#include <cstdlib>

namespace {

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
typedef const __attribute__((address_space(1))) char *gchar_t;     /* global-memory bytes (never flat) */
typedef const __attribute__((address_space(1))) f32x4 *gvec_t;

#ifndef P1_KG
#define P1_KG 2   /* one-part operands: 32-deep K groups per LDS stage (one barrier per 32 * P1_KG k) */
#endif
/* EPI_NORM / EPI_NORM_GELU: the LayerNorm in FRONT of the projection folded into it (norm_fold.h): A holds the
 * un-normalised rows, W the gamma-scaled weights; the epilogue applies the row's 1/std and mean terms. */
enum { EPI_NONE = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_PATCH = 3, EPI_NORM = 4, EPI_NORM_GELU = 5 };
/* _H: one-part planes of fp16 (the reduced modes' Q|K|V).  OUT_F32_OPER / _OPER_MX: fp32 rows AND the same values as the
 * NEXT projection's operand (one-part bf16 planes / MX values + scales) AND the rows' partial sums for its folded
 * LayerNorm -- the residual-stream producers of the reduced modes (patch embedding, output projection, fc2). */
enum { OUT_F32 = 0, OUT_PLANES = 1, OUT_PLANES_H = 2, OUT_F32_OPER = 3, OUT_F32_OPER_MX = 4 };

struct P3Params {
    const char *A;            /* activation planes [K/32][NPL][a_rows][32] bf16 */
    const char *W;            /* weight planes     [K/32][NPL][N][32] bf16 */
    const float *bias;        /* [N] */
    const float *R;           /* residual [a_rows][N] fp32 (EPI_RESID) */
    void *C;                  /* fp32 [a_rows][N], or planes [N/32][NPL][a_rows][32] */
    int row_begin, row_end;   /* rows of the activation matrix this launch covers */
    int N, K;
    int a_rows;               /* rows of the whole activation matrix = the planes' row count */
    int mtiles, ntiles;
    /* EPI_PATCH (patch embedding): GEMM row m = patch (m / np, m % np) lands in token row image * tokens + 1 + patch,
     * plus that token's position embedding (ViT_seq.c:65-80,114-117) */
    const float *pos;         /* [tokens][N] */
    int np, tokens;
    /* EPI_NORM*: the folded LayerNorm of the rows of A (norm_fold.h) */
    const float *colsum;      /* [N]: sum over k of the gamma-scaled, rounded weights of column n */
    const float *stats;       /* [K/128][a_rows][2]: partial (sum x, sum x^2) per 128 columns, left by A's producer */
    double eps;
    /* OUT_F32_OPER*: what the NEXT projection reads */
    void *oper, *oper_scales; /* planes [N/32][1][c_rows][32] bf16 | MX values [N/128][c_rows][128] + scales [N/128][4][c_rows] */
    float *stats_out;         /* [N/128][c_rows][2] */
    int c_rows;               /* rows of the output matrix (= a_rows except for EPI_PATCH: images x tokens) */
};

/* NPL = parts per value: 3 = the exact fp32 split (six products per block, the default fp32 path);
 * 1 = operands rounded to bf16 by their producers (one product per block: BASELINE config 3's
 * bf16-operand mode).  With one part a W fragment feeds 2 MFMAs instead of 12, so a stage holds two
 * 32-deep K groups (one barrier per 64 k) and four fragments are in flight in registers.
 * (The ablation copy of this kernel, with switches that remove one data movement each, is tools/p3_lab_kernel.inc.) */
template <int NW, int BN, int EPI, int OUTK, int NPL = 3>
__global__ __launch_bounds__(64 * NW, 2) void gemm_p3_kernel(const P3Params p)
{
    constexpr int BM = 32 * NW, JT = BN / 16;
    constexpr int KG = NPL == 3 ? 1 : P1_KG;   /* 32-deep K groups per LDS stage (the launcher sizes the LDS from the same macro) */
    constexpr int RING = NPL == 3 ? 2 : 4;          /* W fragments in flight in registers */
    constexpr int F = KG * JT;                      /* W fragments per step */
    constexpr int NT = NPL == 3 ? 6 : 1;            /* products per block */
    constexpr int GROUP = NPL * BN * 64;            /* bytes of one K group in a stage: [part][BN][64] */
    constexpr int STAGE = KG * GROUP;
    constexpr int PW = KG * NPL * BN / 16 / NW;     /* 1-KiB DMA pieces per wave and stage */
    constexpr bool NORM = EPI == EPI_NORM || EPI == EPI_NORM_GELU;
    constexpr bool GELU = EPI == EPI_GELU || EPI == EPI_NORM_GELU;
    constexpr bool OPER = OUTK == OUT_F32_OPER || OUTK == OUT_F32_OPER_MX;
    static_assert(NPL == 1 || NPL == 3, "parts per value");
    static_assert(JT % 2 == 0 && (KG * NPL * BN / 16) % NW == 0 && F % RING == 0, "tile shape");
    static_assert(OUTK != OUT_F32_OPER_MX || NPL == 1, "an MX operand follows one-part operands only");
    static_assert(!OPER || (BN % 128 == 0 && (EPI == EPI_RESID || EPI == EPI_PATCH)), "operand producers: residual-stream epilogues");
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
        aoff[i] = (unsigned)min(m0 + 32 * wave + 16 * i + l15, p.row_end - 1) * 64u + 16u * q;
    const size_t a_plane = (size_t)p.a_rows * 64, w_plane = (size_t)p.N * 64;

    /* W DMA piece pc = 16 rows x 64 B of one part: lane fills physical chunk (lane & 3) of row
     * 16*rb + (lane >> 2) with logical chunk phys ^ f(row >> 2). */
    const unsigned wlane = (unsigned)(lane >> 2) * 64u + 16u * ((lane & 3) ^ swz64(lane >> 4));
    const gchar_t wtile = (gchar_t)p.W + (size_t)n0 * 64;

    /* W fragment j = 2s + b.  Planes out: MFMA row l15 = LDS row 32s + 8*(l15 >> 2) + 4b + (l15 & 3), so that a lane
     * ends up with EIGHT consecutive columns (one 16-byte store of 16-bit values per part).  fp32 rows out (NATURAL):
     * MFMA row l15 = LDS row 32s + 16b + l15 -- a lane holds columns 16j + 4q .. +3 of fragment j and the four lanes of
     * a row write 64 contiguous bytes per store instruction (with the permuted rows a store instruction writes every
     * other 16 bytes of a line).  Every output element sums the same products in the same order either way.
     * The operand producers (OPER) write planes / MX blocks besides their fp32 rows: permuted order. */
    constexpr bool NATURAL = OUTK == OUT_F32;
    const int rl = NATURAL ? l15 : 8 * (l15 >> 2) + (l15 & 3);
    const unsigned woff_e = (unsigned)rl * 64u + 16u * (q ^ swz64(NATURAL ? (l15 >> 2) : 2 * (l15 >> 2)));
    const unsigned woff_o = NATURAL ? woff_e + 16u * 64u
                                    : (unsigned)(rl + 4) * 64u + 16u * (q ^ swz64(2 * (l15 >> 2) + 1));
    /* first column (relative to n0) of the four values a lane holds of fragment j */
    auto frag_col = [&](int j) { return NATURAL ? 16 * j + 4 * q : 32 * (j >> 1) + 8 * q + 4 * (j & 1); };

    /* One-part operands (the reduced modes): the residual goes INTO the accumulators with the bias, (r + bias) + sum
     * instead of r + (bias + sum) -- its load then runs under the prologue's DMA instead of behind the K loop, costs no
     * registers, and the epilogue is stores only.  The sums differ in the last bits (every MFMA rounds at the
     * magnitude of r): far inside what bf16 operands leave, but not the reference's order, so the exact fp32 path
     * (NPL = 3) keeps adding the residual to the finished sum (ViT_seq.c:350,362). */
    constexpr bool R_IN_ACC = EPI == EPI_RESID && NPL == 1;
    f32x4 acc[2][JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        const int col = n0 + frag_col(j);
        const f32x4 bv = NORM ? f32x4{0.0f, 0.0f, 0.0f, 0.0f} : *reinterpret_cast<const f32x4 *>(p.bias + col);   /* NORM: bias in the epilogue */
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            acc[i][j] = bv;
            if (R_IN_ACC) {   /* rows past the end re-read the last row (never stored) */
                const int row = min(m0 + 32 * wave + 16 * i + l15, p.row_end - 1);
                acc[i][j] = *reinterpret_cast<const f32x4 *>(p.R + (size_t)row * p.N + col) + bv;
            }
        }
    }
    /* folded LayerNorm: 1/std and -mean/std of this lane's two rows, from the partial sums their producer left */
    float n_rstd[2] = {1.0f, 1.0f}, n_shift[2] = {0.0f, 0.0f};
    if (NORM) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            row_norm_terms(p.stats, p.K >> 7, p.a_rows, min(m0 + 32 * wave + 16 * i + l15, p.row_end - 1), q, p.K, p.eps, n_rstd[i], n_shift[i]);
    }

    frag_t a0[KG][2][NPL], a1[KG][2][NPL], w[RING][NPL];

    /* Uniform (SGPR) base + 32-bit per-lane offset, the base made opaque per step: otherwise the
     * compiler keeps one 64-bit per-lane induction pointer per load (24 VGPRs, spilled). */
    auto load_a = [&](frag_t (&a)[KG][2][NPL], int kt) {
#pragma unroll
        for (int kg = 0; kg < KG; ++kg)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                gchar_t base = (gchar_t)p.A + (size_t)((kt * KG + kg) * NPL + pl) * a_plane;
                asm volatile("" : "+s"(base));
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    a[kg][i][pl] = __builtin_bit_cast(frag_t, *reinterpret_cast<gvec_t>(base + aoff[i]));
            }
    };
    auto dma_w = [&](int stage, int kt) {
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int pc = wave * PW + i, gp = pc / (BN / 16), rb = pc - gp * (BN / 16);   /* gp = kg * NPL + part */
            gchar_t src = wtile + ((size_t)(kt * KG * NPL + gp) * w_plane + (size_t)rb * 1024);
            asm volatile("" : "+s"(src));
            __builtin_amdgcn_global_load_lds((gptr_t)(src + wlane), (lptr_t)(smem + stage * STAGE + pc * 1024), 16, 0, 0);
        }
    };
    auto read_w = [&](frag_t (&wf)[NPL], const char *stage, int f) {
        const int kg = f / JT, j = f % JT;
#pragma unroll
        for (int o = 0; o < NPL; ++o) {   /* in the order the products need them: part 0, 2, 1 */
            const int pl = (NPL - o) % NPL;
            wf[pl] = __builtin_bit_cast(frag_t, *reinterpret_cast<const f32x4 *>(
                                                    stage + kg * GROUP + ((j & 1) ? woff_o : woff_e) + (j >> 1) * 2048 + pl * (BN * 64)));
        }
    };
    auto mfma_frag = [&](const frag_t (&a)[KG][2][NPL], const frag_t (&wf)[NPL], int f) {
        const int kg = f / JT, j = f % JT;
#pragma unroll
        for (int t = 0; t < NT; ++t) /* per accumulator: smallest terms first */
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[i][j] = mfma_part(wf[term_w<NPL>(t)], a[kg][i][term_a<NPL>(t)], acc[i][j]);
    };
    /* fragment f+RING-1 is fetched under fragment f's MFMAs (left alone, the compiler issues each read
     * right before its use and the matrix pipe waits out the LDS latency) */
    auto interleave = [&]() {
#pragma unroll
        for (int r = 0; r < NPL; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        if (2 * NT > 2 * NPL)
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * NT - 2 * NPL, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    const int nk = p.K / (32 * KG);
    auto step = [&](const frag_t (&au)[KG][2][NPL], frag_t (&al)[KG][2][NPL], int kt) {
        const char *cur = smem + (kt & 1) * STAGE, *nxt = smem + ((kt + 1) & 1) * STAGE;
        const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
        load_a(al, more1 ? kt + 1 : kt);   /* unconditional (no copies at a join): the last step re-reads its own */
#pragma unroll
        for (int f = 0; f <= F - RING; ++f) {
            read_w(w[(f + RING - 1) % RING], cur, f + RING - 1);
            mfma_frag(au, w[f % RING], f);
            interleave();
        }
        __syncthreads();   /* stage kt read by every wave (its last fragments are in registers); stage kt+1 has landed */
        if (more2)
            dma_w(kt & 1, kt + 2);
#pragma unroll
        for (int f = F - RING + 1; f < F; ++f) {
            if (more1)
                read_w(w[(f + RING - 1) % RING], nxt, f + RING - 1 - F);
            mfma_frag(au, w[f % RING], f);
            interleave();
        }
    };

    dma_w(0, 0);
    load_a(a0, 0);
    /* folded LayerNorm: the tile's column terms (colsum, folded bias: 2 x BN floats) go to LDS behind the two W stages --
     * the epilogue takes them by ds_read_b128.  (Loaded from global memory in the epilogue, four 16-byte loads per
     * fragment pair, each a round trip to L2 with nothing to hide it: QKV 0.34 -> 0.43 ms, fc1 0.53 -> 0.64.) */
    float *const ncol = reinterpret_cast<float *>(smem + 2 * STAGE);
    if (NORM && tid < 2 * BN / 4) {
        const int which = tid / (BN / 4), c4 = tid - which * (BN / 4);
        *reinterpret_cast<f32x4 *>(ncol + which * BN + 4 * c4) = *reinterpret_cast<const f32x4 *>((which ? p.bias : p.colsum) + n0 + 4 * c4);
    }
    __syncthreads();
    if (nk > 1)
        dma_w(1, 1);
#pragma unroll
    for (int f = 0; f < RING - 1; ++f)
        read_w(w[f], smem, f);
    for (int kt = 0; kt < nk; kt += 2) {   /* nk is even (launcher) */
        step(a0, a1, kt);
        step(a1, a0, kt + 1);
    }

    /* Epilogue: fragment pair (2s, 2s+1) of row block i = 8 consecutive columns of one row. */
    int rows_[2];
    size_t orow_[2];
    const float *posrow_[2] = {nullptr, nullptr};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        rows_[i] = m0 + 32 * wave + 16 * i + l15;
        orow_[i] = (size_t)rows_[i];
        if (EPI == EPI_PATCH) {
            const int b = rows_[i] / p.np, pp = rows_[i] - b * p.np;
            orow_[i] = (size_t)b * p.tokens + 1 + pp;
            posrow_[i] = p.pos + (size_t)(1 + pp) * p.N;
        }
    }
    /* OPER: this lane's share of a row's (sum, sum of squares) over 128 columns, even and odd elements apart (packed adds / fmas) */
    f32x2 psum[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}}, psq[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
    /* fragment pair s of row block i; nb = the folded-LayerNorm column terms (colsum lo/hi, folded bias lo/hi) */
    auto emit = [&](int i, int s, const f32x4 (&nb)[4]) {
        const int row = rows_[i];
        const int col = n0 + frag_col(2 * s), col_hi = n0 + frag_col(2 * s + 1);   /* planes: col_hi = col + 4 */
        f32x4 lo = acc[i][2 * s], hi = acc[i][2 * s + 1];
        if (NORM) {   /* LN(x) W^T + b = rstd (x W'^T) - rstd mean colsum(W') + b'   (norm_fold.h) */
            const f32x4 r4 = {n_rstd[i], n_rstd[i], n_rstd[i], n_rstd[i]}, s4 = {n_shift[i], n_shift[i], n_shift[i], n_shift[i]};
            lo = __builtin_elementwise_fma(lo, r4, __builtin_elementwise_fma(nb[0], s4, nb[2]));
            hi = __builtin_elementwise_fma(hi, r4, __builtin_elementwise_fma(nb[1], s4, nb[3]));
        }
        if (EPI == EPI_PATCH) {
            lo = lo + *reinterpret_cast<const f32x4 *>(posrow_[i] + col);
            hi = hi + *reinterpret_cast<const f32x4 *>(posrow_[i] + col_hi);
        }
        if (GELU) {
            /* a result that is rounded to ONE bf16 part takes the GELU whose error is matched to that format */
            auto gelu2 = [](f32x2 v) { return (NPL == 1 && OUTK == OUT_PLANES) ? gelu_lowp2<0>(v) : gelu_exact2(v); };
            const f32x2 g0 = gelu2(f32x2{lo[0], lo[1]}), g1 = gelu2(f32x2{lo[2], lo[3]});
            const f32x2 g2 = gelu2(f32x2{hi[0], hi[1]}), g3 = gelu2(f32x2{hi[2], hi[3]});
            lo = f32x4{g0[0], g0[1], g1[0], g1[1]};
            hi = f32x4{g2[0], g2[1], g3[0], g3[1]};
        }
        if (EPI == EPI_RESID && !R_IN_ACC) {
            const float *rp = p.R + (size_t)row * p.N;
            lo = *reinterpret_cast<const f32x4 *>(rp + col) + lo;
            hi = *reinterpret_cast<const f32x4 *>(rp + col_hi) + hi;
        }
        if (OUTK == OUT_PLANES || OUTK == OUT_PLANES_H) {
            frag_t part[3];
            if (NPL == 3) {
                split8(lo, hi, part[0], part[1], part[2]);
            } else if (OUTK == OUT_PLANES_H) {
                half8 hv;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    hv[e] = (_Float16)(e < 4 ? lo[e] : hi[e - 4]);
                part[0] = __builtin_bit_cast(frag_t, hv);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    part[0][e] = (__bf16)(e < 4 ? lo[e] : hi[e - 4]);
            }
            char *dst = static_cast<char *>(p.C) + ((size_t)((n0 >> 5) + s) * NPL * p.a_rows + row) * 64 + 16 * q;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
                *reinterpret_cast<f32x4 *>(dst + pl * a_plane) = __builtin_bit_cast(f32x4, part[pl]);
        } else {
            float *cp = static_cast<float *>(p.C) + orow_[i] * p.N;
            *reinterpret_cast<f32x4 *>(cp + col) = lo;
            *reinterpret_cast<f32x4 *>(cp + col_hi) = hi;
        }
        if (OPER) {
            /* the same eight values as the next projection's operand, and their share of the row's statistics (of the
             * fp32 values, as layer_norm_seq ViT_seq.c:124-131 sums them; fixed order: even and odd columns ascending per
             * lane, their two sums, then the four lanes of the row, per 128 columns -- whatever the tile, the same sums) */
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x2 v = e < 2 ? f32x2{lo[2 * e], lo[2 * e + 1]} : f32x2{hi[2 * e - 4], hi[2 * e - 3]};
                psum[i] = psum[i] + v;
                psq[i] = __builtin_elementwise_fma(v, v, psq[i]);
            }
            if (OUTK == OUT_F32_OPER) {   /* planes [N/32][NPL][c_rows][32]: one rounded part, or the exact three-part split */
                frag_t part[3];
                if (NPL == 3) {
                    split8(lo, hi, part[0], part[1], part[2]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        part[0][e] = (__bf16)(e < 4 ? lo[e] : hi[e - 4]);
                }
                char *dst = static_cast<char *>(p.oper) + ((size_t)((n0 >> 5) + s) * NPL * p.c_rows + orow_[i]) * 64 + 16 * q;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    *reinterpret_cast<f32x4 *>(dst + (size_t)pl * p.c_rows * 64) = __builtin_bit_cast(f32x4, part[pl]);
            } else {
                mx_store_block8(lo, hi, static_cast<char *>(p.oper), static_cast<unsigned char *>(p.oper_scales), p.c_rows, orow_[i],
                                n0 + 32 * s, q, true);
            }
            if ((s & 3) == 3) {
                float su = psum[i][0] + psum[i][1], sq2 = psq[i][0] + psq[i][1];
                su += __shfl_xor(su, 16);
                sq2 += __shfl_xor(sq2, 16);
                su += __shfl_xor(su, 32);
                sq2 += __shfl_xor(sq2, 32);
                if (q == 0)
                    *reinterpret_cast<f32x2 *>(p.stats_out + ((size_t)((n0 >> 7) + (s >> 2)) * p.c_rows + orow_[i]) * 2) = f32x2{su, sq2};
                psum[i] = psq[i] = f32x2{0.0f, 0.0f};
            }
        }
    };
    if (NORM) {   /* column terms read once per fragment pair, shared by the two row blocks */
        auto column_terms = [&](int s, f32x4 (&nb)[4]) {
            const int c_lo = frag_col(2 * s), c_hi = frag_col(2 * s + 1);
            nb[0] = *reinterpret_cast<const f32x4 *>(ncol + c_lo);
            nb[1] = *reinterpret_cast<const f32x4 *>(ncol + c_hi);
            nb[2] = *reinterpret_cast<const f32x4 *>(ncol + BN + c_lo);
            nb[3] = *reinterpret_cast<const f32x4 *>(ncol + BN + c_hi);
        };
        if (m0 + BM <= p.row_end) {   /* every tile but the last row of tiles: straight-line code, no per-lane conditions --
                                       * with a condition around every (pair, row block) the GELU chains of different pairs
                                       * cannot be interleaved and the epilogue ran at the latency of one chain */
#pragma unroll
            for (int s = 0; s < JT / 2; ++s) {
                f32x4 nb[4];
                column_terms(s, nb);
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    emit(i, s, nb);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (rows_[i] >= p.row_end)
                    continue;
#pragma unroll
                for (int s = 0; s < JT / 2; ++s) {
                    f32x4 nb[4];
                    column_terms(s, nb);
                    emit(i, s, nb);
                }
            }
        }
    } else {
        const f32x4 none[4] = {};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (rows_[i] >= p.row_end)
                continue;
#pragma unroll
            for (int s = 0; s < JT / 2; ++s)
                emit(i, s, none);
        }
    }
}

template <int NW, int BN, int EPI, int OUTK, int NPL>
int launch_p3_tile(hipStream_t st, P3Params p)
{
    constexpr int LDS = 2 * (NPL == 3 ? 1 : P1_KG) * NPL * BN * 64 + ((EPI == EPI_NORM || EPI == EPI_NORM_GELU) ? 2 * BN * 4 : 0);
    VH_SET_LDS_ONCE((gemm_p3_kernel<NW, BN, EPI, OUTK, NPL>), LDS);
    p.mtiles = (p.row_end - p.row_begin + 32 * NW - 1) / (32 * NW);
    p.ntiles = p.N / BN;
    hipLaunchKernelGGL((gemm_p3_kernel<NW, BN, EPI, OUTK, NPL>), dim3(p.mtiles * p.ntiles), dim3(64 * NW), LDS, st, p);
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
 * the choice.  (Round 4 tried 128x64 tiles for batch 64's N = 768 projections -- 594 tiles of 128x128 sit two on some CUs
 * and three on others -- and lost: twice the A re-reads per MFMA make that tile L2-bound; profiles/r04_bench_batch64_*.) */
template <int EPI, int OUTK, int NPL>
int launch_p3(hipStream_t st, const P3Params &p, int small_only)
{
    const int rows = p.row_end - p.row_begin;
    const int num_cus = vh_device_cus(vh_current_device());
    const int ntiles = p.N / 256, mtiles = (rows + 255) / 256;
    const long tiles = (long)mtiles * ntiles;
    /* One-part operands writing fp32 rows (the reduced modes' output projection, fc2 and patch embedding): 128x256
     * tiles, 4 waves of 32x256, two workgroups per CU -- one workgroup's stores of fp32 rows run under the other's K
     * loop (measured, ViT-B/16 batch 512: out-proj 0.223 -> 0.213 ms, fc2 0.503 -> 0.491, patch embedding 0.29 -> 0.27
     * against the rule below; the planes-out epilogues of QKV and fc1 measure the same either way). */
    if ((EPI == EPI_RESID || EPI == EPI_PATCH) && NPL == 1 && p.N % 256 == 0 && tiles >= num_cus)
        return launch_p3_tile<4, 256, EPI, OUTK, NPL>(st, p);
    if (p.N % 256 != 0 || small_only || 2 * tiles < 5 * (long)num_cus)
        return launch_p3_tile<4, 128, EPI, OUTK, NPL>(st, p);
    const long full = tiles / num_cus, rem = tiles % num_cus;
    const int rows_big = (int)(full * num_cus / ntiles) * 256;
    auto big_tiles = [&](const P3Params &q) { return launch_p3_tile<8, 256, EPI, OUTK, NPL>(st, q); };
    if (rem == 0 || 4 * rem > 3 * num_cus || rows_big <= 0 || rows_big >= rows)
        return big_tiles(p);
    P3Params big = p, rest = p;
    big.row_end = p.row_begin + rows_big;
    rest.row_begin = big.row_end;
    const int rc = big_tiles(big);
    return rc ? rc : launch_p3_tile<4, 128, EPI, OUTK, NPL>(st, rest);
}

/* fp32 [rows][K] <-> planes [K/32][NPL][rows][32]: one thread per 8 consecutive elements */
template <int NPL>
__global__ void split_rows_kernel(const float *__restrict__ in, char *__restrict__ planes, int rows, int K)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k8 = K >> 3;
    if (idx >= (size_t)rows * k8)
        return;
    const int row = (int)(idx / k8), c8 = (int)(idx - (size_t)row * k8);
    const float *src = in + (size_t)row * K + 8 * c8;
    const f32x4 u = *reinterpret_cast<const f32x4 *>(src), v = *reinterpret_cast<const f32x4 *>(src + 4);
    bf16x8 part[3];
    if (NPL == 3) {
        split8(u, v, part[0], part[1], part[2]);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            part[0][e] = (__bf16)(e < 4 ? u[e] : v[e - 4]);
    }
    char *dst = planes + ((size_t)(c8 >> 2) * NPL * rows + row) * 64 + 16 * (c8 & 3);
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
        *reinterpret_cast<f32x4 *>(dst + (size_t)pl * rows * 64) = __builtin_bit_cast(f32x4, part[pl]);
}

template <int NPL>
__global__ void merge_rows_kernel(const char *__restrict__ planes, float *__restrict__ out, int rows, int K)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k8 = K >> 3;
    if (idx >= (size_t)rows * k8)
        return;
    const int row = (int)(idx / k8), c8 = (int)(idx - (size_t)row * k8);
    const char *src = planes + ((size_t)(c8 >> 2) * NPL * rows + row) * 64 + 16 * (c8 & 3);
    bf16x8 part[3];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
        part[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(src + (size_t)pl * rows * 64));
    float *dst = out + (size_t)row * K + 8 * c8;
#pragma unroll
    for (int e = 0; e < 8; ++e)   /* exact: the parts are disjoint slices of one 24-bit significand */
        dst[e] = NPL == 3 ? ((float)part[0][e] + (float)part[1][e]) + (float)part[2][e] : (float)part[0][e];
}

int check_planes_args(const char *who, const void *a, const void *b, int rows, int cols, int parts)
{
    if (!a || !b || rows <= 0 || cols <= 0 || cols % 32 != 0 || (((uintptr_t)a | (uintptr_t)b) & 15) || (parts != 1 && parts != 3))
        return vh_fail(1, "%s: bad argument (cols %% 32 == 0, 16-byte aligned pointers, parts 1 or 3)", who);
    return 0;
}

} // namespace

extern "C" int vh_launch_split_rows(vh_stream_t s, const float *input, void *planes, int rows, int cols, int parts)
{
    if (int rc = check_planes_args("vh_launch_split_rows", input, planes, rows, cols, parts))
        return rc;
    const size_t threads = (size_t)rows * (cols / 8);
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (parts == 3)
        hipLaunchKernelGGL(split_rows_kernel<3>, grid, dim3(256), 0, (hipStream_t)s, input, static_cast<char *>(planes), rows, cols);
    else
        hipLaunchKernelGGL(split_rows_kernel<1>, grid, dim3(256), 0, (hipStream_t)s, input, static_cast<char *>(planes), rows, cols);
    VH_LAUNCH_CHECK("split_rows_kernel");
    return 0;
}

extern "C" int vh_launch_merge_rows(vh_stream_t s, const void *planes, float *output, int rows, int cols, int parts)
{
    if (int rc = check_planes_args("vh_launch_merge_rows", planes, output, rows, cols, parts))
        return rc;
    const size_t threads = (size_t)rows * (cols / 8);
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (parts == 3)
        hipLaunchKernelGGL(merge_rows_kernel<3>, grid, dim3(256), 0, (hipStream_t)s, static_cast<const char *>(planes), output, rows, cols);
    else
        hipLaunchKernelGGL(merge_rows_kernel<1>, grid, dim3(256), 0, (hipStream_t)s, static_cast<const char *>(planes), output, rows, cols);
    VH_LAUNCH_CHECK("merge_rows_kernel");
    return 0;
}

extern "C" int vh_launch_split3_rows(vh_stream_t s, const float *input, void *planes, int rows, int cols)
{
    return vh_launch_split_rows(s, input, planes, rows, cols, 3);
}

extern "C" int vh_launch_merge3_rows(vh_stream_t s, const void *planes, float *output, int rows, int cols)
{
    return vh_launch_merge_rows(s, planes, output, rows, cols, 3);
}

extern "C" int vh_launch_linear_planes(vh_stream_t s, void *output, int output_planes, const void *weight_planes,
                                       const void *input_planes, int parts, const float *bias, int rowA, int colA,
                                       int colB, int doGelu, const float *residual)
{
    if (!output || !weight_planes || !input_planes || !bias)
        return vh_fail(1, "vh_launch_linear_planes: null pointer argument");
    if (parts != 1 && parts != 3)
        return vh_fail(1, "vh_launch_linear_planes: parts must be 3 (exact fp32 split) or 1 (bf16 operands)");
    const int kstep = parts == 3 ? 64 : 64 * P1_KG;   /* two LDS stages per loop iteration */
    if (rowA <= 0 || colA <= 0 || colB <= 0 || colA % kstep != 0 || colB % 128 != 0)
        return vh_fail(1, "vh_launch_linear_planes: needs colA %% %d == 0 and colB %% 128 == 0 (%d,%d,%d)", kstep, rowA, colA, colB);
    if ((doGelu && residual) || (residual && output_planes) || output_planes < 0 || output_planes > 2 ||
        (output_planes == 2 && (parts != 1 || doGelu)))
        return vh_fail(1, "vh_launch_linear_planes: unsupported epilogue combination");
    if ((((uintptr_t)output | (uintptr_t)weight_planes | (uintptr_t)input_planes | (uintptr_t)bias | (uintptr_t)residual) & 15) != 0)
        return vh_fail(1, "vh_launch_linear_planes: pointers must be 16-byte aligned");
    if ((size_t)rowA * 64 > 0xffffffffull)
        return vh_fail(1, "vh_launch_linear_planes: rowA=%d too large", rowA);
    P3Params p = {};
    p.A = static_cast<const char *>(input_planes);
    p.W = static_cast<const char *>(weight_planes);
    p.bias = bias; p.R = residual; p.C = output;
    p.row_begin = 0; p.row_end = rowA; p.a_rows = rowA;
    p.N = colB; p.K = colA;
    hipStream_t st = (hipStream_t)s;
    const int small_only = residual && colA < 2048;   /* the N = K = E output projection: measured */
#define VH_P3_DISPATCH(NPL)                                                                                          \
    do {                                                                                                             \
        if (doGelu)                                                                                                  \
            return output_planes ? launch_p3<EPI_GELU, OUT_PLANES, NPL>(st, p, small_only)                           \
                                 : launch_p3<EPI_GELU, OUT_F32, NPL>(st, p, small_only);                             \
        if (residual)                                                                                                \
            return launch_p3<EPI_RESID, OUT_F32, NPL>(st, p, small_only);                                            \
        return output_planes ? launch_p3<EPI_NONE, OUT_PLANES, NPL>(st, p, small_only)                               \
                             : launch_p3<EPI_NONE, OUT_F32, NPL>(st, p, small_only);                                 \
    } while (0)
    if (parts == 3)
        VH_P3_DISPATCH(3);
    if (output_planes == 2)
        return launch_p3<EPI_NONE, OUT_PLANES_H, 1>(st, p, small_only);
    VH_P3_DISPATCH(1);
#undef VH_P3_DISPATCH
}

extern "C" int vh_launch_linear_p3(vh_stream_t s, void *output, int output_planes, const void *weight_planes,
                                   const void *input_planes, const float *bias, int rowA, int colA, int colB,
                                   int doGelu, const float *residual)
{
    return vh_launch_linear_planes(s, output, output_planes, weight_planes, input_planes, 3, bias, rowA, colA, colB,
                                   doGelu, residual);
}

/* ---- LayerNorm folded into the projections (norm_fold.h): the consumer and producer launches ---------------------- */

/* LN(A) W^T + b with the LayerNorm folded: `input_planes` = the UN-normalised rows as one-part bf16 planes, `row_stats` =
 * their partial sums [colA/128][rowA][2], `weight_planes` = the gamma-scaled weights, `colsum` / `bias_folded` the column
 * terms (vh_launch_fold_* below).  output_planes: 0 fp32 rows, 1 bf16 planes, 2 fp16 planes (as vh_launch_linear_planes). */
static int linear_planes_norm(vh_stream_t s, void *output, int output_planes, const void *weight_planes, const void *input_planes,
                              const float *row_stats, const float *colsum, const float *bias_folded, double eps, int rowA, int colA,
                              int colB, int doGelu, int parts)
{
    if (!output || !weight_planes || !input_planes || !row_stats || !colsum || !bias_folded)
        return vh_fail(1, "vh_launch_linear_planes_norm: null pointer argument");
    const int kstep = parts == 3 ? 64 : 64 * P1_KG;
    if (rowA <= 0 || colA <= 0 || colB <= 0 || colA % kstep != 0 || colB % 128 != 0)
        return vh_fail(1, "vh_launch_linear_planes_norm: needs colA %% %d == 0 and colB %% 128 == 0 (%d,%d,%d)", kstep, rowA, colA, colB);
    if (output_planes < 0 || output_planes > 2 || (output_planes == 2 && (doGelu || parts == 3)) || (output_planes == 0 && doGelu))
        return vh_fail(1, "vh_launch_linear_planes_norm: unsupported epilogue combination");
    if (colA > 16 * 128 || colA % 128 != 0)
        return vh_fail(1, "vh_launch_linear_planes_norm: colA=%d: one partial sum per 128 columns, at most 16 per row", colA);
    if ((((uintptr_t)output | (uintptr_t)weight_planes | (uintptr_t)input_planes | (uintptr_t)colsum | (uintptr_t)bias_folded) & 15) != 0 ||
        ((uintptr_t)row_stats & 7) != 0)
        return vh_fail(1, "vh_launch_linear_planes_norm: pointers must be 16-byte aligned (row_stats: 8)");
    if ((size_t)rowA * 64 > 0xffffffffull)
        return vh_fail(1, "vh_launch_linear_planes_norm: rowA=%d too large", rowA);
    P3Params p = {};
    p.A = static_cast<const char *>(input_planes);
    p.W = static_cast<const char *>(weight_planes);
    p.bias = bias_folded; p.colsum = colsum; p.stats = row_stats; p.eps = eps; p.C = output;
    p.row_begin = 0; p.row_end = rowA; p.a_rows = rowA; p.c_rows = rowA;
    p.N = colB; p.K = colA;
    hipStream_t st = (hipStream_t)s;
    if (parts == 3) {
        if (doGelu)
            return launch_p3<EPI_NORM_GELU, OUT_PLANES, 3>(st, p, 0);
        return output_planes == 1 ? launch_p3<EPI_NORM, OUT_PLANES, 3>(st, p, 0) : launch_p3<EPI_NORM, OUT_F32, 3>(st, p, 0);
    }
    if (doGelu)
        return launch_p3<EPI_NORM_GELU, OUT_PLANES, 1>(st, p, 0);
    return output_planes == 2 ? launch_p3<EPI_NORM, OUT_PLANES_H, 1>(st, p, 0)
         : output_planes == 1 ? launch_p3<EPI_NORM, OUT_PLANES, 1>(st, p, 0) : launch_p3<EPI_NORM, OUT_F32, 1>(st, p, 0);
}

extern "C" int vh_launch_linear_planes_norm(vh_stream_t s, void *output, int output_planes, const void *weight_planes,
                                            const void *input_planes, const float *row_stats, const float *colsum,
                                            const float *bias_folded, double eps, int rowA, int colA, int colB, int doGelu)
{
    return linear_planes_norm(s, output, output_planes, weight_planes, input_planes, row_stats, colsum, bias_folded, eps, rowA, colA, colB,
                              doGelu, 1);
}

/* the same on three-part planes (the fp32 path as a LAB VARIANT, $VIT_HIP_LN_FOLD=1): exact six-product arithmetic, the
 * un-normalised rows as the operand; output_planes 0 (fp32 rows) or 1 (three-part planes) */
extern "C" int vh_launch_linear_p3_norm(vh_stream_t s, void *output, int output_planes, const void *weight_planes3,
                                        const void *input_planes3, const float *row_stats, const float *colsum,
                                        const float *bias_folded, double eps, int rowA, int colA, int colB, int doGelu)
{
    return linear_planes_norm(s, output, output_planes, weight_planes3, input_planes3, row_stats, colsum, bias_folded, eps, rowA, colA, colB,
                              doGelu, 3);
}

/* output = residual + A W^T + b (fp32 rows, in place allowed) AND the same rows as the next projection's operand:
 * one-part bf16 planes [colB/32][rowA][32] (operand_scales NULL) or MX values + scales, plus the rows' partial sums
 * row_stats_out [colB/128][rowA][2] for that projection's folded LayerNorm. */
static int linear_planes_resid_norm(vh_stream_t s, float *output, const void *weight_planes, const void *input_planes, const float *bias,
                                    const float *residual, int rowA, int colA, int colB, void *operand_out, void *operand_scales_out,
                                    float *row_stats_out, int parts)
{
    if (!output || !weight_planes || !input_planes || !bias || !residual || !operand_out || !row_stats_out)
        return vh_fail(1, "vh_launch_linear_planes_resid_norm: null pointer argument");
    const int kstep = parts == 3 ? 64 : 64 * P1_KG;
    if (rowA <= 0 || colA <= 0 || colB <= 0 || colA % kstep != 0 || colB % 128 != 0 || (parts == 3 && operand_scales_out))
        return vh_fail(1, "vh_launch_linear_planes_resid_norm: needs colA %% %d == 0 and colB %% 128 == 0 (%d,%d,%d)", kstep, rowA, colA, colB);
    if ((((uintptr_t)output | (uintptr_t)weight_planes | (uintptr_t)input_planes | (uintptr_t)bias | (uintptr_t)residual | (uintptr_t)operand_out) & 15) != 0 ||
        ((uintptr_t)row_stats_out & 7) != 0)
        return vh_fail(1, "vh_launch_linear_planes_resid_norm: pointers must be 16-byte aligned (row_stats_out: 8)");
    if ((size_t)rowA * 128 > 0xffffffffull)
        return vh_fail(1, "vh_launch_linear_planes_resid_norm: rowA=%d too large", rowA);
    P3Params p = {};
    p.A = static_cast<const char *>(input_planes);
    p.W = static_cast<const char *>(weight_planes);
    p.bias = bias; p.R = residual; p.C = output;
    p.oper = operand_out; p.oper_scales = operand_scales_out; p.stats_out = row_stats_out;
    p.row_begin = 0; p.row_end = rowA; p.a_rows = rowA; p.c_rows = rowA;
    p.N = colB; p.K = colA;
    hipStream_t st = (hipStream_t)s;
    const int small_only = colA < 2048;   /* as vh_launch_linear_planes: the N = K = E output projection */
    if (parts == 3)
        return launch_p3<EPI_RESID, OUT_F32_OPER, 3>(st, p, small_only);
    return operand_scales_out ? launch_p3<EPI_RESID, OUT_F32_OPER_MX, 1>(st, p, small_only)
                              : launch_p3<EPI_RESID, OUT_F32_OPER, 1>(st, p, small_only);
}

extern "C" int vh_launch_linear_planes_resid_norm(vh_stream_t s, float *output, const void *weight_planes, const void *input_planes,
                                                  const float *bias, const float *residual, int rowA, int colA, int colB,
                                                  void *operand_out, void *operand_scales_out, float *row_stats_out)
{
    return linear_planes_resid_norm(s, output, weight_planes, input_planes, bias, residual, rowA, colA, colB, operand_out,
                                    operand_scales_out, row_stats_out, 1);
}

/* the same on three-part planes (fp32 lab variant): fp32 rows + the exact three-part planes of the same rows + partial sums */
extern "C" int vh_launch_linear_p3_resid_norm(vh_stream_t s, float *output, const void *weight_planes3, const void *input_planes3,
                                              const float *bias, const float *residual, int rowA, int colA, int colB,
                                              void *operand_planes3_out, float *row_stats_out)
{
    return linear_planes_resid_norm(s, output, weight_planes3, input_planes3, bias, residual, rowA, colA, colB, operand_planes3_out,
                                    nullptr, row_stats_out, 3);
}

namespace {

/* eight consecutive k of row m (chunk c of K step kt) -> the NPL parts of planes[K/32][NPL][rows][32] */
template <int NPL>
__device__ __forceinline__ void store_parts8(char *planes, size_t kt, int rows, int m, int c, const f32x4 &u, const f32x4 &v)
{
    bf16x8 part[3];
    if (NPL == 3) {
        split8(u, v, part[0], part[1], part[2]);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            part[0][e] = (__bf16)(e < 4 ? u[e] : v[e - 4]);
    }
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
        *reinterpret_cast<f32x4 *>(planes + ((kt * NPL + pl) * rows + m) * 64 + 16 * c) = __builtin_bit_cast(f32x4, part[pl]);
}

/* One wave per weight row n (Linear layout [N][K]).  MODE 0: out[n][k] = w[n][k] * gamma[k] (the gamma-scaled matrix, fp32,
 * before the mode's rounding).  MODE 1: out[n] = bias[n] + sum_k beta[k] w[n][k] (the folded bias; double accumulation). */
template <int MODE>
__global__ __launch_bounds__(256) void fold_rows_kernel(const float *__restrict__ w, const float *__restrict__ vec, const float *__restrict__ bias,
                                                        float *__restrict__ out, int N, int K)
{
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N)
        return;
    const float *row = w + (size_t)n * K;
    if (MODE == 0) {
        for (int k = lane; k < K; k += 64)
            out[(size_t)n * K + k] = row[k] * vec[k];
    } else {
        double acc = 0.0;
        for (int k = lane; k < K; k += 64)
            acc += (double)vec[k] * (double)row[k];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1)
            acc += __shfl_xor(acc, m);
        if (lane == 0)
            out[n] = (float)((double)bias[n] + acc);
    }
}

/* colsum[n] = sum_k of the operand's OWN values (what the matrix cores multiply): bf16 planes [K/32][parts][N][32]
 * (scales == nullptr; parts 1, or 3: the parts of a value added) or MX values [K/128][N][128] with scales [K/128][4][N].
 * One wave per row n, double accumulation. */
__global__ __launch_bounds__(256) void colsum_operand_kernel(const char *__restrict__ values, const unsigned char *__restrict__ scales,
                                                            float *__restrict__ out, int N, int K, int parts)
{
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N)
        return;
    double acc = 0.0;
    if (!scales) {
        for (int k = lane; k < K; k += 64)
            for (int pl = 0; pl < parts; ++pl)
                acc += (double)(float)*reinterpret_cast<const __bf16 *>(values + (((size_t)(k >> 5) * parts + pl) * N + n) * 64 + 2 * (k & 31));
    } else {
        for (int k = lane; k < K; k += 64) {
            const int ks = k >> 7, blk = (k >> 5) & 3;
            const unsigned sb = scales[((size_t)ks * 4 + 2 * (blk & 1) + (blk >> 1)) * N + n];
            const unsigned char byte = (unsigned char)values[((size_t)ks * N + n) * 128 + (k & 127)];
            /* e4m3 -> fp32: sign, 4 exponent bits (bias 7), 3 significand bits; subnormals by value */
            const int e = (byte >> 3) & 15, m = byte & 7;
            float v = e ? ldexpf(1.0f + m * 0.125f, e - 7) : ldexpf(m * 0.125f, -6);
            if (byte & 0x80)
                v = -v;
            acc += (double)v * (double)ldexpf(1.0f, (int)sb - 127);
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        acc += __shfl_xor(acc, m);
    if (lane == 0)
        out[n] = (float)acc;
}

/* Token 0 of every image (class token + pos_embed[0], ViT_seq.c:90-93,114-117) for the folded path: the fp32 row, the
 * row as the next projection's operand and its partial sums.  One wave per (16 images, 128-column group g). */
template <int NPL>   /* parts of the planes operand (oper_scales == nullptr): 1 rounded, 3 exact */
__global__ __launch_bounds__(64) void cls_rows_operand_kernel(const float *__restrict__ cls, const float *__restrict__ pos,
                                                             float *__restrict__ tokens, char *__restrict__ oper,
                                                             unsigned char *__restrict__ oper_scales, float *__restrict__ stats,
                                                             int n_images, int tokens_per_image, int E, int c_rows)
{
    /* lane (l15 = lane & 15, j = lane >> 4) <-> the epilogue's layout: 8 consecutive columns 128 g + 32 s + 8 j .. + 7 for
     * s = 0..3 in turn; the 16 lanes l15 of a lane group work on 16 different images */
    const int lane = threadIdx.x, l15 = lane & 15, j = lane >> 4;
    const int g = blockIdx.y, b = blockIdx.x * 16 + l15;
    const bool live = b < n_images;
    const size_t row = (size_t)(live ? b : n_images - 1) * tokens_per_image;
    float su = 0.0f, sq = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int col = 128 * g + 32 * s + 8 * j;
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(cls + col) + *reinterpret_cast<const f32x4 *>(pos + col);
        const f32x4 hi = *reinterpret_cast<const f32x4 *>(cls + col + 4) + *reinterpret_cast<const f32x4 *>(pos + col + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = e < 4 ? lo[e] : hi[e - 4];
            su += v;
            sq += v * v;
        }
        if (live) {
            *reinterpret_cast<f32x4 *>(tokens + row * E + col) = lo;
            *reinterpret_cast<f32x4 *>(tokens + row * E + col + 4) = hi;
        }
        if (!oper_scales) {
            if (live)
                store_parts8<NPL>(oper, (size_t)(col >> 5), c_rows, (int)row, j, lo, hi);
        } else {
            mx_store_block8(lo, hi, oper, oper_scales, c_rows, row, 128 * g + 32 * s, j, live);
        }
    }
    su += __shfl_xor(su, 16);
    sq += __shfl_xor(sq, 16);
    su += __shfl_xor(su, 32);
    sq += __shfl_xor(sq, 32);
    if (j == 0 && live)
        *reinterpret_cast<f32x2 *>(stats + ((size_t)g * c_rows + row) * 2) = f32x2{su, sq};
}

} // namespace

/* out[n][k] = weight[n][k] * gamma[k]: the LayerNorm scale moved into the columns of the projection behind it */
extern "C" int vh_launch_fold_gamma(vh_stream_t s, const float *weight, const float *gamma, float *out, int out_features, int in_features)
{
    if (!weight || !gamma || !out || out_features <= 0 || in_features <= 0)
        return vh_fail(1, "vh_launch_fold_gamma: bad argument");
    hipLaunchKernelGGL(fold_rows_kernel<0>, dim3((out_features + 3) / 4), dim3(256), 0, (hipStream_t)s, weight, gamma, (const float *)nullptr, out,
                       out_features, in_features);
    VH_LAUNCH_CHECK("fold_rows_kernel");
    return 0;
}

/* out[n] = bias[n] + sum_k beta[k] weight[n][k]: the LayerNorm shift carried through the projection (ORIGINAL fp32 weights) */
extern "C" int vh_launch_fold_bias(vh_stream_t s, const float *weight, const float *beta, const float *bias, float *out, int out_features,
                                   int in_features)
{
    if (!weight || !beta || !bias || !out || out_features <= 0 || in_features <= 0)
        return vh_fail(1, "vh_launch_fold_bias: bad argument");
    hipLaunchKernelGGL(fold_rows_kernel<1>, dim3((out_features + 3) / 4), dim3(256), 0, (hipStream_t)s, weight, beta, bias, out, out_features,
                       in_features);
    VH_LAUNCH_CHECK("fold_rows_kernel");
    return 0;
}

/* out[n] = sum_k of the values the operand holds for weight row n: one-part bf16 planes (scales NULL) or MX values + scales */
static int colsum_operand(vh_stream_t s, const void *values, const void *scales, float *out, int out_features, int in_features, int parts)
{
    if (!values || !out || out_features <= 0 || in_features <= 0 || in_features % (scales ? 128 : 32) != 0)
        return vh_fail(1, "vh_launch_colsum_operand: bad argument");
    hipLaunchKernelGGL(colsum_operand_kernel, dim3((out_features + 3) / 4), dim3(256), 0, (hipStream_t)s, static_cast<const char *>(values),
                       static_cast<const unsigned char *>(scales), out, out_features, in_features, parts);
    VH_LAUNCH_CHECK("colsum_operand_kernel");
    return 0;
}

extern "C" int vh_launch_colsum_operand(vh_stream_t s, const void *values, const void *scales, float *out, int out_features, int in_features)
{
    return colsum_operand(s, values, scales, out, out_features, in_features, 1);
}

/* the same for three-part planes [K/32][3][N][32] (the fp32 path's weights): the parts of a value added */
extern "C" int vh_launch_colsum_planes3(vh_stream_t s, const void *planes3, float *out, int out_features, int in_features)
{
    return colsum_operand(s, planes3, nullptr, out, out_features, in_features, 3);
}

/* ---- patch embedding on one-part planes (the reduced modes' conv_proj; replaces conv2d.cl:1-80 for them) ----
 * The fp32 path gathers patch rows on load and splits both operands in the K loop (gemm_mfma.hip, A_PATCH).  Where the
 * projections take bf16 operands anyway, an im2row producer writes the patches as one-part planes -- pixels rounded
 * to bf16 once -- and the patch embedding is the planes GEMM with a token-row epilogue: K padded with zeros to the
 * one-part K step (ViT-H/14: 3*14*14 = 588 -> 640), conv weights padded and rounded alike at context creation. */
namespace {

/* Thread i writes bytes [16 i, 16 i + 16) of planes[Kp/32][1][n_rows][32]: (K step kt, row m, chunk c) with c fastest
 * -- stores are contiguous; for patch % 8 == 0 a lane's eight k are 32 contiguous bytes of one image row and the
 * two chunks of a (row, kh) pair up to 64, consecutive patches of an image row to runs of a KiB. */
template <int NPL>   /* 1: pixels rounded to bf16 (reduced modes); 3: the exact three-part split (the fp32 path) */
__global__ void im2row_planes_kernel(const float *__restrict__ images, char *__restrict__ planes, int n_rows, int chans,
                                     int img, int patch, int grid, int K, int Kp)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_rows * (Kp >> 3))
        return;
    const int c = (int)(i & 3);
    const size_t rm = i >> 2;
    const int kt = (int)(rm / n_rows), m = (int)(rm - (size_t)kt * n_rows);
    const int k0 = 32 * kt + 8 * c;
    const int np = grid * grid, b = m / np, pp = m - b * np, oh = pp / grid, ow = pp - oh * grid;
    const float *base = images + ((size_t)b * chans * img + (size_t)oh * patch) * img + (size_t)ow * patch;
    const int pp2 = patch * patch;
    f32x4 u = {0.0f, 0.0f, 0.0f, 0.0f}, v = u;
    if ((patch & 7) == 0 && (img & 3) == 0 && k0 + 8 <= K) {
        const int ic = k0 / pp2, rem = k0 - ic * pp2, kh = rem / patch, kw = rem - kh * patch;
        const float *src = base + ((size_t)ic * img + kh) * img + kw;
        u = *reinterpret_cast<const f32x4 *>(src);
        v = *reinterpret_cast<const f32x4 *>(src + 4);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            float x = 0.0f;
            if (k < K) {
                const int ic = k / pp2, rem = k - ic * pp2, kh = rem / patch, kw = rem - kh * patch;
                x = base[((size_t)ic * img + kh) * img + kw];
            }
            if (e < 4)
                u[e] = x;
            else
                v[e - 4] = x;
        }
    }
    store_parts8<NPL>(planes, (size_t)kt, n_rows, m, c, u, v);
}

/* fp32 [rows][K] -> planes [Kp/32][NPL][rows][32], zero beyond K (the conv weights, once) */
template <int NPL>
__global__ void pad_rows_planes_kernel(const float *__restrict__ in, char *__restrict__ planes, int rows, int K, int Kp)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * (Kp >> 3))
        return;
    const int c = (int)(i & 3);
    const size_t rm = i >> 2;
    const int kt = (int)(rm / rows), r = (int)(rm - (size_t)kt * rows);
    f32x4 u, v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 32 * kt + 8 * c + e;
        const float x = k < K ? in[(size_t)r * K + k] : 0.0f;
        if (e < 4)
            u[e] = x;
        else
            v[e - 4] = x;
    }
    store_parts8<NPL>(planes, (size_t)kt, rows, r, c, u, v);
}

} // namespace

extern "C" int vh_patch_planes_k(int in_chans, int patch_size)
{
    const int K = in_chans * patch_size * patch_size, step = 64 * P1_KG;
    return (K + step - 1) / step * step;
}

extern "C" int vh_launch_conv_weight_planes_parts(vh_stream_t s, const float *conv_w, void *planes, int embed_dim, int in_chans,
                                                  int patch_size, int parts)
{
    if (!conv_w || !planes || embed_dim <= 0 || in_chans <= 0 || patch_size <= 0 || ((uintptr_t)planes & 15) || (parts != 1 && parts != 3))
        return vh_fail(1, "vh_launch_conv_weight_planes: bad argument");
    const int K = in_chans * patch_size * patch_size, Kp = vh_patch_planes_k(in_chans, patch_size);
    const size_t threads = (size_t)embed_dim * (Kp / 8);
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (parts == 3)
        hipLaunchKernelGGL(pad_rows_planes_kernel<3>, grid, dim3(256), 0, (hipStream_t)s, conv_w, static_cast<char *>(planes), embed_dim, K, Kp);
    else
        hipLaunchKernelGGL(pad_rows_planes_kernel<1>, grid, dim3(256), 0, (hipStream_t)s, conv_w, static_cast<char *>(planes), embed_dim, K, Kp);
    VH_LAUNCH_CHECK("pad_rows_planes_kernel");
    return 0;
}

extern "C" int vh_launch_conv_weight_planes(vh_stream_t s, const float *conv_w, void *planes, int embed_dim, int in_chans,
                                            int patch_size)
{
    return vh_launch_conv_weight_planes_parts(s, conv_w, planes, embed_dim, in_chans, patch_size, 1);
}

static int patch_embed_planes(vh_stream_t s, const float *images, const void *conv_w_planes, const float *conv_b,
                              const float *cls_token, const float *pos_embed, float *tokens, int n_images, int in_chans,
                              int img_size, int patch_size, int embed_dim, void *workspace, size_t workspace_bytes,
                              void *operand_out, void *operand_scales_out, float *row_stats_out, int parts = 1)
{
    if (!images || !conv_w_planes || !conv_b || !cls_token || !pos_embed || !tokens || !workspace)
        return vh_fail(1, "vh_launch_patch_embed_planes: null pointer argument");
    if (n_images <= 0 || in_chans <= 0 || img_size <= 0 || patch_size <= 0 || embed_dim <= 0 || img_size % patch_size != 0 ||
        embed_dim % 128 != 0)
        return vh_fail(1, "vh_launch_patch_embed_planes: bad geometry (embed_dim %% 128 == 0)");
    const int grid = img_size / patch_size, K = in_chans * patch_size * patch_size, Kp = vh_patch_planes_k(in_chans, patch_size);
    const long M = (long)n_images * grid * grid;
    const long c_rows = (long)n_images * (grid * grid + 1);
    if ((parts != 1 && parts != 3) || (parts == 3 && operand_scales_out))
        return vh_fail(1, "vh_launch_patch_embed_planes: parts must be 1 (bf16 operands) or 3 (exact fp32 split: planes operand only)");
    if (c_rows * 128 > 0xffffffffl || workspace_bytes < (size_t)M * Kp * 2 * parts ||
        (((uintptr_t)workspace | (uintptr_t)conv_w_planes | (uintptr_t)images | (uintptr_t)conv_b | (uintptr_t)pos_embed | (uintptr_t)tokens |
          (uintptr_t)cls_token | (uintptr_t)operand_out) & 15) || ((uintptr_t)row_stats_out & 7))
        return vh_fail(1, "vh_launch_patch_embed_planes: needs %zu bytes of 16-byte aligned workspace, aligned pointers", (size_t)M * Kp * 2 * parts);
    if (operand_out && !row_stats_out)
        return vh_fail(1, "vh_launch_patch_embed_planes_norm: operand without row statistics");
    hipStream_t st = (hipStream_t)s;
    if (operand_out) {
        if (parts == 3)
            hipLaunchKernelGGL(cls_rows_operand_kernel<3>, dim3((n_images + 15) / 16, embed_dim / 128), dim3(64), 0, st, cls_token, pos_embed,
                               tokens, static_cast<char *>(operand_out), static_cast<unsigned char *>(operand_scales_out), row_stats_out,
                               n_images, grid * grid + 1, embed_dim, (int)c_rows);
        else
            hipLaunchKernelGGL(cls_rows_operand_kernel<1>, dim3((n_images + 15) / 16, embed_dim / 128), dim3(64), 0, st, cls_token, pos_embed,
                               tokens, static_cast<char *>(operand_out), static_cast<unsigned char *>(operand_scales_out), row_stats_out,
                               n_images, grid * grid + 1, embed_dim, (int)c_rows);
        VH_LAUNCH_CHECK("cls_rows_operand_kernel");
    } else if (int rc = vh_cls_rows(st, cls_token, pos_embed, tokens, n_images, grid * grid + 1, embed_dim))
        return rc;
    const size_t threads = (size_t)M * (Kp / 8);
    if (parts == 3)
        hipLaunchKernelGGL(im2row_planes_kernel<3>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, images,
                           static_cast<char *>(workspace), (int)M, in_chans, img_size, patch_size, grid, K, Kp);
    else
        hipLaunchKernelGGL(im2row_planes_kernel<1>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, images,
                           static_cast<char *>(workspace), (int)M, in_chans, img_size, patch_size, grid, K, Kp);
    VH_LAUNCH_CHECK("im2row_planes_kernel");
    P3Params p = {};
    p.A = static_cast<const char *>(workspace);
    p.W = static_cast<const char *>(conv_w_planes);
    p.bias = conv_b; p.C = tokens; p.pos = pos_embed;
    p.row_begin = 0; p.row_end = (int)M; p.a_rows = (int)M; p.c_rows = (int)c_rows;
    p.N = embed_dim; p.K = Kp;
    p.np = grid * grid; p.tokens = grid * grid + 1;
    p.oper = operand_out; p.oper_scales = operand_scales_out; p.stats_out = row_stats_out;
    const int small_only = Kp < 2048;   /* the shape of the output projection: small tiles (measured there) */
    if (parts == 3)
        return operand_out ? launch_p3<EPI_PATCH, OUT_F32_OPER, 3>(st, p, small_only) : launch_p3<EPI_PATCH, OUT_F32, 3>(st, p, small_only);
    if (!operand_out)
        return launch_p3<EPI_PATCH, OUT_F32, 1>(st, p, small_only);
    return operand_scales_out ? launch_p3<EPI_PATCH, OUT_F32_OPER_MX, 1>(st, p, small_only) : launch_p3<EPI_PATCH, OUT_F32_OPER, 1>(st, p, small_only);
}

extern "C" int vh_launch_patch_embed_planes(vh_stream_t s, const float *images, const void *conv_w_planes, const float *conv_b,
                                            const float *cls_token, const float *pos_embed, float *tokens, int n_images,
                                            int in_chans, int img_size, int patch_size, int embed_dim, void *workspace,
                                            size_t workspace_bytes)
{
    return patch_embed_planes(s, images, conv_w_planes, conv_b, cls_token, pos_embed, tokens, n_images, in_chans, img_size, patch_size,
                              embed_dim, workspace, workspace_bytes, nullptr, nullptr, nullptr);
}

/* The fp32 path's patch embedding on the planes kernel (replaces conv2d.cl:1-80 there): the im2row producer writes the exact
 * three-part split of the pixels (workspace: M * Kp * 6 bytes), conv_w_planes3 = vh_launch_conv_weight_planes_parts(.., 3);
 * six products per block as every other fp32 projection -- nothing is split inside a K loop any more. */
extern "C" int vh_launch_patch_embed_planes3(vh_stream_t s, const float *images, const void *conv_w_planes3, const float *conv_b,
                                             const float *cls_token, const float *pos_embed, float *tokens, int n_images,
                                             int in_chans, int img_size, int patch_size, int embed_dim, void *workspace,
                                             size_t workspace_bytes)
{
    return patch_embed_planes(s, images, conv_w_planes3, conv_b, cls_token, pos_embed, tokens, n_images, in_chans, img_size, patch_size,
                              embed_dim, workspace, workspace_bytes, nullptr, nullptr, nullptr, 3);
}

/* ... also leaving the token rows as three-part planes [E/32][3][n*tokens][32] + partial sums (fp32 lab variant of the fold) */
extern "C" int vh_launch_patch_embed_planes3_norm(vh_stream_t s, const float *images, const void *conv_w_planes3, const float *conv_b,
                                                  const float *cls_token, const float *pos_embed, float *tokens, int n_images,
                                                  int in_chans, int img_size, int patch_size, int embed_dim, void *workspace,
                                                  size_t workspace_bytes, void *operand_planes3_out, float *row_stats_out)
{
    if (!operand_planes3_out || !row_stats_out)
        return vh_fail(1, "vh_launch_patch_embed_planes3_norm: null pointer argument");
    return patch_embed_planes(s, images, conv_w_planes3, conv_b, cls_token, pos_embed, tokens, n_images, in_chans, img_size, patch_size,
                              embed_dim, workspace, workspace_bytes, operand_planes3_out, nullptr, row_stats_out, 3);
}

/* The same, also leaving the token rows as the first projection's operand (one-part bf16 planes [E/32][n*tokens][32], or MX
 * values + scales when operand_scales_out is given) and their partial sums row_stats_out [E/128][n*tokens][2] for that
 * projection's folded LayerNorm (norm_fold.h) -- class-token rows included. */
extern "C" int vh_launch_patch_embed_planes_norm(vh_stream_t s, const float *images, const void *conv_w_planes, const float *conv_b,
                                                 const float *cls_token, const float *pos_embed, float *tokens, int n_images,
                                                 int in_chans, int img_size, int patch_size, int embed_dim, void *workspace,
                                                 size_t workspace_bytes, void *operand_out, void *operand_scales_out,
                                                 float *row_stats_out)
{
    if (!operand_out || !row_stats_out)
        return vh_fail(1, "vh_launch_patch_embed_planes_norm: null pointer argument");
    return patch_embed_planes(s, images, conv_w_planes, conv_b, cls_token, pos_embed, tokens, n_images, in_chans, img_size, patch_size,
                              embed_dim, workspace, workspace_bytes, operand_out, operand_scales_out, row_stats_out);
}
