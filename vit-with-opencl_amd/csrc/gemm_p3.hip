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

#ifndef P1_KG
#define P1_KG 2   /* one-part operands: 32-deep K groups per LDS stage (one barrier per 32 * P1_KG k) */
#endif
enum { EPI_NONE = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_PATCH = 3 };
enum { OUT_F32 = 0, OUT_PLANES = 1, OUT_PLANES_H = 2 };   /* _H: one-part planes of fp16 (the reduced modes' Q|K|V) */

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
    int lab_lo, lab_hi, lab_cycles;   /* LAB bits 512 / 1024 only (lab_stagger_start) */
    unsigned *lab_slots;
};

/* NPL = parts per value: 3 = the exact fp32 split (six products per block, the default fp32 path);
 * 1 = operands rounded to bf16 by their producers (one product per block: BASELINE config 3's
 * bf16-operand mode).  With one part a W fragment feeds 2 MFMAs instead of 12, so a stage holds two
 * 32-deep K groups (one barrier per 64 k) and four fragments are in flight in registers.
 * LAB (tools/p3_lab.hip only; 0 in the library): bit 0 skips the W fragment reads after the first step, bit 1
 * the W DMA after the prologue, bit 2 the A loads after the prologue, bit 3 the barrier, bit 4 makes every
 * workgroup load the A rows of tile 0, bit 5 the W rows of tile 0 (operands served by L2 alone), bit 6 drops
 * the epilogue's stores -- throw-away
 * ablations that price each data movement; their results are wrong by construction; bit 7 keeps the one-part
 * residual in the epilogue (the form before R_IN_ACC). */
template <int NW, int BN, int EPI, int OUTK, int NPL = 3, int LAB = 0>
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
    static_assert(NPL == 1 || NPL == 3, "parts per value");
    static_assert(JT % 2 == 0 && (KG * NPL * BN / 16) % NW == 0 && F % RING == 0, "tile shape");
    typedef bf16x8 frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    if (LAB & (512 | 1024))
        lab_stagger_start(p.lab_lo, p.lab_hi, p.lab_cycles, (LAB & 1024) ? p.lab_slots : nullptr);
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

    /* W fragment j = 2s + b.  Planes out: MFMA row l15 = LDS row 32s + 8*(l15 >> 2) + 4b + (l15 & 3), so that a lane
     * ends up with EIGHT consecutive columns (one 16-byte store of 16-bit values per part).  fp32 rows out (NATURAL):
     * MFMA row l15 = LDS row 32s + 16b + l15 -- a lane holds columns 16j + 4q .. +3 of fragment j and the four lanes of
     * a row write 64 contiguous bytes per store instruction (with the permuted rows a store instruction writes every
     * other 16 bytes of a line).  Every output element sums the same products in the same order either way. */
    constexpr bool NATURAL = OUTK == OUT_F32 && !(LAB & 256);
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
    constexpr bool R_IN_ACC = EPI == EPI_RESID && NPL == 1 && !(LAB & 128);
    f32x4 acc[2][JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        const int col = n0 + frag_col(j);
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(p.bias + col);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            acc[i][j] = bv;
            if (R_IN_ACC) {   /* rows past the end re-read the last row (never stored) */
                const int row = min(m0 + 32 * wave + 16 * i + l15, p.row_end - 1);
                acc[i][j] = *reinterpret_cast<const f32x4 *>(p.R + (size_t)row * p.N + col) + bv;
            }
        }
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
        if (!(LAB & 4))
            load_a(al, more1 ? kt + 1 : kt);   /* unconditional (no copies at a join): the last step re-reads its own */
#pragma unroll
        for (int f = 0; f <= F - RING; ++f) {
            if (!(LAB & 1) || kt == 0)
                read_w(w[(f + RING - 1) % RING], cur, f + RING - 1);
            mfma_frag(au, w[f % RING], f);
            interleave();
        }
        if (!(LAB & 8))
            __syncthreads();   /* stage kt read by every wave (its last fragments are in registers); stage kt+1 has landed */
        if (more2 && !(LAB & 2))
            dma_w(kt & 1, kt + 2);
#pragma unroll
        for (int f = F - RING + 1; f < F; ++f) {
            if (more1 && (!(LAB & 1) || kt == 0))
                read_w(w[(f + RING - 1) % RING], nxt, f + RING - 1 - F);
            mfma_frag(au, w[f % RING], f);
            interleave();
        }
    };

    dma_w(0, 0);
    load_a(a0, 0);
    __syncthreads();
    if (nk > 1)
        dma_w(1, 1);
#pragma unroll
    for (int f = 0; f < RING - 1; ++f)
        read_w(w[f], smem, f);
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
        size_t orow = (size_t)row;
        const float *posrow = nullptr;
        if (EPI == EPI_PATCH) {
            const int b = row / p.np, pp = row - b * p.np;
            orow = (size_t)b * p.tokens + 1 + pp;
            posrow = p.pos + (size_t)(1 + pp) * p.N;
        }
#pragma unroll
        for (int s = 0; s < JT / 2; ++s) {
            const int col = n0 + frag_col(2 * s), col_hi = n0 + frag_col(2 * s + 1);   /* planes: col_hi = col + 4 */
            f32x4 lo = acc[i][2 * s], hi = acc[i][2 * s + 1];
            if (EPI == EPI_PATCH) {
                lo = lo + *reinterpret_cast<const f32x4 *>(posrow + col);
                hi = hi + *reinterpret_cast<const f32x4 *>(posrow + col_hi);
            }
            if (EPI == EPI_GELU) {
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
            if (LAB & 64) {   /* keep the values alive, store nothing */
                asm volatile("" ::"v"(lo), "v"(hi));
            } else if (OUTK == OUT_PLANES || OUTK == OUT_PLANES_H) {
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
                float *cp = static_cast<float *>(p.C) + orow * p.N;
                *reinterpret_cast<f32x4 *>(cp + col) = lo;
                *reinterpret_cast<f32x4 *>(cp + col_hi) = hi;
            }
        }
    }
}

template <int NW, int BN, int EPI, int OUTK, int NPL>
int launch_p3_tile(hipStream_t st, P3Params p)
{
    constexpr int LDS = 2 * (NPL == 3 ? 1 : P1_KG) * NPL * BN * 64;
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
 * the choice. */
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

/* ---- patch embedding on one-part planes (the reduced modes' conv_proj; replaces conv2d.cl:1-80 for them) ----
 * The fp32 path gathers patch rows on load and splits both operands in the K loop (gemm_mfma.hip, A_PATCH).  Where the
 * projections take bf16 operands anyway, an im2row producer writes the patches as one-part planes -- pixels rounded
 * to bf16 once -- and the patch embedding is the planes GEMM with a token-row epilogue: K padded with zeros to the
 * one-part K step (ViT-H/14: 3*14*14 = 588 -> 640), conv weights padded and rounded alike at context creation. */
namespace {

/* Thread i writes bytes [16 i, 16 i + 16) of planes[Kp/32][1][n_rows][32]: (K step kt, row m, chunk c) with c fastest
 * -- stores are contiguous; for patch % 8 == 0 a lane's eight k are 32 contiguous bytes of one image row and the
 * two chunks of a (row, kh) pair up to 64, consecutive patches of an image row to runs of a KiB. */
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
    bf16x8 out;
    if ((patch & 7) == 0 && (img & 3) == 0 && k0 + 8 <= K) {
        const int ic = k0 / pp2, rem = k0 - ic * pp2, kh = rem / patch, kw = rem - kh * patch;
        const float *src = base + ((size_t)ic * img + kh) * img + kw;
        const f32x4 u = *reinterpret_cast<const f32x4 *>(src), v = *reinterpret_cast<const f32x4 *>(src + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            out[e] = (__bf16)(e < 4 ? u[e] : v[e - 4]);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            float v = 0.0f;
            if (k < K) {
                const int ic = k / pp2, rem = k - ic * pp2, kh = rem / patch, kw = rem - kh * patch;
                v = base[((size_t)ic * img + kh) * img + kw];
            }
            out[e] = (__bf16)v;
        }
    }
    *reinterpret_cast<f32x4 *>(planes + 16 * i) = __builtin_bit_cast(f32x4, out);
}

/* fp32 [rows][K] -> one-part planes [Kp/32][1][rows][32], zero beyond K (the conv weights, once) */
__global__ void pad_rows_planes_kernel(const float *__restrict__ in, char *__restrict__ planes, int rows, int K, int Kp)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * (Kp >> 3))
        return;
    const int c = (int)(i & 3);
    const size_t rm = i >> 2;
    const int kt = (int)(rm / rows), r = (int)(rm - (size_t)kt * rows);
    bf16x8 out;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 32 * kt + 8 * c + e;
        out[e] = (__bf16)(k < K ? in[(size_t)r * K + k] : 0.0f);
    }
    *reinterpret_cast<f32x4 *>(planes + 16 * i) = __builtin_bit_cast(f32x4, out);
}

} // namespace

extern "C" int vh_patch_planes_k(int in_chans, int patch_size)
{
    const int K = in_chans * patch_size * patch_size, step = 64 * P1_KG;
    return (K + step - 1) / step * step;
}

extern "C" int vh_launch_conv_weight_planes(vh_stream_t s, const float *conv_w, void *planes, int embed_dim, int in_chans,
                                            int patch_size)
{
    if (!conv_w || !planes || embed_dim <= 0 || in_chans <= 0 || patch_size <= 0 || ((uintptr_t)planes & 15))
        return vh_fail(1, "vh_launch_conv_weight_planes: bad argument");
    const int K = in_chans * patch_size * patch_size, Kp = vh_patch_planes_k(in_chans, patch_size);
    const size_t threads = (size_t)embed_dim * (Kp / 8);
    hipLaunchKernelGGL(pad_rows_planes_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)s, conv_w,
                       static_cast<char *>(planes), embed_dim, K, Kp);
    VH_LAUNCH_CHECK("pad_rows_planes_kernel");
    return 0;
}

extern "C" int vh_launch_patch_embed_planes(vh_stream_t s, const float *images, const void *conv_w_planes, const float *conv_b,
                                            const float *cls_token, const float *pos_embed, float *tokens, int n_images,
                                            int in_chans, int img_size, int patch_size, int embed_dim, void *workspace,
                                            size_t workspace_bytes)
{
    if (!images || !conv_w_planes || !conv_b || !cls_token || !pos_embed || !tokens || !workspace)
        return vh_fail(1, "vh_launch_patch_embed_planes: null pointer argument");
    if (n_images <= 0 || in_chans <= 0 || img_size <= 0 || patch_size <= 0 || embed_dim <= 0 || img_size % patch_size != 0 ||
        embed_dim % 128 != 0)
        return vh_fail(1, "vh_launch_patch_embed_planes: bad geometry (embed_dim %% 128 == 0)");
    const int grid = img_size / patch_size, K = in_chans * patch_size * patch_size, Kp = vh_patch_planes_k(in_chans, patch_size);
    const long M = (long)n_images * grid * grid;
    if (M * 64 > 0xffffffffl || workspace_bytes < (size_t)M * Kp * 2 ||
        (((uintptr_t)workspace | (uintptr_t)conv_w_planes | (uintptr_t)images | (uintptr_t)conv_b | (uintptr_t)pos_embed | (uintptr_t)tokens) & 15))
        return vh_fail(1, "vh_launch_patch_embed_planes: needs %zu bytes of 16-byte aligned workspace, aligned pointers", (size_t)M * Kp * 2);
    hipStream_t st = (hipStream_t)s;
    if (int rc = vh_cls_rows(st, cls_token, pos_embed, tokens, n_images, grid * grid + 1, embed_dim))
        return rc;
    const size_t threads = (size_t)M * (Kp / 8);
    hipLaunchKernelGGL(im2row_planes_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, images,
                       static_cast<char *>(workspace), (int)M, in_chans, img_size, patch_size, grid, K, Kp);
    VH_LAUNCH_CHECK("im2row_planes_kernel");
    P3Params p = {};
    p.A = static_cast<const char *>(workspace);
    p.W = static_cast<const char *>(conv_w_planes);
    p.bias = conv_b; p.C = tokens; p.pos = pos_embed;
    p.row_begin = 0; p.row_end = (int)M; p.a_rows = (int)M;
    p.N = embed_dim; p.K = Kp;
    p.np = grid * grid; p.tokens = grid * grid + 1;
    return launch_p3<EPI_PATCH, OUT_F32, 1>(st, p, Kp < 2048);   /* the shape of the output projection: small tiles (measured there) */
}
