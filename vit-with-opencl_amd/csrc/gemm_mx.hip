/*
 * gemm_mx.hip -- the four big projections on block-scaled fp8 ("MX": OCP microscaling, e4m3 elements, one
 * e8m0 power-of-two scale per 32 consecutive K elements) with v_mfma_scale_f32_16x16x128_f8f6f4, which runs at
 * twice the bf16 rate (the non-scaled fp8 MFMA of gemm_mfma.hip runs AT the bf16 rate).  BASELINE config 5
 * ("fp8 weights (CDNA4 fp8 MFMA)"); no reference counterpart (the reference is fp32 throughout: ll.cl:7-86,
 * multihead.cl:3-63).  Opt-in precision VIT_PRECISION_MXFP8_GEMM.
 *
 * Operand format ("MX planes"), weights [N][K] and activations [rows][K] alike:
 *     values[K/128][rows][128]  e4m3 bytes     K step, row, element
 *     scales[K/128][4][rows]    e8m0 bytes     K step, lane group, row        (weights; activations: the same bytes
 *                                                                              at mx_act_scale_index, vit_kernels.h)
 * so a 16-row fragment of one K step is 2 KiB contiguous.  Block b (0..3) of a row's K step = its elements
 * 32b .. 32b+31.  The instruction's operand map, measured on the device (no ISA text at hand; tests pin it with
 * a numpy statement): lane l = (row l & 15, group j = l >> 4) supplies 32 bytes; its bytes 0-15 belong,
 * together with bytes 0-15 of group j ^ 1, to one scale block, its bytes 16-31 likewise to another; the scale of
 * the block {groups 2m, 2m+1; byte half h} is byte 0 of the scale VGPR of lane group m + 2h.  Hence
 *     lane (row, j):  bytes 0-15  = block 2(j>>1)   , half (j & 1)   -> offset 64(j>>1) + 16(j&1)
 *                     bytes 16-31 = block 2(j>>1)+1 , same half      -> + 32
 *                     scale VGPR  = scale of block 2(j&1) + (j>>1)  -> scales[..][j][row] holds exactly that
 * (two 16-byte loads and one byte load per fragment and lane, no shuffling).  A and B operands pair up by
 * (lane group, byte), so any assignment works as long as both sides use the same one.
 *
 * Kernel structure = gemm_p3.hip's: tile 256 x 256 (or 128 x 128), every wave owns 32 rows; A fragments go
 * HBM/L2 -> VGPR directly (coalesced thanks to the format), double-buffered one K step ahead; W goes by LDS-DMA
 * into [256][128 B] rows (16-byte chunks swizzled for the permuted fragment rows, see dma_w) plus its scales, two stages; W
 * fragments are taken through a ring of four in registers; one barrier per K step (128 k = 32 MFMAs per wave)
 * before the last ring's worth of MFMAs.  The same column permutation gives a lane eight consecutive output
 * columns: a planes epilogue quantises (block maximum over the four lanes of a row's 32 columns by two
 * shuffles) and stores 8 bytes per lane, plus the scale byte from one lane in four.
 */
#include "kernelHandler.h"
#include "vit_kernels.h"
#include "gemm_common.h"
#include "norm_fold.h"

#include <cstdint>
#include <cstdlib>

namespace {

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
typedef const __attribute__((address_space(1))) char *gchar_t;
typedef const __attribute__((address_space(1))) f32x4 *gvec_t;
typedef const __attribute__((address_space(1))) unsigned char *gbyte_t;
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

/* EPI_NORM / EPI_NORM_GELU: the LayerNorm in front of the projection folded into it (norm_fold.h) */
enum { EPI_NONE = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_NORM = 4, EPI_NORM_GELU = 5 };
/* _H: one-part fp16 planes [N/32][rows][32] (Q|K|V for attention_p3.hip); OUT_F32_OPER_MX: fp32 rows AND the same values as
 * the next projection's MX operand AND the rows' partial sums for its folded LayerNorm (output projection, fc2) */
enum { OUT_F32 = 0, OUT_MX = 1, OUT_PLANES_H = 2, OUT_F32_OPER_MX = 4 };

struct MxParams {
    const char *A, *As;       /* activation values [K/128][a_rows][128], scales [K/128][4][a_rows] */
    const char *W, *Ws;       /* weight values [K/128][N][128], scales [K/128][4][N] */
    const float *bias, *R;
    void *C, *Cs;             /* fp32 [a_rows][N]; or MX values [N/128][a_rows][128] + scales [N/128][4][a_rows] */
    int row_begin, row_end, N, K, a_rows, mtiles, ntiles;
    /* EPI_NORM*: colsum [N] of the gamma-scaled quantised weights, stats [K/128][a_rows][2] left by A's producer, eps */
    const float *colsum, *stats;
    double eps;
    /* OUT_F32_OPER_MX: the next projection's operand (values [N/128][a_rows][128], scales [N/128][4][a_rows]) and row sums */
    void *oper, *oper_scales;
    float *stats_out;         /* [N/128][a_rows][2] */
};

/* (The ablation copy of this kernel, with switches that remove one data movement each, is tools/mx_lab_kernel.inc.) */
template <int NW, int BN, int EPI, int OUTK>
__global__ __launch_bounds__(64 * NW, 2) void gemm_mx_kernel(const MxParams p)
{
    constexpr int BM = 32 * NW, JT = BN / 16, RING = 4;
    constexpr int VALS = BN * 128;                  /* bytes of W values per stage */
    constexpr int SCS = BN + 32;                    /* stride of the four lane groups' scale runs: 8 banks apart (below) */
    constexpr int STAGE = VALS + 4 * SCS;           /* + its scales [4][BN + 32] */
    constexpr int PW = BN / 8 / NW;                 /* 1-KiB value pieces (8 rows x 128 B) per wave and stage */
    constexpr bool NORM = EPI == EPI_NORM || EPI == EPI_NORM_GELU;
    constexpr bool GELU = EPI == EPI_GELU || EPI == EPI_NORM_GELU;
    constexpr bool OPER = OUTK == OUT_F32_OPER_MX;
    static_assert(JT % RING == 0 && (BN / 8) % NW == 0 && NW >= 4, "tile shape");
    static_assert(!OPER || (EPI == EPI_RESID && BN % 128 == 0), "operand producers: residual-stream epilogues");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tile = xcd_tile(blockIdx.x, p.mtiles * p.ntiles);
    const int m0 = p.row_begin + (tile / p.ntiles) * BM;
    const int n0 = (tile % p.ntiles) * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, j4 = lane >> 4;

    /* A fragment i: rows m0 + 32 wave + 16 i + l15 (clamped), bytes 64 (j>>1) + 16 (j&1) and + 32 */
    unsigned arow[2], aoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        arow[i] = (unsigned)min(m0 + 32 * wave + 16 * i + l15, p.row_end - 1);
        aoff[i] = arow[i] * 128u + 64u * (j4 >> 1) + 16u * (j4 & 1);
    }
    const size_t a_step = (size_t)p.a_rows * 128;
    const size_t w_step = (size_t)p.N * 128, ws_step = (size_t)p.N * 4;

    /* W DMA: value piece pc = rows 8pc .. 8pc+7 (lane fills physical chunk lane & 7 of row 8pc + (lane >> 3) with
     * logical chunk phys ^ f(row)); the scales of the stage: one 16-lane piece per lane group, from waves 0..3.
     * f(r) = 2 ((r >> 3) & 3) + ((r >> 1) & 1): a fragment reads the 16 rows 8a + c (+ 4b), a, c = 0..3 -- the permuted
     * order that gives a lane eight consecutive output columns -- and two 128-byte rows share a 256-byte bank row, so
     * the eight rows of one parity need eight different chunk positions: 2a + (c >> 1).  (Round 3's (r >> 1) & 7 was
     * made for 16 CONSECUTIVE rows and repeats on rows r, r + 16: SQ_LDS_BANK_CONFLICT 22-37 M cycles per launch,
     * profiles/r04_pmc_summary_fp8_mode.txt; the bf16 kernel's 64-byte rows never had the problem.)
     * With row = 8 pc + (lane >> 3): f = 2 (pc & 3) + ((lane >> 4) & 1). */
    const gchar_t wtile = (gchar_t)p.W + (size_t)n0 * 128;
    auto dma_w = [&](int stage, int kt) {
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int pc = wave * PW + i;
            gchar_t src = wtile + (size_t)kt * w_step + (size_t)pc * 1024;
            asm volatile("" : "+s"(src));
            const unsigned off = (unsigned)(lane >> 3) * 128u + 16u * ((lane & 7) ^ (2 * (pc & 3) + ((lane >> 4) & 1)));
            __builtin_amdgcn_global_load_lds((gptr_t)(src + off), (lptr_t)(smem + stage * STAGE + pc * 1024), 16, 0, 0);
        }
        /* scales: lane group jj's run of BN bytes -> LDS at VALS + jj (BN + 32), by wave jj (16 bytes per lane).  The 16
         * lanes of a group read scale bytes of rows 8a + c: dwords 0, 2, 4, 6 of the run -- with the runs BN apart (a
         * multiple of 256 bytes) all four groups hit the same four banks; 32 bytes of padding put them 8 banks apart. */
        if (wave < 4 && lane < BN / 16) {
            gchar_t src = (gchar_t)p.Ws + (size_t)kt * ws_step + (size_t)wave * p.N + n0 + 16 * lane;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + stage * STAGE + VALS + wave * SCS), 16, 0, 0);
        }
    };

    /* W fragment j = 2s + b, MFMA row l15 = LDS row 32s + 8 (l15 >> 2) + 4b + (l15 & 3); chunk c of row r sits at
     * c ^ f(r); this lane's chunks are 4 (j4 >> 1) + (j4 & 1) and + 2 */
    const int rl = 8 * (l15 >> 2) + (l15 & 3);
    const int c0 = 4 * (j4 >> 1) + (j4 & 1);
    unsigned wlo[2], whi[2], wsc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int r = rl + 4 * b, sw = 2 * ((r >> 3) & 3) + ((r >> 1) & 1);
        wlo[b] = (unsigned)r * 128u + 16u * (c0 ^ sw);
        whi[b] = (unsigned)r * 128u + 16u * ((c0 + 2) ^ sw);
        wsc[b] = (unsigned)VALS + (unsigned)j4 * SCS + (unsigned)r;
    }

    /* the residual goes into the accumulators with the bias -- (r + bias) + sum: its load runs under the prologue's DMA
     * and the epilogue is stores only (as gemm_p3.hip does for one-part operands; the rounding differs in the last
     * bits, far inside what e4m3 operands leave) */
    f32x4 acc[2][JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        const int col = n0 + 32 * (j >> 1) + 8 * j4 + 4 * (j & 1);
        const f32x4 bv = NORM ? f32x4{0.0f, 0.0f, 0.0f, 0.0f} : *reinterpret_cast<const f32x4 *>(p.bias + col);   /* NORM: bias in the epilogue */
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            acc[i][j] = bv;
            if (EPI == EPI_RESID)
                acc[i][j] = *reinterpret_cast<const f32x4 *>(p.R + (size_t)arow[i] * p.N + col) + bv;
        }
    }
    /* folded LayerNorm: 1/std and -mean/std of this lane's two rows, from the partial sums their producer left */
    float n_rstd[2] = {1.0f, 1.0f}, n_shift[2] = {0.0f, 0.0f};
    if (NORM) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            row_norm_terms(p.stats, p.K >> 7, p.a_rows, (int)arow[i], j4, p.K, p.eps, n_rstd[i], n_shift[i]);
    }

    /* a fragment = two 16-byte halves (kept apart until the MFMA call) and the lane's scale byte */
    i32x4 a0l[2], a0h[2], a1l[2], a1h[2], wl[RING], wh[RING];
    int a0s[2], a1s[2], ws[RING];
    unsigned a_scales4[2] = {0u, 0u};   /* this lane's A scales of four consecutive K steps (mx_act_scale_index), per row block */

    auto load_a = [&](i32x4 (&al)[2], i32x4 (&ah)[2], int (&as)[2], int kt) {
        gchar_t vb = (gchar_t)p.A + (size_t)kt * a_step;
        asm volatile("" : "+s"(vb));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            al[i] = __builtin_bit_cast(i32x4, *reinterpret_cast<gvec_t>(vb + aoff[i]));
            ah[i] = __builtin_bit_cast(i32x4, *reinterpret_cast<gvec_t>(vb + aoff[i] + 32));
            /* one dword = the scales of K steps 4 (kt >> 2) .. + 3 of (row, lane group), fetched every fourth step; the
             * instruction takes byte 0 of its scale register (op_sel 0) */
            if ((kt & 3) == 0)
                a_scales4[i] = *reinterpret_cast<const __attribute__((address_space(1))) unsigned *>(
                    (gchar_t)p.As + ((((size_t)(kt >> 2) * 4 + j4) * (size_t)p.a_rows + arow[i]) << 2));
            as[i] = (int)(a_scales4[i] >> (8 * (kt & 3)));
        }
    };
    auto read_w = [&](int slot, const char *stage, int j) {
        const int b = j & 1, s = j >> 1;
        wl[slot] = __builtin_bit_cast(i32x4, *reinterpret_cast<const f32x4 *>(stage + wlo[b] + s * 32 * 128));
        wh[slot] = __builtin_bit_cast(i32x4, *reinterpret_cast<const f32x4 *>(stage + whi[b] + s * 32 * 128));
        ws[slot] = *reinterpret_cast<const unsigned char *>(stage + wsc[b] + s * 32);
    };
    /* The instruction through inline asm with the accumulator tied ("+v"): with the builtin, hipcc (ROCm 7.2) does
     * not tie D to C for the scaled form, renames all 128 accumulators every K step and spills 300+ registers.
     * What the compiler does not do for an asm statement (guide 5.7) is handled here: `s_nop 1` in front covers a
     * VALU-written operand; the statements are volatile with a memory clobber, so MFMAs and LDS reads keep their
     * source order; a fragment slot is refilled only after the NEXT fragment's MFMAs (two MFMAs behind its last
     * reader); the epilogue waits out the last MFMA with explicit s_nops. */
    auto mfma_frag = [&](const i32x4 (&al)[2], const i32x4 (&ah)[2], const int (&as)[2], int slot, int j) {
        const i32x8 wv = __builtin_shufflevector(wl[slot], wh[slot], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int i = 0; i < 2; ++i) {  /* D[n][m]: W fragment as the MFMA's A operand */
            const i32x8 av = __builtin_shufflevector(al[i], ah[i], 0, 1, 2, 3, 4, 5, 6, 7);
            asm volatile("s_nop 1\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]"
                         : "+v"(acc[i][j])
                         : "v"(wv), "v"(av), "v"(ws[slot]), "v"(as[i])
                         : "memory");
        }
    };

    const int nk = p.K / 128;
    auto step = [&](const i32x4 (&ul)[2], const i32x4 (&uh)[2], const int (&us)[2], i32x4 (&nl)[2], i32x4 (&nh)[2], int (&ns)[2], int kt) {
        const char *cur = smem + (kt & 1) * STAGE, *nxt = smem + ((kt + 1) & 1) * STAGE;
        const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
        load_a(nl, nh, ns, more1 ? kt + 1 : kt);   /* unconditional (no copies at a join): the last step re-reads its own */
#pragma unroll
        for (int f = 0; f <= JT - RING; ++f) {
            mfma_frag(ul, uh, us, f % RING, f);
            read_w((f + RING - 1) % RING, cur, f + RING - 1);   /* the slot fragment f-1 was in */
        }
        __syncthreads();   /* stage kt read by every wave (its last fragments are in registers); stage kt+1 has landed */
        if (more2)
            dma_w(kt & 1, kt + 2);
#pragma unroll
        for (int f = JT - RING + 1; f < JT; ++f) {
            mfma_frag(ul, uh, us, f % RING, f);
            read_w((f + RING - 1) % RING, nxt, f + RING - 1 - JT);   /* unconditional: behind the last step it reads a stage nobody uses */
        }
    };

    dma_w(0, 0);
    load_a(a0l, a0h, a0s, 0);
    /* folded LayerNorm: the tile's column terms (colsum, folded bias: 2 x BN floats) go to LDS behind the two W stages;
     * the epilogue takes them by ds_read_b128 (from global memory they were a round trip to L2 per fragment pair) */
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
        read_w(f, smem, f);
    for (int kt = 0; kt < nk; kt += 2) {   /* nk is even (launcher) */
        step(a0l, a0h, a0s, a1l, a1h, a1s, kt);
        step(a1l, a1h, a1s, a0l, a0h, a0s, kt + 1);
    }

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   /* the last MFMA's result, before ordinary code reads it */

    /* Epilogue: fragment pair (2s, 2s+1) of row block i = 8 consecutive columns n0 + 32s + 8 j4 .. +7 of one row.
     * Fragment pairs outermost: the folded LayerNorm's column terms are loaded once and serve both row blocks. */
    /* OPER: this lane's share of a row's (sum, sum of squares) over 128 columns, even and odd elements apart (packed adds / fmas) */
    f32x2 psum[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}}, psq[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
#pragma unroll
    for (int s = 0; s < JT / 2; ++s) {
        const int col = n0 + 32 * s + 8 * j4;
        f32x4 nb[4] = {};
        if (NORM) {
            nb[0] = *reinterpret_cast<const f32x4 *>(ncol + (col - n0));
            nb[1] = *reinterpret_cast<const f32x4 *>(ncol + (col - n0) + 4);
            nb[2] = *reinterpret_cast<const f32x4 *>(ncol + BN + (col - n0));
            nb[3] = *reinterpret_cast<const f32x4 *>(ncol + BN + (col - n0) + 4);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = m0 + 32 * wave + 16 * i + l15;
            const bool live = row < p.row_end;
            f32x4 lo = acc[i][2 * s], hi = acc[i][2 * s + 1];
            if (NORM) {   /* LN(x) W^T + b = rstd (x W'^T) - rstd mean colsum(W') + b'   (norm_fold.h) */
                const f32x4 r4 = {n_rstd[i], n_rstd[i], n_rstd[i], n_rstd[i]}, s4 = {n_shift[i], n_shift[i], n_shift[i], n_shift[i]};
                lo = __builtin_elementwise_fma(lo, r4, __builtin_elementwise_fma(nb[0], s4, nb[2]));
                hi = __builtin_elementwise_fma(hi, r4, __builtin_elementwise_fma(nb[1], s4, nb[3]));
            }
            if (GELU) {
                /* a result that is quantised to e4m3 takes the GELU whose error is matched to that format */
                auto gelu2 = [](f32x2 v) { return OUTK == OUT_MX ? gelu_lowp2<1>(v) : gelu_exact2(v); };
                const f32x2 g0 = gelu2(f32x2{lo[0], lo[1]}), g1 = gelu2(f32x2{lo[2], lo[3]});
                const f32x2 g2 = gelu2(f32x2{hi[0], hi[1]}), g3 = gelu2(f32x2{hi[2], hi[3]});
                lo = f32x4{g0[0], g0[1], g1[0], g1[1]};
                hi = f32x4{g2[0], g2[1], g3[0], g3[1]};
            }
            if (OUTK == OUT_MX) {
                /* the row's 32 columns n0 + 32s .. +31 = one scale block, held by the four lanes l15 + 16 j */
                mx_store_block8(lo, hi, static_cast<char *>(p.C), static_cast<unsigned char *>(p.Cs), p.a_rows, arow[i], n0 + 32 * s, j4, live);
            } else if (OUTK == OUT_PLANES_H) {
                if (live) {
                    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
                    half8 hv;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        hv[e] = (_Float16)(e < 4 ? lo[e] : hi[e - 4]);
                    *reinterpret_cast<f32x4 *>(static_cast<char *>(p.C) + ((size_t)(col >> 5) * p.a_rows + row) * 64 + 16 * j4) =
                        __builtin_bit_cast(f32x4, hv);
                }
            } else {
                if (live) {
                    float *cp = static_cast<float *>(p.C) + (size_t)row * p.N + col;
                    *reinterpret_cast<f32x4 *>(cp) = lo;
                    *reinterpret_cast<f32x4 *>(cp + 4) = hi;
                }
                if (OPER) {
                    /* the same eight values as the next projection's MX operand, and their share of the row's statistics
                     * (fixed order: even and odd columns ascending per lane, their two sums, then the four lanes of the row, per 128 columns) */
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f32x2 v = e < 2 ? f32x2{lo[2 * e], lo[2 * e + 1]} : f32x2{hi[2 * e - 4], hi[2 * e - 3]};
                        psum[i] = psum[i] + v;
                        psq[i] = __builtin_elementwise_fma(v, v, psq[i]);
                    }
                    mx_store_block8(lo, hi, static_cast<char *>(p.oper), static_cast<unsigned char *>(p.oper_scales), p.a_rows, arow[i],
                                    n0 + 32 * s, j4, live);
                    if ((s & 3) == 3) {
                        float su = psum[i][0] + psum[i][1], sq2 = psq[i][0] + psq[i][1];
                        su += __shfl_xor(su, 16);
                        sq2 += __shfl_xor(sq2, 16);
                        su += __shfl_xor(su, 32);
                        sq2 += __shfl_xor(sq2, 32);
                        if (j4 == 0 && live)
                            *reinterpret_cast<f32x2 *>(p.stats_out + ((size_t)((n0 >> 7) + (s >> 2)) * p.a_rows + row) * 2) = f32x2{su, sq2};
                        psum[i] = psq[i] = f32x2{0.0f, 0.0f};
                    }
                }
            }
        }
    }
}

template <int NW, int BN, int EPI, int OUTK>
int launch_mx_tile(hipStream_t st, MxParams p)
{
    constexpr int LDS = 2 * (BN * 128 + 4 * (BN + 32)) + ((EPI == EPI_NORM || EPI == EPI_NORM_GELU) ? 2 * BN * 4 : 0);
    VH_SET_LDS_ONCE((gemm_mx_kernel<NW, BN, EPI, OUTK>), LDS);
    p.mtiles = (p.row_end - p.row_begin + 32 * NW - 1) / (32 * NW);
    p.ntiles = p.N / BN;
    hipLaunchKernelGGL((gemm_mx_kernel<NW, BN, EPI, OUTK>), dim3(p.mtiles * p.ntiles), dim3(64 * NW), LDS, st, p);
    VH_LAUNCH_CHECK("gemm_mx_kernel");
    return 0;
}

template <int EPI, int OUTK>
int launch_mx(hipStream_t st, const MxParams &p, int small_only)
{
    const int rows = p.row_end - p.row_begin;
    const int num_cus = vh_device_cus(vh_current_device());
    const int ntiles = p.N / 256, mtiles = (rows + 255) / 256;
    const long tiles = (long)mtiles * ntiles;
    /* 128x256 tiles, 4 waves of 32x256, TWO workgroups per CU wherever the problem fills the chip: a K = 768 product is
     * six K steps long, and with one 256x256 workgroup per CU every prologue (first DMA) and every epilogue (GELU, block
     * maxima, quantisation, stores; residual rows) runs with the matrix pipe idle -- with two, one workgroup's epilogue
     * runs under the other's K loop.  Measured (ViT-B/16, batch 512, same call): fc1 0.418 -> 0.356 ms, out-proj
     * 0.185 -> 0.175, fc2 0.362 -> 0.348, QKV 0.241 -> 0.235; 28.3k -> 29.7k images/s. */
    if (p.N % 256 == 0 && tiles >= num_cus)
        return launch_mx_tile<4, 256, EPI, OUTK>(st, p);
    if (p.N % 256 != 0 || small_only || 2 * tiles < 5 * (long)num_cus)
        return launch_mx_tile<4, 128, EPI, OUTK>(st, p);
    const long full = tiles / num_cus, rem = tiles % num_cus;
    const int rows_big = (int)(full * num_cus / ntiles) * 256;
    if (rem == 0 || 4 * rem > 3 * num_cus || rows_big <= 0 || rows_big >= rows)
        return launch_mx_tile<8, 256, EPI, OUTK>(st, p);
    MxParams big = p, rest = p;
    big.row_end = p.row_begin + rows_big;
    rest.row_begin = big.row_end;
    const int rc = launch_mx_tile<8, 256, EPI, OUTK>(st, big);
    return rc ? rc : launch_mx_tile<4, 128, EPI, OUTK>(st, rest);
}

/* fp32 [rows][K] -> MX planes.  One thread per 4 consecutive values, so that a wave reads 1 KiB of a row in one
 * coalesced instruction and writes 256 contiguous bytes; the 8 lanes of a 32-element block share its maximum through
 * three shuffles.  (One thread per block -- 8 loads of 16 B, 128 B apart from its neighbour's -- ran at 3.5 TB/s.) */
template <bool ACT>   /* ACT: an activation tensor (scale bytes at mx_act_scale_index); otherwise the weights' [K/128][4][rows] */
__global__ void quantize_mx_rows_kernel(const float *__restrict__ in, char *__restrict__ values, unsigned char *__restrict__ scales,
                                        int rows, int K)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nv = K >> 2;                                /* a multiple of 32: the 8 lanes of a block never straddle a wave */
    const bool live = idx < (size_t)rows * nv;
    const size_t i = live ? idx : (size_t)rows * nv - 1;  /* idle lanes repeat the last chunk: the shuffles stay uniform */
    const int row = (int)(i / nv), c4 = (int)(i - (size_t)row * nv);
    const f32x4 v = *reinterpret_cast<const f32x4 *>(in + (size_t)row * K + 4 * c4);
    float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    amax = fmaxf(amax, __shfl_xor(amax, 1));
    amax = fmaxf(amax, __shfl_xor(amax, 2));
    amax = fmaxf(amax, __shfl_xor(amax, 4));
    unsigned sbyte;
    float mult;
    mx_block_scale(amax, sbyte, mult);
    if (!live)
        return;
    const int k = 4 * c4, ks = k >> 7, blk = (k >> 5) & 3;
    *reinterpret_cast<unsigned *>(values + ((size_t)ks * rows + row) * 128 + (k & 127)) = pack_fp8x4(v * mult);
    if ((c4 & 7) == 0)
        scales[ACT ? mx_act_scale_index(ks, blk, (size_t)row, rows) : ((size_t)ks * 4 + 2 * (blk & 1) + (blk >> 1)) * rows + row] = (unsigned char)sbyte;
}

} // namespace

static int quantize_mx(vh_stream_t s, const float *input, void *values, void *scales, int rows, int cols, bool act)
{
    if (!input || !values || !scales || rows <= 0 || cols <= 0 || cols % 128 != 0 ||
        (((uintptr_t)input | (uintptr_t)values) & 15) || (act && ((uintptr_t)scales & 3)))
        return vh_fail(1, "vh_launch_quantize_mx_rows: bad argument (cols %% 128 == 0, 16-byte aligned pointers)");
    const size_t threads = (size_t)rows * (cols / 4);
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (act)
        hipLaunchKernelGGL(quantize_mx_rows_kernel<true>, grid, dim3(256), 0, (hipStream_t)s, input, static_cast<char *>(values),
                           static_cast<unsigned char *>(scales), rows, cols);
    else
        hipLaunchKernelGGL(quantize_mx_rows_kernel<false>, grid, dim3(256), 0, (hipStream_t)s, input, static_cast<char *>(values),
                           static_cast<unsigned char *>(scales), rows, cols);
    VH_LAUNCH_CHECK("quantize_mx_rows_kernel");
    return 0;
}

/* weights [N][K]: scale bytes [K/128][4][N] (the form the GEMM's LDS-DMA moves) */
extern "C" int vh_launch_quantize_mx_rows(vh_stream_t s, const float *input, void *values, void *scales, int rows, int cols)
{
    return quantize_mx(s, input, values, scales, rows, cols, false);
}

/* activations [rows][K]: scale bytes [ceil(K/512)][4][rows][4] (vit_kernels.h mx_act_scale_index; vh_mx_act_scale_bytes of them) */
extern "C" int vh_launch_quantize_mx_act(vh_stream_t s, const float *input, void *values, void *scales, int rows, int cols)
{
    return quantize_mx(s, input, values, scales, rows, cols, true);
}

extern "C" size_t vh_mx_act_scale_bytes(int rows, int cols)
{
    return rows > 0 && cols > 0 ? (size_t)((cols / 128 + 3) / 4) * 16 * (size_t)rows : 0;
}

extern "C" int vh_launch_linear_mx(vh_stream_t s, void *output, void *output_scales, const void *weight_values,
                                   const void *weight_scales, const void *input_values, const void *input_scales,
                                   const float *bias, int rowA, int colA, int colB, int doGelu, const float *residual)
{
    if (!output || !weight_values || !weight_scales || !input_values || !input_scales || !bias)
        return vh_fail(1, "vh_launch_linear_mx: null pointer argument");
    if (rowA <= 0 || colA <= 0 || colB <= 0 || colA % 256 != 0 || colB % 128 != 0)
        return vh_fail(1, "vh_launch_linear_mx: needs colA %% 256 == 0 and colB %% 128 == 0 (%d,%d,%d)", rowA, colA, colB);
    if ((doGelu && residual) || (residual && output_scales))
        return vh_fail(1, "vh_launch_linear_mx: unsupported epilogue combination");
    if ((((uintptr_t)output | (uintptr_t)weight_values | (uintptr_t)input_values | (uintptr_t)bias | (uintptr_t)residual |
          (uintptr_t)weight_scales) & 15) != 0)
        return vh_fail(1, "vh_launch_linear_mx: pointers must be 16-byte aligned");
    if ((size_t)rowA * 128 > 0xffffffffull)
        return vh_fail(1, "vh_launch_linear_mx: rowA=%d too large", rowA);
    MxParams p = {};
    p.A = static_cast<const char *>(input_values); p.As = static_cast<const char *>(input_scales);
    p.W = static_cast<const char *>(weight_values); p.Ws = static_cast<const char *>(weight_scales);
    p.bias = bias; p.R = residual; p.C = output; p.Cs = output_scales;
    p.row_begin = 0; p.row_end = rowA; p.a_rows = rowA; p.N = colB; p.K = colA;
    hipStream_t st = (hipStream_t)s;
    const int small_only = residual && colA < 2048;
    if (doGelu)
        return output_scales ? launch_mx<EPI_GELU, OUT_MX>(st, p, small_only) : launch_mx<EPI_GELU, OUT_F32>(st, p, small_only);
    if (residual)
        return launch_mx<EPI_RESID, OUT_F32>(st, p, small_only);
    if (output_scales == output)   /* see vh_launch_linear_mx_planes_f16 */
        return launch_mx<EPI_NONE, OUT_PLANES_H>(st, p, small_only);
    return output_scales ? launch_mx<EPI_NONE, OUT_MX>(st, p, small_only) : launch_mx<EPI_NONE, OUT_F32>(st, p, small_only);
}

/* The same product written as one-part fp16 planes [colB/32][rowA][32] (no GELU, no residual): the Q|K|V input of
 * vh_launch_attention_planes_f16. */
extern "C" int vh_launch_linear_mx_planes_f16(vh_stream_t s, void *output_planes_f16, const void *weight_values,
                                              const void *weight_scales, const void *input_values, const void *input_scales,
                                              const float *bias, int rowA, int colA, int colB)
{
    /* internal convention: output_scales == output selects the fp16-planes epilogue */
    return vh_launch_linear_mx(s, output_planes_f16, output_planes_f16, weight_values, weight_scales, input_values, input_scales,
                               bias, rowA, colA, colB, 0, nullptr);
}

/* ---- LayerNorm folded into the projections (norm_fold.h), block-scaled fp8 operands -------------------------------- */

/* LN(A) W^T + b with the LayerNorm folded: input = the UN-normalised rows as an MX tensor, row_stats = their partial sums
 * [colA/128][rowA][2], weights = the gamma-scaled matrix quantised to MX, colsum / bias_folded its column terms.
 * output_kind: 0 fp32 rows, 1 MX tensor (output + output_scales; with doGelu: fc1), 2 one-part fp16 planes (Q|K|V). */
extern "C" int vh_launch_linear_mx_norm(vh_stream_t s, void *output, void *output_scales, int output_kind, const void *weight_values,
                                        const void *weight_scales, const void *input_values, const void *input_scales,
                                        const float *row_stats, const float *colsum, const float *bias_folded, double eps, int rowA,
                                        int colA, int colB, int doGelu)
{
    if (!output || !weight_values || !weight_scales || !input_values || !input_scales || !row_stats || !colsum || !bias_folded ||
        (output_kind == 1 && !output_scales))
        return vh_fail(1, "vh_launch_linear_mx_norm: null pointer argument");
    if (rowA <= 0 || colA <= 0 || colB <= 0 || colA % 256 != 0 || colB % 128 != 0)
        return vh_fail(1, "vh_launch_linear_mx_norm: needs colA %% 256 == 0 and colB %% 128 == 0 (%d,%d,%d)", rowA, colA, colB);
    if (output_kind < 0 || output_kind > 2 || (doGelu && output_kind != 1))
        return vh_fail(1, "vh_launch_linear_mx_norm: unsupported epilogue combination");
    if (colA > 16 * 128)
        return vh_fail(1, "vh_launch_linear_mx_norm: colA=%d: at most 16 partial sums per row (colA <= 2048)", colA);
    if ((((uintptr_t)output | (uintptr_t)weight_values | (uintptr_t)input_values | (uintptr_t)colsum | (uintptr_t)bias_folded |
          (uintptr_t)weight_scales) & 15) != 0 || ((uintptr_t)row_stats & 7) != 0)
        return vh_fail(1, "vh_launch_linear_mx_norm: pointers must be 16-byte aligned (row_stats: 8)");
    if ((size_t)rowA * 128 > 0xffffffffull)
        return vh_fail(1, "vh_launch_linear_mx_norm: rowA=%d too large", rowA);
    MxParams p = {};
    p.A = static_cast<const char *>(input_values); p.As = static_cast<const char *>(input_scales);
    p.W = static_cast<const char *>(weight_values); p.Ws = static_cast<const char *>(weight_scales);
    p.bias = bias_folded; p.colsum = colsum; p.stats = row_stats; p.eps = eps; p.C = output; p.Cs = output_scales;
    p.row_begin = 0; p.row_end = rowA; p.a_rows = rowA; p.N = colB; p.K = colA;
    hipStream_t st = (hipStream_t)s;
    if (doGelu)
        return launch_mx<EPI_NORM_GELU, OUT_MX>(st, p, 0);
    return output_kind == 2 ? launch_mx<EPI_NORM, OUT_PLANES_H>(st, p, 0)
         : output_kind == 1 ? launch_mx<EPI_NORM, OUT_MX>(st, p, 0) : launch_mx<EPI_NORM, OUT_F32>(st, p, 0);
}

/* output = residual + A W^T + b (fp32 rows, in place allowed) AND the same rows as the next projection's MX operand
 * (operand_values [colB/128][rowA][128], operand_scales [colB/128][4][rowA]) AND their partial sums row_stats_out
 * [colB/128][rowA][2] for that projection's folded LayerNorm. */
extern "C" int vh_launch_linear_mx_resid_norm(vh_stream_t s, float *output, const void *weight_values, const void *weight_scales,
                                              const void *input_values, const void *input_scales, const float *bias,
                                              const float *residual, int rowA, int colA, int colB, void *operand_values,
                                              void *operand_scales, float *row_stats_out)
{
    if (!output || !weight_values || !weight_scales || !input_values || !input_scales || !bias || !residual || !operand_values ||
        !operand_scales || !row_stats_out)
        return vh_fail(1, "vh_launch_linear_mx_resid_norm: null pointer argument");
    if (rowA <= 0 || colA <= 0 || colB <= 0 || colA % 256 != 0 || colB % 128 != 0)
        return vh_fail(1, "vh_launch_linear_mx_resid_norm: needs colA %% 256 == 0 and colB %% 128 == 0 (%d,%d,%d)", rowA, colA, colB);
    if ((((uintptr_t)output | (uintptr_t)weight_values | (uintptr_t)input_values | (uintptr_t)bias | (uintptr_t)residual |
          (uintptr_t)weight_scales | (uintptr_t)operand_values) & 15) != 0 || ((uintptr_t)row_stats_out & 7) != 0)
        return vh_fail(1, "vh_launch_linear_mx_resid_norm: pointers must be 16-byte aligned (row_stats_out: 8)");
    if ((size_t)rowA * 128 > 0xffffffffull)
        return vh_fail(1, "vh_launch_linear_mx_resid_norm: rowA=%d too large", rowA);
    MxParams p = {};
    p.A = static_cast<const char *>(input_values); p.As = static_cast<const char *>(input_scales);
    p.W = static_cast<const char *>(weight_values); p.Ws = static_cast<const char *>(weight_scales);
    p.bias = bias; p.R = residual; p.C = output;
    p.oper = operand_values; p.oper_scales = operand_scales; p.stats_out = row_stats_out;
    p.row_begin = 0; p.row_end = rowA; p.a_rows = rowA; p.N = colB; p.K = colA;
    return launch_mx<EPI_RESID, OUT_F32_OPER_MX>((hipStream_t)s, p, colA < 2048);
}
