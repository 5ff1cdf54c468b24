/*
 * norm_fold.h -- LayerNorm folded into the projection behind it (the reduced-precision modes).  Internal.
 *
 * Replaces, for those modes, the LayerNorm launches in front of the QKV projection and of fc1
 * (`layerNorm`, layer_norm.cl:3-53; CPU statement layer_norm_seq, ViT_seq.c:120-142, called at :346,358).
 *
 *     LN(x) W^T + b  =  rstd * ( x (gamma . W)^T  -  mean * colsum(gamma . W) )  +  ( beta W^T + b )
 *
 *  - context creation scales the columns of W by gamma (W' = gamma . W, THEN rounded to the mode's operand format),
 *    and stores colsum[n] = sum_k W'[n][k] (of the rounded values the matrix cores will multiply) and the folded bias
 *    b'[n] = b[n] + sum_k beta[k] W[n][k];
 *  - the producers of the residual stream x (patch embedding, output projection, fc2) write, besides the fp32 rows, x in
 *    the next projection's operand format and, per row and per 128 columns, the partial sums (sum x, sum x^2) of the
 *    fp32 values: stats[N/128][rows][2].  No atomics: every partial has one writer, and a row's partials do not
 *    depend on the tile shape or on the row's position in the batch;
 *  - the consuming projection adds the partials in fixed order, forms mean and 1/std exactly as layer_norm_seq does
 *    (var = E[x^2] - mean^2, eps added in double: ViT_seq.c:132-135) and applies the row terms to its accumulators.
 *
 * What changes numerically: the operand that is rounded to bf16 / e4m3 is x, not LN(x).  The two differ by the
 * per-row shift and scale (and gamma, which moves into W): the rounding error of x - mean is that of x, i.e. larger by
 * |x| / |x - mean| -- immaterial while |mean| is of the order of the row's standard deviation (measured on the test
 * rows and printed by tests/test_gpu_fold.py), and the reason the exact fp32 path does not use this.
 */
#ifndef VIT_HIP_NORM_FOLD_H
#define VIT_HIP_NORM_FOLD_H

#include "vit_kernels.h"
#include "gemm_common.h"

/* 1/std and -mean/std of row `row` from its producer's partial sums (groups = K / 128 <= 16).  Called by all four lanes
 * (q = lane >> 4) that hold the row in the GEMM epilogue's layout: lane q takes partials q, q + 4, q + 8, q + 12 (four
 * independent loads, issued together -- a loop over the groups with one load per trip cost a round trip to L2 per trip in
 * every tile's prologue), two shuffles add the four lane sums: ((s0 + s1) + (s2 + s3)), the same bits in every lane. */
__device__ __forceinline__ void row_norm_terms(const float *stats, int groups, int rows, int row, int q, int K, double eps, float &rstd,
                                               float &shift)
{
    float sum = 0.0f, sq = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int g = q + 4 * j;
        const f32x2 t = *reinterpret_cast<const f32x2 *>(stats + ((size_t)min(g, groups - 1) * rows + row) * 2);
        sum += g < groups ? t[0] : 0.0f;
        sq += g < groups ? t[1] : 0.0f;
    }
    sum += __shfl_xor(sum, 16);
    sq += __shfl_xor(sq, 16);
    sum += __shfl_xor(sum, 32);
    sq += __shfl_xor(sq, 32);
    const float mean = sum / (float)K;
    const float var = sq / (float)K - mean * mean;
    rstd = 1.0f / sqrtf((float)((double)var + eps));
    shift = -(mean * rstd);
}

/* One 32-column MX scale block of one row, held as 8 consecutive values (lo, hi) by each of the four lanes
 * l15 + 16 j (j = 0..3) of a wave: block maximum by two shuffles, e8m0 scale, e4m3 values; 8-byte store per lane and
 * the scale byte from lane group 0.  Layout of csrc/gemm_mx.hip for activations: values[K/128][rows][128], scale bytes
 * at mx_act_scale_index (vit_kernels.h).  col32 = first column of the block.
 * All four lanes of the block must call it together (the shuffles); only `live` rows are stored. */
__device__ __forceinline__ void mx_store_block8(const f32x4 &lo, const f32x4 &hi, char *values, unsigned char *scales, int rows,
                                                size_t row, int col32, int j4, bool live)
{
    float amax = 0.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        amax = fmaxf(amax, fmaxf(fabsf(lo[e]), fabsf(hi[e])));
    amax = fmaxf(amax, __shfl_xor(amax, 16));
    amax = fmaxf(amax, __shfl_xor(amax, 32));
    unsigned sbyte;
    float mult;
    mx_block_scale(amax, sbyte, mult);
    if (!live)
        return;
    const int ks = col32 >> 7, blk = (col32 >> 5) & 3;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<u32x2 *>(values + ((size_t)ks * rows + row) * 128 + 32 * blk + 8 * j4) = u32x2{pack_fp8x4(lo * mult), pack_fp8x4(hi * mult)};
    if (j4 == 0)
        scales[mx_act_scale_index(ks, blk, row, rows)] = (unsigned char)sbyte;
}

#endif
