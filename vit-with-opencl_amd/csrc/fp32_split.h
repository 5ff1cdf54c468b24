/*
 * fp32_split.h -- fp32 products on the 16-bit matrix cores: operand decompositions shared by
 * the GEMM (gemm_mfma.hip) and attention (attention_f32.hip) kernels.  Internal.
 */
#ifndef VIT_HIP_FP32_SPLIT_H
#define VIT_HIP_FP32_SPLIT_H

#include "vit_kernels.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

/* Exact 3-way bf16 split of eight fp32 values: x = p0 + p1 + p2 with p0 = bf16(x),
 * p1 = bf16(x - p0), p2 = bf16(x - p0 - p1); the subtractions are exact (Sterbenz / aligned
 * bits).  Left to the compiler's own pairing (v_cvt_pk_bf16_f32, part of the subtractions as
 * v_pk_add_f32; ~5.4 VALU instructions per element): writing the fully packed 4.5-instruction
 * form out with inline asm measured 3 % SLOWER end to end (hazard s_nops, no freedom to
 * interleave with the MFMAs). */
__device__ __forceinline__ void split8(const f32x4 &u, const f32x4 &v, bf16x8 &p0, bf16x8 &p1, bf16x8 &p2)
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = e < 4 ? u[e] : v[e - 4];
        const __bf16 h = (__bf16)x;
        const float r1 = x - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        p0[e] = h;
        p1[e] = m;
        p2[e] = (__bf16)r2;
    }
}

/* The same split of four values (8-byte parts): producers that hold four consecutive elements per lane. */
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split4(const f32x4 &u, bf16x4 &p0, bf16x4 &p1, bf16x4 &p2)
{
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = u[e];
        const __bf16 h = (__bf16)x;
        const float r1 = x - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        p0[e] = h;
        p1[e] = m;
        p2[e] = (__bf16)r2;
    }
}

/* Parts of a split fp32 operand, per decomposition:
 *   NPL = 3: x = p0 + p1 + p2 exactly, bf16 parts; six products (all of weight >= 2^-16)
 *   NPL = 2: x = p0 + p1 + eps, fp16 parts (2 x 11 significant bits; |eps| <= 2^-22 |x|, fp16
 *            subnormals are honoured by the conversion and by the MFMA: tools/f16_denorm_probe.hip);
 *            three products p0q0 + p0q1 + p1q0 -- the "3 x TF32"-style emulation: its truncation
 *            error (7.7e-8 of the result, tools/split3_numerics.py) is an order of magnitude below
 *            the rounding noise of any fp32 accumulation order, but it is not exact. */
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
template <int NPL> struct PartT { typedef bf16x8 type; };
template <> struct PartT<2> { typedef half8 type; };
template <> struct PartT<1> { typedef half8 type; };   /* one fp16 part: operands ROUNDED to fp16 (reduced-precision modes only) */

__device__ __forceinline__ void split_parts(const f32x4 &u, const f32x4 &v, bf16x8 (&o)[3])
{
    split8(u, v, o[0], o[1], o[2]);
}
__device__ __forceinline__ void split_parts(const f32x4 &u, const f32x4 &v, half8 (&o)[2])
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = e < 4 ? u[e] : v[e - 4];
        const _Float16 h = (_Float16)x;
        o[0][e] = h;
        o[1][e] = (_Float16)(x - (float)h);
    }
}
/* The value is made opaque before the conversion: where the fp32 value is itself a product (a probability =
 * exponential x reciprocal sum) the compiler otherwise fuses multiplication and conversion into one v_fma_mix
 * with a single rounding -- for some elements of some kernels and not for others, depending on where the
 * multiplication happens to sit.  Rounded twice (to fp32, then to fp16) everywhere, the attention kernels on rows
 * and on planes, in step or staggered, agree bit for bit. */
__device__ __forceinline__ void split_parts(const f32x4 &u, const f32x4 &v, half8 (&o)[1])
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float x = e < 4 ? u[e] : v[e - 4];
        asm("" : "+v"(x));
        o[0][e] = (_Float16)x;
    }
}
__device__ __forceinline__ f32x4 mfma_part(bf16x8 w, bf16x8 a, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_part(half8 w, half8 a, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(w, a, c, 0, 0, 0);
}
/* the 32x32x16 forms (attention) */
__device__ __forceinline__ f32x16 mfma_part(bf16x8 w, bf16x8 a, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_part(half8 w, half8 a, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(w, a, c, 0, 0, 0);
}
/* (w part, a part) of product t, smallest terms first */
template <int NPL> __device__ __forceinline__ constexpr int n_terms() { return NPL == 3 ? 6 : NPL == 2 ? 3 : 1; }
template <int NPL> __device__ __forceinline__ constexpr int term_w(int t)
{
    return NPL == 3 ? (t == 0 || t == 3 || t == 5 ? 0 : t == 1 ? 2 : 1) : NPL == 2 ? (t == 0 ? 1 : 0) : 0;
}
template <int NPL> __device__ __forceinline__ constexpr int term_a(int t)
{
    return NPL == 3 ? (t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0) : NPL == 2 ? (t == 1 ? 1 : 0) : 0;
}

#endif
