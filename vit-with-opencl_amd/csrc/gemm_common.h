/*
 * gemm_common.h -- pieces shared by the GEMM translation units (gemm_mfma.hip, gemm_p3.hip):
 * the XCD-aware blockIdx -> tile map and the fc1 GELU epilogue.  Internal.
 */
#ifndef VIT_HIP_GEMM_COMMON_H
#define VIT_HIP_GEMM_COMMON_H

#include "vit_kernels.h"

/* 16-byte chunk swizzle of a 64-byte LDS row r (plane rows: 32 bf16): chunk c sits at c ^ f((r >> 2) & 3),
 * f = {0, 2, 3, 1} -- the 16 lanes of every ds_read_b128 group of a 16- or 32-row fragment then hit 16
 * distinct slots of the 256-byte bank row.  Argument: r >> 2. */
__device__ __forceinline__ int swz64(int r4) { return (0x78 >> (2 * (r4 & 3))) & 3; }

/* Bijective XCD remap (blocks b and b+8 share an XCD; which one is not known
 * and not needed): XCD x gets a contiguous run of tiles. */
__device__ __forceinline__ int xcd_tile(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

/* GELU for the fc1 epilogue: 0.5*x*(1+erf(x/sqrt(2))), ViT_seq.c:285 / ll.cl:4.
 * The scalar loop calls libm erff; a device libm erff is two divergent branches of
 * ~50 VALU instructions each, which made this epilogue a third of the fc1 kernel
 * (42 us per 128x128 tile).  GELU only needs erf to ABSOLUTE accuracy (it forms
 * 1 + erf), so one branch-free form serves every x:
 *     erf(a) = sign(a) * (1 - 2^(-t*S(t))),  t = min(|a|, 4),  S(t) = -log2(erfc(t))/t
 * with S a degree-10 least-squares fit on Chebyshev nodes (tools/fit_gelu.py):
 * max |erf - exact| = 8.2e-8, max |gelu - scalar fp32 formula with glibc erff| = 2.4e-7
 * (half an ulp at |x| = 8; that scalar formula is itself 4.5e-7 from exact).
 * The division by sqrt(2) is a multiplication by its fp32 reciprocal. */
__device__ __forceinline__ float gelu_exact(float x)
{
    const float a = x * 0.70710678118654752f;
    const float t = fminf(fabsf(a), 4.0f);
    float s = -1.434945176e-07f;
    s = __builtin_fmaf(s, t, 3.633770575e-06f);
    s = __builtin_fmaf(s, t, -4.095854820e-05f);
    s = __builtin_fmaf(s, t, 2.688577224e-04f);
    s = __builtin_fmaf(s, t, -1.106124371e-03f);
    s = __builtin_fmaf(s, t, 2.616208047e-03f);
    s = __builtin_fmaf(s, t, -3.566097876e-04f);
    s = __builtin_fmaf(s, t, -2.759680524e-02f);
    s = __builtin_fmaf(s, t, 1.482741833e-01f);
    s = __builtin_fmaf(s, t, 9.184474349e-01f);
    s = __builtin_fmaf(s, t, 1.627907038e+00f);
    const float erf_a = copysignf(1.0f - __builtin_amdgcn_exp2f(-(t * s)), a);
    return 0.5f * x * (1.0f + erf_a);
}

/* The same GELU on two values at once: every step is a packed instruction (v_pk_mul_f32,
 * v_pk_fma_f32, ...), half the VALU issue slots of the scalar form, the same bits per element. */
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_exact2(f32x2 x)
{
    const f32x2 a = x * 0.70710678118654752f;
    const f32x2 t = __builtin_elementwise_min(__builtin_elementwise_abs(a), f32x2{4.0f, 4.0f});
    auto k = [](float c) { return f32x2{c, c}; };
    f32x2 s = k(-1.434945176e-07f);
    s = __builtin_elementwise_fma(s, t, k(3.633770575e-06f));
    s = __builtin_elementwise_fma(s, t, k(-4.095854820e-05f));
    s = __builtin_elementwise_fma(s, t, k(2.688577224e-04f));
    s = __builtin_elementwise_fma(s, t, k(-1.106124371e-03f));
    s = __builtin_elementwise_fma(s, t, k(2.616208047e-03f));
    s = __builtin_elementwise_fma(s, t, k(-3.566097876e-04f));
    s = __builtin_elementwise_fma(s, t, k(-2.759680524e-02f));
    s = __builtin_elementwise_fma(s, t, k(1.482741833e-01f));
    s = __builtin_elementwise_fma(s, t, k(9.184474349e-01f));
    s = __builtin_elementwise_fma(s, t, k(1.627907038e+00f));
    const f32x2 ts = t * s;
    const f32x2 e = {__builtin_amdgcn_exp2f(-ts[0]), __builtin_amdgcn_exp2f(-ts[1])};
    const f32x2 erf_a = __builtin_elementwise_copysign(k(1.0f) - e, a);
    return (x * 0.5f) * (k(1.0f) + erf_a);
}

/* GELU for epilogues whose output is ROUNDED to bf16 (KIND 0: 8 significand bits) or to e4m3 (KIND 1: 4 bits) -- the
 * reduced modes' fc1.  The same form with S of a degree matched to the output format (tools/fit_gelu_lowp.py: degree 4
 * on [0, 3.5], error <= 0.18 of a bf16 rounding step; degree 3 on [0, 3], <= 0.03 of an e4m3 step; absolute error
 * <= 1.8e-5 / 1.1e-4 for EVERY x: see the exponent's tail below) and the final algebra folded:
 *     gelu(x) = 0.5 x (1 + sign(a)(1 - e)) = max(x, 0) - 0.5 |x| e,   e = 2^(-t S(t)), t = min(|x| / sqrt 2, c)
 * 6 (5) packed multiply-adds less per pair than gelu_exact2.  The fp32 paths never use it. */
template <int KIND>
__device__ __forceinline__ f32x2 gelu_lowp2(f32x2 x)
{
    auto k = [](float c) { return f32x2{c, c}; };
    const f32x2 a = x * 0.70710678118654752f;
    const float clamp = KIND == 0 ? 3.5f : 3.0f;
    const f32x2 t = __builtin_elementwise_min(__builtin_elementwise_abs(a), k(clamp));
    f32x2 s;
    if (KIND == 0) {
        s = k(1.978939632e-03f);
        s = __builtin_elementwise_fma(s, t, k(-2.493243292e-02f));
        s = __builtin_elementwise_fma(s, t, k(1.416560262e-01f));
        s = __builtin_elementwise_fma(s, t, k(9.218431711e-01f));
        s = __builtin_elementwise_fma(s, t, k(1.627655268e+00f));
    } else {
        s = k(-1.295685954e-02f);
        s = __builtin_elementwise_fma(s, t, k(1.189612895e-01f));
        s = __builtin_elementwise_fma(s, t, k(9.356397390e-01f));
        s = __builtin_elementwise_fma(s, t, k(1.626340985e+00f));
    }
    /* beyond the fitted range the exponent keeps falling, 64 per unit of |a| - clamp (0 inside the range, so nothing
     * changes there): e -> 0 within a fraction of a unit instead of staying at erfc(clamp), which left an error of
     * 0.5 |x| erfc(clamp) growing with |x| (-5.5e-4 at x = -50 for KIND 1).  The bounds above now hold for every x. */
    const f32x2 ts = __builtin_elementwise_fma(k(64.0f), __builtin_elementwise_abs(a) - t, t * s);
    const f32x2 e = {__builtin_amdgcn_exp2f(-ts[0]), __builtin_amdgcn_exp2f(-ts[1])};
    const f32x2 h = (x * 0.5f) * e;
    return __builtin_elementwise_max(x, k(0.0f)) - __builtin_elementwise_abs(h);
}

#endif
