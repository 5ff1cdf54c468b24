/*
 * vit_kernels.h -- internal declarations shared by the HIP translation units.
 * Not part of the public ABI (that is include/kernelHandler.h).
 */
#ifndef VIT_HIP_VIT_KERNELS_H
#define VIT_HIP_VIT_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

/* Four fp32 values -> four OCP e4m3 bytes (gfx950's fp8), round to nearest even, saturating
 * at the format's largest finite value instead of overflowing to NaN. */
#define VH_FP8_MAX 448.0f
__device__ __forceinline__ unsigned pack_fp8x4(f32x4 v)
{
    int r = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v[0], -VH_FP8_MAX, VH_FP8_MAX),
                                            __builtin_amdgcn_fmed3f(v[1], -VH_FP8_MAX, VH_FP8_MAX), 0, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v[2], -VH_FP8_MAX, VH_FP8_MAX),
                                        __builtin_amdgcn_fmed3f(v[3], -VH_FP8_MAX, VH_FP8_MAX), r, true);
    return (unsigned)r;
}

/* Records `msg` as the calling thread's last error and returns `code`. */
int vh_fail(int code, const char *fmt, ...);
/* Converts a hipError_t into the launcher return convention, recording text. */
int vh_hip_status(hipError_t e, const char *what);

/* attention_tiled.hip: the shapes the resident-K/V attention kernel does not take. */
int vh_attention_tiled(void *stream, const float *qkv, void *output, int out_bf16, int n_images, int tokens,
                       int embed_dim, int num_heads);

#define VH_TRY(expr)                                                           \
    do {                                                                       \
        hipError_t vh_try_e_ = (expr);                                         \
        if (vh_try_e_ != hipSuccess)                                           \
            return vh_hip_status(vh_try_e_, #expr);                            \
    } while (0)

/* After a kernel launch: surface launch-configuration errors immediately. */
#define VH_LAUNCH_CHECK(name)                                                  \
    do {                                                                       \
        hipError_t vh_lc_e_ = hipGetLastError();                               \
        if (vh_lc_e_ != hipSuccess)                                            \
            return vh_hip_status(vh_lc_e_, name);                              \
    } while (0)

#endif
