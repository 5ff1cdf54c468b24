/*
 * vit_kernels.h -- internal declarations shared by the HIP translation units.
 * Not part of the public ABI (that is include/kernelHandler.h).
 */
#ifndef VIT_HIP_VIT_KERNELS_H
#define VIT_HIP_VIT_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

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

/* scale byte and multiplier of one 32-element block from its largest magnitude: the smallest power of two
 * that brings the block inside e4m3's range, 2^E with E = ceil(log2(amax / 448)).  (OCP MX v1.0 suggests
 * floor(log2(amax)) - 8, which lets maxima with a significand above 1.75 saturate at 448 -- a 12.5 % clip of the
 * block's largest element; rounding the exponent up instead costs at most one bit of the smallest ones.)  Zero /
 * subnormal maxima take the smallest scale the multiplier can undo. */
__device__ __forceinline__ void mx_block_scale(float amax, unsigned &scale_byte, float &mult)
{
    const unsigned bits = __builtin_bit_cast(unsigned, amax);
    int e = (int)((bits >> 23) & 0xff) - 127 - 8 + ((bits & 0x7fffff) > 0x600000 ? 1 : 0);   /* significand > 1.75 */
    e = e < -126 ? -126 : e;
    scale_byte = (unsigned)(e + 127);
    mult = __builtin_bit_cast(float, (unsigned)(127 - e) << 23);                    /* 2^-e, exact */
}

/* Scale bytes of an ACTIVATION MX tensor [rows][K]: act_scales[ceil(K/512)][4][rows][4] -- K steps in groups of four, lane
 * group g (block b of a K step is read by lane group 2 (b & 1) + (b >> 1)), row, K step within the group.  The lane that
 * owns (row, g) in the GEMM takes its scales of FOUR K steps with one dword load (gemm_mx.hip load_a): with one byte per
 * (K step, group, row) -- the weights' layout, which travels by LDS-DMA and stays as it is -- every fragment and K step
 * cost a 64-lane byte load, as dear on the address path as a 16-byte one (K = 3072: 10 % of fc2).  Bytes of K steps
 * beyond K / 128 in the last group are never written nor used. */
__device__ __forceinline__ size_t mx_act_scale_index(int ks, int blk, size_t row, int rows)
{
    return ((((size_t)(ks >> 2) * 4 + (2 * (blk & 1) + (blk >> 1))) * (size_t)rows + row) << 2) + (size_t)(ks & 3);
}

/* Records `msg` as the calling thread's last error and returns `code`. */
int vh_fail(int code, const char *fmt, ...);
/* Converts a hipError_t into the launcher return convention, recording text. */
int vh_hip_status(hipError_t e, const char *what);

/* attention_tiled.hip: the shapes the resident-K/V attention kernel does not take. */
int vh_attention_tiled(void *stream, const float *qkv, void *output, int out_bf16, int lowp, int n_images, int tokens,
                       int embed_dim, int num_heads);

/* gemm_mfma.hip: token 0 of every image = class token + pos_embed[0] (ViT_seq.c:90-93,114-117) */
int vh_cls_rows(hipStream_t st, const float *cls_token, const float *pos_embed, float *tokens, int n_images,
                int tokens_per_image, int embed_dim);

/* Launch state that is per DEVICE (a process may hold contexts on several GPUs, one host thread each):
 * hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the CU count are cached per device id. */
/* The caches are atomics: several host threads (one per device, or several contexts on one device) may fill a slot at
 * the same time; filling it twice is harmless, a torn read is not.  vh_init refuses device ids >= VH_MAX_DEVICES. */
#define VH_MAX_DEVICES 64
static inline int vh_current_device(void)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= VH_MAX_DEVICES)
        return -1;
    return dev;
}
static inline int vh_device_cus(int dev)
{
    static std::atomic<int> cus[VH_MAX_DEVICES];
    if (dev < 0 || dev >= VH_MAX_DEVICES)
        return 256;
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}
/* once per (kernel instantiation, device): raise the dynamic-LDS limit of `func` to `bytes` */
#define VH_SET_LDS_ONCE(func, bytes)                                                         \
    do {                                                                                     \
        static std::atomic<bool> vh_attr_set_[VH_MAX_DEVICES];                               \
        const int vh_dev_ = vh_current_device();                                             \
        if (vh_dev_ < 0)                                                                     \
            return vh_fail(1, "no current HIP device (or device id >= %d)", VH_MAX_DEVICES); \
        if (!vh_attr_set_[vh_dev_].load(std::memory_order_acquire)) {                        \
            VH_TRY(hipFuncSetAttribute((const void *)(func), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
            vh_attr_set_[vh_dev_].store(true, std::memory_order_release);                    \
        }                                                                                    \
    } while (0)

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
