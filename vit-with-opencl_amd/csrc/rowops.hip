/*
 * rowops.hip -- the memory-bound row operators: LayerNorm and class softmax.
 *
 * LayerNorm replaces `layerNorm` (layer_norm.cl:3-53; host ViT_opencl.c:444-482),
 * where three work-groups per row each recompute the row statistics and reduce
 * through an 8-step LDS tree.  Here: one 64-lane wave per row, the row held in
 * registers (one HBM read, one HBM write = the algorithmic 2*E*4 bytes per
 * row), 16-byte coalesced loads, and __shfl_xor butterflies for the two sums.
 * CPU statement: layer_norm_seq, ViT_seq.c:120-142 (single pass sum / sum of
 * squares, var = E[x^2] - mean^2, eps added in double).
 *
 * Softmax replaces `softMax` (miniSoftMax.cl:1-50; host ViT_opencl.c:750-779):
 * one 256-thread block per row of logits.  CPU statement: Softmax_seq,
 * ViT_seq.c:372-397.
 */
#include "kernelHandler.h"
#include "vit_kernels.h"
#include "fp32_split.h"

namespace {

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        v += __shfl_xor(v, m);
    return v;
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

/* NV = 16-byte chunks per lane; handles embed_dim <= 256*NV, embed_dim % 4 == 0. */
template <int NV, int OUTK> /* OUTK: 0 fp32 rows */
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ in,
                                                        const float *__restrict__ gamma,
                                                        const float *__restrict__ beta,
                                                        void *__restrict__ out, int rows, int E,
                                                        long in_stride, long out_stride, double eps,
                                                        float out_mult)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows)
        return;
    const int nvec = E >> 2;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(in + (size_t)row * in_stride);
    f32x4 x[NV];
    float sum = 0.0f, sq = 0.0f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int idx = c * 64 + lane;
        if (idx < nvec) {
            x[c] = src[idx];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sum += x[c][e];
                sq += x[c][e] * x[c][e];
            }
        }
    }
    sum = wave_sum(sum);
    sq = wave_sum(sq);
    const float mean = sum / (float)E;
    const float var = sq / (float)E - mean * mean;
    const float inv_std = 1.0f / sqrtf((float)((double)var + eps));

    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(gamma);
    const f32x4 *b4 = reinterpret_cast<const f32x4 *>(beta);
    f32x4 *dst = reinterpret_cast<f32x4 *>(static_cast<float *>(out) + (size_t)row * out_stride);
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int idx = c * 64 + lane;
        if (idx < nvec) {
            const f32x4 g = g4[idx], bb = b4[idx];
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                y[e] = (x[c][e] - mean) * inv_std * g[e] + bb[e];
            dst[idx] = y;
        }
    }
}

/* LayerNorm whose only consumer is the pre-split GEMM (gemm_p3.hip): the result is split exactly into three
 * bf16 parts and written as planes [E/32][3][rows][32].  One wave per row as above (same arithmetic, same
 * summation order: identical values), 16 rows per workgroup; the parts go through an LDS image of the
 * planes' own order, [K step][part][16 rows][64 B], so that every global store instruction writes one
 * contiguous, aligned KiB (16 rows x 64 B of one K step and part) instead of eight 64-byte pieces 19 MB apart. */
constexpr int LN3_ROWS = 16;

/* Workgroup barrier for LDS hand-overs only (lgkmcnt): global loads stay in flight across it.  __syncthreads() is a
 * full fence and drains vmcnt, i.e. it would wait for the NEXT row group's loads issued a moment earlier. */
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

/* Persistent: a workgroup walks row groups blockIdx.x, blockIdx.x + gridDim.x, ... and loads the rows of the next
 * group before it normalises, stages and stores the current one -- with one group per workgroup the loads of a
 * workgroup were in flight for less than half of its lifetime and the kernel ran at the load latency, not at the
 * HBM rate.  FULL: every lane holds NV valid chunks (E = 256 NV: 768, 1024, 1280): no per-lane condition anywhere. */
template <int NV, int NPL, bool FULL>   /* NPL parts per value: 3 = exact split, 1 = rounded to bf16 (the bf16-operand mode) */
__global__ __launch_bounds__(64 * LN3_ROWS) void layernorm_p3_kernel(const float *__restrict__ in,
                                                                    const float *__restrict__ gamma,
                                                                    const float *__restrict__ beta,
                                                                    char *__restrict__ planes, int rows, int E,
                                                                    long in_stride, double eps)
{
    extern __shared__ __attribute__((aligned(16))) char ln_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = E >> 2;
    const int ngroups = (rows + LN3_ROWS - 1) / LN3_ROWS, stride = gridDim.x;
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(gamma);
    const f32x4 *b4 = reinterpret_cast<const f32x4 *>(beta);

    auto load_rows = [&](f32x4 (&x)[NV], int group) {   /* unconditional: rows / chunks past the end repeat the last one */
        const int r = min(group * LN3_ROWS + wave, rows - 1);
        const f32x4 *src = reinterpret_cast<const f32x4 *>(in + (size_t)r * in_stride);
#pragma unroll
        for (int c = 0; c < NV; ++c)
            x[c] = src[FULL ? c * 64 + lane : min(c * 64 + lane, nvec - 1)];
    };
    auto process = [&](const f32x4 (&x)[NV], int group) {
        const int row0 = group * LN3_ROWS;
        float sum = 0.0f, sq = 0.0f;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            if (FULL || c * 64 + lane < nvec) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sum += x[c][e];
                    sq += x[c][e] * x[c][e];
                }
            }
        }
        sum = wave_sum(sum);
        sq = wave_sum(sq);
        const float mean = sum / (float)E;
        const float var = sq / (float)E - mean * mean;
        const float inv_std = 1.0f / sqrtf((float)((double)var + eps));
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int idx = c * 64 + lane;
            if (FULL || idx < nvec) {
                const f32x4 g = g4[idx], bb = b4[idx];
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    y[e] = (x[c][e] - mean) * inv_std * g[e] + bb[e];
                bf16x4 part[3];
                if (NPL == 3)
                    split4(y, part[0], part[1], part[2]);
                else
                    part[0] = bf16x4{(__bf16)y[0], (__bf16)y[1], (__bf16)y[2], (__bf16)y[3]};
                char *d = ln_lds + ((size_t)(idx >> 3) * NPL * LN3_ROWS + wave) * 64 + 8 * (idx & 7);
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    *reinterpret_cast<bf16x4 *>(d + pl * LN3_ROWS * 64) = part[pl];
            }
        }
        lds_barrier();
        /* copy-out: piece = (K step, part) = 1 KiB = 16 rows x 64 B; lane l moves 16 bytes of row l / 4 */
        const int pieces = (E >> 5) * NPL;
        if (row0 + (lane >> 2) < rows)
            for (int pc = wave; pc < pieces; pc += LN3_ROWS)
                *reinterpret_cast<f32x4 *>(planes + ((size_t)pc * rows + row0) * 64 + 16 * lane) =
                    *reinterpret_cast<const f32x4 *>(ln_lds + pc * 1024 + 16 * lane);
        lds_barrier();                                   /* the image is free for the next group */
    };

    f32x4 xa[NV], xb[NV];
    int g0 = blockIdx.x;
    if (g0 >= ngroups)
        return;
    load_rows(xa, g0);
    for (;;) {
        const int g1 = g0 + stride;
        load_rows(xb, min(g1, ngroups - 1));
        process(xa, g0);
        if (g1 >= ngroups)
            break;
        const int g2 = g1 + stride;
        load_rows(xa, min(g2, ngroups - 1));
        process(xb, g1);
        if (g2 >= ngroups)
            break;
        g0 = g2;
    }
}

/* LayerNorm whose only consumer is the block-scaled fp8 GEMM (gemm_mx.hip): one wave per row as above; a lane
 * holds 4 consecutive values, the 8 lanes 8m .. 8m+7 one 32-element scale block (three shuffles for its maximum);
 * values and scales go through an LDS image in the MX tensor's own order ([K step][16 rows][128 B], then the scale
 * bytes [K steps / 4][4 lane groups][16 rows][4]: vit_kernels.h mx_act_scale_index) so that the global stores are
 * contiguous 2 KiB runs and one dword per (row, lane group, four K steps). */
template <int NV, bool FULL>   /* persistent and FULL as layernorm_p3_kernel */
__global__ __launch_bounds__(64 * LN3_ROWS) void layernorm_mx_kernel(const float *__restrict__ in, const float *__restrict__ gamma,
                                                                    const float *__restrict__ beta, char *__restrict__ values,
                                                                    unsigned char *__restrict__ scales, int rows, int E,
                                                                    long in_stride, double eps)
{
    extern __shared__ __attribute__((aligned(16))) char ln_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = E >> 2, ksteps = E >> 7, sgroups = (ksteps + 3) >> 2;
    char *lds_scales = ln_lds + ksteps * LN3_ROWS * 128;
    for (int i = threadIdx.x; i < sgroups * 4 * LN3_ROWS; i += blockDim.x)   /* bytes of K steps beyond the last stay zero */
        reinterpret_cast<unsigned *>(lds_scales)[i] = 0u;
    __syncthreads();
    const int ngroups = (rows + LN3_ROWS - 1) / LN3_ROWS, stride = gridDim.x;
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(gamma);
    const f32x4 *b4 = reinterpret_cast<const f32x4 *>(beta);

    auto load_rows = [&](f32x4 (&x)[NV], int group) {
        const int r = min(group * LN3_ROWS + wave, rows - 1);
        const f32x4 *src = reinterpret_cast<const f32x4 *>(in + (size_t)r * in_stride);
#pragma unroll
        for (int c = 0; c < NV; ++c)
            x[c] = src[FULL ? c * 64 + lane : min(c * 64 + lane, nvec - 1)];
    };
    auto process = [&](const f32x4 (&x)[NV], int group) {
        const int row0 = group * LN3_ROWS;
        float sum = 0.0f, sq = 0.0f;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            if (FULL || c * 64 + lane < nvec) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sum += x[c][e];
                    sq += x[c][e] * x[c][e];
                }
            }
        }
        sum = wave_sum(sum);
        sq = wave_sum(sq);
        const float mean = sum / (float)E;
        const float var = sq / (float)E - mean * mean;
        const float inv_std = 1.0f / sqrtf((float)((double)var + eps));
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int idx = c * 64 + lane;
            if (FULL || idx < nvec) {          /* E % 128 == 0: whole groups of 8 lanes are in or out together */
                const f32x4 g = g4[idx], bb = b4[idx];
                f32x4 y;
                float amax = 0.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y[e] = (x[c][e] - mean) * inv_std * g[e] + bb[e];
                    amax = fmaxf(amax, fabsf(y[e]));
                }
                amax = fmaxf(amax, __shfl_xor(amax, 1));
                amax = fmaxf(amax, __shfl_xor(amax, 2));
                amax = fmaxf(amax, __shfl_xor(amax, 4));
                unsigned sbyte;
                float mult;
                mx_block_scale(amax, sbyte, mult);
                const int k = 4 * idx, ks = k >> 7, blk = (k >> 5) & 3;
                *reinterpret_cast<unsigned *>(ln_lds + (ks * LN3_ROWS + wave) * 128 + (k & 127)) = pack_fp8x4(y * mult);
                if ((lane & 7) == 0)
                    lds_scales[((((ks >> 2) * 4 + 2 * (blk & 1) + (blk >> 1)) * LN3_ROWS + wave) << 2) + (ks & 3)] = (char)sbyte;
            }
        }
        lds_barrier();
        /* copy-out: values piece = half a K step's image (8 rows x 128 B = 1 KiB, lane moves 16 B of row l / 8);
         * scales: one dword (four K steps) per (group of K steps, lane group, row) */
        const int vpieces = ksteps * 2;
        for (int pc = wave; pc < vpieces; pc += LN3_ROWS) {
            const int ks = pc >> 1, r = 8 * (pc & 1) + (lane >> 3);
            if (row0 + r < rows)
                *reinterpret_cast<f32x4 *>(values + ((size_t)ks * rows + row0 + r) * 128 + 16 * (lane & 7)) =
                    *reinterpret_cast<const f32x4 *>(ln_lds + (ks * LN3_ROWS + r) * 128 + 16 * (lane & 7));
        }
        if ((int)threadIdx.x < sgroups * 4 * LN3_ROWS) {
            const int pair = threadIdx.x / LN3_ROWS, r = threadIdx.x % LN3_ROWS;   /* pair = (K-step group, lane group) */
            if (row0 + r < rows)
                reinterpret_cast<unsigned *>(scales)[(size_t)pair * rows + row0 + r] = reinterpret_cast<const unsigned *>(lds_scales)[threadIdx.x];
        }
        lds_barrier();
    };

    f32x4 xa[NV], xb[NV];
    int g0 = blockIdx.x;
    if (g0 >= ngroups)
        return;
    load_rows(xa, g0);
    for (;;) {
        const int g1 = g0 + stride;
        load_rows(xb, min(g1, ngroups - 1));
        process(xa, g0);
        if (g1 >= ngroups)
            break;
        const int g2 = g1 + stride;
        load_rows(xa, min(g2, ngroups - 1));
        process(xb, g1);
        if (g2 >= ngroups)
            break;
        g0 = g2;
    }
}

/* Every `stride`-th row of a row-major fp32 matrix / of each plane of a planes tensor, compacted: the class-token
 * rows (row b * tokens of image b) that the classifier reads.  16 bytes per thread. */
__global__ void gather_rows_kernel(const char *__restrict__ src, char *__restrict__ dst, int n_planes, int dst_rows,
                                   size_t src_plane_bytes, int row_bytes, size_t src_row_stride_bytes)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per_row = row_bytes >> 4;
    const size_t total = (size_t)n_planes * dst_rows * per_row;
    if (idx >= total)
        return;
    const int c = (int)(idx % per_row);
    const size_t r = idx / per_row;
    const int plane = (int)(r / dst_rows), row = (int)(r % dst_rows);
    *reinterpret_cast<f32x4 *>(dst + ((size_t)plane * dst_rows + row) * row_bytes + 16 * c) =
        *reinterpret_cast<const f32x4 *>(src + plane * src_plane_bytes + row * src_row_stride_bytes + 16 * c);
}

constexpr int SM_THREADS = 256;
constexpr int SM_MAX_PER_THREAD = 8; /* rows up to 2048 entries stay in registers */

__global__ __launch_bounds__(SM_THREADS) void softmax_kernel(const float *__restrict__ in,
                                                            float *__restrict__ out, int length)
{
    __shared__ float red[SM_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *src = in + (size_t)blockIdx.x * length;
    float *dst = out + (size_t)blockIdx.x * length;

    float v[SM_MAX_PER_THREAD];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < SM_MAX_PER_THREAD; ++i) {
        const int idx = tid + i * SM_THREADS;
        v[i] = idx < length ? src[idx] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    mx = wave_max(mx);
    if (lane == 0)
        red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();

    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < SM_MAX_PER_THREAD; ++i) {
        v[i] = expf(v[i] - mx); /* exp(-inf) = 0 for the padding */
        sum += v[i];
    }
    sum = wave_sum(sum);
    if (lane == 0)
        red[wave] = sum;
    __syncthreads();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
#pragma unroll
    for (int i = 0; i < SM_MAX_PER_THREAD; ++i) {
        const int idx = tid + i * SM_THREADS;
        if (idx < length)
            dst[idx] = v[i] / sum;
    }
}

} // namespace

static int launch_layer_norm(vh_stream_t s, const float *input, const float *weight, const float *bias,
                             void *output, int out_kind, float out_mult, int rows, int embed_dim,
                             long in_row_stride, long out_row_stride, double eps)
{
    if (!input || !weight || !bias || !output)
        return vh_fail(1, "vh_launch_layer_norm: null pointer argument");
    if (rows <= 0 || embed_dim <= 0 || embed_dim % 4 != 0 || embed_dim > 2048)
        return vh_fail(1, "vh_launch_layer_norm: embed_dim=%d must be a multiple of 4, <= 2048", embed_dim);
    if (in_row_stride % 4 != 0 || out_row_stride % 4 != 0 || in_row_stride < embed_dim ||
        out_row_stride < embed_dim)
        return vh_fail(1, "vh_launch_layer_norm: row strides must be multiples of 4 floats and >= embed_dim");
    hipStream_t st = (hipStream_t)s;
    const dim3 grid((rows + 3) / 4), block(256);
    const int nv = (embed_dim / 4 + 63) / 64;
#define VH_LN_K(NV, OUTK)                                                                          \
    hipLaunchKernelGGL((layernorm_kernel<NV, OUTK>), grid, block, 0, st, input, weight, bias, output, \
                       rows, embed_dim, in_row_stride, out_row_stride, eps, out_mult)
#define VH_LN(NV)                                                                                   \
    do {                                                                                            \
        VH_LN_K(NV, 0);                                                                             \
    } while (0)
    if (nv <= 3) VH_LN(3);
    else if (nv <= 4) VH_LN(4);
    else if (nv <= 5) VH_LN(5);
    else VH_LN(8);
#undef VH_LN
#undef VH_LN_K
    VH_LAUNCH_CHECK("layernorm_kernel");
    return 0;
}

extern "C" int vh_launch_layer_norm(vh_stream_t s, const float *input, const float *weight,
                                    const float *bias, float *output, int rows, int embed_dim,
                                    long in_row_stride, long out_row_stride, double eps)
{
    return launch_layer_norm(s, input, weight, bias, output, 0, 1.0f, rows, embed_dim, in_row_stride,
                             out_row_stride, eps);
}

extern "C" int vh_launch_layer_norm_planes(vh_stream_t s, const float *input, const float *weight, const float *bias,
                                           void *out_planes, int parts, int rows, int embed_dim, long in_row_stride,
                                           double eps)
{
    if (!input || !weight || !bias || !out_planes)
        return vh_fail(1, "vh_launch_layer_norm_planes: null pointer argument");
    if (rows <= 0 || embed_dim <= 0 || embed_dim % 32 != 0 || embed_dim > 2048 || ((uintptr_t)out_planes & 15) ||
        (parts != 1 && parts != 3))
        return vh_fail(1, "vh_launch_layer_norm_planes: embed_dim=%d must be a multiple of 32, <= 2048, planes 16-byte "
                          "aligned, parts 1 or 3", embed_dim);
    if (in_row_stride % 4 != 0 || in_row_stride < embed_dim)
        return vh_fail(1, "vh_launch_layer_norm_planes: row stride must be a multiple of 4 floats and >= embed_dim");
    const int nv = (embed_dim / 4 + 63) / 64;
    const size_t lds = (size_t)(embed_dim / 32) * parts * LN3_ROWS * 64;
    const int ngroups = (rows + LN3_ROWS - 1) / LN3_ROWS;
    const int resident = (160 * 1024) / (int)lds < 2 ? 1 : 2;            /* workgroups of 1024 threads per CU */
    const int cap = resident * vh_device_cus(vh_current_device());
    const dim3 grid(ngroups < cap ? ngroups : cap), block(64 * LN3_ROWS);
    hipStream_t st = (hipStream_t)s;
#define VH_LN3_F(NV, NPL, FULL)                                                                           \
    do {                                                                                                  \
        VH_SET_LDS_ONCE((layernorm_p3_kernel<NV, NPL, FULL>), 160 * 1024);                                \
        hipLaunchKernelGGL((layernorm_p3_kernel<NV, NPL, FULL>), grid, block, lds, st, input, weight, bias, \
                           static_cast<char *>(out_planes), rows, embed_dim, in_row_stride, eps);         \
    } while (0)
#define VH_LN3_K(NV, NPL)                                                                                 \
    do {                                                                                                  \
        if (embed_dim == 256 * (NV))                                                                      \
            VH_LN3_F(NV, NPL, true);                                                                      \
        else                                                                                              \
            VH_LN3_F(NV, NPL, false);                                                                     \
    } while (0)
#define VH_LN3(NV)                                                                                        \
    do {                                                                                                  \
        if (parts == 3)                                                                                   \
            VH_LN3_K(NV, 3);                                                                              \
        else                                                                                              \
            VH_LN3_K(NV, 1);                                                                              \
    } while (0)
    if (nv <= 3) VH_LN3(3);
    else if (nv <= 4) VH_LN3(4);
    else if (nv <= 5) VH_LN3(5);
    else VH_LN3(8);
#undef VH_LN3
#undef VH_LN3_K
#undef VH_LN3_F
    VH_LAUNCH_CHECK("layernorm_p3_kernel");
    return 0;
}

extern "C" int vh_launch_layer_norm_p3(vh_stream_t s, const float *input, const float *weight, const float *bias,
                                       void *out_planes, int rows, int embed_dim, long in_row_stride, double eps)
{
    return vh_launch_layer_norm_planes(s, input, weight, bias, out_planes, 3, rows, embed_dim, in_row_stride, eps);
}

extern "C" int vh_launch_layer_norm_mx(vh_stream_t s, const float *input, const float *weight, const float *bias,
                                       void *out_values, void *out_scales, int rows, int embed_dim, long in_row_stride,
                                       double eps)
{
    if (!input || !weight || !bias || !out_values || !out_scales)
        return vh_fail(1, "vh_launch_layer_norm_mx: null pointer argument");
    if (rows <= 0 || embed_dim <= 0 || embed_dim % 128 != 0 || embed_dim > 2048 || ((uintptr_t)out_values & 15) || ((uintptr_t)out_scales & 3))
        return vh_fail(1, "vh_launch_layer_norm_mx: embed_dim=%d must be a multiple of 128, <= 2048, values 16-byte aligned", embed_dim);
    if (in_row_stride % 4 != 0 || in_row_stride < embed_dim)
        return vh_fail(1, "vh_launch_layer_norm_mx: row stride must be a multiple of 4 floats and >= embed_dim");
    const int nv = (embed_dim / 4 + 63) / 64;
    const size_t lds = (size_t)(embed_dim / 128) * LN3_ROWS * 128 + (size_t)((embed_dim / 128 + 3) / 4) * 4 * LN3_ROWS * 4;
    const int ngroups = (rows + LN3_ROWS - 1) / LN3_ROWS;
    const int cap = 2 * vh_device_cus(vh_current_device());               /* two workgroups of 1024 threads per CU */
    const dim3 grid(ngroups < cap ? ngroups : cap), block(64 * LN3_ROWS);
    hipStream_t st = (hipStream_t)s;
#define VH_LNMX_F(NV, FULL)                                                                                    \
    hipLaunchKernelGGL((layernorm_mx_kernel<NV, FULL>), grid, block, lds, st, input, weight, bias,             \
                       static_cast<char *>(out_values), static_cast<unsigned char *>(out_scales), rows,      \
                       embed_dim, in_row_stride, eps)
#define VH_LNMX(NV)                                                                                            \
    do {                                                                                                       \
        if (embed_dim == 256 * (NV))                                                                           \
            VH_LNMX_F(NV, true);                                                                               \
        else                                                                                                   \
            VH_LNMX_F(NV, false);                                                                              \
    } while (0)
    if (nv <= 3) VH_LNMX(3);
    else if (nv <= 4) VH_LNMX(4);
    else if (nv <= 5) VH_LNMX(5);
    else VH_LNMX(8);
#undef VH_LNMX
#undef VH_LNMX_F
    VH_LAUNCH_CHECK("layernorm_mx_kernel");
    return 0;
}

/* dst[p][i][:] = src[p][i * row_stride][:] for n_planes planes of src_rows rows of row_bytes bytes each
 * (n_planes = 1, row_bytes = 4 * cols: a row-major fp32 matrix; planes tensors: row_bytes = 64) */
extern "C" int vh_launch_gather_rows(vh_stream_t s, const void *src, void *dst, int n_planes, int src_rows, int dst_rows,
                                     int row_bytes, int row_stride)
{
    if (!src || !dst || n_planes <= 0 || src_rows <= 0 || dst_rows <= 0 || row_bytes <= 0 || row_bytes % 16 != 0 || row_stride <= 0 ||
        (size_t)(dst_rows - 1) * row_stride >= (size_t)src_rows || (((uintptr_t)src | (uintptr_t)dst) & 15))
        return vh_fail(1, "vh_launch_gather_rows: bad argument (row_bytes %% 16 == 0, (dst_rows-1)*row_stride < src_rows, 16-byte aligned)");
    const size_t threads = (size_t)n_planes * dst_rows * (row_bytes / 16);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)s,
                       static_cast<const char *>(src), static_cast<char *>(dst), n_planes, dst_rows, (size_t)src_rows * row_bytes,
                       row_bytes, (size_t)row_stride * row_bytes);
    VH_LAUNCH_CHECK("gather_rows_kernel");
    return 0;
}

extern "C" int vh_launch_softmax(vh_stream_t s, const float *input, float *output, int rows,
                                 int length)
{
    if (!input || !output)
        return vh_fail(1, "vh_launch_softmax: null pointer argument");
    if (rows <= 0 || length <= 0 || length > SM_THREADS * SM_MAX_PER_THREAD)
        return vh_fail(1, "vh_launch_softmax: length=%d must be in 1..%d", length,
                       SM_THREADS * SM_MAX_PER_THREAD);
    hipLaunchKernelGGL(softmax_kernel, dim3(rows), dim3(SM_THREADS), 0, (hipStream_t)s, input, output,
                       length);
    VH_LAUNCH_CHECK("softmax_kernel");
    return 0;
}
