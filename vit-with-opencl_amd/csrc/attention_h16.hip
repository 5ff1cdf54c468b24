/*
 * attention_h16.hip -- softmax(Q K^T / sqrt(D)) V on fp16 planes for the shapes of ViT-H/14 (head_dim 80, T = 257) in
 * the reduced-precision GEMM modes (BASELINE config 5).
 *
 * Same operator as QKV_TO_SCOREV (multihead.cl:65-137; host ViT_opencl.c:539-565), CPU statement multihead_attn_seq,
 * ViT_seq.c:192-262; the arithmetic of vh_launch_attention_f16 (Q, K, V and the probabilities rounded to fp16, fp32
 * accumulation and softmax).  The streaming kernel (attention_tiled.hip) serves these shapes from fp32 rows: every
 * 64-query workgroup streams the whole K and V of its head through LDS in 5 + 5 chunk steps, each a load latency long
 * -- 0.63 ms per layer where the matrix work is 0.05.  Here Q|K|V arrive as one-part fp16 planes
 * [3E/32][rows][32] (the projection's epilogue, output_planes = 2), and one head's K and V -- 2 x 3 planes x 272 rows x
 * 64 B = 102 KB -- are RESIDENT in LDS:
 *  - persistent grid, one workgroup of 9 waves per CU walking (image, head) items; a wave owns a 16-query tile
 *    (v_mfma_f32_16x16x16_f16, S^T = K Q^T so that the probabilities stay in registers as the B operand of
 *    O^T = V^T P^T, exactly as attention_tiled.hip) and the 17 tiles of T = 257 take two rounds per item;
 *  - K and V take turns as in attention_p3.hip: V of item n lands (LDS-DMA) under Q.K^T of its first round, K of item
 *    n+1 under the second round's softmax and P.V; three workgroup barriers per item;
 *  - head_dim 80 is not a multiple of the planes' 32 columns: head h starts at column 80h = 32 p0 + 16 sh, so the
 *    three planes p0 .. p0+2 are staged and every 16-wide d group g sits in plane (g + sh) >> 1, half (g + sh) & 1;
 *  - K fragment (16 keys x 4 d per lane group): ds_read_b64 from rows whose 16-byte chunks carry gemm_common.h's
 *    swz64 (conflict-free); V fragment (16 d x 4 keys): ds_read_b64_tr_b16 from linear rows, one per MFMA.
 * Output: fp32 rows [rows][E] (the caller quantises / rounds them into the next GEMM's operand: head_dim 80 does not
 * tile the 32-column blocks of those formats).
 */
#include "kernelHandler.h"
#include "vit_kernels.h"
#include "gemm_common.h"

namespace {

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int MAX_LDS = 160 * 1024;

__device__ __forceinline__ half4 to_half4(const f32x4 &v)
{
    half4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = v[e];
        asm("" : "+v"(x));   /* no multiply + convert fusion: rounded twice everywhere (fp32_split.h, split_parts) */
        h[e] = (_Float16)x;
    }
    return h;
}

template <int HD, int NJ, int NW> /* NJ 16-key tiles (T <= 16 NJ), NW waves; two rounds of NW 16-query tiles cover T <= 32 NW */
__global__ __launch_bounds__(64 * NW) void attention_h16_kernel(const char *__restrict__ qkvh, float *__restrict__ out,
                                                               int T, int E, int H, int n_items, float scale_log2e)
{
    constexpr int G = HD / 16;                      /* 16-wide d groups */
    constexpr int PLN = (HD + 16 + 31) / 32;        /* planes staged per operand (any 16-column offset of the head) */
    constexpr int RB = 16 * NJ;                     /* rows per staged plane */
    constexpr int ROUNDS = 2;
    static_assert(HD % 16 == 0 && NJ <= 2 * NW, "shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *Kb = smem, *Vb = smem + PLN * RB * 64;

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int E32 = E >> 5;
    const size_t prow = (size_t)(n_items / H) * T;

    /* LDS-DMA piece p = 16 rows x 64 B of one staged plane; K rows carry the read swizzle, V rows are linear */
    auto dma = [&](int item, int which /* 1 = K, 2 = V */, char *dst) {
        const int b = item / H, h = item - b * H;
        const int p0 = (HD * h) >> 5;
        for (int p = wave; p < PLN * NJ; p += NW) {
            const int plane = p / NJ, rb = p - plane * NJ;
            const int r = 16 * rb + (lane >> 2);
            int c = lane & 3;
            if (which == 1)
                c ^= swz64(lane >> 4);
            const int pl = min(which * E32 + p0 + plane, 3 * E32 - 1);   /* the window's last plane may lie past V's last: clamp (unused columns) */
            const char *src = qkvh + ((size_t)pl * prow + (size_t)b * T + min(r, T - 1)) * 64 + 16 * c;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + p * 1024), 16, 0, 0);
        }
    };

    /* per-lane LDS offsets inside a staged plane, for the two halves of a 64-byte row */
    int kofs[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
        kofs[hf] = l15 * 64 + 16 * ((2 * hf + (g >> 1)) ^ swz64(l15 >> 2)) + 8 * (g & 1);
    const int vrow = (4 * g + (l15 >> 2)) * 64 + 8 * (l15 & 3);     /* transposed read: lane i addresses row i >> 2, columns 4 (i & 3) .. */

    int item = blockIdx.x;
    if (item >= n_items)
        return;
    dma(item, 1, Kb);
    __syncthreads();                                     /* K of the first item */

    for (; item < n_items; item += gridDim.x) {
        const int b = item / H, h = item - b * H;
        const int c0 = HD * h, p0 = c0 >> 5, sh = (c0 >> 4) & 1;
        const int next = item + gridDim.x;
        dma(item, 2, Vb);                                /* Vb is free: every wave finished P.V of the previous item */

#pragma unroll
        for (int rnd = 0; rnd < ROUNDS; ++rnd) {
            const int tile = rnd * NW + wave;
            const bool active = 16 * tile < T;           /* wave-uniform */
            const int q_row = 16 * tile + l15;

            f32x4 S[NJ];
            if (active) {
                /* this lane's query (MFMA column), the 4 d values of its lane group per 16-wide d group */
                half4 qh[G];
                const size_t qr = (size_t)b * T + min(q_row, T - 1);
#pragma unroll
                for (int s = 0; s < G; ++s) {
                    const int idx = s + sh;
                    qh[s] = *reinterpret_cast<const half4 *>(qkvh + ((size_t)(p0 + (idx >> 1)) * prow + qr) * 64 + 32 * (idx & 1) + 8 * g);
                }
                /* S^T = K Q^T: rows = keys of tile j, column = this lane's query */
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int s = 0; s < G; ++s) {
                        const int idx = s + sh;
                        const half4 kf = *reinterpret_cast<const half4 *>(Kb + (idx >> 1) * (RB * 64) + j * 1024 + ((idx & 1) ? kofs[1] : kofs[0]));
                        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(kf, qh[s], acc, 0, 0, 0);
                    }
                    S[j] = acc;
                }
            }
            if (rnd == 0) {
                __syncthreads();                         /* A: V of this item has landed */
            } else {
                __syncthreads();                         /* A': every wave is done with K of this item */
                if (next < n_items)
                    dma(next, 1, Kb);
            }
            if (active) {
                /* row softmax per query: register r of tile j is key 16j + 4g + r (attention_tiled.hip) */
                float mx = -INFINITY;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (16 * j + 4 * g + r >= T)
                            S[j][r] = -INFINITY;
                        mx = fmaxf(mx, S[j][r]);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float off = -mx * scale_log2e;
                float sum = 0.0f;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        S[j][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(S[j][r], scale_log2e, off));
                        sum += S[j][r];
                    }
                sum += __shfl_xor(sum, 16);
                sum += __shfl_xor(sum, 32);
                const float inv = 1.0f / sum;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    S[j] *= inv;

                /* O^T = V^T P^T: rows = d of group dt, column = this lane's query; lane group g contracts keys 16j + 4g .. +3 */
                f32x4 O[G];
#pragma unroll
                for (int dt = 0; dt < G; ++dt)
                    O[dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (16 * j < T) {                    /* uniform */
                        const half4 ph = to_half4(S[j]);
#pragma unroll
                        for (int dt = 0; dt < G; ++dt) {
                            const int idx = dt + sh;
                            const char *vp = Vb + (idx >> 1) * (RB * 64) + j * 1024 + vrow + 32 * (idx & 1);
                            const s16x4 vt = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(vp));
                            O[dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(half4, vt), ph, O[dt], 0, 0, 0);
                        }
                    }
                }
                /* O^T register r of d group dt: d = 16dt + 4g + r, query = lane & 15 */
                if (q_row < T) {
                    float *o = out + ((size_t)b * T + q_row) * E + (size_t)c0 + 4 * g;
#pragma unroll
                    for (int dt = 0; dt < G; ++dt)
                        *reinterpret_cast<f32x4 *>(o + 16 * dt) = O[dt];
                }
            }
        }
        __syncthreads();      /* B: K of the next item has landed; every wave is done with V of this one */
    }
}

template <int HD, int NJ, int NW>
int launch_h16(hipStream_t st, const char *qkvh, float *out, int n_images, int T, int E, int H)
{
    constexpr int PLN = (HD + 16 + 31) / 32;
    const size_t lds = (size_t)2 * PLN * 16 * NJ * 64;
    VH_SET_LDS_ONCE((attention_h16_kernel<HD, NJ, NW>), MAX_LDS);
    const int num_cus = vh_device_cus(vh_current_device());
    const int n_items = n_images * H;
    const int grid = n_items < num_cus ? n_items : num_cus;
    const float c = 1.4426950408889634f / sqrtf((float)HD);
    hipLaunchKernelGGL((attention_h16_kernel<HD, NJ, NW>), dim3(grid), dim3(64 * NW), lds, st, qkvh, out, T, E, H, n_items, c);
    VH_LAUNCH_CHECK("attention_h16_kernel");
    return 0;
}

} // namespace

/* qkv_planes_f16 [3*embed_dim/32][n_images*tokens][32] fp16 (vh_launch_linear_planes / vh_launch_linear_mx_planes_f16 with
 * the fp16-planes output) -> output fp32 rows [n_images*tokens][embed_dim].  head_dim 80 (ViT-H/14), tokens <= 272. */
extern "C" int vh_launch_attention_planes_f16_hd80(vh_stream_t s, const void *qkv_planes_f16, float *output, int n_images,
                                                   int tokens, int embed_dim, int num_heads)
{
    if (!qkv_planes_f16 || !output)
        return vh_fail(1, "vh_launch_attention_planes_f16_hd80: null pointer argument");
    if (n_images <= 0 || tokens <= 0 || num_heads <= 0 || embed_dim != num_heads * 80 || tokens > 272 || embed_dim % 32 != 0)
        return vh_fail(1, "vh_launch_attention_planes_f16_hd80: needs head_dim 80 and 1 <= tokens <= 272 (n=%d tokens=%d embed=%d heads=%d)",
                       n_images, tokens, embed_dim, num_heads);
    if ((((uintptr_t)qkv_planes_f16 | (uintptr_t)output) & 15) != 0)
        return vh_fail(1, "vh_launch_attention_planes_f16_hd80: pointers must be 16-byte aligned");
    return launch_h16<80, 17, 9>((hipStream_t)s, static_cast<const char *>(qkv_planes_f16), output, n_images, tokens, embed_dim,
                                 num_heads);
}
