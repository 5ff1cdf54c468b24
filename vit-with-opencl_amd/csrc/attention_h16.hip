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
 * 64 B = 102 KB -- are RESIDENT in LDS (since round 4 its Q as well: 153 KB):
 *  - persistent grid, one workgroup of 9 waves per CU walking (image, head) items; a wave owns a 16-query tile
 *    (S^T = K Q^T so that the probabilities stay in registers as the B operand of O^T = V^T P^T, as in
 *    attention_tiled.hip) and the 17 tiles of T = 257 take two rounds per item;
 *  - both products on v_mfma_f32_16x16x32_f16 (round 4; before: the 16-deep v_mfma_f32_16x16x16_f16, which gfx950 issues
 *    in the SAME 16 cycles -- half the rate, tools/mfma_f16_rate_probe.hip): Q.K^T contracts d in three 32-deep steps
 *    (80 = 32 + 32 + 16; the lanes of the last step's upper half carry Q = 0), P.V contracts the keys of two 16-key
 *    tiles per instruction (the lane group's four keys of either tile = the S^T registers it already holds);
 *  - K and V take turns as in attention_p3.hip: V of item n lands (LDS-DMA) under Q.K^T of its first round, K of item
 *    n+1 under the second round's softmax and P.V; three workgroup barriers per item;
 *  - Q travels with K (a third staged operand, 3 x 52 KB = 153 KB of LDS): a round's query fragments are ds_read_b128s
 *    instead of global loads whose latency stood in front of each round's first MFMA (round 4; same-box A/B
 *    tools/attn_h16_ab.py: -1 % with bf16 planes out, -5 % with MX out, where it also ended the register spills);
 *  - head_dim 80 is not a multiple of the planes' 32 columns: head h starts at column 80h = 32 p0 + 16 sh, so the
 *    three planes p0 .. p0+2 are staged and every 16-wide d group g sits in plane (g + sh) >> 1, half (g + sh) & 1;
 *  - K fragment (16 keys x 8 d per lane group): ds_read_b128 from rows whose 16-byte chunks carry gemm_common.h's
 *    swz64 (conflict-free); V fragment (16 d x 2 x 4 keys): two ds_read_b64_tr_b16 from linear rows per MFMA.
 * Output: fp32 rows [rows][E] (OUTK 0), or the output projection's operand directly -- one-part bf16 planes (OUTK 4) or
 * the block-scaled fp8 tensor (OUTK 8).  head_dim 80 does not tile those formats' 32-column blocks: columns 64-79 of an
 * even head share a block with columns 0-15 of the next head.  For those outputs a workgroup therefore walks head
 * PAIRS (2k, 2k+1) back to back -- the K / V hand-over pipeline does not care -- and a lane keeps its four values of
 * the even head's last 16 columns in registers (the same lane holds the same query's columns in the odd head: the
 * tile -> wave map does not depend on the head) until the odd head's first 16 arrive; every block then has its 32
 * values in the four lanes of one query, as the scale needs.  Same values as the fp32 rows followed by
 * vh_launch_quantize_mx_rows / vh_launch_split_rows(parts = 1), byte for byte, without the 2 x 4-byte round trip of
 * every output value through HBM.
 */
#include "kernelHandler.h"
#include "vit_kernels.h"
#include "gemm_common.h"

namespace {

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

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

template <int HD, int NJ, int NW, int OUTK> /* NJ 16-key tiles (T <= 16 NJ), NW waves; two rounds of NW 16-query tiles cover T <= 32 NW */
__global__ __launch_bounds__(64 * NW) void attention_h16_kernel(const char *__restrict__ qkvh, float *__restrict__ out,
                                                               unsigned char *__restrict__ out_scales,
                                                               int T, int E, int H, int n_items, float scale_log2e)
{
    constexpr bool PAIRS = OUTK != 0;               /* walk head pairs (H even): blocks that straddle two heads */
    static_assert(OUTK == 0 || OUTK == 4 || OUTK == 8, "output kind");
    static_assert(!PAIRS || HD == 80, "the carried half block is written for head_dim 80");
    constexpr int G = HD / 16;                      /* 16-wide d groups */
    constexpr int GS = (G + 1) / 2;                 /* 32-deep contraction steps of Q.K^T (the last may be half empty) */
    constexpr int PLN = (HD + 16 + 31) / 32;        /* planes staged per operand (any 16-column offset of the head) */
    constexpr int RB = 16 * NJ;                     /* rows per staged plane */
    constexpr int ROUNDS = 2;
    static_assert(HD % 16 == 0 && NJ <= 2 * NW, "shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *Kb = smem, *Vb = smem + PLN * RB * 64, *Qb = smem + 2 * PLN * RB * 64;

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int E32 = E >> 5;
    const size_t prow = (size_t)(n_items / H) * T;

    /* LDS-DMA piece p = 16 rows x 64 B of one staged plane; Q and K rows carry the read swizzle, V rows are linear */
    auto dma = [&](int item, int which /* 0 = Q, 1 = K, 2 = V */, char *dst) {
        const int b = item / H, h = item - b * H;
        const int p0 = (HD * h) >> 5;
        for (int p = wave; p < PLN * NJ; p += NW) {
            const int plane = p / NJ, rb = p - plane * NJ;
            const int r = 16 * rb + (lane >> 2);
            int c = lane & 3;
            if (which != 2)
                c ^= swz64(lane >> 4);
            const int pl = min(which * E32 + p0 + plane, 3 * E32 - 1);   /* the window's last plane may lie past V's last: clamp (unused columns) */
            const char *src = qkvh + ((size_t)pl * prow + (size_t)b * T + min(r, T - 1)) * 64 + 16 * c;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + p * 1024), 16, 0, 0);
        }
    };

    /* Q.K^T step s: this lane group contracts the 16-wide d group 2s + (g >> 1), its 8 values 8 (g & 1) .. +7 -- chunk
     * 2 ((d group + sh) & 1) + (g & 1) of row l15 of plane (d group + sh) >> 1.  Lanes whose d group lies past the head
     * (the upper half of the last step when G is odd) carry Q = 0 and read the step's lower half again (finite values). */
    int dgrp[GS];
    bool dlive[GS];
#pragma unroll
    for (int s2 = 0; s2 < GS; ++s2) {
        dlive[s2] = 2 * s2 + (g >> 1) < G;
        dgrp[s2] = dlive[s2] ? 2 * s2 + (g >> 1) : 2 * s2;
    }
    const int vrow = (4 * g + (l15 >> 2)) * 64 + 8 * (l15 & 3);     /* transposed read: lane i addresses row i >> 2, columns 4 (i & 3) .. */

    /* the workgroup's n-th item: (image, head) number blockIdx.x + n gridDim.x, or -- PAIRS -- head n & 1 of its
     * (n >> 1)-th head pair; n_items beyond the end */
    auto nth_item = [&](int n) -> int {
        if (!PAIRS)
            return min((int)blockIdx.x + n * (int)gridDim.x, n_items);
        const int pair = (int)blockIdx.x + (n >> 1) * (int)gridDim.x;
        return pair < (n_items >> 1) ? 2 * pair + (n & 1) : n_items;
    };
    int n = 0, item = nth_item(0);
    if (item >= n_items)
        return;
    dma(item, 1, Kb);
    dma(item, 0, Qb);
    __syncthreads();                                     /* K and Q of the first item */

    f32x4 carry[2] = {};                                 /* PAIRS: columns 64..79 of the even head, per round */
    for (; item < n_items; item = nth_item(++n)) {
        const int b = item / H, h = item - b * H;
        const int c0 = HD * h, p0 = c0 >> 5, sh = (c0 >> 4) & 1;
        const int next = nth_item(n + 1);
        dma(item, 2, Vb);                                /* Vb is free: every wave finished P.V of the previous item */

#pragma unroll
        for (int rnd = 0; rnd < ROUNDS; ++rnd) {
            const int tile = rnd * NW + wave;
            const bool active = 16 * tile < T;           /* wave-uniform */
            const int q_row = 16 * tile + l15;

            f32x4 S[NJ];
            if (active) {
                /* this lane's query (MFMA column): the 8 d values of its lane group per 32-deep step, from the staged Q planes
                 * (row 16 tile + l15: the K fragment's address pattern; rows >= T hold a copy of row T - 1, never stored) */
                half8 qh[GS];
                int kofs[GS];
#pragma unroll
                for (int s2 = 0; s2 < GS; ++s2) {
                    const int idx = dgrp[s2] + sh, chunk = 2 * (idx & 1) + (g & 1);
                    kofs[s2] = (idx >> 1) * (RB * 64) + l15 * 64 + 16 * (chunk ^ swz64(l15 >> 2));
                    qh[s2] = *reinterpret_cast<const half8 *>(Qb + tile * 1024 + kofs[s2]);
                    if (!dlive[s2])
                        qh[s2] = half8{};
                }
                /* S^T = K Q^T: rows = keys of tile j, column = this lane's query */
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int s2 = 0; s2 < GS; ++s2) {
                        const half8 kf = *reinterpret_cast<const half8 *>(Kb + j * 1024 + kofs[s2]);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qh[s2], acc, 0, 0, 0);
                    }
                    S[j] = acc;
                }
            }
            if (rnd == 0) {
                __syncthreads();                         /* A: V of this item has landed */
            } else {
                __syncthreads();                         /* A': every wave is done with K and Q of this item */
                if (next < n_items) {
                    dma(next, 1, Kb);
                    dma(next, 0, Qb);
                }
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

                /* O^T = V^T P^T: rows = d of group dt, column = this lane's query; one MFMA contracts the key tiles 2t and 2t + 1:
                 * lane group g the keys 32t + 4g .. +3 (slots 0-3) and 32t + 16 + 4g .. +3 (slots 4-7) -- the S^T registers it
                 * holds.  Keys >= T carry P = 0 exactly (exp2 of -inf) against clamped, finite V rows. */
                f32x4 O[G];
#pragma unroll
                for (int dt = 0; dt < G; ++dt)
                    O[dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int t = 0; t < (NJ + 1) / 2; ++t) {
                    if (32 * t < T) {                    /* uniform */
                        const int j1 = 2 * t + 1 < NJ ? 2 * t + 1 : 2 * t;      /* no second tile: P = 0 against the first one's V */
                        const half4 pl = to_half4(S[2 * t]), pu = 2 * t + 1 < NJ ? to_half4(S[j1]) : half4{};
                        const half8 ph = {pl[0], pl[1], pl[2], pl[3], pu[0], pu[1], pu[2], pu[3]};
#pragma unroll
                        for (int dt = 0; dt < G; ++dt) {
                            const int idx = dt + sh;
                            const char *vp = Vb + (idx >> 1) * (RB * 64) + vrow + 32 * (idx & 1);
                            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(vp + 2 * t * 1024));
                            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(vp + j1 * 1024));
                            O[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                __builtin_bit_cast(half8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}), ph, O[dt], 0, 0, 0);
                        }
                    }
                }
                /* O^T register r of d group dt: d = 16dt + 4g + r, query = lane & 15 */
                if (OUTK == 0) {
                    if (q_row < T) {
                        float *o = out + ((size_t)b * T + q_row) * E + (size_t)c0 + 4 * g;
#pragma unroll
                        for (int dt = 0; dt < G; ++dt)
                            *reinterpret_cast<f32x4 *>(o + 16 * dt) = O[dt];
                    }
                } else {
                    /* 32-column blocks of this head: even head (sh = 0): (dt 0, 1), (dt 2, 3), dt 4 is carried; odd head:
                     * (carry, dt 0), (dt 1, 2), (dt 3, 4).  A block's two halves sit at positions 4g + r and 16 + 4g + r
                     * of its row; the four lanes l15 + 16 g of a query hold the block. */
                    const size_t orow = (size_t)b * T + min(q_row, T - 1);
                    auto emit = [&](const f32x4 &lo, const f32x4 &hi, int col /* first column of the block */) {
                        if (OUTK == 4) {
                            if (q_row < T) {
                                typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
                                char *d = reinterpret_cast<char *>(out) + ((size_t)(col >> 5) * prow + orow) * 64 + 8 * g;
                                *reinterpret_cast<bf16x4_t *>(d) = bf16x4_t{(__bf16)lo[0], (__bf16)lo[1], (__bf16)lo[2], (__bf16)lo[3]};
                                *reinterpret_cast<bf16x4_t *>(d + 32) = bf16x4_t{(__bf16)hi[0], (__bf16)hi[1], (__bf16)hi[2], (__bf16)hi[3]};
                            }
                        } else {
                            float amax = 0.0f;
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                amax = fmaxf(amax, fmaxf(fabsf(lo[e]), fabsf(hi[e])));
                            amax = fmaxf(amax, __shfl_xor(amax, 16));
                            amax = fmaxf(amax, __shfl_xor(amax, 32));
                            unsigned sbyte;
                            float mult;
                            mx_block_scale(amax, sbyte, mult);
                            if (q_row < T) {
                                const int ks = col >> 7, blk = (col >> 5) & 3;
                                char *d = reinterpret_cast<char *>(out) + ((size_t)ks * prow + orow) * 128 + 32 * blk + 4 * g;
                                *reinterpret_cast<unsigned *>(d) = pack_fp8x4(lo * mult);
                                *reinterpret_cast<unsigned *>(d + 16) = pack_fp8x4(hi * mult);
                                if (g == 0)
                                    out_scales[mx_act_scale_index(ks, blk, orow, prow)] = (unsigned char)sbyte;
                            }
                        }
                    };
                    if (sh == 0) {
                        emit(O[0], O[1], c0);
                        emit(O[2], O[3], c0 + 32);
                        carry[rnd] = O[4];
                    } else {
                        emit(carry[rnd], O[0], c0 - 16);
                        emit(O[1], O[2], c0 + 16);
                        emit(O[3], O[4], c0 + 48);
                    }
                }
            }
        }
        __syncthreads();      /* B: K of the next item has landed; every wave is done with V of this one */
    }
}

template <int HD, int NJ, int NW, int OUTK>
int launch_h16(hipStream_t st, const char *qkvh, void *out, void *out_scales, int n_images, int T, int E, int H)
{
    constexpr int PLN = (HD + 16 + 31) / 32;
    const size_t lds = (size_t)3 * PLN * 16 * NJ * 64;      /* K, V and Q of one head */
    VH_SET_LDS_ONCE((attention_h16_kernel<HD, NJ, NW, OUTK>), MAX_LDS);
    const int num_cus = vh_device_cus(vh_current_device());
    const int n_items = n_images * H;
    const int units = OUTK ? n_items / 2 : n_items;      /* workgroups walk head pairs when blocks straddle heads */
    const int grid = units < num_cus ? units : num_cus;
    const float c = 1.4426950408889634f / sqrtf((float)HD);
    hipLaunchKernelGGL((attention_h16_kernel<HD, NJ, NW, OUTK>), dim3(grid), dim3(64 * NW), lds, st, qkvh, static_cast<float *>(out),
                       static_cast<unsigned char *>(out_scales), T, E, H, n_items, c);
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
    return launch_h16<80, 17, 9, 0>((hipStream_t)s, static_cast<const char *>(qkv_planes_f16), output, nullptr, n_images, tokens,
                                    embed_dim, num_heads);
}

/* The same attention writing the output projection's operand directly: output_kind 1 = one-part bf16 planes
 * [embed_dim/32][rows][32] (vh_launch_split_rows(parts = 1) of the fp32 result, byte for byte), 2 = the block-scaled fp8
 * tensor (values [embed_dim/128][rows][128] + out_scales [embed_dim/128][4][rows]: vh_launch_quantize_mx_rows of the
 * fp32 result, byte for byte).  num_heads even (blocks straddle head pairs); embed_dim % 128 == 0 for kind 2. */
extern "C" int vh_launch_attention_planes_f16_hd80_operand(vh_stream_t s, const void *qkv_planes_f16, void *output, void *out_scales,
                                                           int output_kind, int n_images, int tokens, int embed_dim, int num_heads)
{
    if (!qkv_planes_f16 || !output || (output_kind == 2 && !out_scales))
        return vh_fail(1, "vh_launch_attention_planes_f16_hd80_operand: null pointer argument");
    if (n_images <= 0 || tokens <= 0 || num_heads <= 0 || (num_heads & 1) || embed_dim != num_heads * 80 || tokens > 272 ||
        (output_kind != 1 && output_kind != 2) || (output_kind == 2 && embed_dim % 128 != 0))
        return vh_fail(1, "vh_launch_attention_planes_f16_hd80_operand: needs head_dim 80, an even number of heads, 1 <= tokens <= 272, "
                          "output_kind 1 or 2 (n=%d tokens=%d embed=%d heads=%d kind=%d)", n_images, tokens, embed_dim, num_heads, output_kind);
    if ((((uintptr_t)qkv_planes_f16 | (uintptr_t)output) & 15) != 0)
        return vh_fail(1, "vh_launch_attention_planes_f16_hd80_operand: pointers must be 16-byte aligned");
    const char *in = static_cast<const char *>(qkv_planes_f16);
    return output_kind == 1 ? launch_h16<80, 17, 9, 4>((hipStream_t)s, in, output, nullptr, n_images, tokens, embed_dim, num_heads)
                            : launch_h16<80, 17, 9, 8>((hipStream_t)s, in, output, out_scales, n_images, tokens, embed_dim, num_heads);
}
