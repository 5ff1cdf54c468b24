/*
 * attention_p3.hip -- softmax(Q K^T / sqrt(D)) V per (image, head) on PRE-SPLIT operands.
 *
 * Replaces QKV_TO_SCOREV (multihead.cl:65-137; host ViT_opencl.c:539-565); CPU statement
 * multihead_attn_seq, ViT_seq.c:192-262.  Same arithmetic as attention_f32.hip -- both products are
 * fp32 products formed from the exact three-part bf16 splits, six v_mfma_f32_32x32x16_bf16 per block,
 * same k assignment, same softmax -- and the same results bit for bit.  What differs is who splits:
 * attention_f32.hip reads fp32 Q|K|V rows and every one of its 7 waves splits every K and V fragment
 * again (176 M VALU instructions per launch: the kernel is VALU-bound at 32 % MFMA busy).  Here the QKV
 * projection's epilogue has already written Q, K and V as planes [3E/32][3][rows][32] (gemm_p3.hip,
 * OUT_PLANES), so K and V fragments go LDS -> MFMA operand with no arithmetic and only the
 * probabilities P are split in the kernel.
 *
 * Design (MI355X / CDNA4): a persistent grid of one workgroup per CU walks the (image, head) items; one
 * wave per 32-query tile (7 waves for T = 197).
 *  - One head's K planes are 2 K steps x 3 parts x T rows x 64 B = 80 KB, V the same: together the whole
 *    LDS, so the two operands take turns instead of rotating three buffers: the K buffer is refilled
 *    (LDS-DMA) with K of item n+1 while item n runs softmax and P.V, the V buffer with V of item n+1 while
 *    item n+1 runs Q.K^T.  Two workgroup barriers per item mark the hand-overs; every DMA has half an item
 *    (~8 us) to land.
 *  - K fragment (32 keys x 16 d, MFMA A operand): one ds_read_b128 per part from 64-byte plane rows whose
 *    16-byte chunks are XOR-swizzled (gemm_common.h swz64; applied to the DMA source address).
 *  - V fragment (32 d x 16 keys, MFMA A operand of O^T = V^T P^T): V is stored [key][d] as it arrives and
 *    read TRANSPOSED with ds_read_b64_tr_b16 -- a 16-lane group takes a 4-key x 16-d block and each lane
 *    receives its d column's 4 keys; two reads per part give the 8 contraction slots, whose keys are the
 *    ones the S^T accumulator registers of this lane half hold (32j + 16t + (e & 3) + 8 (e >> 2) + 4 lh).
 *  - Q fragments (MFMA B operand) come straight from the planes by 16-byte global loads, one item ahead,
 *    into the registers the finished Q.K^T phase has freed.
 *  - S^T = K Q^T puts each query's keys in one lane's registers: register-local softmax plus one lane-half
 *    exchange; an accumulator register is the B operand of the next product after its split -- P never
 *    leaves registers.  The epilogue writes the output projection's planes (16-byte stores after a
 *    v_permlane32_swap per dword), as attention_f32.hip's planes epilogue does.
 */
#include "kernelHandler.h"
#include "vit_kernels.h"
#include "fp32_split.h"
#include "gemm_common.h"

namespace {

constexpr int HD = 64;
constexpr int MAX_LDS = 160 * 1024;
constexpr int MAX_ROWS = 208;            /* 2 buffers x 6 planes x 208 rows x 64 B = 159 744 B */

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));

/* NKT 32-wide key/query tiles: 32*(NKT-1) < T <= 32*NKT.
 * NPL = parts per value of the Q|K|V planes: 3 = the exact bf16 split (six products per block, the fp32 path);
 * 1 = one fp16 part (operands ROUNDED to fp16 by the QKV projection's epilogue, one product per block: the
 * reduced-precision GEMM modes, the arithmetic of attention_f32.hip's NPL = 1).
 * OUTK: 3 = three-part bf16 planes (NPL = 3); 4 = one-part bf16 planes; 0 = fp32 rows [rows][E]; 8 = a block-scaled
 * fp8 (MX) tensor, values [E/128][rows][128] + scales [E/128][4][rows] in out_scales (NPL = 1). */
template <int NKT, int NPL, int OUTK>
__global__ __launch_bounds__(64 * NKT) void attention_p3_kernel(const char *__restrict__ qkv3, char *__restrict__ out3,
                                                               unsigned char *__restrict__ out_scales,
                                                               int T, int E, int H, int n_items, int RB)
{
    static_assert((NPL == 3 && OUTK == 3) || (NPL == 1 && (OUTK == 4 || OUTK == 0 || OUTK == 8)), "parts / output kind");
    typedef typename PartT<NPL>::type part_t;           /* bf16x8 (three parts) or half8 (one part) */
    constexpr int NT = NPL == 3 ? 6 : 1;                 /* products per block */
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int slot = 2 * NPL * RB * 64;                  /* [K step 2][part][RB rows][64 B] */
    char *Kb = smem, *Vb = smem + slot;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int E32 = E >> 5, rb16 = RB >> 4, pieces = 2 * NPL * rb16;
    const size_t prow = (size_t)(n_items / H) * T;       /* rows of the whole activation matrix */

    /* LDS-DMA piece p = 16 rows x 64 B of one (K step, part): lane fills physical chunk (lane & 3) of row
     * 16*rb + (lane >> 2); K rows carry the read swizzle, V rows are linear (read transposed). */
    auto dma = [&](int item, int which /* 1 = K, 2 = V */, char *dst) {
        const int b = item / H, h = item - b * H;
        const int ks0 = which * E32 + 2 * h;
        for (int p = wave; p < pieces; p += NKT) {
            const int gp = p / rb16, rb = p - gp * rb16;                 /* gp = NPL * (K step) + part */
            const int r = 16 * rb + (lane >> 2);
            int c = lane & 3;
            if (which == 1)
                c ^= swz64(lane >> 4);                                   /* (r >> 2) & 3 == (lane >> 4) & 3 */
            const char *src = qkv3 + ((size_t)(ks0 * NPL + gp) * prow + (size_t)b * T + min(r, T - 1)) * 64 + 16 * c;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + p * 1024), 16, 0, 0);
        }
    };

    /* Q fragments of this lane's query (B operand): d = 16g + 8lh .. +7 of part pl */
    const int q = wave * 32 + lr;
    part_t qp[HD / 16][NPL];
    auto load_q = [&](int item) {
        const int b = item / H, h = item - b * H;
        const size_t row = (size_t)b * T + min(q, T - 1);
#pragma unroll
        for (int g = 0; g < HD / 16; ++g)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
                qp[g][pl] = __builtin_bit_cast(part_t, *reinterpret_cast<const f32x4 *>(
                                qkv3 + ((size_t)((2 * h + (g >> 1)) * NPL + pl) * prow + row) * 64 + 16 * (2 * (g & 1) + lh)));
    };

    /* per-lane LDS offsets.  K fragment of key tile j: row 32j + lr, chunk 2(g & 1) + lh of K step g >> 1;
     * the last tile may run past the buffer and uses a clamped row. */
    int kofs[2], kofs_last[2];
    {
        const int rl = min(32 * (NKT - 1) + lr, RB - 1);
#pragma unroll
        for (int gl = 0; gl < 2; ++gl) {
            kofs[gl] = lr * 64 + 16 * ((2 * gl + lh) ^ swz64(lr >> 2));
            kofs_last[gl] = rl * 64 + 16 * ((2 * gl + lh) ^ swz64(rl >> 2));
        }
    }
    /* V transposed read: lane i of a 16-lane group addresses row (i >> 2) of the 4-key block, columns 4(i & 3)..;
     * lanes 16-31 take d columns 16..31; the upper lane half takes the keys 4 further on */
    const int vofs = (4 * lh + ((lane & 15) >> 2)) * 64 + ((lane & 16) + 4 * (lane & 3)) * 2;

    int item = blockIdx.x;
    if (item >= n_items)
        return;
    load_q(item);
    dma(item, 1, Kb);
    __syncthreads();                                     /* vmcnt(0) + barrier: K and Q of the first item */

    for (; item < n_items; item += gridDim.x) {
        const int b = item / H, h = item - b * H;
        const int next = item + gridDim.x;
        dma(item, 2, Vb);                                /* Vb is free: every wave finished P.V of the previous item */

        /* S^T tiles: rows = keys of tile j, column = this lane's query.  The K fragment of step f + 1 (3 parts)
         * is fetched under the six MFMAs of step f (left alone, the compiler issues every read right before its
         * use and the matrix pipe waits out the LDS latency 28 times per item). */
        f32x16 s[NKT];
        part_t kp[3][NPL];
        auto read_k = [&](part_t (&k3)[NPL], int f) {
            const int j = f / (HD / 16), g = f % (HD / 16);
#pragma unroll
            for (int o = 0; o < NPL; ++o) {              /* in the order the products need them: part 0, 2, 1 */
                const int pl = (NPL - o) % NPL;
                const char *base = Kb + ((g >> 1) * NPL + pl) * RB * 64;
                k3[pl] = __builtin_bit_cast(part_t, (j < NKT - 1)
                    ? *reinterpret_cast<const f32x4 *>(base + kofs[g & 1] + j * 32 * 64)
                    : *reinterpret_cast<const f32x4 *>(base + kofs_last[g & 1]));
            }
        };
        read_k(kp[0], 0);
        if (NKT * (HD / 16) > 1)
            read_k(kp[1], 1);
#pragma unroll
        for (int f = 0; f < NKT * (HD / 16); ++f) {
            const int j = f / (HD / 16), g = f % (HD / 16);
            if (g == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    s[j][r] = 0.0f;
            }
            if (f + 2 < NKT * (HD / 16))
                read_k(kp[(f + 2) % 3], f + 2);          /* two steps (12 MFMAs) ahead: one step does not cover the LDS latency */
#pragma unroll
            for (int t = 0; t < NT; ++t)
                s[j] = mfma_part(kp[f % 3][term_w<NPL>(t)], qp[g][term_a<NPL>(t)], s[j]);
            if (NPL == 3) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        __syncthreads();                                 /* V of this item has landed; K buffer and Q registers are free */
        if (next < n_items) {
            dma(next, 1, Kb);
            load_q(next);
        }

        /* Row softmax over keys (ViT_seq.c:211, :216-234), exactly as attention_f32.hip: scale, max subtraction and
         * change of base folded into one fma + exp2, one multiplication by the reciprocal of the row sum. */
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 32 * (NKT - 1) + (r & 3) + 8 * (r >> 2) + 4 * lh;
            s[NKT - 1][r] = key < T ? s[NKT - 1][r] : -INFINITY;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < NKT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                mx = fmaxf(mx, s[j][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float c1 = 1.44269504088896340736f / sqrtf((float)HD);
        const float c2 = -mx * c1;
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < NKT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][r], c1, c2));
                s[j][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32);
        const float inv_sum = 1.0f / sum;
#pragma unroll
        for (int j = 0; j < NKT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                s[j][r] = s[j][r] * inv_sum;

        /* O^T = V^T P^T: rows = d (two 32-wide tiles), column = this lane's query; 16 keys per MFMA, contraction
         * slot (lh, e) = key 32j + 16t + (e & 3) + 8*(e >> 2) + 4*lh = the key of accumulator register 8t + e. */
        f32x16 o[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o[0][r] = 0.0f;
            o[1][r] = 0.0f;
        }
        /* step f = (key tile j, 16-key half t, d tile dt); the V fragment of step f + 1 (3 parts x 2 transposed
         * reads) is fetched under the MFMAs of step f.  A half whose keys are all >= T carries P = 0 and is skipped
         * (only the last tile can have one); its prefetch reads a clamped, valid row block. */
        part_t vq[2][NPL], pp[NPL];
        auto read_v = [&](part_t (&v3)[NPL], int f) {
            const int j = f >> 2, t = (f >> 1) & 1, dt = f & 1;
            const int r0 = (j == NKT - 1) ? min(32 * j + 16 * t, RB - 16) : 32 * j + 16 * t;
#pragma unroll
            for (int o3 = 0; o3 < NPL; ++o3) {
                const int pl = (NPL - o3) % NPL;
                const char *vp = Vb + (dt * NPL + pl) * RB * 64 + r0 * 64 + vofs;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(vp));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(vp + 8 * 64));
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                v3[pl] = __builtin_bit_cast(part_t, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
            }
        };
        read_v(vq[0], 0);
#pragma unroll
        for (int f = 0; f < 4 * NKT; ++f) {
            const int j = f >> 2, t = (f >> 1) & 1, dt = f & 1;
            if (f + 1 < 4 * NKT)
                read_v(vq[(f + 1) & 1], f + 1);
            if (dt == 0)
                split_parts(f32x4{s[j][8 * t], s[j][8 * t + 1], s[j][8 * t + 2], s[j][8 * t + 3]},
                            f32x4{s[j][8 * t + 4], s[j][8 * t + 5], s[j][8 * t + 6], s[j][8 * t + 7]}, pp);
            if (!(j == NKT - 1 && 32 * j + 16 * t >= T)) {
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
                    o[dt] = mfma_part(vq[f & 1][term_w<NPL>(tt)], pp[term_a<NPL>(tt)], o[dt]);
            }
        }

        /* Output.  A lane holds d = 32dt + 8g + 4lh .. +3 of its query in o[dt][4g .. 4g+3].
         * Planes [E/32][parts][rows][32] of the output projection (K step 2h + dt): 8 bytes per part and g; one
         * half-wave exchange per dword gives each half 16 contiguous bytes.  fp32 rows: 16-byte stores as they are. */
        if (q < T) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            constexpr int OPL = OUTK == 3 ? 3 : 1;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                if (OUTK == 0) {
                    float *row = reinterpret_cast<float *>(out3) + ((size_t)b * T + q) * E + h * HD + 32 * dt + 4 * lh;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<f32x4 *>(row + 8 * g) = f32x4{o[dt][4 * g], o[dt][4 * g + 1], o[dt][4 * g + 2], o[dt][4 * g + 3]};
                    continue;
                }
                if (OUTK == 8) {
                    /* columns 64h + 32dt .. +31 of this query = one scale block, held by the lane pair (lh = 0, 1):
                     * byte 8g + 4lh + e.  After the exchange the lower lane stores bytes 0..15, the upper 16..31. */
                    float amax = 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        amax = fmaxf(amax, fabsf(o[dt][r]));
                    amax = fmaxf(amax, __shfl_xor(amax, 32));
                    unsigned sbyte;
                    float mult;
                    mx_block_scale(amax, sbyte, mult);
                    unsigned x[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        x[g] = pack_fp8x4(f32x4{o[dt][4 * g], o[dt][4 * g + 1], o[dt][4 * g + 2], o[dt][4 * g + 3]} * mult);
                    const auto r02 = __builtin_amdgcn_permlane32_swap(x[0], x[2], false, false);
                    const auto r13 = __builtin_amdgcn_permlane32_swap(x[1], x[3], false, false);
                    const int col = h * HD + 32 * dt, ks = col >> 7, blk = (col >> 5) & 3;
                    const size_t row = (size_t)b * T + q;
                    *reinterpret_cast<u32x4 *>(out3 + ((size_t)ks * prow + row) * 128 + 32 * blk + 16 * lh) =
                        u32x4{r02[0], r02[1], r13[0], r13[1]};
                    if (lh == 0)
                        out_scales[mx_act_scale_index(ks, blk, row, prow)] = (unsigned char)sbyte;
                    continue;
                }
                u32x2 pg[4][OPL];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 part[3];
                    const f32x4 v4 = f32x4{o[dt][4 * g], o[dt][4 * g + 1], o[dt][4 * g + 2], o[dt][4 * g + 3]};
                    if (OUTK == 3) {
                        split4(v4, part[0], part[1], part[2]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            part[0][e] = (__bf16)v4[e];
                    }
#pragma unroll
                    for (int pl = 0; pl < OPL; ++pl)
                        pg[g][pl] = __builtin_bit_cast(u32x2, part[pl]);
                }
                char *d3 = out3 + ((size_t)(2 * h + dt) * OPL * prow + (size_t)b * T + q) * 64 + 16 * lh;
#pragma unroll
                for (int pl = 0; pl < OPL; ++pl)
#pragma unroll
                    for (int g = 0; g < 4; g += 2) {
                        const auto r0 = __builtin_amdgcn_permlane32_swap(pg[g][pl][0], pg[g + 1][pl][0], false, false);
                        const auto r1 = __builtin_amdgcn_permlane32_swap(pg[g][pl][1], pg[g + 1][pl][1], false, false);
                        *reinterpret_cast<u32x4 *>(d3 + (size_t)pl * prow * 64 + 16 * g) = u32x4{r0[0], r1[0], r0[1], r1[1]};
                    }
            }
        }

        __syncthreads();      /* K and Q of the next item have landed; every wave is done with V of this one */
    }
}

template <int NKT, int NPL, int OUTK>
int launch_k(hipStream_t st, const char *qkv3, char *out3, unsigned char *out_scales, int n_images, int T, int E, int H)
{
    const int RB = (T + 15) / 16 * 16;
    const size_t lds = (size_t)2 * 2 * NPL * RB * 64;
    VH_SET_LDS_ONCE((attention_p3_kernel<NKT, NPL, OUTK>), MAX_LDS);
    const int num_cus = vh_device_cus(vh_current_device());
    const int n_items = n_images * H;
    const int grid = n_items < num_cus ? n_items : num_cus;
    hipLaunchKernelGGL((attention_p3_kernel<NKT, NPL, OUTK>), dim3(grid), dim3(64 * NKT), lds, st, qkv3, out3, out_scales, T, E, H, n_items, RB);
    VH_LAUNCH_CHECK("attention_p3_kernel");
    return 0;
}


template <int NPL, int OUTK>
int launch_t(hipStream_t st, const char *in, char *out, unsigned char *out_scales, int n_images, int tokens, int embed_dim, int num_heads)
{
    switch ((tokens + 31) / 32) {
    case 1: return launch_k<1, NPL, OUTK>(st, in, out, out_scales, n_images, tokens, embed_dim, num_heads);
    case 2: return launch_k<2, NPL, OUTK>(st, in, out, out_scales, n_images, tokens, embed_dim, num_heads);
    case 3: return launch_k<3, NPL, OUTK>(st, in, out, out_scales, n_images, tokens, embed_dim, num_heads);
    case 4: return launch_k<4, NPL, OUTK>(st, in, out, out_scales, n_images, tokens, embed_dim, num_heads);
    case 5: return launch_k<5, NPL, OUTK>(st, in, out, out_scales, n_images, tokens, embed_dim, num_heads);
    case 6: return launch_k<6, NPL, OUTK>(st, in, out, out_scales, n_images, tokens, embed_dim, num_heads);
    default: return launch_k<7, NPL, OUTK>(st, in, out, out_scales, n_images, tokens, embed_dim, num_heads);
    }
}

int check_args(const char *who, const void *in, const void *out, int n_images, int tokens, int embed_dim, int num_heads)
{
    if (!in || !out)
        return vh_fail(1, "%s: null pointer argument", who);
    if (n_images <= 0 || tokens <= 0 || num_heads <= 0 || embed_dim != num_heads * HD || tokens > MAX_ROWS)
        return vh_fail(1, "%s: needs head_dim 64 and 1 <= tokens <= %d (n=%d tokens=%d embed=%d heads=%d)", who,
                       MAX_ROWS, n_images, tokens, embed_dim, num_heads);
    if ((((uintptr_t)in | (uintptr_t)out) & 15) != 0)
        return vh_fail(1, "%s: pointers must be 16-byte aligned", who);
    return 0;
}

} // namespace

/* qkv_planes [3E/32][3][n_images*tokens][32] (the QKV projection written by vh_launch_linear_p3 with
 * output_planes) -> out_planes [E/32][3][n_images*tokens][32]; head_dim 64, tokens <= 208. */
extern "C" int vh_launch_attention_planes(vh_stream_t s, const void *qkv_planes, void *out_planes, int n_images,
                                          int tokens, int embed_dim, int num_heads)
{
    if (int rc = check_args("vh_launch_attention_planes", qkv_planes, out_planes, n_images, tokens, embed_dim, num_heads))
        return rc;
    return launch_t<3, 3>((hipStream_t)s, static_cast<const char *>(qkv_planes), static_cast<char *>(out_planes), nullptr,
                          n_images, tokens, embed_dim, num_heads);
}

/* The reduced-precision GEMM modes: qkv_planes_f16 [3E/32][n_images*tokens][32] fp16 (vh_launch_linear_planes /
 * vh_launch_linear_mx with output_planes = 2) -> output: one-part bf16 planes [E/32][rows][32] (output_planes = 1, the
 * next GEMM's operand in the bf16 mode) or fp32 rows [rows][embed_dim] (output_planes = 0).  Arithmetic of
 * vh_launch_attention_f16: Q, K, V and the probabilities rounded to fp16, fp32 accumulation and softmax. */
extern "C" int vh_launch_attention_planes_f16(vh_stream_t s, const void *qkv_planes_f16, void *output, int output_planes,
                                              int n_images, int tokens, int embed_dim, int num_heads)
{
    if (int rc = check_args("vh_launch_attention_planes_f16", qkv_planes_f16, output, n_images, tokens, embed_dim, num_heads))
        return rc;
    const char *in = static_cast<const char *>(qkv_planes_f16);
    char *out = static_cast<char *>(output);
    return output_planes ? launch_t<1, 4>((hipStream_t)s, in, out, nullptr, n_images, tokens, embed_dim, num_heads)
                         : launch_t<1, 0>((hipStream_t)s, in, out, nullptr, n_images, tokens, embed_dim, num_heads);
}

/* The same attention writing the output projection's operand of the block-scaled fp8 mode directly: an MX tensor
 * (values [embed_dim/128][rows][128], scales [embed_dim/128][4][rows]; vh_launch_quantize_mx_rows of the fp32 result,
 * byte for byte).  embed_dim % 128 == 0. */
extern "C" int vh_launch_attention_planes_f16_mx(vh_stream_t s, const void *qkv_planes_f16, void *out_values, void *out_scales,
                                                 int n_images, int tokens, int embed_dim, int num_heads)
{
    if (int rc = check_args("vh_launch_attention_planes_f16_mx", qkv_planes_f16, out_values, n_images, tokens, embed_dim, num_heads))
        return rc;
    if (!out_scales || embed_dim % 128 != 0)
        return vh_fail(1, "vh_launch_attention_planes_f16_mx: needs out_scales and embed_dim %% 128 == 0");
    return launch_t<1, 8>((hipStream_t)s, static_cast<const char *>(qkv_planes_f16), static_cast<char *>(out_values),
                          static_cast<unsigned char *>(out_scales), n_images, tokens, embed_dim, num_heads);
}
