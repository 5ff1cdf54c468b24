/*
 * attention_tiled.hip -- softmax(Q K^T / sqrt(D)) V for the shapes the resident-K/V kernel
 * (attention_f32.hip: head_dim 64, T <= 208) does not take: any head_dim that is a multiple
 * of 16 up to 128 and any T up to 512 -- ViT-H/14 is T = 257, D = 80, where one head's K and V
 * (2 x 82 KB) no longer fit a CU's LDS together.
 *
 * Same operator as QKV_TO_SCOREV (multihead.cl:65-137; host ViT_opencl.c:539-565), CPU
 * statement multihead_attn_seq, ViT_seq.c:192-262, and the same arithmetic plan as the
 * resident kernel (S^T = K Q^T so that the probabilities stay in registers as the B operand
 * of O^T = V^T P^T; exact two-pass softmax per query, scaled after the dot product,
 * normalised before P.V), on v_mfma_f32_16x16x4_f32 so that D = 80 tiles evenly.
 *
 *  - One workgroup = 4 waves = 64 queries of one (image, head); a wave owns 16 queries
 *    (the MFMA column, lane & 15) and holds their whole score row in registers:
 *    4 x ceil(T/16) accumulators per lane, each lane group (lane >> 4) four keys of a tile.
 *  - K, then V, stream through two 32-key LDS buffers (global_load_dwordx4 -> ds_write_b128,
 *    the next chunk's loads in flight under the current chunk's MFMAs).  Rows are padded to
 *    D + 4 words: the K fragment (ds_read_b128, 16 keys x one 16-byte chunk per lane group)
 *    and the V fragment (ds_read_b32, 16 consecutive d of four keys) are both conflict-free.
 *  - The contraction index of an MFMA is permuted instead of moving data: in K Q^T lane
 *    group g contracts d = 16s + 4g + e in step e (a ds_read_b128 of K and a 16-byte load
 *    of Q give e = 0..3), in V^T P^T it contracts key 16j + 4g + r in step r -- exactly the
 *    key whose probability accumulator register r of tile j already holds.
 *
 * Input rows are the fused projection output [Q(E) | K(E) | V(E)]; output is
 * [n_images*T][E] with heads concatenated (ViT_seq.c:252-258).
 */
#include "kernelHandler.h"
#include "vit_kernels.h"

namespace {

constexpr int KC = 64;   /* keys per LDS chunk (four 16-key MFMA tiles) */
constexpr int TPC = KC / 16;
constexpr int QB_MIN = 64;   /* queries per workgroup = 16 x NW waves, NW = 4 or 6 (whichever leaves fewer idle waves in the last block) */

/* Workgroup barrier for LDS hand-overs only: this wave's LDS operations have completed (lgkmcnt), global loads stay
 * in flight.  __syncthreads() is a full workgroup fence -- it also drains vmcnt, i.e. it waits at every chunk step
 * for the prefetch issued a few hundred cycles earlier, which put the whole load latency into each of the 18 steps. */
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

typedef _Float16 half4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ half4 to_half4(const f32x4 &v)
{
    half4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = v[e];
        asm("" : "+v"(x));   /* no multiply + convert fusion (fp32_split.h, split_parts): rounded twice everywhere */
        h[e] = (_Float16)x;
    }
    return h;
}

/* LOWP (the reduced-precision GEMM modes only): Q, K, V and the probabilities are rounded to fp16 in registers and
 * the four contraction steps of a lane group become ONE v_mfma_f32_16x16x16_f16 -- its lane layout (row l & 15,
 * k = 4 (l >> 4) + i) is exactly the permuted contraction described above.  fp32 accumulation and softmax; the
 * arithmetic of vh_launch_attention_f16 for head dimensions and token counts the resident kernels do not take. */
template <int D, int NJ, bool OUTBF16, bool LOWP, int NW> /* NJ 16-key tiles: T <= 16*NJ; NW waves of 16 queries */
__global__ __launch_bounds__(64 * NW, 2) void attention_tiled_kernel(const float *__restrict__ qkv,
                                                                 void *__restrict__ out, int T, int E, int H,
                                                                 int n_qblocks, float scale_log2e)
{
    constexpr int DS = D + 4;                    /* LDS row stride in words */
    constexpr int NC = (NJ + TPC - 1) / TPC;     /* chunks */
    constexpr int DT = D / 16;                   /* 16-wide d tiles */
    constexpr int PIECES = KC * D / 4;           /* 16-byte pieces per chunk */
    constexpr int NTH = 64 * NW, QB = 16 * NW;
    constexpr int PPT = (PIECES + NTH - 1) / NTH;    /* per thread */
    static_assert(D % 16 == 0 && D <= 128, "head_dim");

    __shared__ __attribute__((aligned(16))) float lds[2][KC * DS];

    const int item = blockIdx.x / n_qblocks, qb = blockIdx.x - item * n_qblocks;
    const int img = item / H, h = item - img * H;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q_row = qb * QB + wave * 16 + l15;
    const bool active = qb * QB + wave * 16 < T;   /* wave-uniform: any of its 16 queries exists */
    const size_t ld = (size_t)3 * E;
    const float *base = qkv + (size_t)img * T * ld + (size_t)h * D;

    /* this lane's query (MFMA column), the d values its lane group contracts */
    f32x4 qreg[DT];
    half4 qh[DT];
    {
        const float *qp = base + (size_t)min(q_row, T - 1) * ld + 4 * g;
#pragma unroll
        for (int s = 0; s < DT; ++s) {
            qreg[s] = *reinterpret_cast<const f32x4 *>(qp + 16 * s);
            if (LOWP)
                qh[s] = to_half4(qreg[s]);
        }
    }

    /* One chunk is in flight in registers, one step ahead of the chunk being consumed: a chunk's MFMAs take a few hundred
     * cycles, its global loads one to two microseconds.  (Two chunks in flight were tried: the compiler still places the
     * loads one step ahead, no effect -- DESIGN 5.) */
    f32x4 stage[1][PPT];
    /* piece p of a chunk = 16 bytes; threads past the last piece repeat it (same bytes to the same LDS address): a
     * per-thread condition here becomes a branch on EXEC, and at its join the compiler drains every load in flight */
    int prow[PPT], pch[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int p = min(tid + NTH * i, PIECES - 1);
        prow[i] = p / (D / 4);
        pch[i] = p - prow[i] * (D / 4);
    }
    auto load_chunk = [&](int c, const float *src) {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int r = min(KC * c + prow[i], T - 1);   /* rows past T: a finite copy, masked below */
            stage[0][i] = *reinterpret_cast<const f32x4 *>(src + (size_t)r * ld + 4 * pch[i]);
        }
    };
    auto store_chunk = [&](int c) {   /* chunk c: registers stage[c & 1] -> LDS buffer c & 1 */
#pragma unroll
        for (int i = 0; i < PPT; ++i)
            *reinterpret_cast<f32x4 *>(&lds[c & 1][prow[i] * DS + 4 * pch[i]]) = stage[0][i];
    };

    /* ---- pass 1: S^T = K Q^T, all key tiles of this wave's 16 queries into registers ---- */
    f32x4 S[NJ];
    load_chunk(0, base + E);
    store_chunk(0);
    lds_barrier();
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (KC * c < T) { /* uniform */
            const bool more = c + 1 < NC && KC * (c + 1) < T;
            if (more)
                load_chunk(c + 1, base + E);
#pragma unroll
            for (int jj = 0; jj < TPC; ++jj) {
                const int j = TPC * c + jj;
                if (j < NJ) {
                    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (active && 16 * j < T) {
                        const float *kp = &lds[c & 1][(jj * 16 + l15) * DS + 4 * g];
#pragma unroll
                        for (int s = 0; s < DT; ++s) {
                            const f32x4 kf = *reinterpret_cast<const f32x4 *>(kp + 16 * s);
                            if (LOWP) {
                                acc = __builtin_amdgcn_mfma_f32_16x16x16f16(to_half4(kf), qh[s], acc, 0, 0, 0);
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], qreg[s][e], acc, 0, 0, 0);
                            }
                        }
                    }
                    S[j] = acc;
                }
            }
            if (more)
                store_chunk(c + 1);
            lds_barrier();
        } else {
#pragma unroll
            for (int jj = 0; jj < TPC; ++jj)
                if (TPC * c + jj < NJ)
                    S[TPC * c + jj] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    }

    /* the first two V chunks on their way while the softmax runs */
    load_chunk(0, base + 2 * E);

    /* ---- row softmax per query: register r of tile j is key 16j + 4g + r ---- */
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
    /* exp((s - max)/sqrt(D)) = exp2(s*c - max*c), c = log2(e)/sqrt(D): one fma + v_exp_f32
     * (the scalar loop scales, subtracts, then calls expf: ViT_seq.c:212,224) */
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
    const float inv = 1.0f / sum; /* the scalar loop divides each entry (ViT_seq.c:232) */
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        S[j] *= inv;

    /* ---- pass 2: O^T = V^T P^T ---- */
    f32x4 O[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
        O[dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    store_chunk(0);
    lds_barrier();
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (KC * c < T) {
            const bool more = c + 1 < NC && KC * (c + 1) < T;
            if (more)
                load_chunk(c + 1, base + 2 * E);
#pragma unroll
            for (int jj = 0; jj < TPC; ++jj) {
                const int j = TPC * c + jj;
                if (j < NJ && active && 16 * j < T) {
                    const float *vp = &lds[c & 1][(jj * 16 + 4 * g) * DS + l15];
                    if (LOWP) {
                        const half4 ph = to_half4(S[j]);
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt)
                            O[dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(
                                to_half4(f32x4{vp[16 * dt], vp[DS + 16 * dt], vp[2 * DS + 16 * dt], vp[3 * DS + 16 * dt]}), ph, O[dt], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int dt = 0; dt < DT; ++dt)
                                O[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[r * DS + 16 * dt], S[j][r], O[dt], 0, 0, 0);
                    }
                }
            }
            if (more)
                store_chunk(c + 1);
            lds_barrier();
        }
    }

    /* O^T register r of d tile dt: d = 16dt + 4g + r, query = lane & 15 */
    if (q_row < T) {
        const size_t o = ((size_t)img * T + q_row) * E + (size_t)h * D + 4 * g;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            if (OUTBF16) {
                typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                const bf16x4 v = {(__bf16)O[dt][0], (__bf16)O[dt][1], (__bf16)O[dt][2], (__bf16)O[dt][3]};
                *reinterpret_cast<bf16x4 *>(static_cast<__bf16 *>(out) + o + 16 * dt) = v;
            } else {
                *reinterpret_cast<f32x4 *>(static_cast<float *>(out) + o + 16 * dt) = O[dt];
            }
        }
    }
}

template <int D, int NJ, int NW>
int launch_nw(hipStream_t st, const float *qkv, void *out, int out_bf16, int lowp, int n_images, int T, int E, int H)
{
    const int n_qblocks = (T + 16 * NW - 1) / (16 * NW);
    const dim3 grid((unsigned)(n_images * H * n_qblocks)), block(64 * NW);
    const float c = 1.4426950408889634f / sqrtf((float)D);
    if (lowp)
        hipLaunchKernelGGL((attention_tiled_kernel<D, NJ, false, true, NW>), grid, block, 0, st, qkv, out, T, E, H, n_qblocks, c);
    else if (out_bf16)
        hipLaunchKernelGGL((attention_tiled_kernel<D, NJ, true, false, NW>), grid, block, 0, st, qkv, out, T, E, H, n_qblocks, c);
    else
        hipLaunchKernelGGL((attention_tiled_kernel<D, NJ, false, false, NW>), grid, block, 0, st, qkv, out, T, E, H, n_qblocks, c);
    VH_LAUNCH_CHECK("attention_tiled_kernel");
    return 0;
}

/* Four waves (64 queries) per workgroup: measured on ViT-H/14 (T = 257: 5 query blocks, the last with one busy
 * wave) against 2, 3 and 6 waves -- 0.71 ms against 0.90, 0.92 and 0.95: fewer passes over K and V do not pay for
 * fewer workgroups in flight. */
template <int D, int NJ>
int launch_nj(hipStream_t st, const float *qkv, void *out, int out_bf16, int lowp, int n_images, int T, int E, int H)
{
    return launch_nw<D, NJ, 4>(st, qkv, out, out_bf16, lowp, n_images, T, E, H);
}

template <int D>
int launch_d(hipStream_t st, const float *qkv, void *out, int out_bf16, int lowp, int n_images, int T, int E, int H)
{
    if (T <= 128)
        return launch_nj<D, 8>(st, qkv, out, out_bf16, lowp, n_images, T, E, H);
    if (T <= 272)
        return launch_nj<D, 17>(st, qkv, out, out_bf16, lowp, n_images, T, E, H);
    return launch_nj<D, 32>(st, qkv, out, out_bf16, lowp, n_images, T, E, H);
}

} // namespace

/* Called by vh_launch_attention / vh_launch_attention_bf16 (attention_f32.hip) for the shapes
 * outside the resident-K/V kernel; arguments are already checked for null / positivity. */
int vh_attention_tiled(vh_stream_t s, const float *qkv, void *output, int out_bf16, int lowp, int n_images, int tokens,
                       int embed_dim, int num_heads)
{
    if (lowp && out_bf16)
        return vh_fail(1, "vh_launch_attention: the fp16-operand streaming kernel writes fp32 rows only");
    const int D = embed_dim / num_heads;
    if (D * num_heads != embed_dim || tokens > 512 || ((size_t)n_images * num_heads * ((tokens + QB_MIN - 1) / QB_MIN)) >> 31)
        return vh_fail(1, "vh_launch_attention: embed=%d heads=%d tokens=%d outside the supported range "
                          "(head_dim 64 | 80 | 128, tokens <= 512)", embed_dim, num_heads, tokens);
    if ((((uintptr_t)qkv | (uintptr_t)output) & 15) != 0 || embed_dim % 4 != 0)
        return vh_fail(1, "vh_launch_attention: qkv / output must be 16-byte aligned");
    hipStream_t st = (hipStream_t)s;
    switch (D) {
    case 64: return launch_d<64>(st, qkv, output, out_bf16, lowp, n_images, tokens, embed_dim, num_heads);
    case 80: return launch_d<80>(st, qkv, output, out_bf16, lowp, n_images, tokens, embed_dim, num_heads);
    case 128: return launch_d<128>(st, qkv, output, out_bf16, lowp, n_images, tokens, embed_dim, num_heads);
    default:
        return vh_fail(1, "vh_launch_attention: head_dim %d is not built (64, 80, 128)", D);
    }
}
