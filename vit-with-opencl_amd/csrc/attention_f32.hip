/*
 * attention_f32.hip -- softmax(Q K^T / sqrt(D)) V per (image, head), fp32.
 *
 * Replaces QKV_TO_SCOREV (multihead.cl:65-137; host ViT_opencl.c:539-565): the
 * reference runs one 256-thread work-group per (token, head), re-reads K 197x
 * from global memory and reduces through LDS trees.  CPU statement:
 * multihead_attn_seq, ViT_seq.c:192-262.
 *
 * Design (MI355X / CDNA4), one workgroup per (image, head), one wave per
 * 32-query tile (7 waves for T = 197):
 *  - K and V head slices ([T][64] each) are staged once into LDS (K rows padded
 *    to 68 floats so the ds_read_b128 fragment reads are conflict-free; V rows
 *    are read 32 consecutive floats per half-wave, conflict-free unpadded).
 *  - S^T = K Q^T on v_mfma_f32_32x32x2_f32: with the key index on the MFMA row
 *    and the query on the lane, each lane ends up holding, for ITS query
 *    (lane & 31), all keys of its half (lane >> 5) in registers: the row
 *    softmax is register-local plus one lane-half exchange.
 *  - The normalised P never leaves registers: an S^T accumulator register is
 *    exactly the B operand (k = key pair {klo, klo+4}, column = query) of the
 *    next product O^T = V^T P^T, whose A operand V[key][d] is read from LDS
 *    with the lane on d.
 *  - Numerics follow the scalar loop: scores scaled after the dot product,
 *    max-subtracted exp, normalised before the P.V product.
 *
 * Input rows are the fused projection output [Q(E) | K(E) | V(E)]; output is
 * [n_images*T][E] with heads concatenated (ViT_seq.c:252-258).
 */
#include "kernelHandler.h"
#include "vit_kernels.h"

namespace {

constexpr int HD = 64;        /* head dim this kernel is specialised for */
constexpr int KLD = HD + 4;   /* padded K row (floats) */

template <int NKT> /* number of 32-wide key/query tiles: T <= 32*NKT */
__global__ __launch_bounds__(64 * NKT) void attention_f32_kernel(const float *__restrict__ qkv,
                                                                float *__restrict__ out, int T,
                                                                int E, int H)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ks = smem;                  /* [NKT*32][KLD] */
    float *Vs = smem + NKT * 32 * KLD; /* [NKT*32][HD]  */

    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const size_t ld = (size_t)3 * E;
    const float *base = qkv + (size_t)b * T * ld + (size_t)h * HD;

    /* Stage K and V (zero rows beyond T so the padded products stay finite). */
    for (int idx = tid; idx < NKT * 32 * (HD / 4); idx += 64 * NKT) {
        const int row = idx >> 4, c4 = (idx & 15) * 4;
        f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
        if (row < T) {
            kv = *reinterpret_cast<const f32x4 *>(base + row * ld + E + c4);
            vv = *reinterpret_cast<const f32x4 *>(base + row * ld + 2 * E + c4);
        }
        *reinterpret_cast<f32x4 *>(Ks + row * KLD + c4) = kv;
        *reinterpret_cast<f32x4 *>(Vs + row * HD + c4) = vv;
    }

    /* This lane's query row; d = 8c + 4*lh + e in element e of chunk c. */
    const int q = wave * 32 + lr;
    const int qc = min(q, T - 1);
    f32x4 qf[HD / 8];
#pragma unroll
    for (int c = 0; c < HD / 8; ++c)
        qf[c] = *reinterpret_cast<const f32x4 *>(base + qc * ld + 8 * c + 4 * lh);

    __syncthreads();

    /* Waves w and w+4 share a SIMD.  Running in lockstep they would both want the
     * matrix pipe (QK^T, PV) and then both the VALU (softmax).  Holding the second
     * wave of each SIMD back by about one MFMA phase makes the phases complementary:
     * one wave's softmax runs under the other's MFMAs.  While it sleeps its partner
     * has the pipe to itself, so nothing is lost.  Speed only; results are unchanged. */
    if (NKT > 4 && __builtin_amdgcn_readfirstlane(wave) >= 4) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while ((long long)(__builtin_amdgcn_s_memtime() - t0) < (long long)(NKT * 32 * 64))
            __builtin_amdgcn_s_sleep(32);
    }

    /* S^T tiles: rows = keys of tile j, column = this lane's query. */
    f32x16 s[NKT];
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            s[j][r] = 0.0f;
#pragma unroll
        for (int c = 0; c < HD / 8; ++c) {
            const f32x4 kf = *reinterpret_cast<const f32x4 *>(Ks + (32 * j + lr) * KLD + 8 * c + 4 * lh);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                s[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[c][e], s[j], 0, 0, 0);
        }
    }

    /* Row softmax over keys (ViT_seq.c:211, :216-234): scale after the dot product
     * (division by sqrt(64) = 8 is exactly a multiplication by 0.125), subtract the row
     * maximum, exponentiate, normalise.  exp() is exp2 on a two-term product
     * x*log2(e) = t + r (t rounded, r the exact remainder plus the low part of log2 e),
     * exp2(t) * (1 + r ln 2): ~1 ulp like libm expf at a third of the instructions;
     * the normalisation multiplies by one correctly rounded reciprocal per row instead
     * of dividing 197 times (<= 1 ulp per probability). */
    const float inv_scale = 1.0f / sqrtf((float)HD);
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NKT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 32 * j + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float v = key < T ? s[j][r] * inv_scale : -INFINITY;
            s[j][r] = v;
            mx = fmaxf(mx, v);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < NKT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 32 * j + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float x = fmaxf(s[j][r] - mx, -120.0f);                    /* padding is -inf */
            const float t = x * 1.44269502162933349609375f;                 /* fp32(log2 e) */
            const float rem = __builtin_fmaf(x, 1.44269502162933349609375f, -t) +
                              x * 1.925963033500011e-8f;                       /* log2 e - fp32(log2 e) */
            const float e2 = __builtin_amdgcn_exp2f(t);
            const float e = key < T ? __builtin_fmaf(e2, rem * 0.693147182464599609375f, e2) : 0.0f;
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

    /* O^T = V^T P^T: rows = d (two 32-wide tiles), column = this lane's query. */
    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        o[0][r] = 0.0f;
        o[1][r] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < NKT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 32 * j + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float v0 = Vs[key * HD + lr];
            const float v1 = Vs[key * HD + 32 + lr];
            o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[j][r], o[0], 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[j][r], o[1], 0, 0, 0);
        }

    if (q < T) {
        float *dst = out + ((size_t)b * T + q) * E + (size_t)h * HD + 4 * lh;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {o[dt][4 * g], o[dt][4 * g + 1], o[dt][4 * g + 2], o[dt][4 * g + 3]};
                *reinterpret_cast<f32x4 *>(dst + dt * 32 + 8 * g) = v;
            }
    }
}

template <int NKT>
int launch(hipStream_t st, const float *qkv, float *out, int n_images, int T, int E, int H)
{
    const size_t lds = sizeof(float) * NKT * 32 * (KLD + HD);
    static bool attr_set = false;
    if (!attr_set) {
        VH_TRY(hipFuncSetAttribute((const void *)attention_f32_kernel<NKT>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((attention_f32_kernel<NKT>), dim3(n_images * H), dim3(64 * NKT), lds, st, qkv,
                       out, T, E, H);
    VH_LAUNCH_CHECK("attention_f32_kernel");
    return 0;
}

} // namespace

extern "C" int vh_launch_attention(vh_stream_t s, const float *qkv, float *output, int n_images,
                                   int tokens, int embed_dim, int num_heads)
{
    if (!qkv || !output)
        return vh_fail(1, "vh_launch_attention: null pointer argument");
    if (n_images <= 0 || tokens <= 0 || num_heads <= 0 || embed_dim != num_heads * HD)
        return vh_fail(1, "vh_launch_attention: needs head_dim == %d (embed=%d heads=%d)", HD,
                       embed_dim, num_heads);
    if (tokens > 32 * 7)
        return vh_fail(1, "vh_launch_attention: tokens=%d exceeds the %d this kernel holds in registers",
                       tokens, 32 * 7);
    hipStream_t st = (hipStream_t)s;
    switch ((tokens + 31) / 32) {
    case 1: return launch<1>(st, qkv, output, n_images, tokens, embed_dim, num_heads);
    case 2: return launch<2>(st, qkv, output, n_images, tokens, embed_dim, num_heads);
    case 3: return launch<3>(st, qkv, output, n_images, tokens, embed_dim, num_heads);
    case 4: return launch<4>(st, qkv, output, n_images, tokens, embed_dim, num_heads);
    case 5: return launch<5>(st, qkv, output, n_images, tokens, embed_dim, num_heads);
    case 6: return launch<6>(st, qkv, output, n_images, tokens, embed_dim, num_heads);
    default: return launch<7>(st, qkv, output, n_images, tokens, embed_dim, num_heads);
    }
}
