/*
 * attention_f32.hip -- softmax(Q K^T / sqrt(D)) V per (image, head), fp32.
 *
 * Replaces QKV_TO_SCOREV (multihead.cl:65-137; host ViT_opencl.c:539-565): the
 * reference runs one 256-thread work-group per (token, head), re-reads K 197x
 * from global memory and reduces through LDS trees.  CPU statement:
 * multihead_attn_seq, ViT_seq.c:192-262.
 *
 * Design (MI355X / CDNA4): a persistent grid of one workgroup per CU walks the
 * (image, head) items; one wave per 32-query tile (7 waves for T = 197).
 *  - K and V head slices ([T][64] fp32 = 50 KB each) live in three rotating LDS
 *    buffers filled by LDS-DMA (global_load_lds_dwordx4, no staging registers, no
 *    ds_write): while item n computes, V_n and K_{n+1} stream in; V_{n+1} reuses
 *    K_n's buffer.  All HBM latency sits under the MFMAs of the current item.
 *    Hand-offs: one s_barrier per item (buffers free / next K visible) and one LDS
 *    arrival counter (every wave's share of V_n has landed) polled before P.V.
 *  - K rows are 256 B = the LDS bank row, so the 16 lanes of a ds_read_b128 group
 *    (16 different keys, same d chunk) would hit one slot; chunk c of row r is kept
 *    at c ^ (r & 15) instead (applied to the DMA source address and to the reads).
 *    V is read 32 consecutive floats per half-wave: conflict-free as is.
 *  - S^T = K Q^T on the matrix cores: with the key index on the MFMA row and
 *    the query on the lane, each lane ends up holding, for ITS query (lane & 31),
 *    all keys of its half (lane >> 5) in registers: the row softmax is register-local
 *    plus one lane-half exchange.
 *  - Both products are fp32 products; by default they run like the GEMMs' (gemm_mfma.hip):
 *    every operand fragment is split exactly into three bf16 parts in registers and the
 *    six partial products of weight >= 2^-16 go through v_mfma_f32_32x32x16_bf16 (template
 *    NPL = 3) -- the same accuracy as v_mfma_f32_32x32x2_f32 (NPL = 0, VIT_HIP_ATTN_MFMA=fp32)
 *    at 2.67x its rate, which turns the kernel from MFMA-bound to VALU-bound.
 *  - The normalised P never leaves registers: an S^T accumulator register is exactly
 *    the B operand (k = key pair {klo, klo+4}, column = query) of the next product
 *    O^T = V^T P^T, whose A operand V[key][d] is read from LDS with the lane on d.
 *  - Waves w and w+4 share a SIMD; the second one is held back by about one MFMA
 *    phase so that one wave's softmax (VALU) runs under the other's MFMAs.
 *  - Numerics follow the scalar loop: scores scaled after the dot product,
 *    max-subtracted exponential, normalised before the P.V product (see the softmax
 *    comment for the rounding differences).
 *
 * Input rows are the fused projection output [Q(E) | K(E) | V(E)]; output is
 * [n_images*T][E] with heads concatenated (ViT_seq.c:252-258).
 */
#include "kernelHandler.h"
#include "vit_kernels.h"
#include "fp32_split.h"

#include <cstdlib>

namespace {

constexpr int HD = 64;                    /* head dim this kernel is specialised for */
constexpr int MAX_LDS = 160 * 1024;
#ifndef LATE_SCALE
#define LATE_SCALE 100   /* per cent of the default hold-back of a SIMD's second wave (tuning) */
#endif
constexpr int MAX_ROWS = (MAX_LDS - 64) / (3 * HD * 4) / 8 * 8; /* rows per buffer with 3 buffers: 208 */

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

/* NPL: how the two fp32 products Q.K^T and P.V reach the matrix cores.  0: v_mfma_f32_32x32x2_f32
 * (fp32 operands, 1/16 of the 16-bit rate).  3: every operand fragment split exactly into three
 * bf16 parts in registers, six v_mfma_f32_32x32x16_bf16 per block (the GEMMs' SPLIT3, same
 * accuracy, 2.67x the fp32 MFMA rate -- the kernel turns from MFMA-bound to VALU-bound).
 * 2: two fp16 parts, three products (the opt-in emulation mode, fp32_split.h).  1: operands rounded
 * to fp16, one product -- only for the reduced-precision GEMM modes (bf16 / fp8 operands), whose
 * tolerances it sits far inside (11-bit operands against their 8- and 4-bit ones). */
template <int NKT, int OUTK, int NPL> /* NKT 32-wide key/query tiles: 32*(NKT-1) < T <= 32*NKT; OUTK: 0 fp32, 1 bf16,
                                         * 3 the three-part bf16 split as planes [E/32][3][n_images*T][32] (gemm_p3.hip) */
__global__ __launch_bounds__(64 * NKT) void attention_f32_kernel(const float *__restrict__ qkv,
                                                                void *__restrict__ out, int T,
                                                                int E, int H, int n_items, int RB)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int buf_f = RB * HD;                                          /* floats per buffer */
    unsigned *v_ready = reinterpret_cast<unsigned *>(smem + 3 * buf_f); /* arrivals of V shares */

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const size_t ld = (size_t)3 * E;
    const int pieces = RB / 4;                                          /* 1-KiB DMA pieces per buffer */

    /* One LDS-DMA piece = 4 rows x 256 B; lane l fills 16-byte chunk (l & 15) of row
     * 4p + (l >> 4).  which: 1 = K (swizzled), 2 = V (linear). */
    auto dma_piece = [&](const float *base, int which, float *dst, int p) {
        const int r = 4 * p + (lane >> 4);
        int c = lane & 15;
        if (which == 1)
            c ^= r & 15;
        const float *src = base + (size_t)min(r, T - 1) * ld + 4 * c; /* rows >= T: finite duplicates */
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + p * 4 * HD), 16, 0, 0);
    };
    auto head_base = [&](int item, int which) {
        const int b = item / H, h = item - b * H;
        return qkv + (size_t)b * T * ld + (size_t)h * HD + (size_t)which * E;
    };
    auto dma = [&](int item, int which, float *dst) {
        const float *base = head_base(item, which);
        for (int p = wave; p < pieces; p += NKT)
            dma_piece(base, which, dst, p);
    };

    /* Query fragments: lane's row q = 32*wave + (lane & 31); element e of chunk c is
     * d = 8c + 4*(lane >> 5) + e.  They are loaded one item ahead (after P.V, when the
     * P registers are dead) and waited for at the end-of-item barrier, so that no
     * ordinary load is outstanding while the DMAs of an item are in flight -- the
     * compiler would otherwise wait for everything (vmcnt(0)) at the fragments' first use. */
    const int q = wave * 32 + lr;
    const int qc = min(q, T - 1);
    f32x4 qf[HD / 8];
    auto load_q = [&](int it) {
        const int b = it / H, h = it - b * H;
        if (NPL == 0) {
            const float *row = qkv + ((size_t)b * T + qc) * ld + (size_t)h * HD + 4 * lh;
#pragma unroll
            for (int c = 0; c < HD / 8; ++c)
                qf[c] = *reinterpret_cast<const f32x4 *>(row + 8 * c);
        } else { /* 16-deep MFMA groups: lane half lh holds d = 16g + 8lh .. +7 in qf[2g], qf[2g+1] */
            const float *row = qkv + ((size_t)b * T + qc) * ld + (size_t)h * HD + 8 * lh;
#pragma unroll
            for (int g = 0; g < HD / 16; ++g) {
                qf[2 * g] = *reinterpret_cast<const f32x4 *>(row + 16 * g);
                qf[2 * g + 1] = *reinterpret_cast<const f32x4 *>(row + 16 * g + 4);
            }
        }
    };

    /* Per-lane LDS byte offsets (item-independent).  K fragment of key row r, d chunk
     * (2c + lh): row r at r*256 B, chunk swizzled with r & 15.  V: lane reads
     * V[key][lr] and V[key][32 + lr] with key = k0 + 4*lh + rr. */
    int kofs[HD / 8], kofs_last[HD / 8];
    {
        const int rl = min(32 * (NKT - 1) + lr, RB - 1);
#pragma unroll
        for (int c = 0; c < HD / 8; ++c) {
            /* NPL == 0: chunk 2c + lh (d = 8c + 4lh ..); split forms: chunk 4g + 2lh + hh for c = 2g + hh */
            const int chunk = NPL == 0 ? 2 * c + lh : 4 * (c >> 1) + 2 * lh + (c & 1);
            kofs[c] = lr * (HD * 4) + 16 * (chunk ^ (lr & 15));
            kofs_last[c] = rl * (HD * 4) + 16 * (chunk ^ (rl & 15));
        }
    }
    const int vofs = (4 * lh * HD + lr) * 4;

    int item = blockIdx.x;
    if (item >= n_items)
        return;
    if (tid == 0)
        *v_ready = 0;
    load_q(item);
    dma(item, 1, smem); /* K of the first item into buffer 0 */
    __syncthreads();    /* vmcnt(0) + barrier */

    unsigned n = 0;     /* local item counter: K in buffer (2n)%3, V in (2n+1)%3, free (2n+2)%3 */
    for (; item < n_items; item += gridDim.x, ++n) {
        /* Buffer roles of this item.  The byte offsets are made opaque so that the
         * per-lane LDS addresses below are formed per item (one add each) instead of
         * being hoisted out of the loop for all three buffers -- that hoisting costs
         * well over a hundred VGPRs and spills. */
        unsigned kb = ((2 * n) % 3) * buf_f * 4, vb = ((2 * n + 1) % 3) * buf_f * 4;
        asm volatile("" : "+s"(kb), "+s"(vb));
        const char *Kb = reinterpret_cast<const char *>(smem) + kb;
        const char *Vb = reinterpret_cast<const char *>(smem) + vb;
        float *Vs = smem + ((2 * n + 1) % 3) * buf_f;
        float *Kn = smem + ((2 * n + 2) % 3) * buf_f;
        const int b = item / H, h = item - b * H;
        const int next = item + gridDim.x;

        /* V_n and K_{n+1}: both target buffers were released by the barrier that ended
         * the previous item.  The second wave of each SIMD issues its share at once (it
         * is about to be held back anyway); the first waves interleave theirs with the
         * QK^T tiles so that issuing ~15 DMA instructions does not delay their MFMAs. */
        /* hold-back of a SIMD's second wave ~ one Q.K^T phase of the first: 64 cycles per key with the
         * fp32 MFMA, about a third of that on the split forms */
        constexpr int LATE_CYCLES_PER_KEY = (NPL == 0 ? 64 : NPL == 3 ? 24 : NPL == 2 ? 14 : 8) * LATE_SCALE / 100;
        const bool late = NKT > 4 && wave >= 4;
        const float *v_src = head_base(item, 2);
        const float *k_src = head_base(next < n_items ? next : item, 1);
        if (late) {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            dma(item, 2, Vs);
            if (next < n_items)
                dma(next, 1, Kn);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0)
                __hip_atomic_fetch_add(v_ready, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            while ((long long)(__builtin_amdgcn_s_memtime() - t0) < (long long)(NKT * 32 * LATE_CYCLES_PER_KEY))
                __builtin_amdgcn_s_sleep(32);
        }

        /* split forms: this item's query fragments into parts, once (qf is free for the next item's loads) */
        typename PartT<NPL>::type qp[NPL == 0 ? 1 : HD / 16][NPL == 0 ? 1 : NPL];
        if constexpr (NPL != 0) {
#pragma unroll
            for (int g = 0; g < HD / 16; ++g)
                split_parts(qf[2 * g], qf[2 * g + 1], qp[g]);
        }

        /* S^T tiles: rows = keys of tile j, column = this lane's query. */
        f32x16 s[NKT];
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                s[j][r] = 0.0f;
            /* rows 32j + lr: lane offset + compile-time j*8 KiB; only the last tile can
             * run past the buffer and uses the clamped-row offsets */
            auto kfrag = [&](int c) {
                return (j < NKT - 1) ? *reinterpret_cast<const f32x4 *>(Kb + kofs[c] + j * 32 * HD * 4)
                                     : *reinterpret_cast<const f32x4 *>(Kb + kofs_last[c]);
            };
            if constexpr (NPL == 0) {
#pragma unroll
                for (int c = 0; c < HD / 8; ++c) {
                    const f32x4 kf = kfrag(c);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        s[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[c][e], s[j], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int g = 0; g < HD / 16; ++g) {
                    typename PartT<NPL>::type kp[NPL];
                    split_parts(kfrag(2 * g), kfrag(2 * g + 1), kp);
#pragma unroll
                    for (int t = 0; t < n_terms<NPL>(); ++t)
                        s[j] = mfma_part(kp[term_w<NPL>(t)], qp[g][term_a<NPL>(t)], s[j]);
                }
            }
            if (!late) {
                for (int p = wave + NKT * j; p < pieces; p += NKT * NKT) {
                    dma_piece(v_src, 2, Vs, p);
                    if (next < n_items)
                        dma_piece(k_src, 1, Kn, p);
                }
            }
        }

        if (!late) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0)
                __hip_atomic_fetch_add(v_ready, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }

        /* Row softmax over keys (ViT_seq.c:211, :216-234).  The scalar loop computes
         * expf(score/sqrt(D) - max) / sum; here the scale (an exact power of two for
         * D = 64), the max subtraction and the change of base are one fma per element,
         *     e = 2^(s * c1 + c2),  c1 = log2(e)/sqrt(D),  c2 = -max * c1,
         * followed by one multiplication with a correctly rounded reciprocal of the row
         * sum.  The rounding of c2 is a common factor of the whole row and cancels in the
         * normalisation; per element the result is within ~|x| * 6e-8 relative of the
         * scalar value, far below the 2e-5 operator tolerance.  Keys >= T exist only in
         * the last tile; they are set to -inf, which the fma carries to 2^-inf = 0. */
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

        /* All NKT shares of V_n landed?  (Arrivals were posted at least one MFMA phase
         * ago; this poll normally passes at once.) */
        const unsigned want = (unsigned)NKT * (n + 1);
        while (__hip_atomic_load(v_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want)
            __builtin_amdgcn_s_sleep(1);

        /* O^T = V^T P^T: rows = d (two 32-wide tiles), column = this lane's query.
         * Register groups whose keys are all >= T carry P = 0 and are skipped. */
        f32x16 o[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o[0][r] = 0.0f;
            o[1][r] = 0.0f;
        }
        if constexpr (NPL == 0) {
    #pragma unroll
            for (int j = 0; j < NKT; ++j)
    #pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (j == NKT - 1 && 32 * j + 8 * g >= T)
                        continue;
    #pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int r = 4 * g + rr;
                        /* keys k0 + rr + 4*lh with k0 = 32j + 8g < T: all 8 rows exist (RB = T up to 8) */
                        const char *vp = Vb + vofs + (32 * j + 8 * g + rr) * (HD * 4);
                        const float v0 = *reinterpret_cast<const float *>(vp);
                        const float v1 = *reinterpret_cast<const float *>(vp + 32 * 4);
                        o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[j][r], o[0], 0, 0, 0);
                        o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[j][r], o[1], 0, 0, 0);
                    }
                }
        } else {
            /* 16 keys per MFMA: contraction slot (lh, e) is key 32j + 16t + (e & 3) + 8*(e >> 2) + 4*lh
             * -- the key whose probability accumulator register 8t + e of tile j already holds. */
#pragma unroll
            for (int j = 0; j < NKT; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (j == NKT - 1 && 32 * j + 16 * t >= T)
                        continue;
                    typename PartT<NPL>::type pp[NPL];
                    split_parts(f32x4{s[j][8 * t], s[j][8 * t + 1], s[j][8 * t + 2], s[j][8 * t + 3]},
                                f32x4{s[j][8 * t + 4], s[j][8 * t + 5], s[j][8 * t + 6], s[j][8 * t + 7]}, pp);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        /* rows 32j + 16t + 4lh + {0..3, 8..11} exist: RB is T rounded up to 16 */
                        const char *vp = Vb + vofs + (32 * j + 16 * t) * (HD * 4) + dt * 32 * 4;
                        f32x4 vlo, vhi;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            vlo[e] = *reinterpret_cast<const float *>(vp + e * (HD * 4));
                            vhi[e] = *reinterpret_cast<const float *>(vp + (8 + e) * (HD * 4));
                        }
                        typename PartT<NPL>::type vq[NPL];
                        split_parts(vlo, vhi, vq);
#pragma unroll
                        for (int tt = 0; tt < n_terms<NPL>(); ++tt)
                            o[dt] = mfma_part(vq[term_w<NPL>(tt)], pp[term_a<NPL>(tt)], o[dt]);
                    }
                }
        }

        if (next < n_items) {
            asm volatile("" ::: "memory"); /* not above the P.V reads: P's registers must be dead */
            load_q(next);
        }

        if (OUTK == 3 || OUTK == 4) {
            /* Consumed only by the planes output projection: planes [E/32][NP][rows][32] (NP = 3: the exact split;
             * OUTK == 4: NP = 1, values rounded to bf16 for the bf16-operand mode), K step 2h + dt.
             * A lane holds d = 8g + 4lh .. +3 (8 bytes per part); one half-wave exchange per dword
             * (v_permlane32_swap) gives the lower half the 16 bytes of group g and the upper half those of
             * group g+1.  Lanes l and l + 32 share a query, so the guard keeps pairs together. */
            if (q < T) {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const size_t prow = (size_t)(n_items / H) * T;
                constexpr int NP = OUTK == 3 ? 3 : 1;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    u32x2 pg[4][3];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        bf16x4 part[3];
                        const f32x4 v = {o[dt][4 * g], o[dt][4 * g + 1], o[dt][4 * g + 2], o[dt][4 * g + 3]};
                        if (NP == 3)
                            split4(v, part[0], part[1], part[2]);
                        else
                            part[0] = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl)
                            pg[g][pl] = __builtin_bit_cast(u32x2, part[pl]);
                    }
                    char *d3 = static_cast<char *>(out) + ((size_t)(2 * h + dt) * NP * prow + (size_t)b * T + q) * 64 + 16 * lh;
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                        for (int g = 0; g < 4; g += 2) {
                            const auto r0 = __builtin_amdgcn_permlane32_swap(pg[g][pl][0], pg[g + 1][pl][0], false, false);
                            const auto r1 = __builtin_amdgcn_permlane32_swap(pg[g][pl][1], pg[g + 1][pl][1], false, false);
                            *reinterpret_cast<u32x4 *>(d3 + (size_t)pl * prow * 64 + 16 * g) = u32x4{r0[0], r1[0], r0[1], r1[1]};
                        }
                }
            }
        } else if (q < T) {
            const size_t off = ((size_t)b * T + q) * E + (size_t)h * HD + 4 * lh;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {o[dt][4 * g], o[dt][4 * g + 1], o[dt][4 * g + 2], o[dt][4 * g + 3]};
                    *reinterpret_cast<f32x4 *>(static_cast<float *>(out) + off + dt * 32 + 8 * g) = v;
                }
        }

        /* End of item: every wave is done with K_n and V_n (their buffers are free) and
         * has waited for its share of K_{n+1}, which the barrier publishes. */
        __syncthreads();
    }
}

template <int NKT, int OUTK, int NPL>
int launch_k(hipStream_t st, const float *qkv, void *out, int n_images, int T, int E, int H)
{
    /* rows per buffer: whole register groups of keys (8 for the fp32 MFMA, 16 for the split forms), whole 4-row DMA pieces */
    const int RB = NPL == 0 ? (T + 7) / 8 * 8 : (T + 15) / 16 * 16;
    const size_t lds = sizeof(float) * 3 * RB * HD + 64;
    VH_SET_LDS_ONCE((attention_f32_kernel<NKT, OUTK, NPL>), MAX_LDS);
    /* Persistent grid: one workgroup per CU (the three K/V buffers fill a CU's LDS),
     * each walking (image, head) items blockIdx.x, blockIdx.x + grid, ... */
    const int num_cus = vh_device_cus(vh_current_device());
    const int n_items = n_images * H;
    const int grid = n_items < num_cus ? n_items : num_cus;
    hipLaunchKernelGGL((attention_f32_kernel<NKT, OUTK, NPL>), dim3(grid), dim3(64 * NKT), lds, st, qkv, out,
                       T, E, H, n_items, RB);
    VH_LAUNCH_CHECK("attention_f32_kernel");
    return 0;
}

/* arith: 0 = fp32 MFMA, 3 = exact three-part bf16 split (default), 2 = two fp16 parts (emulation mode),
 * 1 = operands rounded to fp16 (the bf16 / fp8 GEMM modes).  out_kind: 0 = fp32 rows, 3 = three-part planes
 * (exact split), 4 = one-part (bf16) planes. */
template <int NKT>
int launch(hipStream_t st, const float *qkv, void *out, int out_kind, int arith, int n_images, int T, int E, int H)
{
    if (out_kind == 3)
        return launch_k<NKT, 3, 3>(st, qkv, out, n_images, T, E, H);
    if (out_kind == 4)
        return launch_k<NKT, 4, 1>(st, qkv, out, n_images, T, E, H);
    switch (arith) {
    case 0: return launch_k<NKT, 0, 0>(st, qkv, out, n_images, T, E, H);
    case 1: return launch_k<NKT, 0, 1>(st, qkv, out, n_images, T, E, H);
    case 2: return launch_k<NKT, 0, 2>(st, qkv, out, n_images, T, E, H);
    default: return launch_k<NKT, 0, 3>(st, qkv, out, n_images, T, E, H);
    }
}

} // namespace

static int launch_attention(vh_stream_t s, const float *qkv, void *output, int out_bf16, int arith, int n_images,
                            int tokens, int embed_dim, int num_heads)
{
    if (!qkv || !output)
        return vh_fail(1, "vh_launch_attention: null pointer argument");
    if (n_images <= 0 || tokens <= 0 || num_heads <= 0 || embed_dim <= 0)
        return vh_fail(1, "vh_launch_attention: non-positive dimension (n=%d tokens=%d embed=%d heads=%d)",
                       n_images, tokens, embed_dim, num_heads);
    /* Resident K/V (this file) where one head's K and V fit the LDS three times over;
     * otherwise the streaming kernel (VIT_HIP_ATTN=tiled forces it, for tests). */
    static int force_tiled = -1;
    if (force_tiled < 0) {
        const char *env = getenv("VIT_HIP_ATTN");
        force_tiled = (env && env[0] == 't') ? 1 : 0;
    }
    if (embed_dim != num_heads * HD || tokens > MAX_ROWS || (force_tiled && out_bf16 < 3))   /* arith 1: fp16 operands there too */
        return vh_attention_tiled(s, qkv, output, out_bf16, arith == 1 && out_bf16 == 0, n_images, tokens, embed_dim, num_heads);
    static int native = -1;
    if (native < 0) {
        const char *env = getenv("VIT_HIP_GEMM_FP32");   /* "native": the fp32 matrix instruction here too */
        native = (env && env[0] == 'n') ? 1 : 0;
    }
    if (native && arith == 3 && out_bf16 == 0)
        arith = 0;
    hipStream_t st = (hipStream_t)s;
    switch ((tokens + 31) / 32) {
    case 1: return launch<1>(st, qkv, output, out_bf16, arith, n_images, tokens, embed_dim, num_heads);
    case 2: return launch<2>(st, qkv, output, out_bf16, arith, n_images, tokens, embed_dim, num_heads);
    case 3: return launch<3>(st, qkv, output, out_bf16, arith, n_images, tokens, embed_dim, num_heads);
    case 4: return launch<4>(st, qkv, output, out_bf16, arith, n_images, tokens, embed_dim, num_heads);
    case 5: return launch<5>(st, qkv, output, out_bf16, arith, n_images, tokens, embed_dim, num_heads);
    case 6: return launch<6>(st, qkv, output, out_bf16, arith, n_images, tokens, embed_dim, num_heads);
    default: return launch<7>(st, qkv, output, out_bf16, arith, n_images, tokens, embed_dim, num_heads);
    }
}

extern "C" int vh_launch_attention(vh_stream_t s, const float *qkv, float *output, int n_images,
                                   int tokens, int embed_dim, int num_heads)
{
    return launch_attention(s, qkv, output, 0, 3, n_images, tokens, embed_dim, num_heads);
}

/* The fp32 attention writing its output as the three-part split planes [E/32][3][n_images*tokens][32] that
 * vh_launch_linear_p3 reads (same values as vh_launch_attention, split exactly). */
extern "C" int vh_launch_attention_p3(vh_stream_t s, const float *qkv, void *out_planes, int n_images,
                                      int tokens, int embed_dim, int num_heads)
{
    if (embed_dim != num_heads * HD || tokens > MAX_ROWS)
        return vh_fail(1, "vh_launch_attention_p3: needs head_dim 64 and tokens <= %d", MAX_ROWS);
    return launch_attention(s, qkv, out_planes, 3, 3, n_images, tokens, embed_dim, num_heads);
}

/* The bf16-operand mode's attention writing one-part planes [E/32][1][n_images*tokens][32] (bf16) for
 * vh_launch_linear_planes(parts = 1): the arithmetic of vh_launch_attention_f16 (Q, K, V, P rounded to fp16 for the
 * two products, fp32 softmax), the result rounded to bf16. */
extern "C" int vh_launch_attention_planes_bf16(vh_stream_t s, const float *qkv, void *out_planes, int n_images,
                                               int tokens, int embed_dim, int num_heads)
{
    if (embed_dim != num_heads * HD || tokens > MAX_ROWS)
        return vh_fail(1, "vh_launch_attention_planes_bf16: needs head_dim 64 and tokens <= %d", MAX_ROWS);
    return launch_attention(s, qkv, out_planes, 4, 1, n_images, tokens, embed_dim, num_heads);
}

/* The emulation mode's attention: Q.K^T and P.V on two fp16 parts / three products (kernelHandler.h,
 * vh_launch_linear_h2); shapes outside the resident-K/V kernel use the streaming fp32 kernel. */
extern "C" int vh_launch_attention_h2(vh_stream_t s, const float *qkv, float *output, int n_images,
                                      int tokens, int embed_dim, int num_heads)
{
    return launch_attention(s, qkv, output, 0, 2, n_images, tokens, embed_dim, num_heads);
}

/* fp32 in, fp32 out, Q / K / V / P rounded to fp16 for the two products: the fp8-operand GEMM mode. */
extern "C" int vh_launch_attention_f16(vh_stream_t s, const float *qkv, float *output, int n_images,
                                       int tokens, int embed_dim, int num_heads)
{
    return launch_attention(s, qkv, output, 0, 1, n_images, tokens, embed_dim, num_heads);
}
