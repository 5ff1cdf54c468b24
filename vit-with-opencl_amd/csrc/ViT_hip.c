/*
 * ViT_hip.c -- host side of the ViT forward pass, in C, over the C-ABI device
 * shim (include/kernelHandler.h).  This file is the counterpart of the
 * reference's ViT_opencl.c host orchestration (Encoder :710-748, ViT_opencl
 * :794-986) and of its device-memory management (:125-357), redesigned:
 *
 *  - the whole batch moves through each operator at once (M = n*tokens rows)
 *    instead of one image at a time through 113 event-chained launches;
 *  - weights are uploaded once per context into one HBM slab, activations live
 *    in one arena sized at creation (the reference creates and destroys 14
 *    cl_mem objects per image, ViT_opencl.c:929,962);
 *  - one in-order stream, so no event graph is needed;
 *  - the residual adds are folded into the projection epilogues and the final
 *    LayerNorm runs on the class-token rows only (the reference normalises all
 *    197 rows and uses one, ViT_opencl.c:951-955 / ViT_seq.c:506-511).
 *
 * Operator order per layer (ViT_seq.c:330-370):
 *   y = LN1(x); qkv = y Win^T + bin; a = attention(qkv); x = x + (a Wout^T + bout);
 *   y = LN2(x); h = gelu(y W1^T + b1); x = x + (h W2^T + b2)
 */
#include "ViT_opencl.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define TRY(expr)                 \
    do {                          \
        int try_rc_ = (expr);     \
        if (try_rc_ != 0) {       \
            rc = try_rc_;         \
            goto fail;            \
        }                         \
    } while (0)

struct vit_hip_ctx
{
    vit_config cfg;
    int device;
    int max_batch;
    int tokens;
    int n_tensors;
    vh_stream_t stream;

    float *w_slab;      /* all weights, one allocation */
    float **w;          /* device pointer per tensor index */
    int precision;      /* VIT_PRECISION_F32 or VIT_PRECISION_BF16_GEMM */
    void *w16_slab;     /* bf16 copies of the four big matrices of every layer */
    void **w16;         /* per tensor index (NULL where no bf16 copy exists) */
    /* F32: the same four matrices pre-split into three bf16 planes each (vh_launch_linear_w3) */
    void *w3_slab;
    void **w3;          /* per tensor index (NULL: use the fp32 tensor) */
    float *w3_scale;    /* F32_FP16X2: per tensor index, the power of two its fp16 parts were scaled by */
    int use_p3;         /* F32: GEMM inputs travel as three-part bf16 planes (y, attn, hid hold 6 bytes per value) */
    int cls_only_last;  /* use_p3: the last layer's output projection and MLP run on the class-token rows only */
    /* FP8_GEMM: block-scaled e4m3 copies of the same four matrices (values, then their e8m0 block scales) */
    void *w8_slab;
    void **w8, **w8s;   /* per tensor index: values, scales */
    /* conv_proj as planes [Kp/32][parts][E][32]: one part (bf16) for BF16_GEMM / FP8_GEMM (vh_launch_patch_embed_planes),
     * the exact three-part split for the fp32 path on planes (vh_launch_patch_embed_planes3) */
    void *wconv16;
    /* BF16_GEMM / FP8_GEMM with the LayerNorms folded into the projections behind them (csrc/norm_fold.h): the QKV and fc1
     * operand copies hold gamma-scaled weights; per such matrix, colsum [N] of the rounded values and the folded bias [N] */
    int ln_fold;
    float *fold_slab;
    float **fold_cs, **fold_b;   /* per tensor index (in_proj and fc1 weights only) */
    size_t w_slab_bytes, planes_bytes, wconv16_bytes, fold_bytes;   /* sizes of w_slab, of the mode's repacked slab, of wconv16, of fold_slab */

    /* activation arena (rows = max_batch * tokens) */
    float *x;           /* residual stream      [rows][E]   */
    float *y;           /* LayerNorm output     [rows][E]   */
    float *attn;        /* attention output     [rows][E]   */
    float *qkv;         /* fused Q|K|V          [rows][3E]  */
    float *hid;         /* MLP hidden           [rows][F]   */
    size_t ws_bytes;    /* size of hid, which doubles as the patch-gather workspace */
    float *stats;       /* ln_fold: partial (sum, sum of squares) of the residual rows, [E/128][rows][2] */
    float *cls;         /* normalised CLS rows  [max_batch][E] */
    float *d_logits;    /* [max_batch][classes] */
    float *d_probs;     /* [max_batch][classes] */
    /* host-pointer API: two staging slots so that gathering + uploading chunk k+1
     * overlaps the compute of chunk k (pinned host memory, second stream, events) */
    float *d_images[2]; /* [max_batch][C][H][W] */
    float *h_images[2];
    float *h_logits[2];
    float *h_probs[2];
    vh_stream_t copy_stream;
    vh_event_t up_done[2], comp_done[2], out_done[2];

    /* optional per-operator timing with HIP events on the launch stream */
    vh_event_t *prof_ev;   /* 2 events per recorded launch */
    int *prof_class;       /* operator class per recorded launch */
    int prof_cap;          /* launches the pool can hold */
    int prof_used;         /* launches recorded since enable */
    unsigned prof_mask;    /* operator classes to record (bit = vit_op_class); 0 = all */
};

static int prof_begin(vit_hip_ctx *ctx, vh_stream_t s, int op_class)
{
    if (!ctx->prof_ev || ctx->prof_used >= ctx->prof_cap)
        return -1;
    if (ctx->prof_mask && !(ctx->prof_mask & (1u << op_class)))
        return -1;
    const int slot = ctx->prof_used++;
    ctx->prof_class[slot] = op_class;
    vh_event_record(ctx->prof_ev[2 * slot], s);
    return slot;
}

static void prof_end(vit_hip_ctx *ctx, vh_stream_t s, int slot)
{
    if (slot >= 0)
        vh_event_record(ctx->prof_ev[2 * slot + 1], s);
}

/* Launch `call` bracketed by two events when profiling is enabled. */
#define OP(op_class, call)                                  \
    do {                                                    \
        const int op_slot_ = prof_begin(ctx, s, op_class);  \
        TRY(call);                                          \
        prof_end(ctx, s, op_slot_);                         \
    } while (0)

static void prof_release(vit_hip_ctx *ctx)
{
    if (ctx->prof_ev) {
        for (int i = 0; i < 2 * ctx->prof_cap; ++i)
            if (ctx->prof_ev[i])
                vh_event_destroy(ctx->prof_ev[i]);
        free(ctx->prof_ev);
        free(ctx->prof_class);
    }
    ctx->prof_ev = NULL;
    ctx->prof_class = NULL;
    ctx->prof_cap = ctx->prof_used = 0;
}

static double wall_seconds(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

const vit_config *vit_hip_config(const vit_hip_ctx *ctx) { return &ctx->cfg; }
int vit_hip_device(const vit_hip_ctx *ctx) { return ctx->device; }
float *vit_hip_logits_buffer(vit_hip_ctx *ctx) { return ctx->d_logits; }   /* [max_batch][classes], vit_gather_rccl.c */
vh_stream_t vit_hip_stream(const vit_hip_ctx *ctx) { return ctx->stream; }
int vit_hip_max_batch(const vit_hip_ctx *ctx) { return ctx->max_batch; }
const float *vit_hip_weight(const vit_hip_ctx *ctx, int idx)
{
    return (idx >= 0 && idx < ctx->n_tensors) ? ctx->w[idx] : NULL;
}

void vit_hip_destroy(vit_hip_ctx *ctx)
{
    if (!ctx)
        return;
    if (ctx->stream || ctx->w_slab)   /* a context that never reached the device (refused header, bad arguments) leaves the
                                       * device -- and the caller's error text -- alone */
        vh_set_device(ctx->device);
    if (ctx->stream)
        vh_stream_sync(ctx->stream);
    prof_release(ctx);
    if (ctx->w16_slab)
        vh_free(ctx->w16_slab);
    free(ctx->w16);
    if (ctx->w3_slab)
        vh_free(ctx->w3_slab);
    free(ctx->w3);
    free(ctx->w3_scale);
    if (ctx->w8_slab)
        vh_free(ctx->w8_slab);
    if (ctx->wconv16)
        vh_free(ctx->wconv16);
    free(ctx->w8);
    free(ctx->w8s);
    free(ctx->fold_cs);
    free(ctx->fold_b);
    float *dev[] = {ctx->w_slab, ctx->x, ctx->y, ctx->attn, ctx->qkv, ctx->hid, ctx->fold_slab, ctx->stats,
                    ctx->cls, ctx->d_logits, ctx->d_probs, ctx->d_images[0], ctx->d_images[1]};
    for (size_t i = 0; i < sizeof(dev) / sizeof(dev[0]); ++i)
        if (dev[i])
            vh_free(dev[i]);
    float *host[] = {ctx->h_images[0], ctx->h_images[1], ctx->h_logits[0], ctx->h_logits[1],
                     ctx->h_probs[0], ctx->h_probs[1]};
    for (size_t i = 0; i < sizeof(host) / sizeof(host[0]); ++i)
        if (host[i])
            vh_host_free(host[i]);
    for (int i = 0; i < 2; ++i) {
        if (ctx->up_done[i]) vh_event_destroy(ctx->up_done[i]);
        if (ctx->comp_done[i]) vh_event_destroy(ctx->comp_done[i]);
        if (ctx->out_done[i]) vh_event_destroy(ctx->out_done[i]);
    }
    if (ctx->copy_stream)
        vh_stream_destroy(ctx->copy_stream);
    if (ctx->stream)
        vh_stream_destroy(ctx->stream);
    free(ctx->w);
    free(ctx);
}

static const int BIG[4] = {2, 4, 8, 10};   /* in_proj, out_proj, fc1, fc2 weights within a layer's 12 tensors */

/* bytes per weight of the mode's repacked copy of the four big matrices (0: none) */
static size_t repack_bytes_per_weight(const vit_hip_ctx *ctx)
{
    const vit_config *cfg = &ctx->cfg;
    switch (ctx->precision) {
    case VIT_PRECISION_BF16_GEMM: return 2;                                                   /* one-part planes */
    case VIT_PRECISION_F32: return (cfg->embed_dim % 128 == 0 && cfg->mlp_hidden % 128 == 0) ? 6 : 0;   /* three-part planes */
    case VIT_PRECISION_F32_FP16X2: return 4;                                                  /* two fp16 parts */
    default: return 0;                                                                        /* FP8: values + scales, below */
    }
}

/* The sizes of a context's weight slabs: w_slab, the mode's repacked slab, wconv16, fold_slab.  A function of (cfg,
 * precision, ln_fold) alone and free of side effects: vit_hip_create_from_planes checks a file's header against it
 * BEFORE anything is allocated. */
static void weight_slab_sizes(const vit_hip_ctx *ctx, size_t sizes[4])
{
    const vit_config *cfg = &ctx->cfg;
    size_t total = 0;
    for (int i = 0; i < ctx->n_tensors; ++i)
        total += align_up(vit_config_tensor_size(cfg, i) * sizeof(float), 256);
    const size_t per = repack_bytes_per_weight(ctx);
    size_t planes = 0, fold = 0;
    for (int l = 0; l < cfg->depth; ++l)
        for (int k = 0; k < 4; ++k) {
            const size_t cnt = vit_config_tensor_size(cfg, 4 + 12 * l + BIG[k]);
            planes += ctx->precision == VIT_PRECISION_FP8_GEMM ? align_up(cnt, 256) + align_up(cnt / 32, 256) : align_up(cnt * per, 256);
            if (ctx->ln_fold && (k == 0 || k == 2))   /* in_proj and fc1: colsum and folded bias, one float per output feature each */
                fold += 2 * align_up(vit_config_tensor_size(cfg, 4 + 12 * l + BIG[k] + 1) * sizeof(float), 256);
        }
    sizes[0] = total;
    sizes[1] = planes;
    const size_t conv_parts = (ctx->precision == VIT_PRECISION_BF16_GEMM || ctx->precision == VIT_PRECISION_FP8_GEMM) ? 1
                            : (ctx->precision == VIT_PRECISION_F32 && per == 6) ? 3 : 0;
    sizes[2] = cfg->embed_dim % 128 == 0 ? (size_t)vh_patch_planes_k(cfg->in_chans, cfg->patch_size) * (size_t)cfg->embed_dim * 2 * conv_parts : 0;
    sizes[3] = fold;
}

/* Allocate the weight slabs and fix every tensor's place in them.  Depends on (cfg, precision, ln_fold) only, so a planes
 * file written by one context (vit_hip_export_planes) drops into another's slabs byte for byte. */
static int layout_weights(vit_hip_ctx *ctx)
{
    int rc = 0;
    const vit_config *cfg = &ctx->cfg;
    const int n_tensors = ctx->n_tensors;
    size_t sizes[4];
    weight_slab_sizes(ctx, sizes);
    const size_t total = sizes[0], planes = sizes[1];
    ctx->w_slab_bytes = total;
    TRY(vh_malloc((void **)&ctx->w_slab, total));
    size_t off = 0;
    for (int i = 0; i < n_tensors; ++i) {
        ctx->w[i] = (float *)((char *)ctx->w_slab + off);
        off += align_up(vit_config_tensor_size(cfg, i) * sizeof(float), 256);
    }
    const size_t per = repack_bytes_per_weight(ctx);
    ctx->planes_bytes = planes;
    ctx->fold_bytes = sizes[3];
    if (ctx->fold_bytes) {
        TRY(vh_malloc((void **)&ctx->fold_slab, ctx->fold_bytes));
        size_t fo = 0;
        for (int l = 0; l < cfg->depth; ++l)
            for (int k = 0; k < 4; k += 2) {
                const int idx = 4 + 12 * l + BIG[k];
                const size_t nb = align_up(vit_config_tensor_size(cfg, idx + 1) * sizeof(float), 256);
                ctx->fold_cs[idx] = (float *)((char *)ctx->fold_slab + fo);
                ctx->fold_b[idx] = (float *)((char *)ctx->fold_slab + fo + nb);
                fo += 2 * nb;
            }
    }
    if (planes) {
        void *slab = NULL;
        TRY(vh_malloc(&slab, planes));
        if (ctx->precision == VIT_PRECISION_BF16_GEMM)
            ctx->w16_slab = slab;
        else if (ctx->precision == VIT_PRECISION_FP8_GEMM)
            ctx->w8_slab = slab;
        else
            ctx->w3_slab = slab;
        size_t o = 0;
        for (int l = 0; l < cfg->depth; ++l)
            for (int k = 0; k < 4; ++k) {
                const int idx = 4 + 12 * l + BIG[k];
                const size_t cnt = vit_config_tensor_size(cfg, idx);
                if (ctx->precision == VIT_PRECISION_BF16_GEMM) {
                    ctx->w16[idx] = (char *)slab + o;
                    o += align_up(cnt * 2, 256);
                } else if (ctx->precision == VIT_PRECISION_FP8_GEMM) {
                    /* values (1 byte per weight) then the e8m0 scales (1 byte per 32 weights), both 256-byte aligned */
                    ctx->w8[idx] = (char *)slab + o;
                    ctx->w8s[idx] = (char *)slab + o + align_up(cnt, 256);
                    o += align_up(cnt, 256) + align_up(cnt / 32, 256);
                } else {
                    ctx->w3[idx] = (char *)slab + o;
                    o += align_up(cnt * per, 256);
                }
            }
    }
    if (sizes[2]) {
        /* the reduced modes' patch embedding: conv weights rounded to bf16 planes, K padded to the one-part K step */
        ctx->wconv16_bytes = sizes[2];
        TRY(vh_malloc(&ctx->wconv16, ctx->wconv16_bytes));
    }
    return 0;
fail:
    return rc;
}

/* Upload the caller's tensors and repack the four big matrices of every layer into the mode's GEMM operand format. */
static int fill_weights(vit_hip_ctx *ctx, const Network *networks)
{
    int rc = 0;
    const vit_config *cfg = &ctx->cfg;
    for (int i = 0; i < ctx->n_tensors; ++i)
        TRY(vh_h2d(ctx->w[i], networks[i].data, networks[i].size * sizeof(float), ctx->stream));
    float *d_amax = NULL, *d_scaled = NULL;
    if (ctx->precision == VIT_PRECISION_F32_FP16X2)
        TRY(vh_malloc((void **)&d_amax, sizeof(float)));
    if (ctx->ln_fold)   /* gamma-scaled copy of one matrix at a time, before its rounding (the largest is fc1 / in_proj) */
        TRY(vh_malloc((void **)&d_scaled, (size_t)cfg->embed_dim * (size_t)(cfg->mlp_hidden > 3 * cfg->embed_dim ? cfg->mlp_hidden : 3 * cfg->embed_dim) * sizeof(float)));
    for (int l = 0; l < cfg->depth && rc == 0; ++l)
        for (int k = 0; k < 4 && rc == 0; ++k) {
            const int idx = 4 + 12 * l + BIG[k];
            const int out_f = (int)networks[idx + 1].size, in_f = (int)(networks[idx].size / networks[idx + 1].size);
            if (ctx->ln_fold && (k == 0 || k == 2)) {
                /* the LayerNorm in front of this projection moves into it (csrc/norm_fold.h): W' = gamma . W rounded to the
                 * mode's operand format, colsum of the ROUNDED values, bias' = bias + beta W^T from the fp32 weights.
                 * ln1 (tensors 0, 1 of the layer) feeds in_proj, ln2 (6, 7) feeds fc1. */
                const float *gamma = ctx->w[4 + 12 * l + (k == 0 ? 0 : 6)], *beta = ctx->w[4 + 12 * l + (k == 0 ? 1 : 7)];
                if ((rc = vh_launch_fold_gamma(ctx->stream, ctx->w[idx], gamma, d_scaled, out_f, in_f)) != 0)
                    break;
                if (ctx->precision == VIT_PRECISION_BF16_GEMM)
                    rc = vh_launch_split_rows(ctx->stream, d_scaled, ctx->w16[idx], out_f, in_f, 1);
                else if (ctx->precision == VIT_PRECISION_F32)
                    rc = vh_launch_split3_planes(ctx->stream, d_scaled, ctx->w3[idx], out_f, in_f);
                else
                    rc = vh_launch_quantize_mx_rows(ctx->stream, d_scaled, ctx->w8[idx], ctx->w8s[idx], out_f, in_f);
                if (rc == 0)
                    rc = ctx->precision == VIT_PRECISION_BF16_GEMM
                             ? vh_launch_colsum_operand(ctx->stream, ctx->w16[idx], NULL, ctx->fold_cs[idx], out_f, in_f)
                         : ctx->precision == VIT_PRECISION_F32
                             ? vh_launch_colsum_planes3(ctx->stream, ctx->w3[idx], ctx->fold_cs[idx], out_f, in_f)
                             : vh_launch_colsum_operand(ctx->stream, ctx->w8[idx], ctx->w8s[idx], ctx->fold_cs[idx], out_f, in_f);
                if (rc == 0)
                    rc = vh_launch_fold_bias(ctx->stream, ctx->w[idx], beta, ctx->w[idx + 1], ctx->fold_b[idx], out_f, in_f);
            } else if (ctx->precision == VIT_PRECISION_BF16_GEMM) {
                /* one-part planes [K/32][1][N][32] (gemm_p3.hip); everything else (norms, biases, embeddings, classifier) stays fp32 */
                rc = vh_launch_split_rows(ctx->stream, ctx->w[idx], ctx->w16[idx], out_f, in_f, 1);
            } else if (ctx->precision == VIT_PRECISION_F32 && ctx->w3_slab) {
                /* the constant GEMM operand split once (exact 3-way bf16 split, 6 bytes per weight) */
                rc = vh_launch_split3_planes(ctx->stream, ctx->w[idx], ctx->w3[idx], out_f, in_f);
            } else if (ctx->precision == VIT_PRECISION_FP8_GEMM) {
                rc = vh_launch_quantize_mx_rows(ctx->stream, ctx->w[idx], ctx->w8[idx], ctx->w8s[idx], out_f, in_f);
            } else if (ctx->precision == VIT_PRECISION_F32_FP16X2) {
                /* two fp16 parts of w * 2^k per weight (4 bytes), k per tensor such that max|w| * 2^k lands in
                 * [8192, 16384): the low part stays clear of fp16's subnormals, nothing overflows */
                float amax = 0.0f;
                if ((rc = vh_memset(d_amax, 0, sizeof(float), ctx->stream)) != 0 ||
                    (rc = vh_launch_absmax(ctx->stream, ctx->w[idx], networks[idx].size, d_amax)) != 0 ||
                    (rc = vh_d2h(&amax, d_amax, sizeof(float), ctx->stream)) != 0 ||
                    (rc = vh_stream_sync(ctx->stream)) != 0)
                    break;
                int e = 0;
                float scale = 1.0f;
                if (amax > 0.0f && amax < 3.0e38f) {
                    (void)frexpf(amax, &e);                 /* amax = m * 2^e, m in [0.5, 1) */
                    scale = ldexpf(1.0f, 14 - e);           /* amax * scale in [8192, 16384) */
                }
                ctx->w3_scale[idx] = scale;
                rc = vh_launch_split2h_planes(ctx->stream, ctx->w[idx], ctx->w3[idx], out_f, in_f, scale);
            }
        }
    if (rc == 0 && d_scaled)
        rc = vh_stream_sync(ctx->stream);   /* the scratch copy is freed below */
    if (d_amax)
        vh_free(d_amax);
    if (d_scaled)
        vh_free(d_scaled);
    d_amax = d_scaled = NULL;
    if (rc)
        return rc;
    if (ctx->wconv16)
        TRY(vh_launch_conv_weight_planes_parts(ctx->stream, ctx->w[1], ctx->wconv16, cfg->embed_dim, cfg->in_chans, cfg->patch_size,
                                               ctx->precision == VIT_PRECISION_F32 ? 3 : 1));
    return 0;
fail:
    if (d_amax)
        vh_free(d_amax);
    if (d_scaled)
        vh_free(d_scaled);
    return rc;
}

static int alloc_arena(vit_hip_ctx *ctx);

/* $VIT_HIP_PRECISION: F32 unless asked otherwise: "bf16" -> BF16_GEMM, "fp16x2" -> F32_FP16X2, "fp8" -> FP8_GEMM */
static int env_precision(void)
{
    const char *env = getenv("VIT_HIP_PRECISION");
    return (env && env[0] == 'b') ? VIT_PRECISION_BF16_GEMM
         : (env && strncmp(env, "fp16x2", 6) == 0) ? VIT_PRECISION_F32_FP16X2
         : (env && strncmp(env, "fp8", 3) == 0) ? VIT_PRECISION_FP8_GEMM : VIT_PRECISION_F32;
}

int vit_hip_create(vit_hip_ctx **out, const vit_config *cfg, const Network *networks,
                   int n_tensors, int device, int max_batch)
{
    return vit_hip_create_ex(out, cfg, networks, n_tensors, device, max_batch, env_precision());
}

int vit_hip_precision(const vit_hip_ctx *ctx) { return ctx->precision; }
int vit_hip_ln_fold(const vit_hip_ctx *ctx) { return ctx ? ctx->ln_fold : 0; }

/* The reduced modes fold every LayerNorm but the final one into the projection behind it (csrc/norm_fold.h) unless
 * $VIT_HIP_LN_FOLD=0 (the separate LayerNorm launches of rounds 1-3: the A/B of tests and bench).  The fp32 paths never fold. */
static int want_ln_fold(const vit_config *cfg, int precision)
{
    const char *env = getenv("VIT_HIP_LN_FOLD");
    if (precision == VIT_PRECISION_F32) {
        /* LAB VARIANT, off unless asked for ($VIT_HIP_LN_FOLD=1): the same fold on the exact three-part planes.  It keeps the
         * 1e-4 parity on the goldens, but puts it behind the x - mean cancellation for ~1 % (docs/LABBOOK.md R4.6). */
        const char *p3 = getenv("VIT_HIP_P3"), *native = getenv("VIT_HIP_GEMM_FP32");
        return env && env[0] == '1' && !(p3 && p3[0] == '0') && !(native && native[0] == 'n') && cfg->embed_dim % 128 == 0 &&
               cfg->mlp_hidden % 128 == 0 && cfg->embed_dim / 128 <= 16;
    }
    if (precision != VIT_PRECISION_BF16_GEMM && precision != VIT_PRECISION_FP8_GEMM)
        return 0;
    if (env && env[0] == '0')
        return 0;
    return cfg->embed_dim % 128 == 0 && cfg->embed_dim / 128 <= 16;   /* row_norm_terms: at most 16 partial sums per row */
}

/* Argument checks shared by both ways of making a context, and the empty context itself. */
static int ctx_new(vit_hip_ctx **out, const vit_config *cfg, int n_tensors, int device, int max_batch, int precision, int ln_fold)
{
    if (!out)
        return 1;
    *out = NULL;
    if (!cfg || max_batch <= 0)
        return 1;
    if (precision != VIT_PRECISION_F32 && precision != VIT_PRECISION_BF16_GEMM && precision != VIT_PRECISION_FP8_GEMM &&
        precision != VIT_PRECISION_F32_FP16X2)
        return 1;
    if (precision == VIT_PRECISION_F32_FP16X2 && (cfg->embed_dim % 128 != 0 || cfg->mlp_hidden % 128 != 0))
        return 2;
    if (precision == VIT_PRECISION_FP8_GEMM && (cfg->embed_dim % 256 != 0 || cfg->mlp_hidden % 256 != 0))
        return 2;
    if (precision == VIT_PRECISION_BF16_GEMM && (cfg->embed_dim % 128 != 0 || cfg->mlp_hidden % 128 != 0))
        return 2;
    if (cfg->depth <= 0 || cfg->embed_dim <= 0 || cfg->num_heads <= 0 || cfg->patch_size <= 0 || cfg->img_size <= 0 ||
        cfg->in_chans <= 0 || cfg->num_classes <= 0 || cfg->mlp_hidden <= 0)
        return 2;
    /* sane sizes: what a context may be asked to allocate is bounded whatever a caller or a file header says */
    if (cfg->depth > 256 || cfg->embed_dim > 16384 || cfg->mlp_hidden > 65536 || cfg->img_size > 4096 || cfg->in_chans > 64 ||
        cfg->num_classes > (1 << 20) || cfg->num_heads > 1024 || max_batch > (1 << 20))
        return 2;
    if (n_tensors != vit_config_num_tensors(cfg))
        return 2;
    if (cfg->embed_dim % cfg->num_heads != 0 || cfg->img_size % cfg->patch_size != 0)
        return 2;
    if (ln_fold && ((precision != VIT_PRECISION_BF16_GEMM && precision != VIT_PRECISION_FP8_GEMM && precision != VIT_PRECISION_F32) ||
                    cfg->embed_dim % 128 != 0))
        return 2;
    vit_hip_ctx *ctx = (vit_hip_ctx *)calloc(1, sizeof(*ctx));
    if (!ctx)
        return 4;
    ctx->cfg = *cfg;
    ctx->device = device;
    ctx->max_batch = max_batch;
    ctx->tokens = vit_config_tokens(cfg);
    ctx->n_tensors = n_tensors;
    ctx->precision = precision;
    ctx->ln_fold = ln_fold;
    ctx->fold_cs = (float **)calloc((size_t)n_tensors, sizeof(float *));
    ctx->fold_b = (float **)calloc((size_t)n_tensors, sizeof(float *));
    ctx->w = (float **)calloc((size_t)n_tensors, sizeof(float *));
    ctx->w16 = (void **)calloc((size_t)n_tensors, sizeof(void *));
    ctx->w3 = (void **)calloc((size_t)n_tensors, sizeof(void *));
    ctx->w3_scale = (float *)calloc((size_t)n_tensors, sizeof(float));
    ctx->w8 = (void **)calloc((size_t)n_tensors, sizeof(void *));
    ctx->w8s = (void **)calloc((size_t)n_tensors, sizeof(void *));
    if (!ctx->w || !ctx->w16 || !ctx->w3 || !ctx->w3_scale || !ctx->w8 || !ctx->w8s || !ctx->fold_cs || !ctx->fold_b) {
        free(ctx->fold_cs);
        free(ctx->fold_b);
        free(ctx->w);
        free(ctx->w16);
        free(ctx->w3);
        free(ctx->w3_scale);
        free(ctx->w8);
        free(ctx->w8s);
        free(ctx);
        return 4;
    }
    *out = ctx;
    return 0;
}

int vit_hip_create_ex(vit_hip_ctx **out, const vit_config *cfg, const Network *networks,
                      int n_tensors, int device, int max_batch, int precision)
{
    int rc = 0;
    if (!out)
        return 1;
    *out = NULL;
    if (!networks || !cfg)
        return 1;
    if (n_tensors != vit_config_num_tensors(cfg))
        return 2;
    /* The reference never checks its tensors (a missing file is a NULL
     * dereference, Network.c:144-148); here a wrong count is an error. */
    for (int i = 0; i < n_tensors; ++i)
        if (!networks[i].data || networks[i].size != vit_config_tensor_size(cfg, i)) {
            fprintf(stderr, "vit_hip_create: tensor %d has %zu elements, expected %zu\n", i,
                    networks[i].data ? networks[i].size : (size_t)0, vit_config_tensor_size(cfg, i));
            return 3;
        }
    vit_hip_ctx *ctx = NULL;
    if ((rc = ctx_new(&ctx, cfg, n_tensors, device, max_batch, precision, want_ln_fold(cfg, precision))) != 0)
        return rc;
    TRY(vh_init(device));
    TRY(vh_stream_create(&ctx->stream));
    TRY(layout_weights(ctx));
    TRY(fill_weights(ctx, networks));
    TRY(alloc_arena(ctx));
    *out = ctx;
    return 0;
fail:
    vit_hip_destroy(ctx);
    return rc;
}

/* ---- repacked weights on disk (the offline half of the weight-format tooling) --------------------------
 * vit_hip_export_planes writes what a context holds in HBM after its repack -- the fp32 slab (all tensors, the
 * reference's order, 256-byte aligned) followed by the mode's operand slab for the four big matrices of every layer
 * (three-part bf16 planes, one-part planes, two fp16 parts, or MX values + scales: csrc/gemm_p3.hip, gemm_mx.hip) and,
 * in the reduced modes, the conv_proj planes -- behind a header that pins model shape and precision.  Both slabs'
 * layouts follow from (config, precision) alone (layout_weights), so vit_hip_create_from_planes is three reads into
 * three allocations: no fp32 -> format pass, no per-tensor files (the reference's loader opens 152 of them,
 * Network.c:134-218). */
/* Bumped whenever a repack kernel changes what it writes for the same (config, precision): the slab sizes alone would
 * not notice (csrc/gemm_p3.hip planes, csrc/gemm_mx.hip MX values / scale order, the fold terms of csrc/norm_fold.h). */
#define VIT_PLANES_LAYOUT_VERSION 2

struct planes_header
{
    char magic[8];                 /* "VITPLN02" */
    unsigned header_bytes;
    int precision, n_tensors;
    int cfg_ints[8];               /* img, patch, chans, classes, embed, depth, heads, mlp_hidden */
    double eps;
    unsigned long long w_slab_bytes, planes_bytes, wconv16_bytes, scale_floats;   /* scale_floats = n_tensors (w3_scale) */
    unsigned long long fold_bytes; /* colsum + folded bias of the gamma-scaled matrices (ln_fold) */
    int ln_fold, layout_version;
    unsigned long long checksum;   /* of everything behind the header, in file order (planes_hash) */
};

/* FNV-1a over 64-bit words (the payload slabs are all multiples of 8 bytes; a tail is taken byte-wise) */
static unsigned long long planes_hash(unsigned long long h, const void *data, size_t bytes)
{
    const unsigned long long *w = (const unsigned long long *)data;
    for (size_t i = 0; i < bytes / 8; ++i)
        h = (h ^ w[i]) * 0x100000001b3ull;
    const unsigned char *t = (const unsigned char *)data + (bytes & ~(size_t)7);
    for (size_t i = 0; i < (bytes & 7); ++i)
        h = (h ^ t[i]) * 0x100000001b3ull;
    return h;
}
#define PLANES_HASH_SEED 0xcbf29ce484222325ull

static int copy_file_and_device(vit_hip_ctx *ctx, FILE *fp, void *dev, size_t bytes, int to_file, unsigned long long *hash)
{
    enum { CHUNK = 64 << 20 };
    int rc = 0;
    void *host = NULL;
    if (bytes == 0)
        return 0;
    TRY(vh_host_alloc(&host, bytes < CHUNK ? bytes : (size_t)CHUNK));
    for (size_t off = 0; off < bytes && rc == 0; off += CHUNK) {
        const size_t n = bytes - off < CHUNK ? bytes - off : (size_t)CHUNK;
        if (to_file) {
            if ((rc = vh_d2h(host, (char *)dev + off, n, ctx->stream)) != 0 || (rc = vh_stream_sync(ctx->stream)) != 0)
                break;
            *hash = planes_hash(*hash, host, n);
            if (fwrite(host, 1, n, fp) != n)
                rc = vh_set_error(120, "vit_hip_export_planes: short write");
        } else {
            if (fread(host, 1, n, fp) != n) {
                rc = vh_set_error(121, "vit_hip_create_from_planes: file is shorter than its header says");
                break;
            }
            *hash = planes_hash(*hash, host, n);
            if ((rc = vh_h2d((char *)dev + off, host, n, ctx->stream)) != 0 || (rc = vh_stream_sync(ctx->stream)) != 0)
                break;
        }
    }
fail:
    if (host)
        vh_host_free(host);
    return rc;
}

static void *planes_slab(const vit_hip_ctx *ctx)
{
    return ctx->precision == VIT_PRECISION_BF16_GEMM ? ctx->w16_slab : ctx->precision == VIT_PRECISION_FP8_GEMM ? ctx->w8_slab : ctx->w3_slab;
}

/* Written to `path`.tmp and renamed over `path` when complete: an interrupted export never leaves a truncated file
 * under the name a later run will open. */
int vit_hip_export_planes(vit_hip_ctx *ctx, const char *path)
{
    int rc = 0;
    if (!ctx || !path)
        return vh_set_error(1, "vit_hip_export_planes: null argument");
    TRY(vh_set_device(ctx->device));
    const size_t plen = strlen(path);
    char *tmp = (char *)malloc(plen + 5);
    if (!tmp)
        return vh_set_error(4, "vit_hip_export_planes: out of host memory");
    memcpy(tmp, path, plen);
    memcpy(tmp + plen, ".tmp", 5);
    FILE *fp = fopen(tmp, "wb");
    if (!fp) {
        free(tmp);
        return vh_set_error(122, "vit_hip_export_planes: cannot open the file for writing");
    }
    struct planes_header h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "VITPLN02", 8);
    h.header_bytes = (unsigned)sizeof(h);
    h.precision = ctx->precision;
    h.n_tensors = ctx->n_tensors;
    const vit_config *c = &ctx->cfg;
    const int ints[8] = {c->img_size, c->patch_size, c->in_chans, c->num_classes, c->embed_dim, c->depth, c->num_heads, c->mlp_hidden};
    memcpy(h.cfg_ints, ints, sizeof(ints));
    h.eps = c->eps;
    h.w_slab_bytes = ctx->w_slab_bytes;
    h.planes_bytes = ctx->planes_bytes;
    h.wconv16_bytes = ctx->wconv16_bytes;
    h.fold_bytes = ctx->fold_bytes;
    h.ln_fold = ctx->ln_fold;
    h.layout_version = VIT_PLANES_LAYOUT_VERSION;
    h.scale_floats = (unsigned long long)ctx->n_tensors;
    unsigned long long hash = planes_hash(PLANES_HASH_SEED, ctx->w3_scale, sizeof(float) * (size_t)ctx->n_tensors);
    /* the header goes first with a zero checksum and is rewritten once the payload has been hashed */
    if (fwrite(&h, sizeof(h), 1, fp) != 1 || fwrite(ctx->w3_scale, sizeof(float), (size_t)ctx->n_tensors, fp) != (size_t)ctx->n_tensors)
        rc = vh_set_error(120, "vit_hip_export_planes: short write");
    if (rc == 0)
        rc = copy_file_and_device(ctx, fp, ctx->w_slab, ctx->w_slab_bytes, 1, &hash);
    if (rc == 0)
        rc = copy_file_and_device(ctx, fp, planes_slab(ctx), ctx->planes_bytes, 1, &hash);
    if (rc == 0)
        rc = copy_file_and_device(ctx, fp, ctx->wconv16, ctx->wconv16_bytes, 1, &hash);
    if (rc == 0)
        rc = copy_file_and_device(ctx, fp, ctx->fold_slab, ctx->fold_bytes, 1, &hash);
    if (rc == 0) {
        h.checksum = hash;
        if (fseek(fp, 0, SEEK_SET) != 0 || fwrite(&h, sizeof(h), 1, fp) != 1)
            rc = vh_set_error(120, "vit_hip_export_planes: short write");
    }
    if (fclose(fp) != 0 && rc == 0)
        rc = vh_set_error(120, "vit_hip_export_planes: short write");
    if (rc == 0 && rename(tmp, path) != 0)
        rc = vh_set_error(122, "vit_hip_export_planes: cannot move the finished file into place");
    if (rc != 0)
        remove(tmp);
    free(tmp);
    return rc;
fail:
    return rc;
}

int vit_hip_create_from_planes(vit_hip_ctx **out, const char *path, int device, int max_batch)
{
    int rc = 0;
    if (!out)
        return 1;
    *out = NULL;
    if (!path)
        return vh_set_error(1, "vit_hip_create_from_planes: null path");
    FILE *fp = fopen(path, "rb");
    if (!fp)
        return vh_set_error(123, "vit_hip_create_from_planes: cannot open the file");
    struct planes_header h;
    vit_hip_ctx *ctx = NULL;
    if (fread(&h, sizeof(h), 1, fp) != 1 || memcmp(h.magic, "VITPLN02", 8) != 0 || h.header_bytes != sizeof(h) ||
        h.layout_version != VIT_PLANES_LAYOUT_VERSION) {
        fclose(fp);
        return vh_set_error(124, "vit_hip_create_from_planes: not a planes file of this library version (magic, header size or operand-layout version)");
    }
    vit_config cfg = {h.cfg_ints[0], h.cfg_ints[1], h.cfg_ints[2], h.cfg_ints[3], h.cfg_ints[4], h.cfg_ints[5], h.cfg_ints[6],
                      h.cfg_ints[7], h.eps};
    /* ctx_new bounds every dimension and allocates only the per-tensor pointer tables (n_tensors is tied to depth) */
    if (h.n_tensors <= 0 || h.n_tensors > 4 + 12 * 256 + 4 ||
        (rc = ctx_new(&ctx, &cfg, h.n_tensors, device, max_batch, h.precision, h.ln_fold ? 1 : 0)) != 0) {
        fclose(fp);
        return vh_set_error(rc ? rc : 2, "vit_hip_create_from_planes: the header's model shape or precision is not one this library takes");
    }
    {   /* the slab sizes this library derives from (shape, precision, fold) against the header's, BEFORE anything is allocated */
        size_t sizes[4];
        weight_slab_sizes(ctx, sizes);
        if (h.w_slab_bytes != sizes[0] || h.planes_bytes != sizes[1] || h.wconv16_bytes != sizes[2] || h.fold_bytes != sizes[3] ||
            h.scale_floats != (unsigned long long)ctx->n_tensors) {
            rc = vh_set_error(125, "vit_hip_create_from_planes: slab sizes in the file do not match this library's layout");
            goto fail;
        }
    }
    if (fread(ctx->w3_scale, sizeof(float), (size_t)ctx->n_tensors, fp) != (size_t)ctx->n_tensors) {
        rc = vh_set_error(124, "vit_hip_create_from_planes: truncated header");
        goto fail;
    }
    unsigned long long hash = planes_hash(PLANES_HASH_SEED, ctx->w3_scale, sizeof(float) * (size_t)ctx->n_tensors);
    TRY(vh_init(device));
    TRY(vh_stream_create(&ctx->stream));
    TRY(layout_weights(ctx));
    TRY(copy_file_and_device(ctx, fp, ctx->w_slab, ctx->w_slab_bytes, 0, &hash));
    TRY(copy_file_and_device(ctx, fp, planes_slab(ctx), ctx->planes_bytes, 0, &hash));
    TRY(copy_file_and_device(ctx, fp, ctx->wconv16, ctx->wconv16_bytes, 0, &hash));
    TRY(copy_file_and_device(ctx, fp, ctx->fold_slab, ctx->fold_bytes, 0, &hash));
    if (hash != h.checksum) {
        rc = vh_set_error(126, "vit_hip_create_from_planes: payload checksum mismatch (corrupt file)");
        goto fail;
    }
    TRY(alloc_arena(ctx));
    fclose(fp);
    *out = ctx;
    return 0;
fail:
    fclose(fp);
    vit_hip_destroy(ctx);
    return rc;
}

/* The activation arena and the host-pointer path's staging, sized for max_batch images. */
static int alloc_arena(vit_hip_ctx *ctx)
{
    int rc = 0;
    const vit_config *cfg = &ctx->cfg;
    const int precision = ctx->precision, max_batch = ctx->max_batch;
    const size_t E = (size_t)cfg->embed_dim, F = (size_t)cfg->mlp_hidden, NC = (size_t)cfg->num_classes;
    const size_t rows = (size_t)max_batch * ctx->tokens;
    const size_t img = (size_t)cfg->in_chans * cfg->img_size * cfg->img_size;
    {   /* the default fp32 path: every GEMM input is written by its producer as the exact three-part
         * bf16 split (csrc/gemm_p3.hip), 6 bytes per value; VIT_HIP_P3=0 keeps fp32 activations and
         * the in-loop split (csrc/gemm_mfma.hip) */
        const char *env_p3 = getenv("VIT_HIP_P3");
        const char *env_native = getenv("VIT_HIP_GEMM_FP32");   /* "native": the fp32 matrix instruction (gemm_mfma.hip) */
        ctx->use_p3 = ctx->w3_slab && precision == VIT_PRECISION_F32 && !(env_p3 && env_p3[0] == '0') &&
                      !(env_native && env_native[0] == 'n') && rows * 64 <= 0xffffffffull;
    }
    if (precision == VIT_PRECISION_F32 && ctx->ln_fold && !ctx->use_p3)
        return vh_set_error(2, "vit_hip_create: $VIT_HIP_LN_FOLD=1 on the fp32 path needs the planes path (batch too large for it)");
    {
        const char *env_ll = getenv("VIT_HIP_LAST_LAYER");
        ctx->cls_only_last = ctx->use_p3 && !ctx->ln_fold && env_ll && strcmp(env_ll, "cls") == 0;
    }
    const size_t act = ctx->use_p3 ? 6 : sizeof(float);   /* bytes per GEMM-input value */
    TRY(vh_malloc((void **)&ctx->x, rows * E * sizeof(float)));
    TRY(vh_malloc((void **)&ctx->y, rows * E * act));
    TRY(vh_malloc((void **)&ctx->attn, rows * E * act));
    TRY(vh_malloc((void **)&ctx->qkv, rows * 3 * E * act));
    {   /* patch geometries that need gathered rows (H/14) borrow the MLP hidden buffer, idle at that point */
        size_t ws = vh_patch_embed_workspace(max_batch, cfg->in_chans, cfg->img_size, cfg->patch_size, cfg->embed_dim);
        if (ctx->wconv16) {   /* the im2row producer's planes: patches x Kp x 2 bytes x parts (a small MLP can be smaller than that) */
            const size_t grid = (size_t)(cfg->img_size / cfg->patch_size);
            const size_t planes = (size_t)max_batch * grid * grid * (size_t)vh_patch_planes_k(cfg->in_chans, cfg->patch_size) * 2 *
                                  (precision == VIT_PRECISION_F32 ? 3 : 1);
            ws = ws > planes ? ws : planes;
        }
        const size_t hid_bytes = rows * F * act;
        ctx->ws_bytes = ws > hid_bytes ? ws : hid_bytes;
        TRY(vh_malloc((void **)&ctx->hid, ctx->ws_bytes));
    }
    if (ctx->ln_fold)
        TRY(vh_malloc((void **)&ctx->stats, rows * (E / 128) * 2 * sizeof(float)));
    TRY(vh_malloc((void **)&ctx->cls, (size_t)max_batch * E * sizeof(float)));
    TRY(vh_malloc((void **)&ctx->d_logits, (size_t)max_batch * NC * sizeof(float)));
    TRY(vh_malloc((void **)&ctx->d_probs, (size_t)max_batch * NC * sizeof(float)));
    TRY(vh_stream_create(&ctx->copy_stream));
    for (int i = 0; i < 2; ++i) {
        TRY(vh_malloc((void **)&ctx->d_images[i], (size_t)max_batch * img * sizeof(float)));
        TRY(vh_host_alloc((void **)&ctx->h_images[i], (size_t)max_batch * img * sizeof(float)));
        TRY(vh_host_alloc((void **)&ctx->h_logits[i], (size_t)max_batch * NC * sizeof(float)));
        TRY(vh_host_alloc((void **)&ctx->h_probs[i], (size_t)max_batch * NC * sizeof(float)));
        TRY(vh_event_create(&ctx->up_done[i]));
        TRY(vh_event_create(&ctx->comp_done[i]));
        TRY(vh_event_create(&ctx->out_done[i]));
    }
    TRY(vh_stream_sync(ctx->stream));
    return 0;
fail:
    return rc;
}

/* ---- one encoder layer per arithmetic (ViT_seq.c:330-370; Encoder ViT_opencl.c:710-748) ------------------------------
 * Every function enqueues layer l for n images on stream s: y = LN1(x); qkv = y Win^T + bin; a = attention(qkv);
 * x += a Wout^T + bout; y = LN2(x); h = gelu(y W1^T + b1); x += h W2^T + b2 -- in its mode's operand formats.  With
 * ctx->ln_fold the two LayerNorms are not launched: whoever wrote x also left it as the projection's operand in ctx->y
 * with the rows' partial sums in ctx->stats, and the QKV / fc1 launches apply the row terms (csrc/norm_fold.h). */

/* Q|K|V as one-part fp16 planes for the planes attention kernels (head_dim 64 with T <= 208; head_dim 80 with T <= 272:
 * ViT-H/14), fp32 rows for the streaming kernel */
static int qkv_as_planes(const vit_hip_ctx *ctx)
{
    const int E = ctx->cfg.embed_dim, H = ctx->cfg.num_heads, T = ctx->tokens;
    return (E == 64 * H && T <= 208) || (E == 80 * H && T <= 272);
}

/* The reduced modes' attention on fp16-rounded operands, writing the output projection's operand: one-part bf16 planes
 * (attn_scales NULL) or an MX tensor.  The planes kernels write it themselves; the streaming kernel (and head_dim 80 with
 * an odd head count) leaves fp32 rows in the idle MLP buffer, which are then rounded / quantised (one timed operator). */
static int attention_reduced(vit_hip_ctx *ctx, vh_stream_t s, int n, char *attn_scales)
{
    int rc = 0;
    const vit_config *c = &ctx->cfg;
    const int E = c->embed_dim, T = ctx->tokens, H = c->num_heads, rows = n * T;
    if (E == 64 * H && T <= 208)
        OP(VIT_OP_ATTENTION, attn_scales ? vh_launch_attention_planes_f16_mx(s, ctx->qkv, ctx->attn, attn_scales, n, T, E, H)
                                         : vh_launch_attention_planes_f16(s, ctx->qkv, ctx->attn, 1, n, T, E, H));
    else if (qkv_as_planes(ctx) && (H & 1) == 0)
        OP(VIT_OP_ATTENTION, vh_launch_attention_planes_f16_hd80_operand(s, ctx->qkv, ctx->attn, attn_scales, attn_scales ? 2 : 1, n, T, E, H));
    else
        OP(VIT_OP_ATTENTION, (rc = qkv_as_planes(ctx) ? vh_launch_attention_planes_f16_hd80(s, ctx->qkv, ctx->hid, n, T, E, H)
                                                      : vh_launch_attention_f16(s, ctx->qkv, ctx->hid, n, T, E, H)) != 0 ? rc :
                             attn_scales ? vh_launch_quantize_mx_act(s, ctx->hid, ctx->attn, attn_scales, rows, E)
                                         : vh_launch_split_rows(s, ctx->hid, ctx->attn, rows, E, 1));
    return 0;
fail:
    return rc;
}

/* FP8_GEMM: y, attn and hid hold MX tensors (values, then the scales, in the same allocations) */
static int layer_fp8(vit_hip_ctx *ctx, vh_stream_t s, int n, int l)
{
    int rc = 0;
    const vit_config *c = &ctx->cfg;
    const int E = c->embed_dim, F = c->mlp_hidden, rows = n * ctx->tokens, fold = ctx->ln_fold, last = l == c->depth - 1;
    float **lw = ctx->w + 4 + 12 * l;
    void **l8 = ctx->w8 + 4 + 12 * l, **l8s = ctx->w8s + 4 + 12 * l;
    float **cs = ctx->fold_cs + 4 + 12 * l, **bf = ctx->fold_b + 4 + 12 * l;
    char *ys = (char *)ctx->y + align_up((size_t)rows * E, 256), *as_ = (char *)ctx->attn + align_up((size_t)rows * E, 256);
    char *hs = (char *)ctx->hid + align_up((size_t)rows * F, 256);
    const int kind = qkv_as_planes(ctx) ? 2 : 0;
    if (!fold)
        OP(VIT_OP_LAYER_NORM, vh_launch_layer_norm_mx(s, ctx->x, lw[0], lw[1], ctx->y, ys, rows, E, E, c->eps));
    OP(VIT_OP_QKV, fold ? vh_launch_linear_mx_norm(s, ctx->qkv, NULL, kind, l8[2], l8s[2], ctx->y, ys, ctx->stats, cs[2], bf[2], c->eps, rows, E, 3 * E, 0)
                 : kind ? vh_launch_linear_mx_planes_f16(s, ctx->qkv, l8[2], l8s[2], ctx->y, ys, lw[3], rows, E, 3 * E)
                        : vh_launch_linear_mx(s, ctx->qkv, NULL, l8[2], l8s[2], ctx->y, ys, lw[3], rows, E, 3 * E, 0, NULL));
    TRY(attention_reduced(ctx, s, n, as_));
    OP(VIT_OP_OUT_PROJ, fold ? vh_launch_linear_mx_resid_norm(s, ctx->x, l8[4], l8s[4], ctx->attn, as_, lw[5], ctx->x, rows, E, E, ctx->y, ys, ctx->stats)
                             : vh_launch_linear_mx(s, ctx->x, NULL, l8[4], l8s[4], ctx->attn, as_, lw[5], rows, E, E, 0, ctx->x));
    if (!fold)
        OP(VIT_OP_LAYER_NORM, vh_launch_layer_norm_mx(s, ctx->x, lw[6], lw[7], ctx->y, ys, rows, E, E, c->eps));
    OP(VIT_OP_FC1, fold ? vh_launch_linear_mx_norm(s, ctx->hid, hs, 1, l8[8], l8s[8], ctx->y, ys, ctx->stats, cs[8], bf[8], c->eps, rows, E, F, 1)
                        : vh_launch_linear_mx(s, ctx->hid, hs, l8[8], l8s[8], ctx->y, ys, lw[9], rows, E, F, 1, NULL));
    /* nothing reads the operand behind the last layer: the final LayerNorm takes the fp32 rows */
    OP(VIT_OP_FC2, fold && !last ? vh_launch_linear_mx_resid_norm(s, ctx->x, l8[10], l8s[10], ctx->hid, hs, lw[11], ctx->x, rows, F, E, ctx->y, ys, ctx->stats)
                                 : vh_launch_linear_mx(s, ctx->x, NULL, l8[10], l8s[10], ctx->hid, hs, lw[11], rows, F, E, 0, ctx->x));
    return 0;
fail:
    return rc;
}

/* BF16_GEMM: operands as one-part planes [K/32][1][rows][32] written by their producers; the fp32 path's kernel with one
 * product per block (gemm_p3.hip, NPL = 1) */
static int layer_bf16(vit_hip_ctx *ctx, vh_stream_t s, int n, int l)
{
    int rc = 0;
    const vit_config *c = &ctx->cfg;
    const int E = c->embed_dim, F = c->mlp_hidden, rows = n * ctx->tokens, fold = ctx->ln_fold, last = l == c->depth - 1;
    float **lw = ctx->w + 4 + 12 * l;
    void **lw16 = ctx->w16 + 4 + 12 * l;
    float **cs = ctx->fold_cs + 4 + 12 * l, **bf = ctx->fold_b + 4 + 12 * l;
    const int kind = qkv_as_planes(ctx) ? 2 : 0;
    if (!fold)
        OP(VIT_OP_LAYER_NORM, vh_launch_layer_norm_planes(s, ctx->x, lw[0], lw[1], ctx->y, 1, rows, E, E, c->eps));
    OP(VIT_OP_QKV, fold ? vh_launch_linear_planes_norm(s, ctx->qkv, kind, lw16[2], ctx->y, ctx->stats, cs[2], bf[2], c->eps, rows, E, 3 * E, 0)
                        : vh_launch_linear_planes(s, ctx->qkv, kind, lw16[2], ctx->y, 1, lw[3], rows, E, 3 * E, 0, NULL));
    TRY(attention_reduced(ctx, s, n, NULL));
    OP(VIT_OP_OUT_PROJ, fold ? vh_launch_linear_planes_resid_norm(s, ctx->x, lw16[4], ctx->attn, lw[5], ctx->x, rows, E, E, ctx->y, NULL, ctx->stats)
                             : vh_launch_linear_planes(s, ctx->x, 0, lw16[4], ctx->attn, 1, lw[5], rows, E, E, 0, ctx->x));
    if (!fold)
        OP(VIT_OP_LAYER_NORM, vh_launch_layer_norm_planes(s, ctx->x, lw[6], lw[7], ctx->y, 1, rows, E, E, c->eps));
    OP(VIT_OP_FC1, fold ? vh_launch_linear_planes_norm(s, ctx->hid, 1, lw16[8], ctx->y, ctx->stats, cs[8], bf[8], c->eps, rows, E, F, 1)
                        : vh_launch_linear_planes(s, ctx->hid, 1, lw16[8], ctx->y, 1, lw[9], rows, E, F, 1, NULL));
    OP(VIT_OP_FC2, fold && !last ? vh_launch_linear_planes_resid_norm(s, ctx->x, lw16[10], ctx->hid, lw[11], ctx->x, rows, F, E, ctx->y, NULL, ctx->stats)
                                 : vh_launch_linear_planes(s, ctx->x, 0, lw16[10], ctx->hid, 1, lw[11], rows, F, E, 0, ctx->x));
    return 0;
fail:
    return rc;
}

/* F32_FP16X2: the fp32 layer with the four projections on two fp16 parts / three products */
static int layer_fp16x2(vit_hip_ctx *ctx, vh_stream_t s, int n, int l)
{
    int rc = 0;
    const vit_config *c = &ctx->cfg;
    const int E = c->embed_dim, F = c->mlp_hidden, T = ctx->tokens, rows = n * T;
    float **lw = ctx->w + 4 + 12 * l;
    void **l3 = ctx->w3 + 4 + 12 * l;
    const float *ws = ctx->w3_scale + 4 + 12 * l;
    OP(VIT_OP_LAYER_NORM, vh_launch_layer_norm(s, ctx->x, lw[0], lw[1], ctx->y, rows, E, E, E, c->eps));
    OP(VIT_OP_QKV, vh_launch_linear_h2(s, ctx->qkv, l3[2], ws[2], ctx->y, lw[3], rows, E, 3 * E, 0, NULL));
    OP(VIT_OP_ATTENTION, vh_launch_attention_h2(s, ctx->qkv, ctx->attn, n, T, E, c->num_heads));
    OP(VIT_OP_OUT_PROJ, vh_launch_linear_h2(s, ctx->x, l3[4], ws[4], ctx->attn, lw[5], rows, E, E, 0, ctx->x));
    OP(VIT_OP_LAYER_NORM, vh_launch_layer_norm(s, ctx->x, lw[6], lw[7], ctx->y, rows, E, E, E, c->eps));
    OP(VIT_OP_FC1, vh_launch_linear_h2(s, ctx->hid, l3[8], ws[8], ctx->y, lw[9], rows, E, F, 1, NULL));
    OP(VIT_OP_FC2, vh_launch_linear_h2(s, ctx->x, l3[10], ws[10], ctx->hid, lw[11], rows, F, E, 0, ctx->x));
    return 0;
fail:
    return rc;
}

/* F32, the default: GEMM inputs as exact three-part planes written by LayerNorm, attention and the fc1 epilogue.  ln_fold
 * here is the LAB VARIANT ($VIT_HIP_LN_FOLD=1, docs/LABBOOK.md R4.6): ctx->y then holds the split of x itself.
 * *cls_rows is set when the last layer ran on the class-token rows only (opt-in): the final LayerNorm then reads them
 * compacted at the start of the Q|K|V buffer. */
static int layer_f32_planes(vit_hip_ctx *ctx, vh_stream_t s, int n, int l, int *cls_rows)
{
    int rc = 0;
    const vit_config *c = &ctx->cfg;
    const int E = c->embed_dim, F = c->mlp_hidden, T = ctx->tokens, rows = n * T, fold = ctx->ln_fold, last = l == c->depth - 1;
    float **lw = ctx->w + 4 + 12 * l;
    void **l3 = ctx->w3 + 4 + 12 * l;
    float **cs = ctx->fold_cs + 4 + 12 * l, **bf = ctx->fold_b + 4 + 12 * l;
    /* head_dim 64, T <= 208: Q, K, V too travel as planes (only the probabilities are split inside the attention kernel);
     * other shapes: the streaming kernel, fp32 rows in, fp32 out (into the idle MLP buffer), then split */
    const int planes_attn = E == 64 * c->num_heads && T <= 208;
    if (!fold)
        OP(VIT_OP_LAYER_NORM, vh_launch_layer_norm_p3(s, ctx->x, lw[0], lw[1], ctx->y, rows, E, E, c->eps));
    OP(VIT_OP_QKV, fold ? vh_launch_linear_p3_norm(s, ctx->qkv, planes_attn, l3[2], ctx->y, ctx->stats, cs[2], bf[2], c->eps, rows, E, 3 * E, 0)
                        : vh_launch_linear_p3(s, ctx->qkv, planes_attn, l3[2], ctx->y, lw[3], rows, E, 3 * E, 0, NULL));
    OP(VIT_OP_ATTENTION, planes_attn ? vh_launch_attention_planes(s, ctx->qkv, ctx->attn, n, T, E, c->num_heads)
                         : (rc = vh_launch_attention(s, ctx->qkv, ctx->hid, n, T, E, c->num_heads)) != 0 ? rc
                         : vh_launch_split3_rows(s, ctx->hid, ctx->attn, rows, E));
    if (last && ctx->cls_only_last && T >= 4) {
        /* Opt-in (vit_hip_set_last_layer_cls_only / $VIT_HIP_LAST_LAYER=cls).  The classifier reads row 0 of every
         * image only (ViT_seq.c:511), and behind the last attention no operator mixes rows: the output projection,
         * LayerNorm and MLP of the last layer are evaluated for the n class-token rows instead of n * T, in the
         * Q|K|V buffer the attention has just released.  Same kernels, same k order: identical logits, bit for
         * bit; the residual stream of the other rows (vit_hip_read_tokens) is NOT updated by this layer. */
        char *scratch = (char *)ctx->qkv;
        float *x_cls = (float *)scratch;
        char *attn_cls = scratch + align_up((size_t)n * E * 4, 256);
        char *y_cls = attn_cls + align_up((size_t)n * E * 6, 256);
        char *hid_cls = y_cls + align_up((size_t)n * E * 6, 256);
        OP(VIT_OP_OUT_PROJ, (rc = vh_launch_gather_rows(s, ctx->attn, attn_cls, 3 * (E / 32), rows, n, 64, T)) != 0 ? rc :
                            (rc = vh_launch_gather_rows(s, ctx->x, x_cls, 1, rows, n, 4 * E, T)) != 0 ? rc :
                            vh_launch_linear_p3(s, x_cls, 0, l3[4], attn_cls, lw[5], n, E, E, 0, x_cls));
        OP(VIT_OP_LAYER_NORM, vh_launch_layer_norm_p3(s, x_cls, lw[6], lw[7], y_cls, n, E, E, c->eps));
        OP(VIT_OP_FC1, vh_launch_linear_p3(s, hid_cls, 1, l3[8], y_cls, lw[9], n, E, F, 1, NULL));
        OP(VIT_OP_FC2, vh_launch_linear_p3(s, x_cls, 0, l3[10], hid_cls, lw[11], n, F, E, 0, x_cls));
        *cls_rows = 1;
        return 0;
    }
    OP(VIT_OP_OUT_PROJ, fold ? vh_launch_linear_p3_resid_norm(s, ctx->x, l3[4], ctx->attn, lw[5], ctx->x, rows, E, E, ctx->y, ctx->stats)
                             : vh_launch_linear_p3(s, ctx->x, 0, l3[4], ctx->attn, lw[5], rows, E, E, 0, ctx->x));
    if (!fold)
        OP(VIT_OP_LAYER_NORM, vh_launch_layer_norm_p3(s, ctx->x, lw[6], lw[7], ctx->y, rows, E, E, c->eps));
    OP(VIT_OP_FC1, fold ? vh_launch_linear_p3_norm(s, ctx->hid, 1, l3[8], ctx->y, ctx->stats, cs[8], bf[8], c->eps, rows, E, F, 1)
                        : vh_launch_linear_p3(s, ctx->hid, 1, l3[8], ctx->y, lw[9], rows, E, F, 1, NULL));
    OP(VIT_OP_FC2, fold && !last ? vh_launch_linear_p3_resid_norm(s, ctx->x, l3[10], ctx->hid, lw[11], ctx->x, rows, F, E, ctx->y, ctx->stats)
                                 : vh_launch_linear_p3(s, ctx->x, 0, l3[10], ctx->hid, lw[11], rows, F, E, 0, ctx->x));
    return 0;
fail:
    return rc;
}

/* F32 with fp32 activation rows ($VIT_HIP_P3=0, $VIT_HIP_GEMM_FP32=native, or shapes the planes cannot take): round 1's
 * kernels, operands split inside the K loop (pre-split weight planes when built) or the native fp32 MFMA */
static int layer_f32_rows(vit_hip_ctx *ctx, vh_stream_t s, int n, int l)
{
    int rc = 0;
    const vit_config *c = &ctx->cfg;
    const int E = c->embed_dim, F = c->mlp_hidden, T = ctx->tokens, rows = n * T;
    float **lw = ctx->w + 4 + 12 * l;   /* ln1 w,b; in w,b; out w,b; ln2 w,b; fc1 w,b; fc2 w,b */
    void **l3 = ctx->w3 + 4 + 12 * l;   /* pre-split weight planes, when built */
    OP(VIT_OP_LAYER_NORM, vh_launch_layer_norm(s, ctx->x, lw[0], lw[1], ctx->y, rows, E, E, E, c->eps));
    OP(VIT_OP_QKV, l3[2] ? vh_launch_linear_w3(s, ctx->qkv, l3[2], ctx->y, lw[3], rows, E, 3 * E, 0, NULL)
                         : vh_launch_linear(s, ctx->qkv, lw[2], ctx->y, lw[3], rows, E, 3 * E, 0, NULL));
    OP(VIT_OP_ATTENTION, vh_launch_attention(s, ctx->qkv, ctx->attn, n, T, E, c->num_heads));
    OP(VIT_OP_OUT_PROJ, l3[4] ? vh_launch_linear_w3(s, ctx->x, l3[4], ctx->attn, lw[5], rows, E, E, 0, ctx->x)
                              : vh_launch_linear(s, ctx->x, lw[4], ctx->attn, lw[5], rows, E, E, 0, ctx->x));
    OP(VIT_OP_LAYER_NORM, vh_launch_layer_norm(s, ctx->x, lw[6], lw[7], ctx->y, rows, E, E, E, c->eps));
    OP(VIT_OP_FC1, l3[8] ? vh_launch_linear_w3(s, ctx->hid, l3[8], ctx->y, lw[9], rows, E, F, 1, NULL)
                         : vh_launch_linear(s, ctx->hid, lw[8], ctx->y, lw[9], rows, E, F, 1, NULL));
    OP(VIT_OP_FC2, l3[10] ? vh_launch_linear_w3(s, ctx->x, l3[10], ctx->hid, lw[11], rows, F, E, 0, ctx->x)
                          : vh_launch_linear(s, ctx->x, lw[10], ctx->hid, lw[11], rows, F, E, 0, ctx->x));
    return 0;
fail:
    return rc;
}

/* patch embedding + class token + position embedding (ViT_seq.c:437-443), in the form the mode's first layer reads */
static int patch_embedding(vit_hip_ctx *ctx, vh_stream_t s, const float *d_images, int n)
{
    int rc = 0;
    const vit_config *c = &ctx->cfg;
    const int E = c->embed_dim;
    float **w = ctx->w;
    /* ln_fold: y receives the token rows as the first projection's operand -- planes, or MX values with their scales behind
     * them -- and stats their partial sums (class-token rows included) */
    char *const y_scales = (char *)ctx->y + align_up((size_t)n * ctx->tokens * E, 256);
    if (ctx->ln_fold && ctx->precision == VIT_PRECISION_F32)   /* lab variant: the fold on three-part planes */
        OP(VIT_OP_PATCH_EMBED, vh_launch_patch_embed_planes3_norm(s, d_images, ctx->wconv16, w[2], w[0], w[3], ctx->x, n, c->in_chans,
                                                                  c->img_size, c->patch_size, E, ctx->hid, ctx->ws_bytes, ctx->y, ctx->stats));
    else if (ctx->ln_fold)
        OP(VIT_OP_PATCH_EMBED, vh_launch_patch_embed_planes_norm(s, d_images, ctx->wconv16, w[2], w[0], w[3], ctx->x, n, c->in_chans,
                                                                 c->img_size, c->patch_size, E, ctx->hid, ctx->ws_bytes, ctx->y,
                                                                 ctx->precision == VIT_PRECISION_FP8_GEMM ? y_scales : NULL, ctx->stats));
    else if (ctx->wconv16 && ctx->precision != VIT_PRECISION_F32)   /* reduced modes: im2row to one-part planes (in the MLP buffer, idle here) + the planes GEMM */
        OP(VIT_OP_PATCH_EMBED, vh_launch_patch_embed_planes(s, d_images, ctx->wconv16, w[2], w[0], w[3], ctx->x, n, c->in_chans,
                                                            c->img_size, c->patch_size, E, ctx->hid, ctx->ws_bytes));
    else if (ctx->wconv16 && ctx->use_p3)   /* the fp32 path on planes: im2row writes the exact three-part split, six products per block */
        OP(VIT_OP_PATCH_EMBED, vh_launch_patch_embed_planes3(s, d_images, ctx->wconv16, w[2], w[0], w[3], ctx->x, n, c->in_chans,
                                                             c->img_size, c->patch_size, E, ctx->hid, ctx->ws_bytes));
    else
        OP(VIT_OP_PATCH_EMBED, vh_launch_patch_embed_ws(s, d_images, w[1], w[2], w[0], w[3], ctx->x, n, c->in_chans,
                                                        c->img_size, c->patch_size, E, ctx->hid, ctx->ws_bytes));
    return 0;
fail:
    return rc;
}

int vit_hip_forward_device(vit_hip_ctx *ctx, const float *d_images, int n, float *d_logits,
                           float *d_probs, vh_stream_t stream)
{
    int rc = 0;
    if (!ctx || !d_images || n <= 0 || n > ctx->max_batch)
        return 1;
    TRY(vh_set_device(ctx->device));   /* the current device is per host thread */
    const vit_config *c = &ctx->cfg;
    const int E = c->embed_dim, T = ctx->tokens, NC = c->num_classes;
    vh_stream_t s = stream ? stream : ctx->stream;

    TRY(patch_embedding(ctx, s, d_images, n));
    int cls_rows = 0;   /* the last layer ran on the class-token rows only (opt-in, fp32 path on planes) */
    for (int l = 0; l < c->depth; ++l) {
        switch (ctx->precision) {
        case VIT_PRECISION_FP8_GEMM: TRY(layer_fp8(ctx, s, n, l)); break;
        case VIT_PRECISION_BF16_GEMM: TRY(layer_bf16(ctx, s, n, l)); break;
        case VIT_PRECISION_F32_FP16X2: TRY(layer_fp16x2(ctx, s, n, l)); break;
        default:
            if (ctx->use_p3)
                TRY(layer_f32_planes(ctx, s, n, l, &cls_rows));
            else
                TRY(layer_f32_rows(ctx, s, n, l));
        }
    }

    /* final LayerNorm on the class-token rows -- row i * stride of the residual stream, or compacted at the start of the
     * Q|K|V buffer -- classifier, softmax (ViT_seq.c:506-515) */
    float **tw = ctx->w + 4 + 12 * c->depth;
    float *logits = d_logits ? d_logits : ctx->d_logits;
    const float *final_x = cls_rows ? (const float *)ctx->qkv : ctx->x;
    const long final_stride = cls_rows ? (long)E : (long)T * E;
    OP(VIT_OP_LAYER_NORM, vh_launch_layer_norm(s, final_x, tw[0], tw[1], ctx->cls, n, E, final_stride, E, c->eps));
    OP(VIT_OP_HEAD, vh_launch_linear(s, logits, tw[2], ctx->cls, tw[3], n, E, NC, 0, NULL));
    if (d_probs)
        OP(VIT_OP_SOFTMAX, vh_launch_softmax(s, logits, d_probs, n, NC));
    return 0;
fail:
    return rc;
}

int vit_hip_set_last_layer_cls_only(vit_hip_ctx *ctx, int on)
{
    if (!ctx)
        return -1;
    const int before = ctx->cls_only_last;
    ctx->cls_only_last = on && ctx->use_p3 && !ctx->ln_fold;
    return before;
}

/* Debug/test hook: copy the residual stream ([n*tokens][E]) to the host. */
int vit_hip_read_tokens(vit_hip_ctx *ctx, int n, float *host_out)
{
    int rc = vh_set_device(ctx->device);
    if (rc)
        return rc;
    rc = vh_d2h(host_out, ctx->x, (size_t)n * ctx->tokens * ctx->cfg.embed_dim * sizeof(float),
                    ctx->stream);
    return rc ? rc : vh_stream_sync(ctx->stream);
}

int vit_hip_profile_enable(vit_hip_ctx *ctx, int max_forwards)
{
    int rc = 0;
    if (!ctx)
        return 1;
    TRY(vh_set_device(ctx->device));
    TRY(vh_stream_sync(ctx->stream));
    prof_release(ctx);
    if (max_forwards <= 0)
        return 0;
    const int per_forward = 1 + 7 * ctx->cfg.depth + 3;
    ctx->prof_cap = per_forward * max_forwards;
    ctx->prof_ev = (vh_event_t *)calloc((size_t)2 * ctx->prof_cap, sizeof(vh_event_t));
    ctx->prof_class = (int *)calloc((size_t)ctx->prof_cap, sizeof(int));
    if (!ctx->prof_ev || !ctx->prof_class) {
        prof_release(ctx);
        return 4;
    }
    for (int i = 0; i < 2 * ctx->prof_cap; ++i)
        TRY(vh_event_create(&ctx->prof_ev[i]));
    return 0;
fail:
    prof_release(ctx);
    return rc;
}

/* Restrict recording to the operator classes in `op_mask` (bit i = vit_op_class i; 0 = all): every
 * recorded launch costs two event packets between kernels, which a timed run may not want. */
int vit_hip_profile_select(vit_hip_ctx *ctx, unsigned op_mask)
{
    if (!ctx)
        return 1;
    ctx->prof_mask = op_mask;
    return 0;
}

int vit_hip_profile_read(vit_hip_ctx *ctx, double ms_sum[VIT_OP_COUNT], long launches[VIT_OP_COUNT])
{
    int rc = 0;
    if (!ctx || !ms_sum || !launches)
        return 1;
    TRY(vh_set_device(ctx->device));
    for (int k = 0; k < VIT_OP_COUNT; ++k) {
        ms_sum[k] = 0.0;
        launches[k] = 0;
    }
    for (int i = 0; i < ctx->prof_used; ++i) {
        float ms = 0.0f;
        TRY(vh_event_sync(ctx->prof_ev[2 * i + 1]));
        TRY(vh_event_elapsed_ms(&ms, ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]));
        ms_sum[ctx->prof_class[i]] += ms;
        launches[ctx->prof_class[i]] += 1;
    }
    ctx->prof_used = 0;
    return 0;
fail:
    return rc;
}

/* Host-pointer forward, software-pipelined over chunks of max_batch images:
 *   host     : gather chunk k into pinned slot k&1   | scatter outputs of chunk k-1
 *   copy strm: H2D chunk k                            (after compute of chunk k-2 released the slot)
 *   compute  : forward chunk k, D2H its logits/probs  (after the H2D)
 * so PCIe and the gather of the separately malloc'd images (Network.c:90) hide under
 * the previous chunk's kernels. */
static void scatter_outputs(vit_hip_ctx *ctx, int slot, int first, int m, float *logits, float **probs)
{
    const size_t NC = (size_t)ctx->cfg.num_classes;
    if (logits)
        memcpy(logits + (size_t)first * NC, ctx->h_logits[slot], (size_t)m * NC * sizeof(float));
    if (probs)
        for (int i = 0; i < m; ++i)
            memcpy(probs[first + i], ctx->h_probs[slot] + (size_t)i * NC, NC * sizeof(float));
}

/* Gather of the separately allocated host images (Network.c:90) into one pinned staging slot, on
 * several host threads: a single memcpy stream moves ~3 GB/s, which would cap the host-pointer path
 * below the device-resident rate. */
struct gather_job
{
    float *dst;
    const ImageData *images;
    int first, count;
    size_t img;
};

static void *gather_worker(void *arg)
{
    const struct gather_job *j = (const struct gather_job *)arg;
    for (int i = 0; i < j->count; ++i)
        memcpy(j->dst + (size_t)(j->first + i) * j->img, j->images[j->first + i].data, j->img * sizeof(float));
    return NULL;
}

static void gather_images(float *dst, const ImageData *images, int m, size_t img)
{
    enum { MAX_THREADS = 8 };
    int nt = m / 16;
    if (nt > MAX_THREADS)
        nt = MAX_THREADS;
    struct gather_job jobs[MAX_THREADS];
    pthread_t tid[MAX_THREADS];
    int started = 0;
    if (nt < 2) {
        struct gather_job all = {dst, images, 0, m, img};
        gather_worker(&all);
        return;
    }
    const int per = (m + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
        const int first = t * per, count = first >= m ? 0 : (m - first < per ? m - first : per);
        jobs[t] = (struct gather_job){dst, images, first, count, img};
        if (count == 0)
            break;
        if (t == nt - 1 || pthread_create(&tid[started], NULL, gather_worker, &jobs[t]) != 0)
            gather_worker(&jobs[t]);          /* the last share (or a failed create) runs here */
        else
            ++started;
    }
    for (int t = 0; t < started; ++t)
        pthread_join(tid[t], NULL);
}

int vit_hip_forward(vit_hip_ctx *ctx, const ImageData *images, int n, float *logits, float **probs)
{
    int rc = 0;
    if (!ctx || !images || n <= 0)
        return 1;
    TRY(vh_set_device(ctx->device));
    const vit_config *c = &ctx->cfg;
    const size_t img = (size_t)c->in_chans * c->img_size * c->img_size;
    const size_t NC = (size_t)c->num_classes;
    for (int i = 0; i < n; ++i)
        if (!images[i].data || images[i].c != c->in_chans || images[i].h != c->img_size ||
            images[i].w != c->img_size)
            return 5;

    int prev_first = 0, prev_m = 0, k = 0;
    for (int first = 0; first < n; first += ctx->max_batch, ++k) {
        const int m = (n - first < ctx->max_batch) ? n - first : ctx->max_batch;
        const int s = k & 1;
        /* slot s was last used by chunk k-2, whose outputs were waited for below */
        gather_images(ctx->h_images[s], images + first, m, img);
        if (k >= 2)
            TRY(vh_stream_wait_event(ctx->copy_stream, ctx->comp_done[s]));
        TRY(vh_h2d(ctx->d_images[s], ctx->h_images[s], (size_t)m * img * sizeof(float), ctx->copy_stream));
        TRY(vh_event_record(ctx->up_done[s], ctx->copy_stream));

        TRY(vh_stream_wait_event(ctx->stream, ctx->up_done[s]));
        TRY(vit_hip_forward_device(ctx, ctx->d_images[s], m, ctx->d_logits, probs ? ctx->d_probs : NULL,
                                   ctx->stream));
        TRY(vh_event_record(ctx->comp_done[s], ctx->stream));
        if (logits)
            TRY(vh_d2h(ctx->h_logits[s], ctx->d_logits, (size_t)m * NC * sizeof(float), ctx->stream));
        if (probs)
            TRY(vh_d2h(ctx->h_probs[s], ctx->d_probs, (size_t)m * NC * sizeof(float), ctx->stream));
        TRY(vh_event_record(ctx->out_done[s], ctx->stream));

        if (k >= 1) { /* finish chunk k-1 while chunk k runs */
            TRY(vh_event_sync(ctx->out_done[s ^ 1]));
            scatter_outputs(ctx, s ^ 1, prev_first, prev_m, logits, probs);
        }
        prev_first = first;
        prev_m = m;
    }
    TRY(vh_event_sync(ctx->out_done[(k - 1) & 1]));
    scatter_outputs(ctx, (k - 1) & 1, prev_first, prev_m, logits, probs);
    return 0;
fail:
    vh_stream_sync(ctx->copy_stream);
    vh_stream_sync(ctx->stream);
    return rc;
}

/* ---- several GPUs behind one call -------------------------------------------------
 * Images never interact (the reference processes them strictly one at a time, ViT_opencl.c:926), so
 * the batch is cut into contiguous shards, one per device; every device holds a full replica of the
 * weights (346 MB for ViT-B/16) and is driven by its own host thread, context and stream.  A thread
 * writes its shard's outputs straight into the caller's arrays, so inside one process the "gather of
 * the class logits" is this scatter -- no collective.  (bench.py's one-process-per-GPU form gathers
 * with RCCL instead.) */

void vit_shard_range(int total, int shard, int n_shards, int *lo, int *hi)
{
    const int per = n_shards > 0 ? (total + n_shards - 1) / n_shards : total;
    int a = shard * per, b;
    if (a > total)
        a = total;
    b = a + per;
    if (b > total)
        b = total;
    *lo = a;
    *hi = b;
}

struct shard_job
{
    int (*fn)(void *arg, int shard, int lo, int hi);
    void *arg;
    int shard, lo, hi, rc;
    double ms;          /* wall time of fn on its thread */
    char err[256];
};

static void *shard_worker(void *p)
{
    struct shard_job *j = (struct shard_job *)p;
    const double t0 = wall_seconds();
    j->rc = j->fn(j->arg, j->shard, j->lo, j->hi);
    j->ms = 1e3 * (wall_seconds() - t0);
    if (j->rc != 0)
        snprintf(j->err, sizeof(j->err), "%s", vh_last_error());   /* the error text is per thread */
    return NULL;
}

/* Run fn(arg, s, lo_s, hi_s) for the n_shards contiguous shards of [0, total), each non-empty shard
 * on its own host thread (shard 0 on the caller's); returns 0 or the first failing shard's status. */
int vit_shard_run(int total, int n_shards, int (*fn)(void *arg, int shard, int lo, int hi), void *arg)
{
    return vit_shard_run_timed(total, n_shards, fn, arg, NULL);
}

/* The same; ms_per_shard[s] (may be NULL) receives the wall time shard s's fn took on its thread (0 for an empty shard). */
int vit_shard_run_timed(int total, int n_shards, int (*fn)(void *arg, int shard, int lo, int hi), void *arg, double *ms_per_shard)
{
    enum { MAX_SHARDS = 64 };
    if (total < 0 || n_shards <= 0 || n_shards > MAX_SHARDS || !fn)
        return 1;
    struct shard_job jobs[MAX_SHARDS];
    pthread_t tid[MAX_SHARDS];
    int threaded[MAX_SHARDS];
    for (int s = 0; s < n_shards; ++s) {
        jobs[s] = (struct shard_job){fn, arg, s, 0, 0, 0, 0.0, ""};
        vit_shard_range(total, s, n_shards, &jobs[s].lo, &jobs[s].hi);
        threaded[s] = 0;
    }
    for (int s = 1; s < n_shards; ++s)
        if (jobs[s].hi > jobs[s].lo)
            threaded[s] = pthread_create(&tid[s], NULL, shard_worker, &jobs[s]) == 0;
    for (int s = 0; s < n_shards; ++s)
        if (!threaded[s] && jobs[s].hi > jobs[s].lo)
            shard_worker(&jobs[s]);               /* shard 0, and any shard whose thread could not start */
    for (int s = 1; s < n_shards; ++s)
        if (threaded[s])
            pthread_join(tid[s], NULL);
    if (ms_per_shard)
        for (int s = 0; s < n_shards; ++s)
            ms_per_shard[s] = jobs[s].ms;
    for (int s = 0; s < n_shards; ++s)
        if (jobs[s].rc != 0) {
            fprintf(stderr, "vit_shard_run: shard %d [%d, %d) failed with status %d: %s\n", s, jobs[s].lo, jobs[s].hi,
                    jobs[s].rc, jobs[s].err);
            return jobs[s].rc;
        }
    return 0;
}

struct vit_hip_multi
{
    int n_devices;
    vit_hip_ctx **ctx;
    /* creation arguments, read by the per-device threads */
    const vit_config *cfg;
    const Network *networks;
    const int *devices;
    int n_tensors, max_batch, precision;
    /* forward arguments */
    const ImageData *images;
    float *logits;
    float **probs;
    void *gather;   /* RCCL communicators of vit_hip_forward_device_multi (vit_gather_rccl.c), made on first use */
    double enqueue_ms[64];   /* vit_hip_forward_device_multi: host time each device's thread spent enqueuing its shard, last call */
};

void vit_gather_release(void *state);
void **vit_hip_multi_gather_slot(vit_hip_multi *m) { return &m->gather; }
double *vit_hip_multi_enqueue_ms_slot(vit_hip_multi *m) { return m->enqueue_ms; }

int vit_hip_multi_last_enqueue_ms(const vit_hip_multi *m, double *ms, int capacity)
{
    if (!m || !ms || capacity < m->n_devices)
        return -1;
    for (int d = 0; d < m->n_devices; ++d)
        ms[d] = m->enqueue_ms[d];
    return m->n_devices;
}

static int multi_create_one(void *arg, int shard, int lo, int hi)
{
    vit_hip_multi *m = (vit_hip_multi *)arg;
    (void)lo;
    (void)hi;
    return vit_hip_create_ex(&m->ctx[shard], m->cfg, m->networks, m->n_tensors, m->devices[shard], m->max_batch,
                             m->precision);
}

int vit_hip_create_multi(vit_hip_multi **out, const vit_config *cfg, const Network *networks, int n_tensors,
                         const int *devices, int n_devices, int max_batch_per_device, int precision)
{
    if (!out || !cfg || !networks || !devices || n_devices <= 0 || n_devices > 64 || max_batch_per_device <= 0)
        return 1;
    *out = NULL;
    vit_hip_multi *m = (vit_hip_multi *)calloc(1, sizeof(*m));
    if (!m)
        return 4;
    m->ctx = (vit_hip_ctx **)calloc((size_t)n_devices, sizeof(*m->ctx));
    int *devs = (int *)malloc(sizeof(int) * (size_t)n_devices);
    if (!m->ctx || !devs) {
        free(m->ctx);
        free(devs);
        free(m);
        return 4;
    }
    memcpy(devs, devices, sizeof(int) * (size_t)n_devices);
    m->n_devices = n_devices;
    m->cfg = cfg;
    m->networks = networks;
    m->devices = devs;
    m->n_tensors = n_tensors;
    m->max_batch = max_batch_per_device;
    m->precision = precision;
    /* one "shard" per device: the replicas are built side by side (weight upload + repack each) */
    const int rc = vit_shard_run(n_devices, n_devices, multi_create_one, m);
    m->cfg = NULL;
    m->networks = NULL;
    if (rc != 0) {
        vit_hip_destroy_multi(m);
        return rc;
    }
    *out = m;
    return 0;
}

void vit_hip_destroy_multi(vit_hip_multi *m)
{
    if (!m)
        return;
    vit_gather_release(m->gather);
    for (int d = 0; d < m->n_devices; ++d)
        vit_hip_destroy(m->ctx[d]);
    free((void *)m->devices);
    free(m->ctx);
    free(m);
}

int vit_hip_multi_devices(const vit_hip_multi *m) { return m ? m->n_devices : 0; }
vit_hip_ctx *vit_hip_multi_ctx(const vit_hip_multi *m, int i) { return (m && i >= 0 && i < m->n_devices) ? m->ctx[i] : NULL; }

static int multi_forward_one(void *arg, int shard, int lo, int hi)
{
    vit_hip_multi *m = (vit_hip_multi *)arg;
    const size_t NC = (size_t)vit_hip_config(m->ctx[shard])->num_classes;
    return vit_hip_forward(m->ctx[shard], m->images + lo, hi - lo, m->logits ? m->logits + (size_t)lo * NC : NULL,
                           m->probs ? m->probs + lo : NULL);
}

/* Not re-entrant on one vit_hip_multi (like vit_hip_forward on one context). */
int vit_hip_forward_multi(vit_hip_multi *m, const ImageData *images, int n, float *logits, float **probs)
{
    if (!m || !images || n <= 0)
        return 1;
    m->images = images;
    m->logits = logits;
    m->probs = probs;
    return vit_shard_run(n, m->n_devices, multi_forward_one, m);
}

/* $VIT_HIP_DEVICES: "all", or a comma-separated list of device ids; returns the count written. */
static int parse_devices(const char *env, int *out, int capacity)
{
    int n = 0;
    if (!env || !*env)
        return 0;
    if (strcmp(env, "all") == 0) {
        const int have = vh_device_count();
        for (int d = 0; d < have && n < capacity; ++d)
            out[n++] = d;
        return n;
    }
    for (const char *p = env; *p && n < capacity;) {
        char *end = NULL;
        const long v = strtol(p, &end, 10);
        if (end == p)
            break;
        out[n++] = (int)v;
        p = (*end == ',') ? end + 1 : end;
        if (*end != ',' && *end != '\0')
            break;
    }
    return n;
}

/* Wall-clock split of the calling thread's last ViT_opencl(): context creation (the reference's "setup time",
 * ViT_opencl.c:910) and everything after it up to the return (forward + teardown). */
static _Thread_local double last_setup_s, last_forward_s;
void vit_hip_last_call_seconds(double *setup_s, double *forward_s)
{
    if (setup_s)
        *setup_s = last_setup_s;
    if (forward_s)
        *forward_s = last_forward_s;
}

/* The drop-in entry point (reference ViT_opencl.c:794).  Same observable
 * behaviour: fills probabilities[i][0..999]; prints a setup-time line and a
 * throughput line where the reference prints "setup time" / "picture #i". */
void ViT_opencl(ImageData *image, Network *networks, float **probabilities)
{
    if (!image || !networks || !probabilities) {
        printf("[%s:%d] ViT_opencl: NULL argument\n", __FILE__, __LINE__);
        exit(EXIT_FAILURE);
    }
    const double t0 = wall_seconds();
    vit_config cfg;
    vit_config_preset(&cfg, "vit_b_16");
    const int n = image[0].n;
    if (n <= 0) {   /* the reference's per-image loop simply does not run (ViT_opencl.c:926); no device is touched */
        printf("setup time: 0.000000 sec (no images)\n\n");
        last_setup_s = last_forward_s = 0.0;
        return;
    }
    int device = 0;
    const char *env = getenv("VIT_HIP_DEVICE");
    if (env && *env)
        device = atoi(env);
    int devices[64];
    const int n_devices = parse_devices(getenv("VIT_HIP_DEVICES"), devices, 64);
    const int per_device = n_devices > 1 ? (n + n_devices - 1) / n_devices : n;
    int chunk = per_device < 512 ? per_device : 512;
    const char *envb = getenv("VIT_HIP_MAX_BATCH");
    if (envb && atoi(envb) > 0)
        chunk = atoi(envb) < per_device ? atoi(envb) : per_device;
    if (n_devices > 1) {
        /* $VIT_HIP_DEVICES names several GPUs: contiguous shards of the images, one replica per device */
        const int precision = env_precision();
        vit_hip_multi *m = NULL;
        int rcm = vit_hip_create_multi(&m, &cfg, networks, vit_config_num_tensors(&cfg), devices, n_devices, chunk, precision);
        if (rcm != 0) {
            printf("[%s:%d] vit_hip_create_multi failed (%d): %s\n", __FILE__, __LINE__, rcm, vh_last_error());
            exit(EXIT_FAILURE);
        }
        const double t1m = wall_seconds();
        printf("setup time: %.6f sec (%d devices)\n\n", t1m - t0, n_devices);
        VH_CHECK(vit_hip_forward_multi(m, image, n, NULL, probabilities));
        const double t2m = wall_seconds();
        printf("pictures #0..#%d: %.6f sec (%.1f images/sec)\n\n", n - 1, t2m - t1m,
               (double)n / (t2m - t1m > 0 ? t2m - t1m : 1e-9));
        vit_hip_destroy_multi(m);
        last_setup_s = t1m - t0;
        last_forward_s = wall_seconds() - t1m;
        return;
    }
    if (n_devices == 1)
        device = devices[0];

    vit_hip_ctx *ctx = NULL;
    int rc = vit_hip_create(&ctx, &cfg, networks, vit_config_num_tensors(&cfg), device, chunk);
    if (rc != 0) {
        printf("[%s:%d] vit_hip_create failed (%d): %s\n", __FILE__, __LINE__, rc, vh_last_error());
        exit(EXIT_FAILURE);
    }
    const double t1 = wall_seconds();
    printf("setup time: %.6f sec (%s)\n\n", t1 - t0, vh_device_name());
    VH_CHECK(vit_hip_forward(ctx, image, n, NULL, probabilities));
    const double t2 = wall_seconds();
    printf("pictures #0..#%d: %.6f sec (%.1f images/sec)\n\n", n - 1, t2 - t1,
           (double)n / (t2 - t1 > 0 ? t2 - t1 : 1e-9));
    vit_hip_destroy(ctx);
    last_setup_s = t1 - t0;
    last_forward_s = wall_seconds() - t1;
}
