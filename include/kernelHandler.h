/*
 * kernelHandler.h -- C ABI of the HIP device shim (libvit_hip.so).
 *
 * This is the replacement for the reference's OpenCL runtime glue
 * (MulticoreMainProject/kernelHandler.h:6-18, kernelHandler.c:15,35) plus the
 * clSetKernelArg/clEnqueueNDRangeKernel wrappers in ViT_opencl.c:361-779.
 * The reference loads .cl text at run time (get_source_code) and JIT-builds it
 * (build_error prints the log); here every kernel is compiled ahead of time
 * for gfx950 and reached through one `vh_launch_*` entry per kernel.
 *
 * Everything is extern "C" with plain pointers and sizes: host code stays C.
 * Device pointers are ordinary `float *` values that must not be dereferenced
 * on the host.  All launchers are asynchronous on `stream` (0 = the null
 * stream) and return 0 on success or a non-zero hipError_t value; the text of
 * the most recent failure on the calling thread is at vh_last_error().
 *
 * There is no CPU fallback anywhere behind this header: without a usable
 * gfx950 device vh_init() fails and every launcher returns an error.
 */
#ifndef VIT_HIP_KERNELHANDLER_H
#define VIT_HIP_KERNELHANDLER_H

#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *vh_stream_t; /* hipStream_t */
typedef void *vh_event_t;  /* hipEvent_t  */

/* Mirrors CHECK_ERROR (reference kernelHandler.h:6-10): print where, then
 * exit(EXIT_FAILURE).  Used by the drop-in ViT_opencl(), which has no way to
 * return a status; the extended vit_hip_* API returns the code instead. */
#define VH_CHECK(err)                                                          \
    do {                                                                       \
        int vh_check_err_ = (err);                                             \
        if (vh_check_err_ != 0) {                                              \
            printf("[%s:%d] HIP error %d: %s\n", __FILE__, __LINE__,           \
                   vh_check_err_, vh_last_error());                            \
            exit(EXIT_FAILURE);                                                \
        }                                                                      \
    } while (0)

/* ---- runtime (replaces clGetPlatformIDs..clCreateCommandQueue, ViT_opencl.c:799-861) ---- */
int vh_device_count(void);                 /* number of visible HIP devices (0 if none / no driver) */
int vh_init(int device);                   /* select `device`; fails unless it is a gfx950 part */
const char *vh_last_error(void);           /* text of the last failure on this thread ("" if none) */
int vh_set_error(int code, const char *message);   /* record `message` as this thread's last error; returns code (1 if 0) */
const char *vh_device_name(void);          /* e.g. "AMD Instinct MI355X (gfx950, 256 CUs)" */
int vh_set_device(int device);             /* make `device` current for the calling thread (after vh_init) */

int vh_stream_create(vh_stream_t *out);
int vh_stream_destroy(vh_stream_t s);
int vh_stream_sync(vh_stream_t s);         /* replaces clFinish */
int vh_device_sync(void);

int vh_event_create(vh_event_t *out);
int vh_event_destroy(vh_event_t e);
int vh_event_record(vh_event_t e, vh_stream_t s);
int vh_stream_wait_event(vh_stream_t s, vh_event_t e); /* work queued on s after this waits for e */
int vh_event_sync(vh_event_t e);
int vh_event_elapsed_ms(float *ms, vh_event_t start, vh_event_t stop);

/* ---- memory (replaces clCreateBuffer / clEnqueueWriteBuffer / clEnqueueReadBuffer,
 *      ViT_opencl.c:125-357, :371, :775) ---- */
int vh_malloc(void **out, size_t bytes);
int vh_free(void *p);
int vh_host_alloc(void **out, size_t bytes); /* pinned host memory for async copies */
int vh_host_free(void *p);
int vh_memset(void *dst, int value, size_t bytes, vh_stream_t s);
int vh_h2d(void *dst, const void *src, size_t bytes, vh_stream_t s); /* async w.r.t. host iff src is pinned */
int vh_d2h(void *dst, const void *src, size_t bytes, vh_stream_t s);
int vh_d2d(void *dst, const void *src, size_t bytes, vh_stream_t s);

/* ---- kernels: one launcher per HIP kernel ---- */

/* Patch embedding, fused with flatten/transpose, class-token prepend and
 * position-embedding add.  Replaces Conv2d + postConv2d (ViT_opencl.c:361-442;
 * kernels conv2d_kernel conv2d.cl:1 and postprocess conv2d.cl:39; CPU
 * Conv2d_seq..pos_emb_seq ViT_seq.c:25-118).
 *   images   [n_images][in_chans][img][img]  fp32, contiguous
 *   conv_w   [embed][in_chans][patch][patch], conv_b [embed]
 *   cls      [embed], pos [tokens][embed]   (tokens = (img/patch)^2 + 1)
 *   tokens   [n_images][tokens][embed]      output */
int vh_launch_patch_embed(vh_stream_t s, const float *images, const float *conv_w,
                          const float *conv_b, const float *cls_token, const float *pos_embed,
                          float *tokens, int n_images, int in_chans, int img_size,
                          int patch_size, int embed_dim);

/* The same for patch geometries whose rows cannot be gathered on load (patch % 4 != 0 or
 * in_chans*patch*patch % 32 != 0 -- ViT-H/14: 3*14*14 = 588): patches are gathered once
 * into zero-padded rows in `workspace` (vh_patch_embed_workspace() bytes, 16-byte aligned;
 * 0 for geometries vh_launch_patch_embed takes directly, which then ignores the workspace). */
size_t vh_patch_embed_workspace(int n_images, int in_chans, int img_size, int patch_size, int embed_dim);
int vh_launch_patch_embed_ws(vh_stream_t s, const float *images, const float *conv_w,
                             const float *conv_b, const float *cls_token, const float *pos_embed,
                             float *tokens, int n_images, int in_chans, int img_size,
                             int patch_size, int embed_dim, void *workspace, size_t workspace_bytes);

/* The patch embedding of the reduced-precision modes (bf16 / fp8 GEMM operands; conv2d.cl:1-80 for them): an im2row
 * producer writes the patches as one-part bf16 planes [Kp/32][1][n_images*grid^2][32] into `workspace` (Kp =
 * vh_patch_planes_k(): in_chans*patch^2 padded with zeros to the one-part K step; Kp * 2 bytes per patch row), and
 * the planes GEMM (vh_launch_linear_planes, parts = 1) runs with a token-row epilogue.  `conv_w_planes` =
 * [Kp/32][1][embed][32], written once by vh_launch_conv_weight_planes.  fp32 accumulation, bias, position
 * embedding and class-token rows as vh_launch_patch_embed; pixels and weights are rounded to bf16.  embed % 128 == 0. */
int vh_patch_planes_k(int in_chans, int patch_size);
int vh_launch_conv_weight_planes(vh_stream_t s, const float *conv_w, void *planes, int embed_dim, int in_chans, int patch_size);
/* ... with `parts` parts per value: 1 as above, 3 = the exact fp32 split [Kp/32][3][embed][32] for vh_launch_patch_embed_planes3 */
int vh_launch_conv_weight_planes_parts(vh_stream_t s, const float *conv_w, void *planes, int embed_dim, int in_chans, int patch_size,
                                       int parts);
/* The fp32 path's patch embedding on the planes kernel: the im2row producer writes the exact three-part split of the pixels
 * (workspace: n_images*grid^2 * Kp * 6 bytes), six bf16 products per block as vh_launch_linear_p3 -- no operand is split inside
 * a K loop.  Same arguments as vh_launch_patch_embed_planes. */
int vh_launch_patch_embed_planes3(vh_stream_t s, const float *images, const void *conv_w_planes3, const float *conv_b,
                                  const float *cls_token, const float *pos_embed, float *tokens, int n_images, int in_chans,
                                  int img_size, int patch_size, int embed_dim, void *workspace, size_t workspace_bytes);
int vh_launch_patch_embed_planes(vh_stream_t s, const float *images, const void *conv_w_planes, const float *conv_b,
                                 const float *cls_token, const float *pos_embed, float *tokens, int n_images,
                                 int in_chans, int img_size, int patch_size, int embed_dim, void *workspace,
                                 size_t workspace_bytes);

/* Row LayerNorm, y = (x-mean)*inv_std*w + b with var = E[x^2]-mean^2 and
 * inv_std = 1/sqrt(var+eps) (eps added in double, ViT_seq.c:21,135).
 * Replaces layer_norm (ViT_opencl.c:444-482; layerNorm layer_norm.cl:3; CPU
 * layer_norm_seq ViT_seq.c:120).  Row r is read at input + r*in_row_stride and
 * written at output + r*out_row_stride (strides in floats), so the final
 * norm can run on the class-token rows only. */
int vh_launch_layer_norm(vh_stream_t s, const float *input, const float *weight,
                         const float *bias, float *output, int rows, int embed_dim,
                         long in_row_stride, long out_row_stride, double eps);

/* output[rowA][colB] = input[rowA][colA] . weight[colB][colA]^T + bias[colB],
 * optionally followed by exact-erf GELU, optionally adding `residual`
 * (same shape as output; may alias output).  Argument order and names follow
 * the reference kernel linear_layer (ll.cl:7-16) and its host wrapper
 * (ViT_opencl.c:622-672); it also replaces QKV (multihead.cl:3, colB = 3*embed,
 * output rows are Q|K|V side by side) and encoderResidual (layer_norm.cl:55).
 * CPU: linear_layer_seq ViT_seq.c:295, gelu :283, residual loops :348,:360. */
int vh_launch_linear(vh_stream_t s, float *output, const float *weight, const float *input,
                     const float *bias, int rowA, int colA, int colB, int doGelu,
                     const float *residual);

/* The same product with the weight pre-split: `weight_planes` holds the exact three-way split
 * w = p0 + p1 + p2 as bfloat16, laid out [colA/32][3][colB][32] (K step, part, row, element: what one K
 * step of a tile reads is contiguous), written once by vh_launch_split3_planes (the fp32 GEMM forms
 * every product from such parts on the bf16 matrix cores; splitting the constant operand ahead
 * of time leaves only the activations to split in the inner loop).  Same results as
 * vh_launch_linear bit for bit.  colB % 128 == 0. */
int vh_launch_split3_planes(vh_stream_t s, const float *weight, void *planes, int rows, int cols);
int vh_launch_linear_w3(vh_stream_t s, float *output, const void *weight_planes, const float *input,
                        const float *bias, int rowA, int colA, int colB, int doGelu,
                        const float *residual);

/* ---- both operands pre-split ("P3"): the default fp32 path of the four big projections ----
 * The exact three-way split x = p0 + p1 + p2 of an fp32 matrix [rows][cols], stored as bfloat16 planes
 *     planes[cols/32][3][rows][32]      (K step, part, row, element; 6 bytes per value)
 * -- the layout vh_launch_split3_planes gives the weights.  When the PRODUCER of a GEMM input writes
 * this format (vh_launch_layer_norm_p3, vh_launch_attention_p3, vh_launch_linear_p3 with output_planes),
 * the GEMM's K loop contains no split arithmetic at all: matrix instructions, LDS reads and loads only
 * (csrc/gemm_p3.hip).  Same six partial products per block, in the same order, as vh_launch_linear /
 * vh_launch_linear_w3: results are identical bit for bit.  No reference counterpart (format plumbing
 * around ll.cl:7 / multihead.cl:3 / layer_norm.cl:3); rows < 2^26, cols % 32 == 0, 16-byte aligned. */
int vh_launch_split3_rows(vh_stream_t s, const float *input, void *planes, int rows, int cols);
int vh_launch_merge3_rows(vh_stream_t s, const void *planes, float *output, int rows, int cols); /* exact inverse */
/* The same format with `parts` parts per value: 3 as above, or 1 = values rounded to bfloat16 (the operands of
 * the bf16-operand mode, BASELINE config 3), planes[cols/32][1][rows][32]: the same GEMM kernel takes both. */
int vh_launch_split_rows(vh_stream_t s, const float *input, void *planes, int rows, int cols, int parts);
int vh_launch_merge_rows(vh_stream_t s, const void *planes, float *output, int rows, int cols, int parts);
int vh_launch_layer_norm_planes(vh_stream_t s, const float *input, const float *weight, const float *bias,
                                void *out_planes, int parts, int rows, int embed_dim, long in_row_stride, double eps);
int vh_launch_attention_planes_bf16(vh_stream_t s, const float *qkv, void *out_planes, int n_images, int tokens,
                                    int embed_dim, int num_heads);   /* one-part planes, arithmetic of vh_launch_attention_f16 */
/* colA % 64 == 0 (parts 3) or % 128 == 0 (parts 1); fp32 accumulation, bias, GELU, residual as vh_launch_linear.
 * output_planes: 0 = fp32 rows [rowA][colB]; 1 = planes of `parts` parts (no residual); 2 (parts 1, no GELU, no
 * residual) = one-part planes of FP16 values [colB/32][rowA][32] -- the Q|K|V input of vh_launch_attention_planes_f16 */
int vh_launch_linear_planes(vh_stream_t s, void *output, int output_planes, const void *weight_planes,
                            const void *input_planes, int parts, const float *bias, int rowA, int colA, int colB,
                            int doGelu, const float *residual);
/* vh_launch_layer_norm writing planes [embed_dim/32][3][rows][32] */
int vh_launch_layer_norm_p3(vh_stream_t s, const float *input, const float *weight, const float *bias,
                            void *out_planes, int rows, int embed_dim, long in_row_stride, double eps);
/* vh_launch_attention writing planes [embed_dim/32][3][n_images*tokens][32] (head_dim 64, tokens <= 208) */
int vh_launch_attention_p3(vh_stream_t s, const float *qkv, void *out_planes, int n_images, int tokens,
                           int embed_dim, int num_heads);
/* The same attention on PRE-SPLIT Q, K, V: qkv_planes [3*embed_dim/32][3][n_images*tokens][32] (the QKV projection
 * written by vh_launch_linear_p3 with output_planes), out_planes as above; same results bit for bit, no split
 * arithmetic left but that of the probabilities (csrc/attention_p3.hip).  head_dim 64, tokens <= 208. */
int vh_launch_attention_planes(vh_stream_t s, const void *qkv_planes, void *out_planes, int n_images, int tokens,
                               int embed_dim, int num_heads);
/* The reduced-precision modes' attention on planes: qkv_planes_f16 [3*embed_dim/32][n_images*tokens][32] fp16 (the QKV
 * projection with output_planes = 2, or vh_launch_linear_mx_planes_f16); output = one-part bf16 planes
 * [embed_dim/32][rows][32] (output_planes != 0) or fp32 rows [rows][embed_dim].  Arithmetic of vh_launch_attention_f16
 * (same results bit for bit on the same fp16 values); head_dim 64, tokens <= 208. */
int vh_launch_attention_planes_f16(vh_stream_t s, const void *qkv_planes_f16, void *output, int output_planes,
                                   int n_images, int tokens, int embed_dim, int num_heads);
/* The same arithmetic for ViT-H/14's shape (head_dim 80, tokens <= 272) with one head's K and V resident in LDS
 * (csrc/attention_h16.hip): qkv_planes_f16 as above -> fp32 rows [n_images*tokens][embed_dim]. */
int vh_launch_attention_planes_f16_hd80(vh_stream_t s, const void *qkv_planes_f16, float *output, int n_images,
                                        int tokens, int embed_dim, int num_heads);
/* ... writing the output projection's operand directly: output_kind 1 = one-part bf16 planes [embed_dim/32][rows][32], 2 =
 * the block-scaled fp8 tensor (values + out_scales); the fp32 result followed by vh_launch_split_rows(parts = 1) /
 * vh_launch_quantize_mx_rows, byte for byte.  head_dim 80, num_heads even, tokens <= 272. */
int vh_launch_attention_planes_f16_hd80_operand(vh_stream_t s, const void *qkv_planes_f16, void *output, void *out_scales,
                                                int output_kind, int n_images, int tokens, int embed_dim, int num_heads);
/* vh_launch_linear on planes: input_planes [colA/32][3][rowA][32], weight_planes [colA/32][3][colB][32];
 * output fp32 [rowA][colB], or (output_planes != 0, no residual) planes [colB/32][3][rowA][32].
 * colA % 64 == 0, colB % 128 == 0. */
int vh_launch_linear_p3(vh_stream_t s, void *output, int output_planes, const void *weight_planes,
                        const void *input_planes, const float *bias, int rowA, int colA, int colB,
                        int doGelu, const float *residual);

/* fp32 emulation with two fp16 parts per operand and three products (the "3 x TF32" scheme on
 * fp16: a = a0 + a1 + eps, |eps| <= 2^-22 |a|; a.w ~ a0w0 + a0w1 + a1w0, fp32 accumulation).  NOT
 * exact -- operands keep 22 of their 24 significant bits -- but its truncation error (~8e-8 of the
 * result) is an order of magnitude below the rounding noise of an fp32 accumulation, at half
 * the matrix-core work of the exact split.  `weight_planes` = [colA/32][2][colB][32] fp16 of
 * weight * weight_scale (a power of two that lifts the low part out of fp16's subnormal range;
 * vh_launch_split2h_planes); inputs must stay below 65504 in magnitude.  Opt-in (ViT_opencl.h). */
int vh_launch_split2h_planes(vh_stream_t s, const float *weight, void *planes, int rows, int cols, float scale);
int vh_launch_linear_h2(vh_stream_t s, float *output, const void *weight_planes, float weight_scale,
                        const float *input, const float *bias, int rowA, int colA, int colB,
                        int doGelu, const float *residual);
int vh_launch_attention_h2(vh_stream_t s, const float *qkv, float *output, int n_images, int tokens,
                           int embed_dim, int num_heads);   /* Q.K^T and P.V likewise */

/* Scaled-dot-product attention over the fused QKV rows produced by
 * vh_launch_linear (row = Q[embed] | K[embed] | V[embed]); per (image, head):
 * softmax(Q K^T / sqrt(head_dim)) V, heads concatenated.  Replaces
 * QKV_TO_SCOREV (multihead.cl:65-137, ViT_opencl.c:539-565); CPU
 * multihead_attn_seq ViT_seq.c:192-262.
 *   qkv    [n_images*tokens][3*embed],  output [n_images*tokens][embed] */
int vh_launch_attention(vh_stream_t s, const float *qkv, float *output, int n_images,
                        int tokens, int embed_dim, int num_heads);

/* Every row_stride-th row compacted: dst[p][i][:] = src[p][i * row_stride][:] over n_planes planes of src_rows rows of
 * row_bytes bytes (a row-major fp32 matrix: n_planes 1, row_bytes 4 * cols; a planes tensor: row_bytes 64).  Used to
 * pull out the class-token rows (ViT_seq.c:511 reads row 0 of every image only).  No reference counterpart. */
int vh_launch_gather_rows(vh_stream_t s, const void *src, void *dst, int n_planes, int src_rows, int dst_rows,
                          int row_bytes, int row_stride);

/* Row softmax with max subtraction: output[r][i] = exp(x-max)/sum.  Replaces
 * Softmax (ViT_opencl.c:750-779; softMax miniSoftMax.cl:1); CPU Softmax_seq
 * ViT_seq.c:372. */
int vh_launch_softmax(vh_stream_t s, const float *input, float *output, int rows, int length);

/* ---- reduced-precision attention (the bf16- and fp8-operand modes, BASELINE configs 3 and 5) ----
 * fp32 in, fp32 out; Q, K, V and P rounded to fp16 for the two products (11-bit operands, fp32 accumulation and
 * softmax): far inside those modes' tolerances.  The bf16-operand mode itself runs on one-part planes:
 * vh_launch_layer_norm_planes / vh_launch_attention_planes_bf16 / vh_launch_linear_planes with parts = 1 (above). */
int vh_launch_attention_f16(vh_stream_t s, const float *qkv, float *output, int n_images,
                            int tokens, int embed_dim, int num_heads);

int vh_launch_absmax(vh_stream_t s, const float *input, size_t count, float *amax); /* atomic max |x| into *amax (fp16-pair mode: weight ranges) */

/* ---- block-scaled fp8 ("MX": OCP microscaling) operand variants -- BASELINE config 5 at the fp8 matrix rate ----
 * No reference counterpart.  An MX tensor [rows][cols] (cols % 128 == 0) is e4m3 elements with one e8m0
 * power-of-two scale per 32 consecutive elements of a row (scale = the smallest 2^E with max|block| / 2^E <= 448,
 * elements = e4m3_nearest_even(x / scale)), stored as
 *     values[cols/128][rows][128] bytes      scales[cols/128][4][rows] bytes
 * where scales[k][g][r] is the scale of block 2 (g & 1) + (g >> 1) of row r's K step k: the order in which the four
 * lane groups of v_mfma_scale_f32_16x16x128_f8f6f4 consume them (csrc/gemm_mx.hip; tests/mx_ref.py is the numpy
 * statement both are checked against byte for byte).  Scales are per block and computed where the tensor is
 * produced: no calibration pass. */
int vh_launch_quantize_mx_rows(vh_stream_t s, const float *input, void *values, void *scales, int rows, int cols);
/* An ACTIVATION tensor [rows][cols] as MX: values as above, scale bytes as [ceil(cols/512)][4][rows][4] -- K steps in
 * groups of four, lane group, row, K step within the group -- so that the GEMM's lane takes its scales of four K steps
 * with one dword load (vh_mx_act_scale_bytes(rows, cols) bytes, 4-byte aligned).  Every producer of a GEMM input writes
 * this form (vh_launch_layer_norm_mx, the fc1 / attention epilogues, the folded-LayerNorm producers); vh_launch_linear_mx*
 * take it for `input_scales`.  Weights keep [cols/128][4][rows] (vh_launch_quantize_mx_rows). */
int vh_launch_quantize_mx_act(vh_stream_t s, const float *input, void *values, void *scales, int rows, int cols);
size_t vh_mx_act_scale_bytes(int rows, int cols);
/* vh_launch_layer_norm writing its result as an MX tensor (embed_dim % 128 == 0) */
int vh_launch_layer_norm_mx(vh_stream_t s, const float *input, const float *weight, const float *bias,
                            void *out_values, void *out_scales, int rows, int embed_dim, long in_row_stride, double eps);
/* output (fp32 [rowA][colB]; or MX values + output_scales when output_scales != NULL, no residual) =
 *   input_mx . weight_mx^T + bias [+GELU | +residual]; fp32 accumulation.  colA % 256 == 0, colB % 128 == 0. */
int vh_launch_linear_mx(vh_stream_t s, void *output, void *output_scales, const void *weight_values,
                        const void *weight_scales, const void *input_values, const void *input_scales,
                        const float *bias, int rowA, int colA, int colB, int doGelu, const float *residual);
/* vh_launch_attention_planes_f16 writing an MX tensor (= vh_launch_quantize_mx_rows of its fp32 result, byte for byte) */
int vh_launch_attention_planes_f16_mx(vh_stream_t s, const void *qkv_planes_f16, void *out_values, void *out_scales,
                                      int n_images, int tokens, int embed_dim, int num_heads);
/* the same product (no GELU, no residual) written as one-part fp16 planes [colB/32][rowA][32] */
int vh_launch_linear_mx_planes_f16(vh_stream_t s, void *output_planes_f16, const void *weight_values,
                                   const void *weight_scales, const void *input_values, const void *input_scales,
                                   const float *bias, int rowA, int colA, int colB);

/* ---- LayerNorm folded into the projection behind it (the reduced-precision modes; csrc/norm_fold.h) ----
 * Replaces the `layerNorm` launches in front of the QKV projection and fc1 (layer_norm.cl:3-53; ViT_seq.c:120-142,346,358):
 *     LN(x) W^T + b = rstd (x (gamma.W)^T - mean colsum(gamma.W)) + (beta W^T + b)
 * The producers of the residual stream x -- patch embedding, output projection, fc2 -- leave x, besides the fp32 rows,
 * as the next projection's operand (one-part bf16 planes, or an MX tensor) and, per row and 128 columns, the partial sums
 * (sum x, sum x^2): row_stats[cols/128][rows][2] (no atomics: one writer per partial, fixed summation order).  The
 * consuming projection multiplies the un-normalised rows by the gamma-scaled weights and applies the row terms in its
 * epilogue; mean and 1/std are formed as layer_norm_seq forms them (eps added in double). */
/* context-creation helpers: out[n][k] = weight[n][k] gamma[k];  out[n] = bias[n] + sum_k beta[k] weight[n][k];
 * out[n] = sum_k of the operand's own (rounded / quantised) values of weight row n (scales NULL: one-part bf16 planes) */
int vh_launch_fold_gamma(vh_stream_t s, const float *weight, const float *gamma, float *out, int out_features, int in_features);
int vh_launch_fold_bias(vh_stream_t s, const float *weight, const float *beta, const float *bias, float *out, int out_features,
                        int in_features);
int vh_launch_colsum_operand(vh_stream_t s, const void *values, const void *scales, float *out, int out_features, int in_features);
/* vh_launch_patch_embed_planes that also leaves the token rows (class-token rows included) as the first projection's
 * operand -- one-part bf16 planes [embed/32][n*tokens][32], or MX values + scales when operand_scales_out != NULL -- and
 * their partial sums row_stats_out [embed/128][n*tokens][2] */
int vh_launch_patch_embed_planes_norm(vh_stream_t s, const float *images, const void *conv_w_planes, const float *conv_b,
                                      const float *cls_token, const float *pos_embed, float *tokens, int n_images, int in_chans,
                                      int img_size, int patch_size, int embed_dim, void *workspace, size_t workspace_bytes,
                                      void *operand_out, void *operand_scales_out, float *row_stats_out);
/* LN(A) W^T + b, folded: input_planes = the un-normalised rows (one-part bf16 planes), weight_planes = gamma-scaled.
 * output_planes as vh_launch_linear_planes (0 fp32 rows, 1 bf16 planes, 2 fp16 planes); doGelu needs output_planes 1. */
int vh_launch_linear_planes_norm(vh_stream_t s, void *output, int output_planes, const void *weight_planes,
                                 const void *input_planes, const float *row_stats, const float *colsum, const float *bias_folded,
                                 double eps, int rowA, int colA, int colB, int doGelu);
/* output = residual + A W^T + b (fp32 rows; in place allowed), the same rows as the next projection's operand (bf16
 * planes, or MX when operand_scales_out != NULL) and their partial sums row_stats_out [colB/128][rowA][2] */
int vh_launch_linear_planes_resid_norm(vh_stream_t s, float *output, const void *weight_planes, const void *input_planes,
                                       const float *bias, const float *residual, int rowA, int colA, int colB,
                                       void *operand_out, void *operand_scales_out, float *row_stats_out);
/* the same three on the exact THREE-part planes (the fp32 path's lab variant, $VIT_HIP_LN_FOLD=1): six-product arithmetic on the
 * un-normalised rows; output_planes 0 (fp32 rows) or 1 (three-part planes) */
int vh_launch_colsum_planes3(vh_stream_t s, const void *planes3, float *out, int out_features, int in_features);
int vh_launch_patch_embed_planes3_norm(vh_stream_t s, const float *images, const void *conv_w_planes3, const float *conv_b,
                                       const float *cls_token, const float *pos_embed, float *tokens, int n_images, int in_chans,
                                       int img_size, int patch_size, int embed_dim, void *workspace, size_t workspace_bytes,
                                       void *operand_planes3_out, float *row_stats_out);
int vh_launch_linear_p3_norm(vh_stream_t s, void *output, int output_planes, const void *weight_planes3, const void *input_planes3,
                             const float *row_stats, const float *colsum, const float *bias_folded, double eps, int rowA, int colA,
                             int colB, int doGelu);
int vh_launch_linear_p3_resid_norm(vh_stream_t s, float *output, const void *weight_planes3, const void *input_planes3, const float *bias,
                                   const float *residual, int rowA, int colA, int colB, void *operand_planes3_out, float *row_stats_out);
/* the same two on block-scaled fp8 operands.  output_kind: 0 fp32 rows, 1 MX tensor (the only kind doGelu takes), 2 one-part
 * fp16 planes */
int vh_launch_linear_mx_norm(vh_stream_t s, void *output, void *output_scales, int output_kind, const void *weight_values,
                             const void *weight_scales, const void *input_values, const void *input_scales, const float *row_stats,
                             const float *colsum, const float *bias_folded, double eps, int rowA, int colA, int colB, int doGelu);
int vh_launch_linear_mx_resid_norm(vh_stream_t s, float *output, const void *weight_values, const void *weight_scales,
                                   const void *input_values, const void *input_scales, const float *bias, const float *residual,
                                   int rowA, int colA, int colB, void *operand_values, void *operand_scales, float *row_stats_out);

#ifdef __cplusplus
}
#endif

#endif /* VIT_HIP_KERNELHANDLER_H */
