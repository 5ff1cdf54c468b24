/*
 * vit_seq_port.c -- CPU restatement ("port") of the reference's scalar ViT
 * forward pass.  TEST INFRASTRUCTURE ONLY: nothing under oracle/ is part of
 * the product; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it, and only as the checker / the CPU timing leg.
 *
 * Parity status: PINNED for ViT-B/16.  tests/test_oracle.py requires this
 * file's logits and probabilities to be bit-identical to golden vectors that
 * oracle/ref_harness produced by running the reference's own, unmodified
 * ViT_seq.c (compiled in place into oracle/_ref/) on the same synthetic inputs
 * and, when oracle/_ref/ is present, re-checks that equality live, stage by
 * stage.  For ViT-L/16 and ViT-H/14 the reference has no code at all (its
 * shape is #define'd, ViT_seq.c:10-21); the same loops with other bounds are
 * "parity unpinned" there.
 *
 * Every function cites the reference lines it restates.  The arithmetic is
 * fp32 with strictly sequential accumulation in the reference's index order;
 * build with -ffp-contract=off and without -ffast-math (see oracle/Makefile).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ViT_opencl.h" /* ImageData, Network, vit_config */

/* ViT_seq.c:25-57 -- stride == kernel == patch convolution, output [E][g][g]. */
void port_conv2d(const vit_config *c, const float *input, float *output,
                 const float *weight, const float *bias)
{
    const int g = c->img_size / c->patch_size, P = c->patch_size, S = c->img_size;
    for (int oc = 0; oc < c->embed_dim; ++oc)
        for (int oh = 0; oh < g; ++oh)
            for (int ow = 0; ow < g; ++ow) {
                float sum = bias[oc];
                for (int ic = 0; ic < c->in_chans; ++ic)
                    for (int kh = 0; kh < P; ++kh)
                        for (int kw = 0; kw < P; ++kw) {
                            int ih = oh * P + kh, iw = ow * P + kw;
                            sum += input[(ic * S + ih) * S + iw] *
                                   weight[((oc * c->in_chans + ic) * P + kh) * P + kw];
                        }
                output[(oc * g + oh) * g + ow] = sum;
            }
}

/* ViT_seq.c:59-118 -- [E][g*g] -> [g*g][E], class token in row 0, + pos_embed. */
void port_tokens(const vit_config *c, const float *conv_out, float *tokens,
                 const float *cls, const float *pos)
{
    const int g = c->img_size / c->patch_size, np = g * g, E = c->embed_dim;
    for (int j = 0; j < E; ++j)
        tokens[j] = cls[j];
    for (int p = 0; p < np; ++p)
        for (int oc = 0; oc < E; ++oc)
            tokens[(size_t)(1 + p) * E + oc] = conv_out[(size_t)oc * np + p];
    for (size_t i = 0; i < (size_t)(np + 1) * E; ++i)
        tokens[i] = tokens[i] + pos[i];
}

/* ViT_seq.c:120-142.  `eps` is a double literal there, so var + eps is a
 * double add that is narrowed to float at the sqrtf call. */
void port_layer_norm(const vit_config *c, const float *input, float *output,
                     const float *weight, const float *bias, int tokens)
{
    const int E = c->embed_dim;
    for (int t = 0; t < tokens; ++t) {
        float sum = 0.0f, sum_sq = 0.0f;
        for (int i = 0; i < E; ++i) {
            float val = input[(size_t)t * E + i];
            sum += val;
            sum_sq += val * val;
        }
        float mean = sum / E;
        float var = sum_sq / E - mean * mean;
        float inv_std = 1.0f / sqrtf(var + c->eps);
        for (int i = 0; i < E; ++i) {
            size_t idx = (size_t)t * E + i;
            output[idx] = (input[idx] - mean) * inv_std * weight[i] + bias[i];
        }
    }
}

/* ViT_seq.c:295-309 (also the out-projection loop :268-279 and, with three
 * row offsets, the fused QKV loop :156-172): sum starts at the bias. */
void port_linear(const float *input, float *output, int tokens, int in_features,
                 int out_features, const float *weight, const float *bias)
{
    for (int t = 0; t < tokens; ++t)
        for (int o = 0; o < out_features; ++o) {
            float sum = bias[o];
            for (int i = 0; i < in_features; ++i)
                sum += input[(size_t)t * in_features + i] * weight[(size_t)o * in_features + i];
            output[(size_t)t * out_features + o] = sum;
        }
}

/* ViT_seq.c:192-262 -- per-head scaled-dot-product attention over fused
 * [Q | K | V] rows (row stride 3E); heads are concatenated in `attn` ([tokens][E]). */
void port_attention(const vit_config *c, const float *qkv, float *attn, int tokens)
{
    const int E = c->embed_dim, H = c->num_heads, D = E / H;
    const float *Q = qkv, *K = qkv + E, *V = qkv + 2 * E;
    const size_t ld = (size_t)3 * E;
    float *scores = (float *)malloc(sizeof(float) * (size_t)tokens * tokens);

    for (int h = 0; h < H; ++h) {
        const int off = h * D;
        /* :200-213 */
        for (int i = 0; i < tokens; ++i)
            for (int j = 0; j < tokens; ++j) {
                float score = 0.0f;
                for (int d = 0; d < D; ++d)
                    score += Q[i * ld + off + d] * K[j * ld + off + d];
                scores[(size_t)i * tokens + j] = score / sqrtf((float)D);
            }
        /* :216-234 */
        for (int i = 0; i < tokens; ++i) {
            float *row = scores + (size_t)i * tokens;
            float max_val = row[0];
            for (int j = 1; j < tokens; ++j)
                if (row[j] > max_val)
                    max_val = row[j];
            float sum_exp = 0.0f;
            for (int j = 0; j < tokens; ++j) {
                row[j] = expf(row[j] - max_val);
                sum_exp += row[j];
            }
            for (int j = 0; j < tokens; ++j)
                row[j] /= sum_exp;
        }
        /* :238-258 */
        for (int i = 0; i < tokens; ++i)
            for (int d = 0; d < D; ++d) {
                float sum = 0.0f;
                for (int j = 0; j < tokens; ++j)
                    sum += scores[(size_t)i * tokens + j] * V[j * ld + off + d];
                attn[(size_t)i * E + off + d] = sum;
            }
    }
    free(scores);
}

/* ViT_seq.c:144-281. */
void port_mha(const vit_config *c, const float *input, float *output, const float *in_w,
              const float *in_b, const float *out_w, const float *out_b, int tokens)
{
    const int E = c->embed_dim;
    float *qkv = (float *)malloc(sizeof(float) * (size_t)tokens * 3 * E);
    float *attn = (float *)malloc(sizeof(float) * (size_t)tokens * E);

    /* :156-172 -- rows [0,E) of in_w give Q, [E,2E) K, [2E,3E) V. */
    port_linear(input, qkv, tokens, E, 3 * E, in_w, in_b);
    port_attention(c, qkv, attn, tokens);
    /* :268-279 */
    port_linear(attn, output, tokens, E, E, out_w, out_b);
    free(attn);
    free(qkv);
}

/* ViT_seq.c:283-286. */
float port_gelu(float x)
{
    return 0.5f * x * (1.0f + erff(x / sqrtf(2.0f)));
}

/* ViT_seq.c:310-327. */
void port_mlp(const vit_config *c, const float *input, float *output, const float *fc1_w,
              const float *fc1_b, const float *fc2_w, const float *fc2_b, int tokens)
{
    const int E = c->embed_dim, F = c->mlp_hidden;
    float *hid = (float *)malloc(sizeof(float) * (size_t)tokens * F);
    port_linear(input, hid, tokens, E, F, fc1_w, fc1_b);
    for (size_t i = 0; i < (size_t)tokens * F; ++i)
        hid[i] = port_gelu(hid[i]);
    port_linear(hid, output, tokens, F, E, fc2_w, fc2_b);
    free(hid);
}

/* ViT_seq.c:330-370 -- w points at the layer's 12 tensors (ln1 w,b; in w,b;
 * out w,b; ln2 w,b; fc1 w,b; fc2 w,b). */
void port_encoder(const vit_config *c, const float *input, float *output, const Network *w,
                  int tokens)
{
    const size_t n = (size_t)tokens * c->embed_dim;
    float *ln = (float *)malloc(sizeof(float) * n);
    float *tmp = (float *)malloc(sizeof(float) * n);
    float *res = (float *)malloc(sizeof(float) * n);

    port_layer_norm(c, input, ln, w[0].data, w[1].data, tokens);
    port_mha(c, ln, tmp, w[2].data, w[3].data, w[4].data, w[5].data, tokens);
    for (size_t i = 0; i < n; ++i)
        res[i] = input[i] + tmp[i];
    port_layer_norm(c, res, ln, w[6].data, w[7].data, tokens);
    port_mlp(c, ln, tmp, w[8].data, w[9].data, w[10].data, w[11].data, tokens);
    for (size_t i = 0; i < n; ++i)
        output[i] = res[i] + tmp[i];

    free(res);
    free(tmp);
    free(ln);
}

/* ViT_seq.c:372-397. */
void port_softmax(const float *logits, float *probabilities, int length)
{
    float max_val = logits[0];
    for (int i = 1; i < length; ++i)
        if (logits[i] > max_val)
            max_val = logits[i];
    float sum_exp = 0.0f;
    for (int i = 0; i < length; ++i) {
        probabilities[i] = expf(logits[i] - max_val);
        sum_exp += probabilities[i];
    }
    for (int i = 0; i < length; ++i)
        probabilities[i] /= sum_exp;
}

/* One image through the whole model (ViT_seq.c:433-517).  Any of logits /
 * probs / tokens_out may be NULL; tokens_out ([tokens][E]) receives the
 * residual stream after `stop_after_layers` encoder layers (-1 = all) so
 * tests can compare intermediate activations. */
void port_forward_image(const vit_config *c, const float *image, const Network *nets,
                        float *logits, float *probs, float *tokens_out, int stop_after_layers)
{
    const int g = c->img_size / c->patch_size, T = g * g + 1, E = c->embed_dim;
    const size_t n = (size_t)T * E;
    float *conv = (float *)malloc(sizeof(float) * (size_t)E * g * g);
    float *x = (float *)malloc(sizeof(float) * n);
    float *y = (float *)malloc(sizeof(float) * n);

    port_conv2d(c, image, conv, nets[1].data, nets[2].data);
    port_tokens(c, conv, x, nets[0].data, nets[3].data);

    int layers = c->depth;
    if (stop_after_layers >= 0 && stop_after_layers < layers)
        layers = stop_after_layers;
    for (int l = 0; l < layers; ++l) {
        port_encoder(c, x, y, nets + 4 + 12 * l, T);
        float *t = x; x = y; y = t;
    }
    if (tokens_out)
        memcpy(tokens_out, x, sizeof(float) * n);

    if ((logits || probs) && layers == c->depth) {
        const int tail = 4 + 12 * c->depth;
        float *lg = (float *)malloc(sizeof(float) * c->num_classes);
        /* :506 normalises all rows but only row 0 is used (:511); rows are
         * independent, so normalising row 0 alone gives the same bits. */
        port_layer_norm(c, x, y, nets[tail].data, nets[tail + 1].data, 1);
        port_linear(y, lg, 1, E, c->num_classes, nets[tail + 2].data, nets[tail + 3].data);
        if (logits)
            memcpy(logits, lg, sizeof(float) * c->num_classes);
        if (probs)
            port_softmax(lg, probs, c->num_classes);
        free(lg);
    }
    free(y);
    free(x);
    free(conv);
}

/* Same call shape as the reference entry point ViT_seq (ViT_seq.c:402). */
void port_ViT_seq(const vit_config *c, ImageData *image, Network *networks, float **probabilities)
{
    for (int i = 0; i < image->n; ++i)
        port_forward_image(c, image[i].data, networks, NULL, probabilities[i], NULL, -1);
}
