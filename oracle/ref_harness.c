/*
 * ref_harness.c -- drives the reference's own, unmodified ViT_seq.c (built in
 * place from /root/reference into oracle/_ref/libvitseq_ref.so by
 * oracle/Makefile; no reference source is copied into this repository) on the
 * deterministic synthetic inputs of vit_synth_*().  TEST INFRASTRUCTURE ONLY.
 *
 *   ref_harness full   <first_image> <count> <seed_base> <out.bin>
 *   ref_harness full_rounded <first_image> <count> <seed_base> <out.bin>   (weights rounded to 1e-6
 *                       like the reference loader does, Network.c:208-211: the file-based flow)
 *   ref_harness stages <seed_base> <out.bin>
 *   ref_harness time   <first_image> <count> <seed_base>       (prints seconds/image)
 *   ref_harness full_file <images.bin> <seed_base> <out.bin>   (images from a file in the reference's own format,
 *                       Network.c:41-71: int n, c, h, w, then n*c*h*w floats -- e.g. its Data/input-1.bin; synthetic weights)
 *
 * Logits are not observable through the reference's call surface
 * (ViT_seq.c:509-515 keeps them in a local).  The reference library is built
 * -fPIC, so its call to Softmax_seq goes through the PLT; this executable
 * defines Softmax_seq itself, records the logits, and forwards to the
 * reference's real Softmax_seq found with dlsym(RTLD_NEXT).
 *
 * Output file: a sequence of records { char name[32]; uint64 count; float data[count]; }.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ViT_opencl.h" /* ImageData, Network, vit_config, vit_synth_* (layout-identical types) */

/* Reference entry points (ViT_seq.c); declared here because ViT_seq.h:5 has
 * the ViT_seq prototype commented out and Network.h defines globals. */
void ViT_seq(ImageData *image, Network *networks, float **probabilities);   /* :402 */
void Conv2d_seq(float *input, float *output, Network weight, Network bias); /* :25 */
void flatten_transpose_seq(float *input, float *output);                    /* :59 */
void class_token_seq(float *patch_tokens, float *final_tokens, Network cls_tk); /* :83 */
void pos_emb_seq(float *input, float *output, Network pos_emb);             /* :107 */
void layer_norm_seq(float *input, float *output, Network weight, Network bias); /* :120 */
void multihead_attn_seq(float *input, float *output, Network in_weight, Network in_bias,
                        Network out_weight, Network out_bias);              /* :144 */
float gelu(float x);                                                        /* :283 */
void linear_layer_seq(float *input, float *output, int tokens, int in_features,
                      int out_features, Network weight, Network bias);      /* :295 */
void mlp_block_seq(float *input, float *output, Network fc1_weight, Network fc1_bias,
                   Network fc2_weight, Network fc2_bias);                   /* :310 */
void Encoder_seq(float *input, float *output, Network ln1_w, Network ln1_b, Network attn_w,
                 Network attn_b, Network attn_out_w, Network attn_out_b, Network ln2_w,
                 Network ln2_b, Network mlp1_w, Network mlp1_b, Network mlp2_w,
                 Network mlp2_b);                                           /* :330 */

static float *g_logit_sink = NULL; /* where the hook stores the next image's logits */
static int g_hook_calls = 0;

/* Interposes ViT_seq.c:372. */
void Softmax_seq(float *logits, float *probabilities, int length)
{
    static void (*real)(float *, float *, int) = NULL;
    if (!real) {
        real = (void (*)(float *, float *, int))dlsym(RTLD_NEXT, "Softmax_seq");
        if (!real) {
            fprintf(stderr, "ref_harness: reference Softmax_seq not found\n");
            exit(2);
        }
    }
    if (g_logit_sink) {
        memcpy(g_logit_sink, logits, sizeof(float) * (size_t)length);
        g_logit_sink += length;
    }
    ++g_hook_calls;
    real(logits, probabilities, length);
}

static void put(FILE *f, const char *name, const float *data, uint64_t count)
{
    char tag[32] = {0};
    strncpy(tag, name, sizeof(tag) - 1);
    fwrite(tag, 1, sizeof(tag), f);
    fwrite(&count, sizeof(count), 1, f);
    fwrite(data, sizeof(float), count, f);
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static int g_round_weights = 0; /* emulate load_weights' rounding (reference Network.c:208-211) */

static Network *make_weights(const vit_config *cfg, unsigned long long seed_base)
{
    int n = vit_config_num_tensors(cfg);
    Network *nets = (Network *)calloc((size_t)n, sizeof(Network));
    for (int i = 0; i < n; ++i) {
        nets[i].size = vit_config_tensor_size(cfg, i);
        nets[i].data = (float *)malloc(sizeof(float) * nets[i].size);
        vit_synth_tensor(cfg, i, seed_base, nets[i].data);
        if (g_round_weights)
            for (size_t k = 0; k < nets[i].size; ++k)
                nets[i].data[k] = roundf(nets[i].data[k] * 1000000.0f) / 1000000.0f;
    }
    return nets;
}

static ImageData *make_images(const vit_config *cfg, int first, int count)
{
    size_t per = (size_t)cfg->in_chans * cfg->img_size * cfg->img_size;
    ImageData *im = (ImageData *)calloc((size_t)count, sizeof(ImageData));
    for (int i = 0; i < count; ++i) {
        im[i].n = count; im[i].c = cfg->in_chans; im[i].h = cfg->img_size; im[i].w = cfg->img_size;
        im[i].data = (float *)malloc(sizeof(float) * per);
        vit_synth_image(cfg, first + i, im[i].data);
    }
    return im;
}

/* images from a file in the format of the reference's load_image_data (Network.c:41-71) */
static ImageData *read_images(const vit_config *cfg, const char *path, int *count)
{
    FILE *f = fopen(path, "rb");
    int hdr[4];
    if (!f || fread(hdr, sizeof(int), 4, f) != 4 || hdr[0] <= 0 || hdr[1] != cfg->in_chans || hdr[2] != cfg->img_size ||
        hdr[3] != cfg->img_size) {
        fprintf(stderr, "ref_harness: %s is not an image file of this model's shape\n", path);
        exit(2);
    }
    const size_t per = (size_t)hdr[1] * hdr[2] * hdr[3];
    ImageData *im = (ImageData *)calloc((size_t)hdr[0], sizeof(ImageData));
    for (int i = 0; i < hdr[0]; ++i) {
        im[i].n = hdr[0]; im[i].c = hdr[1]; im[i].h = hdr[2]; im[i].w = hdr[3];
        im[i].data = (float *)malloc(sizeof(float) * per);
        if (fread(im[i].data, sizeof(float), per, f) != per) {
            fprintf(stderr, "ref_harness: %s is truncated\n", path);
            exit(2);
        }
    }
    fclose(f);
    *count = hdr[0];
    return im;
}

static const char *g_image_file = NULL;

static int run_full(int first, int count, unsigned long long seed_base, const char *out, int timing_only)
{
    vit_config cfg;
    vit_config_preset(&cfg, "vit_b_16");
    Network *nets = make_weights(&cfg, seed_base);
    ImageData *im = g_image_file ? read_images(&cfg, g_image_file, &count) : make_images(&cfg, first, count);
    const int C = cfg.num_classes;
    float *logits = (float *)malloc(sizeof(float) * (size_t)count * C);
    float *probs = (float *)malloc(sizeof(float) * (size_t)count * C);
    float **rows = (float **)malloc(sizeof(float *) * (size_t)count);
    for (int i = 0; i < count; ++i)
        rows[i] = probs + (size_t)i * C;

    g_logit_sink = logits;
    double t0 = now_s();
    ViT_seq(im, nets, rows);
    double dt = now_s() - t0;
    g_logit_sink = NULL;
    if (g_hook_calls != count) {
        fprintf(stderr, "ref_harness: Softmax_seq hook ran %d times for %d images\n", g_hook_calls, count);
        return 3;
    }
    fprintf(stderr, "ref_harness: %d image(s) in %.3f s = %.3f s/image\n", count, dt, dt / count);
    if (timing_only) {
        fprintf(stderr, "SECONDS_PER_IMAGE %.6f\n", dt / count);
        return 0;
    }
    FILE *f = fopen(out, "wb");
    if (!f) { perror(out); return 1; }
    float secs = (float)(dt / count);
    put(f, "logits", logits, (uint64_t)count * C);
    put(f, "probs", probs, (uint64_t)count * C);
    put(f, "seconds_per_image", &secs, 1);
    fclose(f);
    return 0;
}

static int run_stages(unsigned long long seed_base, const char *out)
{
    vit_config cfg;
    vit_config_preset(&cfg, "vit_b_16");
    Network *w = make_weights(&cfg, seed_base);
    ImageData *im = make_images(&cfg, 0, 1);
    const int E = cfg.embed_dim, T = vit_config_tokens(&cfg), NP = T - 1;
    const size_t n = (size_t)T * E;
    float *conv = (float *)malloc(sizeof(float) * (size_t)E * NP);
    float *flat = (float *)malloc(sizeof(float) * (size_t)E * NP);
    float *cat = (float *)malloc(sizeof(float) * n);
    float *tok = (float *)malloc(sizeof(float) * n);
    float *ln = (float *)malloc(sizeof(float) * n);
    float *mha = (float *)malloc(sizeof(float) * n);
    float *mlp = (float *)malloc(sizeof(float) * n);
    float *enc = (float *)malloc(sizeof(float) * n);
    float head[1000], sm[1000], gl[4001];

    FILE *f = fopen(out, "wb");
    if (!f) { perror(out); return 1; }

    Conv2d_seq(im[0].data, conv, w[1], w[2]);
    put(f, "conv", conv, (uint64_t)E * NP);
    flatten_transpose_seq(conv, flat);
    class_token_seq(flat, cat, w[0]);
    pos_emb_seq(cat, tok, w[3]);
    put(f, "tokens", tok, n);
    layer_norm_seq(tok, ln, w[4], w[5]);
    put(f, "ln", ln, n);
    multihead_attn_seq(ln, mha, w[6], w[7], w[8], w[9]);
    put(f, "mha", mha, n);
    mlp_block_seq(ln, mlp, w[12], w[13], w[14], w[15]);
    put(f, "mlp", mlp, n);
    Encoder_seq(tok, enc, w[4], w[5], w[6], w[7], w[8], w[9], w[10], w[11], w[12], w[13], w[14], w[15]);
    put(f, "enc0", enc, n);
    linear_layer_seq(ln, head, 1, E, cfg.num_classes, w[150], w[151]);
    put(f, "head", head, 1000);
    Softmax_seq(head, sm, 1000); /* goes through the hook into the reference's own */
    put(f, "softmax", sm, 1000);
    for (int i = 0; i <= 4000; ++i)
        gl[i] = gelu((float)(i - 2000) * (1.0f / 256.0f)); /* x in [-7.8125, 7.8125] */
    put(f, "gelu", gl, 4001);
    fclose(f);
    return 0;
}

int main(int argc, char **argv)
{
    /* The reference prints six debug lines per layer to stdout (ViT_seq.c:173-181). */
    if (!freopen("/dev/null", "w", stdout))
        return 1;
    if (argc == 6 && strcmp(argv[1], "full_rounded") == 0) {
        g_round_weights = 1; /* weights as the reference's loader would deliver them from disk */
        return run_full(atoi(argv[2]), atoi(argv[3]), strtoull(argv[4], NULL, 10), argv[5], 0);
    }
    if (argc == 5 && strcmp(argv[1], "full_file") == 0) {
        g_image_file = argv[2];
        return run_full(0, 0, strtoull(argv[3], NULL, 10), argv[4], 0);
    }
    if (argc == 6 && strcmp(argv[1], "full") == 0)
        return run_full(atoi(argv[2]), atoi(argv[3]), strtoull(argv[4], NULL, 10), argv[5], 0);
    if (argc == 5 && strcmp(argv[1], "time") == 0)
        return run_full(atoi(argv[2]), atoi(argv[3]), strtoull(argv[4], NULL, 10), NULL, 1);
    if (argc == 4 && strcmp(argv[1], "stages") == 0)
        return run_stages(strtoull(argv[2], NULL, 10), argv[3]);
    fprintf(stderr, "usage: ref_harness full <first> <count> <seed_base> <out.bin>\n"
                    "       ref_harness time <first> <count> <seed_base>\n"
                    "       ref_harness stages <seed_base> <out.bin>\n");
    return 64;
}
