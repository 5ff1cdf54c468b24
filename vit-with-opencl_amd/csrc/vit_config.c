/*
 * vit_config.c -- model-shape bookkeeping and deterministic synthetic data.
 * Pure C, no device code; linked into libvit_hip.so and (for the synthetic
 * data only) into the oracle harness so both sides see identical inputs.
 *
 * Tensor index map follows the reference's literal indices
 * (ViT_seq.c:437-513; ViT_opencl.c:159 `base = 4 + 12*layer`, :280-295).
 */
#include "ViT_opencl.h"

#include <stdint.h>
#include <string.h>

int vit_config_preset(vit_config *cfg, const char *name)
{
    if (!cfg || !name)
        return -1;
    cfg->img_size = 224;
    cfg->in_chans = 3;
    cfg->num_classes = 1000;
    cfg->eps = 1e-6;
    if (strcmp(name, "vit_b_16") == 0) {
        /* ViT_seq.c:10-21 */
        cfg->patch_size = 16; cfg->embed_dim = 768; cfg->depth = 12;
        cfg->num_heads = 12; cfg->mlp_hidden = 3072;
    } else if (strcmp(name, "vit_l_16") == 0) {
        cfg->patch_size = 16; cfg->embed_dim = 1024; cfg->depth = 24;
        cfg->num_heads = 16; cfg->mlp_hidden = 4096;
    } else if (strcmp(name, "vit_h_14") == 0) {
        cfg->patch_size = 14; cfg->embed_dim = 1280; cfg->depth = 32;
        cfg->num_heads = 16; cfg->mlp_hidden = 5120;
    } else {
        return -1;
    }
    return 0;
}

int vit_config_tokens(const vit_config *cfg)
{
    int g = cfg->img_size / cfg->patch_size;
    return g * g + 1;
}

int vit_config_num_tensors(const vit_config *cfg)
{
    return 4 + 12 * cfg->depth + 4;
}

size_t vit_config_tensor_size(const vit_config *cfg, int idx)
{
    const size_t E = (size_t)cfg->embed_dim, F = (size_t)cfg->mlp_hidden;
    const size_t P = (size_t)cfg->patch_size, C = (size_t)cfg->in_chans;
    const int tail = 4 + 12 * cfg->depth;
    if (idx < 0 || idx >= tail + 4)
        return 0;
    if (idx == 0) return E;                                      /* class_token */
    if (idx == 1) return E * C * P * P;                          /* conv_proj.weight */
    if (idx == 2) return E;                                      /* conv_proj.bias */
    if (idx == 3) return (size_t)vit_config_tokens(cfg) * E;     /* pos_embedding */
    if (idx >= tail) {
        switch (idx - tail) {
        case 0: case 1: return E;                                /* encoder.ln w,b */
        case 2: return (size_t)cfg->num_classes * E;             /* head.weight */
        default: return (size_t)cfg->num_classes;                /* head.bias */
        }
    }
    switch ((idx - 4) % 12) {
    case 0: case 1: return E;            /* ln_1 w,b */
    case 2: return 3 * E * E;            /* in_proj_weight */
    case 3: return 3 * E;                /* in_proj_bias */
    case 4: return E * E;                /* out_proj.weight */
    case 5: return E;                    /* out_proj.bias */
    case 6: case 7: return E;            /* ln_2 w,b */
    case 8: return F * E;                /* mlp.0.weight */
    case 9: return F;                    /* mlp.0.bias */
    case 10: return E * F;               /* mlp.3.weight */
    default: return E;                   /* mlp.3.bias */
    }
}

/* splitmix64 finaliser used as a counter-based generator: value i of stream
 * `seed` depends on (seed, i) only, so any slice can be produced in parallel
 * and a numpy twin (host/synth.py) reproduces it bit for bit. */
static inline uint64_t vit_mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

void vit_synth_fill(float *dst, size_t count, unsigned long long seed, float scale, float offset)
{
    const uint64_t base = vit_mix64((uint64_t)seed);
    for (size_t i = 0; i < count; ++i) {
        uint32_t u24 = (uint32_t)(vit_mix64(base + (uint64_t)i) >> 40); /* 24 random bits */
        float u = (float)(int32_t)u24 * (1.0f / 8388608.0f) - 1.0f;     /* exact, [-1,1) */
        dst[i] = offset + scale * u;
    }
}

/* Scales chosen so the uniform std (scale/sqrt(3)) matches the measured std of
 * the reference's real tensors. */
void vit_synth_tensor(const vit_config *cfg, int idx, unsigned long long seed_base, float *dst)
{
    const size_t n = vit_config_tensor_size(cfg, idx);
    const int tail = 4 + 12 * cfg->depth;
    float scale = 0.03f, offset = 0.0f;
    if (idx == 0) scale = 0.02f;                 /* class token */
    else if (idx == 1) scale = 0.016f;           /* conv weight, std 0.0092 */
    else if (idx == 2) scale = 0.05f;            /* conv bias */
    else if (idx == 3) scale = 0.088f;           /* pos embedding, std 0.051 */
    else if (idx >= tail) {
        switch (idx - tail) {
        case 0: scale = 0.2f; offset = 0.7f; break;   /* encoder.ln weight */
        case 1: scale = 0.05f; break;
        case 2: scale = 0.064f; break;                /* head weight, std 0.037 */
        default: scale = 0.035f; break;
        }
    } else {
        switch ((idx - 4) % 12) {
        case 0: case 6: scale = 0.25f; offset = 0.3f; break; /* LN gamma */
        case 1: case 7: scale = 0.05f; break;                /* LN beta */
        case 2: scale = 0.04f; break;                        /* in_proj weight */
        case 3: scale = 0.1f; break;                         /* in_proj bias, std 0.059 */
        case 4: scale = 0.043f; break;                       /* out_proj weight, std 0.025 */
        case 5: scale = 0.035f; break;
        case 8: scale = 0.04f; break;                        /* fc1 weight */
        case 9: scale = 0.035f; offset = -0.026f; break;     /* fc1 bias */
        case 10: scale = 0.04f; break;                       /* fc2 weight */
        default: scale = 0.02f; break;
        }
    }
    vit_synth_fill(dst, n, seed_base + (unsigned long long)idx, scale, offset);
}

void vit_synth_image(const vit_config *cfg, int image_index, float *dst)
{
    const size_t n = (size_t)cfg->in_chans * cfg->img_size * cfg->img_size;
    vit_synth_fill(dst, n, 1000ull + (unsigned long long)image_index, 2.0f, 0.0f);
}
