/*
 * vit_main.c -- a small POSIX driver over this repository's headers with the flow of the reference's
 * Main.c (Main.c:16-72) for any image count: load_image_data -> load_weights -> ViT_opencl ->
 * result file.  The reference's own Main.c and comparator.c are exercised unmodified by
 * oracle/_ref/ref_main (oracle/Makefile); this driver exists because those two hard-code 100
 * images and their file names (Main.c:20,43; comparator.c:9,30-31).
 *
 *   vit_main [image_file] [network_dir] [result_file] [planes_file]
 *   defaults: ./Data/input-100.bin ./Network ./Data/opencl_result.txt (no planes file)
 *
 * With a fourth argument the driver shows the offline half of the weight tooling: if `planes_file` exists the context
 * is built from it (vit_hip_create_from_planes: ONE file of already repacked operands, the 152 fp32 files of
 * `network_dir` are not read at all); otherwise the weights are loaded as usual, the context is created with
 * $VIT_HIP_PRECISION's arithmetic and the file is written (vit_hip_export_planes) for the next run.  A file that does
 * not load (truncated, corrupt, another library version) or that holds another precision than $VIT_HIP_PRECISION asks
 * for is rebuilt from `network_dir` and rewritten.  $VIT_HIP_DEVICE picks the GPU, as for ViT_opencl.
 *
 * Result lines have Main.c's format ("[%d] label: %d / prob: %.6f", Main.c:71); the
 * arg-max restarts for every image (Main.c:59 declares pred_idx outside the loop, so
 * class 0 can leak from image to image there).
 * Exit status 0, or 100 on I/O trouble.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ViT_opencl.h"

#define NUM_TENSORS 152
#define NUM_CLASSES 1000

static double wall(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* $VIT_HIP_PRECISION as the library reads it (vit_hip_create): fp32 unless "bf16", "fp16x2" or "fp8" */
static int requested_precision(void)
{
    const char *env = getenv("VIT_HIP_PRECISION");
    if (env && env[0] == 'b')
        return VIT_PRECISION_BF16_GEMM;
    if (env && strncmp(env, "fp16x2", 6) == 0)
        return VIT_PRECISION_F32_FP16X2;
    if (env && strncmp(env, "fp8", 3) == 0)
        return VIT_PRECISION_FP8_GEMM;
    return VIT_PRECISION_F32;
}

int main(int argc, char **argv)
{
    const char *image_file = argc > 1 ? argv[1] : "./Data/input-100.bin";
    const char *network_dir = argc > 2 ? argv[2] : "./Network";
    const char *result_file = argc > 3 ? argv[3] : "./Data/opencl_result.txt";

    const char *planes_file = argc > 4 ? argv[4] : NULL;

    ImageData *images = load_image_data(image_file);
    if (images == NULL)
        return 100;
    const int n = images->n;
    float **probabilities = (float **)malloc(sizeof(float *) * (size_t)(n > 0 ? n : 1));
    if (probabilities == NULL) {
        fprintf(stderr, "vit_main: out of memory (%d result rows)\n", n);
        return 101;
    }
    for (int i = 0; i < n; ++i) {
        probabilities[i] = (float *)malloc(sizeof(float) * NUM_CLASSES);
        if (probabilities[i] == NULL) {
            fprintf(stderr, "vit_main: out of memory (result row %d of %d)\n", i, n);
            return 101;
        }
    }

    printf("=====================Start========================\n");
    const double t0 = wall();
    if (n <= 0) {
        /* nothing to classify (load_image_data refuses such a header, so this is belt and braces): like the reference's
         * per-image loop (ViT_opencl.c:926), no device is touched and no context is asked for a batch of zero */
    } else if (planes_file != NULL) {
        const int chunk = n < 512 ? n : 512;
        const int device = getenv("VIT_HIP_DEVICE") ? atoi(getenv("VIT_HIP_DEVICE")) : 0;   /* as ViT_opencl honours it */
        const int want = requested_precision();
        vit_hip_ctx *ctx = NULL;
        FILE *probe = fopen(planes_file, "rb");
        if (probe != NULL) {
            fclose(probe);
            if (vit_hip_create_from_planes(&ctx, planes_file, device, chunk) != 0) {
                /* truncated, corrupt or written by another library version: say so and rebuild it from the weight files */
                fprintf(stderr, "%s: %s -- rebuilding it from %s\n", planes_file, vh_last_error(), network_dir);
                ctx = NULL;
            } else if (vit_hip_precision(ctx) != want) {
                /* the file silently winning over $VIT_HIP_PRECISION would be a surprise: the request wins, the file is rewritten */
                fprintf(stderr, "%s holds precision %d, $VIT_HIP_PRECISION asks for %d -- rebuilding it from %s\n", planes_file,
                        vit_hip_precision(ctx), want, network_dir);
                vit_hip_destroy(ctx);
                ctx = NULL;
            } else {
                printf("context from %s: %.4f sec (no weight files read)\n", planes_file, wall() - t0);
            }
        }
        if (ctx == NULL) {
            Network network[NUM_TENSORS];
            vit_config cfg;
            load_weights(network_dir, network, NUM_TENSORS);
            vit_config_preset(&cfg, "vit_b_16");
            if (vit_hip_create(&ctx, &cfg, network, NUM_TENSORS, device, chunk) != 0 || vit_hip_export_planes(ctx, planes_file) != 0) {
                fprintf(stderr, "%s: %s\n", planes_file, vh_last_error());
                return 100;
            }
            printf("context from %s, repacked operands written to %s: %.4f sec\n", network_dir, planes_file, wall() - t0);
        }
        VH_CHECK(vit_hip_forward(ctx, images, n, NULL, probabilities));
        vit_hip_destroy(ctx);
    } else {
        Network network[NUM_TENSORS];
        load_weights(network_dir, network, NUM_TENSORS);
        ViT_opencl(images, network, probabilities);
    }
    printf("Elapsed time: %.4f sec\n", wall() - t0);

    if (vit_write_result_file(result_file, probabilities, n, NUM_CLASSES) != 0) {
        fprintf(stderr, "cannot write %s\n", result_file);
        return 100;
    }

    printf("wrote %d result line(s) to %s\n", n, result_file);
    return 0;
}
