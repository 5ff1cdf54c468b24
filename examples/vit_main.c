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
 * $VIT_HIP_PRECISION's arithmetic and the file is written (vit_hip_export_planes) for the next run.
 *
 * Result lines have Main.c's format ("[%d] label: %d / prob: %.6f", Main.c:71); the
 * arg-max restarts for every image (Main.c:59 declares pred_idx outside the loop, so
 * class 0 can leak from image to image there).
 * Exit status 0, or 100 on I/O trouble.
 */
#include <stdio.h>
#include <stdlib.h>
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
    float **probabilities = (float **)malloc(sizeof(float *) * (size_t)n);
    for (int i = 0; i < n; ++i)
        probabilities[i] = (float *)malloc(sizeof(float) * NUM_CLASSES);

    printf("=====================Start========================\n");
    const double t0 = wall();
    if (planes_file != NULL) {
        const int chunk = n < 512 ? n : 512;
        vit_hip_ctx *ctx = NULL;
        FILE *probe = fopen(planes_file, "rb");
        if (probe != NULL) {
            fclose(probe);
            if (vit_hip_create_from_planes(&ctx, planes_file, 0, chunk) != 0) {
                fprintf(stderr, "%s: %s\n", planes_file, vh_last_error());
                return 100;
            }
            printf("context from %s: %.4f sec (no weight files read)\n", planes_file, wall() - t0);
        } else {
            Network network[NUM_TENSORS];
            vit_config cfg;
            load_weights(network_dir, network, NUM_TENSORS);
            vit_config_preset(&cfg, "vit_b_16");
            if (vit_hip_create(&ctx, &cfg, network, NUM_TENSORS, 0, chunk) != 0 || vit_hip_export_planes(ctx, planes_file) != 0) {
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
