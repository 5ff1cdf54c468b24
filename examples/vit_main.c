/*
 * vit_main.c -- a POSIX driver with the same flow as the reference's Main.c
 * (Main.c:16-92) and the same acceptance check as its comparator.c, written against
 * this repository's headers.  It exists to exercise the drop-in boundary end to end
 * from C: load_image_data -> load_weights -> ViT_opencl -> result file -> compare.
 *
 *   vit_main [image_file] [network_dir] [result_file] [answer_file]
 *   defaults: ./Data/input-100.bin ./Network ./Data/opencl_result.txt ./Data/answer_result.txt
 *
 * Result lines have Main.c's format ("[%d] label: %d / prob: %.6f", Main.c:71); the
 * arg-max restarts for every image (Main.c:59 declares pred_idx outside the loop, so
 * class 0 can leak from image to image there).
 * The check follows comparator.c:74-86: one error per label mismatch and one per
 * |dprob| > 0.01; exit status = number of errors (0 = pass), 100 on I/O trouble.
 */
#include <math.h>
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

static int compare_files(const char *result, const char *answer, int count)
{
    FILE *fr = fopen(result, "r"), *fa = fopen(answer, "r");
    if (!fr || !fa) {
        fprintf(stderr, "compare: cannot open %s or %s\n", result, answer);
        if (fr) fclose(fr);
        if (fa) fclose(fa);
        return 100;
    }
    int errors = 0;
    char lr[1024], la[1024];
    for (int line = 0; line < count; ++line) {
        int label_r, label_a;
        float prob_r, prob_a;
        if (!fgets(lr, sizeof lr, fr) || !fgets(la, sizeof la, fa)) {
            fprintf(stderr, "Line %d: not enough lines\n", line);
            ++errors;
            break;
        }
        if (sscanf(lr, "[%*d] label: %d / prob: %f", &label_r, &prob_r) != 2 ||
            sscanf(la, "[%*d] label: %d / prob: %f", &label_a, &prob_a) != 2) {
            fprintf(stderr, "Line %d: parse error\n", line);
            ++errors;
            continue;
        }
        if (label_r != label_a) {
            fprintf(stderr, "Line %d: label mismatch (result %d, answer %d)\n", line, label_r, label_a);
            ++errors;
        }
        if (fabs(prob_r - prob_a) > 0.01f) {
            fprintf(stderr, "Line %d: probability mismatch (result %.6f, answer %.6f)\n", line, prob_r, prob_a);
            ++errors;
        }
    }
    fclose(fr);
    fclose(fa);
    return errors;
}

int main(int argc, char **argv)
{
    const char *image_file = argc > 1 ? argv[1] : "./Data/input-100.bin";
    const char *network_dir = argc > 2 ? argv[2] : "./Network";
    const char *result_file = argc > 3 ? argv[3] : "./Data/opencl_result.txt";
    const char *answer_file = argc > 4 ? argv[4] : "./Data/answer_result.txt";

    ImageData *images = load_image_data(image_file);
    if (images == NULL)
        return 100;
    Network network[NUM_TENSORS];
    load_weights(network_dir, network, NUM_TENSORS);

    const int n = images->n;
    float **probabilities = (float **)malloc(sizeof(float *) * (size_t)n);
    for (int i = 0; i < n; ++i)
        probabilities[i] = (float *)malloc(sizeof(float) * NUM_CLASSES);

    printf("=====================Start========================\n");
    const double t0 = wall();
    ViT_opencl(images, network, probabilities);
    printf("Elapsed time: %.4f sec\n", wall() - t0);

    if (vit_write_result_file(result_file, probabilities, n, NUM_CLASSES) != 0) {
        fprintf(stderr, "cannot write %s\n", result_file);
        return 100;
    }

    const int errors = compare_files(result_file, answer_file, n);
    if (errors == 0)
        printf("Comparator: result and answer agree on all %d images.\n", n);
    else
        printf("Comparator: %d difference(s).\n", errors);
    return errors;
}
