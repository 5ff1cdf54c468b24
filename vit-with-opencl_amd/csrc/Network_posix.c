/*
 * Network_posix.c -- POSIX implementations of the reference's two loaders with
 * the same names, argument meaning, on-disk formats and error behaviour, so
 * the reference's Main.c resolves them from this library:
 *
 *   load_image_data  <- reference Network.c:26-109
 *       file = int32 header {n, c, h, w} then n*c*h*w little-endian fp32;
 *       returns an array of n ImageData, each with its own malloc'd buffer and
 *       the same n/c/h/w; NULL (after perror) on any failure.
 *   load_weights     <- reference Network.c:134-218
 *       every "Weight_<idx>_<name>.bin" in `directory` is read whole as fp32,
 *       each value rounded to 6 decimals (roundf(x*1e6f)/1e6f, :208-211) and
 *       stored at network[idx]; missing indices stay {NULL, 0}; an unopenable
 *       directory is perror + exit(EXIT_FAILURE) (:137-141).
 *
 * Deliberate difference: a file that cannot be opened is skipped (the
 * reference calls fseek on the NULL FILE*, Network.c:171-178).
 * Also provides the writers used to build synthetic Data/ and Network/ trees.
 */
#include "Network.h"

#include <dirent.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

ImageData *load_image_data(const char *filename)
{
    FILE *f = fopen(filename, "rb");
    if (f == NULL) {
        perror("load_image_data: open");
        return NULL;
    }
    int header[4];
    if (fread(header, sizeof(int), 4, f) != 4) {
        perror("load_image_data: header");
        fclose(f);
        return NULL;
    }
    const int n = header[0], c = header[1], h = header[2], w = header[3];
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0) {
        fprintf(stderr, "load_image_data: bad header %d %d %d %d\n", n, c, h, w);
        fclose(f);
        return NULL;
    }
    const size_t image_size = (size_t)c * h * w;
    ImageData *images = (ImageData *)malloc((size_t)n * sizeof(ImageData));
    if (images == NULL) {
        perror("load_image_data: malloc");
        fclose(f);
        return NULL;
    }
    for (int i = 0; i < n; ++i) {
        images[i].n = n;
        images[i].c = c;
        images[i].h = h;
        images[i].w = w;
        images[i].data = (float *)malloc(image_size * sizeof(float));
        if (images[i].data == NULL || fread(images[i].data, sizeof(float), image_size, f) != image_size) {
            perror("load_image_data: image data");
            for (int j = 0; j <= i; ++j)
                free(images[j].data);
            free(images);
            fclose(f);
            return NULL;
        }
    }
    fclose(f);
    return images;
}

/* "Weight_<digits>_..." -> digits, or -1 (reference Network.c:111-132). */
static int weight_index(const char *name)
{
    if (strncmp(name, "Weight_", 7) != 0)
        return -1;
    const char *start = name + 7;
    const char *end = strchr(start, '_');
    if (!end || end == start || end - start > 15)
        return -1;
    char buf[16] = {0};
    memcpy(buf, start, (size_t)(end - start));
    return atoi(buf);
}

void load_weights(const char *directory, Network network[], int count)
{
    DIR *dir = opendir(directory);
    if (!dir) {
        perror("load_weights: opendir");
        exit(EXIT_FAILURE);
    }
    for (int i = 0; i < count; ++i) {
        network[i].data = NULL;
        network[i].size = 0;
    }
    struct dirent *entry;
    while ((entry = readdir(dir)) != NULL) {
        const char *ext = strrchr(entry->d_name, '.');
        if (!ext || strcmp(ext, ".bin") != 0)
            continue;
        const int idx = weight_index(entry->d_name);
        if (idx < 0 || idx >= count)
            continue;
        char path[1024];
        snprintf(path, sizeof(path), "%s/%s", directory, entry->d_name);
        FILE *fp = fopen(path, "rb");
        if (!fp)
            continue;
        fseek(fp, 0, SEEK_END);
        const long bytes = ftell(fp);
        rewind(fp);
        if (bytes < 0) {
            fclose(fp);
            continue;
        }
        const size_t n = (size_t)bytes / sizeof(float);
        float *buf = (float *)malloc(n ? n * sizeof(float) : sizeof(float));
        if (!buf) {
            perror("load_weights: malloc");
            fclose(fp);
            exit(EXIT_FAILURE);
        }
        if (fread(buf, sizeof(float), n, fp) != n) {
            perror("load_weights: read");
            free(buf);
            fclose(fp);
            continue;
        }
        fclose(fp);
        for (size_t i = 0; i < n; ++i)
            buf[i] = roundf(buf[i] * 1000000.0f) / 1000000.0f;
        free(network[idx].data);
        network[idx].data = buf;
        network[idx].size = n;
    }
    closedir(dir);
}

/* ---- writers (no reference counterpart; used to build synthetic trees) ---- */

int vit_write_image_file(const char *filename, const ImageData *images, int n)
{
    FILE *f = fopen(filename, "wb");
    if (!f)
        return -1;
    int header[4] = {n, images[0].c, images[0].h, images[0].w};
    const size_t image_size = (size_t)images[0].c * images[0].h * images[0].w;
    int ok = fwrite(header, sizeof(int), 4, f) == 4;
    for (int i = 0; ok && i < n; ++i)
        ok = fwrite(images[i].data, sizeof(float), image_size, f) == image_size;
    return (fclose(f) == 0 && ok) ? 0 : -1;
}

int vit_write_weight_file(const char *directory, int idx, const char *name, const float *data, size_t count)
{
    char path[1024];
    snprintf(path, sizeof(path), "%s/Weight_%d_%s.bin", directory, idx, name);
    FILE *f = fopen(path, "wb");
    if (!f)
        return -1;
    const int ok = fwrite(data, sizeof(float), count, f) == count;
    return (fclose(f) == 0 && ok) ? 0 : -1;
}
