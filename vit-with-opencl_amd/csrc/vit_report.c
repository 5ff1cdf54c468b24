/*
 * vit_report.c -- result file in the reference's format and a stricter comparison
 * than the reference's comparator.
 *
 * Main.c:59-72 writes one "[%d] label: %d / prob: %.6f" line per image (with pred_idx
 * declared outside the image loop, so class 0 can leak from one image to the next;
 * here the arg-max restarts per image).  comparator.c:74-86 accepts a result when the
 * label is equal and |dprob| <= 0.01.  vit_compare_rows() reports what that check
 * hides: the largest and mean absolute difference, top-1 agreement, top-1 agreement
 * discounting rows whose reference margin is within twice the stated tolerance (a flipped
 * near-tie is not a wrong answer), and top-5 overlap.  Host code only.
 */
#include "ViT_opencl.h"

#include <math.h>
#include <stdio.h>

static int argmax_row(const float *row, int classes)
{
    int best = 0;
    for (int j = 1; j < classes; ++j)
        if (row[j] > row[best])
            best = j;
    return best;
}

int vit_write_result_file(const char *path, float *const *probabilities, int n, int classes)
{
    if (!path || !probabilities || n < 0 || classes <= 0)
        return 1;
    FILE *f = fopen(path, "w");
    if (!f)
        return 2;
    for (int i = 0; i < n; ++i) {
        const int pred = argmax_row(probabilities[i], classes);
        fprintf(f, "[%d] label: %d / prob: %.6f\n", i, pred, probabilities[i][pred]);
    }
    return fclose(f) == 0 ? 0 : 2;
}

static void top5(const float *row, int classes, int out[5])
{
    for (int k = 0; k < 5; ++k)
        out[k] = -1;
    for (int j = 0; j < classes; ++j) {
        int pos = 5;
        while (pos > 0 && (out[pos - 1] < 0 || row[j] > row[out[pos - 1]]))
            --pos;
        if (pos < 5) {
            for (int k = 4; k > pos; --k)
                out[k] = out[k - 1];
            out[pos] = j;
        }
    }
}

int vit_compare_rows(const float *got, const float *want, int n, int classes, double tolerance,
                     vit_compare_report *rep)
{
    if (!got || !want || !rep || n <= 0 || classes <= 0 || !(tolerance >= 0.0))
        return 1;
    double max_abs = 0.0, sum_abs = 0.0, overlap = 0.0;
    int equal = 0, equal_or_tie = 0, nonfinite = 0;
    for (int i = 0; i < n; ++i) {
        const float *g = got + (size_t)i * classes, *w = want + (size_t)i * classes;
        double row_max = 0.0;
        for (int j = 0; j < classes; ++j) {
            const double d = fabs((double)g[j] - (double)w[j]);
            if (!(d == d) || isinf(d)) {
                ++nonfinite;
                continue;
            }
            sum_abs += d;
            if (d > row_max)
                row_max = d;
        }
        if (row_max > max_abs)
            max_abs = row_max;
        const int ag = argmax_row(g, classes), aw = argmax_row(w, classes);
        if (ag == aw) {
            ++equal;
            ++equal_or_tie;
        } else if ((double)w[aw] - (double)w[ag] <= 2.0 * tolerance) {
            ++equal_or_tie; /* the reference itself separates the two classes by no more than the tolerance allows */
        }
        int tg[5], tw[5];
        top5(g, classes, tg);
        top5(w, classes, tw);
        int common = 0;
        const int kk = classes < 5 ? classes : 5;
        for (int a = 0; a < kk; ++a)
            for (int b = 0; b < kk; ++b)
                common += tg[a] == tw[b];
        overlap += (double)common / kk;
    }
    rep->rows = n;
    rep->classes = classes;
    rep->max_abs_diff = max_abs;
    rep->mean_abs_diff = sum_abs / ((double)n * classes);
    rep->top1_equal = equal;
    rep->top1_equal_or_near_tie = equal_or_tie;
    rep->top5_overlap = overlap / n;
    rep->nonfinite = nonfinite;
    return 0;
}
