/*
 * vit_gather_rccl.c -- the device-resident multi-GPU forward with the classifier gather over RCCL, in C.
 *
 * The reference takes exactly one device (ViT_opencl.c:803) and serialises images (ViT_opencl.c:926); here the batch
 * is sharded over the devices of one process (vit_hip_create_multi: a full replica per device) and the ONLY exchange
 * of the path is the gather of every shard's [n_g][classes] fp32 logits into one buffer on the first device --
 * grouped ncclSend / ncclRecv between the devices' compute streams (RCCL over xGMI: every peer reaches device 0 in
 * one hop; at 4000 bytes per image the message is ~2 MB per peer).  The host-pointer entry vit_hip_forward_multi needs
 * no collective (its outputs are host arrays); this entry is for callers whose images and results live in HBM.
 *
 * librccl is opened at run time (dlopen) the first time this entry is used: the drop-in ViT_opencl and every
 * single-GPU user carry no dependency on it.  A process that already holds an RCCL (PyTorch bundles one next to its
 * HIP runtime) gets THAT copy, so that the communicators run on the HIP runtime this library is bound to.
 */
#define _GNU_SOURCE
#include "ViT_opencl.h"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef void *rccl_comm_t;
enum { RCCL_FLOAT32 = 7 };   /* ncclFloat32, rccl.h */

struct vit_gather_state
{
    void *lib;
    int (*CommInitAll)(rccl_comm_t *comms, int ndev, const int *devlist);
    int (*CommDestroy)(rccl_comm_t comm);
    int (*GroupStart)(void);
    int (*GroupEnd)(void);
    int (*Send)(const void *buf, size_t count, int dtype, int peer, rccl_comm_t comm, vh_stream_t stream);
    int (*Recv)(void *buf, size_t count, int dtype, int peer, rccl_comm_t comm, vh_stream_t stream);
    const char *(*GetErrorString)(int);
    rccl_comm_t *comms;
    int n;
};

/* ViT_hip.c: where a vit_hip_multi keeps this state, and a context's own logits buffer */
void **vit_hip_multi_gather_slot(vit_hip_multi *m);
float *vit_hip_logits_buffer(vit_hip_ctx *ctx);
int vit_hip_device(const vit_hip_ctx *ctx);
double *vit_hip_multi_enqueue_ms_slot(vit_hip_multi *m);

static int fail(int code, const char *what, const char *detail)
{
    char msg[400];
    snprintf(msg, sizeof(msg), "%s%s%s", what, detail ? ": " : "", detail ? detail : "");
    return vh_set_error(code, msg);
}

void vit_gather_release(void *state)
{
    struct vit_gather_state *g = (struct vit_gather_state *)state;
    if (!g)
        return;
    if (g->comms) {
        for (int i = 0; i < g->n; ++i)
            if (g->comms[i])
                g->CommDestroy(g->comms[i]);
        free(g->comms);
    }
    /* the library stays loaded: RCCL keeps process-wide state behind its communicators */
    free(g);
}

static int gather_state(vit_hip_multi *m, struct vit_gather_state **out)
{
    void **slot = vit_hip_multi_gather_slot(m);
    if (*slot) {
        *out = (struct vit_gather_state *)*slot;
        return 0;
    }
    struct vit_gather_state *g = (struct vit_gather_state *)calloc(1, sizeof(*g));
    if (!g)
        return fail(4, "vit_hip_forward_device_multi: out of host memory", NULL);
    /* an RCCL already in the process first (its HIP runtime is the one this library is bound to), then the system's */
    static const char *names[] = {"librccl.so", "librccl.so.1"};
    for (int i = 0; i < 2 && !g->lib; ++i)
        g->lib = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);
    for (int i = 1; i >= 0 && !g->lib; --i)
        g->lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!g->lib) {
        const char *e = dlerror();
        free(g);
        return fail(110, "vit_hip_forward_device_multi: cannot load librccl", e);
    }
#define SYM(field, name)                                                                  \
    do {                                                                                  \
        *(void **)(&g->field) = dlsym(g->lib, name);                                      \
        if (!g->field) {                                                                  \
            free(g);                                                                      \
            return fail(111, "vit_hip_forward_device_multi: librccl lacks", name);       \
        }                                                                                 \
    } while (0)
    SYM(CommInitAll, "ncclCommInitAll");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    const int n = vit_hip_multi_devices(m);
    int devs[64];
    for (int i = 0; i < n; ++i)
        devs[i] = vit_hip_device(vit_hip_multi_ctx(m, i));
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (devs[i] == devs[j]) {
                free(g);
                return fail(112, "vit_hip_forward_device_multi: a device appears twice (RCCL needs distinct devices per communicator)", NULL);
            }
    g->comms = (rccl_comm_t *)calloc((size_t)n, sizeof(rccl_comm_t));
    g->n = n;
    if (!g->comms) {
        free(g);
        return fail(4, "vit_hip_forward_device_multi: out of host memory", NULL);
    }
    const int rc = g->CommInitAll(g->comms, n, devs);
    if (rc != 0) {
        const char *e = g->GetErrorString(rc);
        free(g->comms);
        free(g);
        return fail(113, "ncclCommInitAll failed", e);
    }
    *slot = g;
    *out = g;
    return 0;
}

/* One device's share of the enqueue stage: its shard's ~7 launches per layer, asynchronous on its own stream. */
struct enqueue_job
{
    vit_hip_multi *m;
    const float *const *d_images;
    const int *counts;
    float *d_logits_root;
};

static int enqueue_shard(void *arg, int g, int lo, int hi)
{
    const struct enqueue_job *j = (const struct enqueue_job *)arg;
    (void)lo;
    (void)hi;
    if (j->counts[g] == 0)
        return 0;
    vit_hip_ctx *ctx = vit_hip_multi_ctx(j->m, g);
    /* shard 0 writes straight into the gathered buffer */
    return vit_hip_forward_device(ctx, j->d_images[g], j->counts[g], g == 0 ? j->d_logits_root : vit_hip_logits_buffer(ctx), NULL,
                                  vit_hip_stream(ctx));
}

/* Wait for everything this call may have put on any device's stream.  Every return path behind the first enqueue goes
 * through here: the entry is synchronous on failure too, so the caller may free d_logits_root and the images. */
static int sync_all(vit_hip_multi *m, int rc)
{
    for (int g = vit_hip_multi_devices(m) - 1; g >= 0; --g) {   /* the root (device 0) last: its stream holds the receives */
        vit_hip_ctx *ctx = vit_hip_multi_ctx(m, g);
        int src = vh_set_device(vit_hip_device(ctx));
        if (src == 0)
            src = vh_stream_sync(vit_hip_stream(ctx));
        if (rc == 0)
            rc = src;
    }
    return rc;
}

/* d_images[g]: shard g's images ([counts[g]][C][H][W] fp32) resident on device g of `m`; d_logits_root / d_probs_root:
 * [sum counts][classes] fp32 on device 0 of `m`, shard after shard (probs may be NULL).  Synchronous on return, on
 * success and on failure alike.
 *
 * The shards are enqueued CONCURRENTLY, one host thread per device (vit_shard_run_timed, as vit_hip_forward_multi
 * does): a forward is ~90 launches, ~0.4 ms of host time, and from one thread device 7 would start ~3 ms late -- nothing
 * at 86 ms per step, a sixth of the fp8 mode's step.  The per-shard host enqueue times of the last call are kept for
 * vit_hip_multi_last_enqueue_ms. */
int vit_hip_forward_device_multi(vit_hip_multi *m, const float *const *d_images, const int *counts, float *d_logits_root,
                                 float *d_probs_root)
{
    if (!m || !d_images || !counts || !d_logits_root)
        return fail(1, "vit_hip_forward_device_multi: null argument", NULL);
    const int n = vit_hip_multi_devices(m);
    int total = 0;
    for (int g = 0; g < n; ++g) {
        if (counts[g] < 0 || (counts[g] > 0 && !d_images[g]) || counts[g] > vit_hip_max_batch(vit_hip_multi_ctx(m, g)))
            return fail(1, "vit_hip_forward_device_multi: shard larger than its context's max_batch, or missing images", NULL);
        total += counts[g];
    }
    if (total == 0)
        return 0;
    struct vit_gather_state *gs = NULL;
    int rc = gather_state(m, &gs);
    if (rc)
        return rc;
    if (!gs)
        return fail(114, "vit_hip_forward_device_multi: no gather state", NULL);
    const size_t NC = (size_t)vit_hip_config(vit_hip_multi_ctx(m, 0))->num_classes;

    /* every shard's forward, asynchronous on its device's stream, entered from its own host thread */
    struct enqueue_job job = {m, d_images, counts, d_logits_root};
    rc = vit_shard_run_timed(n, n, enqueue_shard, &job, vit_hip_multi_enqueue_ms_slot(m));
    if (rc)
        return sync_all(m, rc);   /* the other shards' kernels still write their logits buffers: wait them out */

    /* the gather: grouped point-to-point on the compute streams, so it is ordered behind each shard's kernels */
    if (n > 1) {
        int nrc = gs->GroupStart();
        const int opened = nrc == 0;
        size_t offset = (size_t)counts[0] * NC;
        for (int g = 1; g < n && nrc == 0; ++g) {
            if (counts[g] == 0)
                continue;
            vit_hip_ctx *ctx = vit_hip_multi_ctx(m, g);
            nrc = gs->Recv(d_logits_root + offset, (size_t)counts[g] * NC, RCCL_FLOAT32, g, gs->comms[0],
                           vit_hip_stream(vit_hip_multi_ctx(m, 0)));
            if (nrc == 0)
                nrc = gs->Send(vit_hip_logits_buffer(ctx), (size_t)counts[g] * NC, RCCL_FLOAT32, 0, gs->comms[g], vit_hip_stream(ctx));
            offset += (size_t)counts[g] * NC;
        }
        if (opened) {   /* a group that was opened is always closed, whatever happened inside it */
            const int erc = gs->GroupEnd();
            if (nrc == 0)
                nrc = erc;
        }
        if (nrc != 0)
            return sync_all(m, fail(114, "RCCL gather failed", gs->GetErrorString(nrc)));
    }
    /* class softmax of all rows on the root (miniSoftMax.cl; ViT_seq.c:372-397), behind the receives on its stream */
    vit_hip_ctx *root = vit_hip_multi_ctx(m, 0);
    rc = vh_set_device(vit_hip_device(root));
    if (rc == 0 && d_probs_root)
        rc = vh_launch_softmax(vit_hip_stream(root), d_logits_root, d_probs_root, total, (int)NC);
    return sync_all(m, rc);
}
