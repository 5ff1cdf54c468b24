/*
 * kernelHandler.hip -- the thin C-ABI runtime shim over HIP.
 *
 * Takes the place of the reference's kernelHandler.c (get_source_code :15,
 * build_error :35, CHECK_ERROR kernelHandler.h:6-10) and of the OpenCL
 * platform/context/queue/buffer plumbing in ViT_opencl.c:125-357,799-861.
 * Kernels are AOT-compiled code objects inside this library, so there is no
 * source loading, no JIT and no dependence on the current directory.
 */
#include "kernelHandler.h"
#include "vit_kernels.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace {
thread_local char g_err[512] = "";
thread_local char g_devname[256] = "";
} // namespace

int vh_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code ? code : 1;
}

int vh_hip_status(hipError_t e, const char *what)
{
    if (e == hipSuccess)
        return 0;
    snprintf(g_err, sizeof(g_err), "%s: %s (%s)", what, hipGetErrorString(e), hipGetErrorName(e));
    return (int)e;
}

extern "C" {

const char *vh_last_error(void) { return g_err; }
int vh_set_error(int code, const char *message) { return vh_fail(code, "%s", message ? message : ""); }

int vh_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int vh_init(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return vh_fail(100, "vh_init: no HIP device available (%s); this library has no CPU fallback",
                       e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    }
    if (device < 0 || device >= n)
        return vh_fail(101, "vh_init: device %d out of range (0..%d)", device, n - 1);
    if (device >= VH_MAX_DEVICES)
        return vh_fail(101, "vh_init: device %d is beyond the %d devices this library keeps launch state for", device,
                       VH_MAX_DEVICES);
    VH_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    VH_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return vh_fail(102, "vh_init: device %d is %s; the kernels in this library are built for gfx950 only",
                       device, prop.gcnArchName);
    /* some driver stacks (no amdgpu.ids file) report an empty marketing name */
    snprintf(g_devname, sizeof(g_devname), "%s (%s, %d CUs)", prop.name[0] ? prop.name : "AMD Instinct accelerator",
             prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}

const char *vh_device_name(void) { return g_devname; }

/* HIP's current device is per host thread: every context entry point re-selects its device. */
int vh_set_device(int device) { VH_TRY(hipSetDevice(device)); return 0; }

int vh_stream_create(vh_stream_t *out)
{
    hipStream_t s;
    VH_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = (vh_stream_t)s;
    return 0;
}
int vh_stream_destroy(vh_stream_t s) { VH_TRY(hipStreamDestroy((hipStream_t)s)); return 0; }
int vh_stream_sync(vh_stream_t s) { VH_TRY(hipStreamSynchronize((hipStream_t)s)); return 0; }
int vh_device_sync(void) { VH_TRY(hipDeviceSynchronize()); return 0; }

int vh_event_create(vh_event_t *out)
{
    hipEvent_t e;
    VH_TRY(hipEventCreate(&e));
    *out = (vh_event_t)e;
    return 0;
}
int vh_event_destroy(vh_event_t e) { VH_TRY(hipEventDestroy((hipEvent_t)e)); return 0; }
int vh_event_record(vh_event_t e, vh_stream_t s) { VH_TRY(hipEventRecord((hipEvent_t)e, (hipStream_t)s)); return 0; }
int vh_stream_wait_event(vh_stream_t s, vh_event_t e)
{
    VH_TRY(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)e, 0));
    return 0;
}
int vh_event_sync(vh_event_t e) { VH_TRY(hipEventSynchronize((hipEvent_t)e)); return 0; }
int vh_event_elapsed_ms(float *ms, vh_event_t start, vh_event_t stop)
{
    VH_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return 0;
}

int vh_malloc(void **out, size_t bytes) { VH_TRY(hipMalloc(out, bytes)); return 0; }
int vh_free(void *p) { VH_TRY(hipFree(p)); return 0; }
int vh_host_alloc(void **out, size_t bytes) { VH_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault)); return 0; }
int vh_host_free(void *p) { VH_TRY(hipHostFree(p)); return 0; }
int vh_memset(void *dst, int value, size_t bytes, vh_stream_t s)
{
    VH_TRY(hipMemsetAsync(dst, value, bytes, (hipStream_t)s));
    return 0;
}
int vh_h2d(void *dst, const void *src, size_t bytes, vh_stream_t s)
{
    VH_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)s));
    return 0;
}
int vh_d2h(void *dst, const void *src, size_t bytes, vh_stream_t s)
{
    VH_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)s));
    return 0;
}
int vh_d2d(void *dst, const void *src, size_t bytes, vh_stream_t s)
{
    VH_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
    return 0;
}

} // extern "C"
