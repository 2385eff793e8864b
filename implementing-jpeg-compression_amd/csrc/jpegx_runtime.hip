// jpegx_runtime.hip -- library / device management of libjpegx.so: error string, init, memory,
// streams, events, debug counters.  Part of the C ABI (include/jpegx.h).
#include "jpegx_internal.h"

namespace jpegx_detail {
thread_local char g_err[512] = "";
thread_local unsigned long long *g_counters = nullptr;
}  // namespace jpegx_detail

extern "C" {

// used by the other translation units of libjpegx.so (jpegx_entropy.hip); not part of the public ABI
void jpegx_internal_set_error(const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg ? msg : ""); }

const char *jpegx_last_error(void) { return g_err; }
int jpegx_version(void) { return JPEGX_VERSION; }

int jpegx_init(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return fail(JPEGX_E_NODEVICE, "no usable HIP device");
    if (device < 0 || device >= n) return fail(JPEGX_E_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(nullptr));   // forces context creation
    return JPEGX_OK;
}

int jpegx_shutdown(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return JPEGX_OK;
    HIP_TRY(hipDeviceSynchronize());
    return JPEGX_OK;
}

int jpegx_device_count(int *count)
{
    if (!count) return fail(JPEGX_E_INVALID, "null count pointer");
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        *count = 0;
        return fail(JPEGX_E_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    return JPEGX_OK;
}

int jpegx_set_device(int device) { HIP_TRY(hipSetDevice(device)); return JPEGX_OK; }
int jpegx_get_device(int *device)
{
    if (!device) return fail(JPEGX_E_INVALID, "null device pointer");
    HIP_TRY(hipGetDevice(device));
    return JPEGX_OK;
}

namespace {
// current device of the calling thread switched to `device` for the lifetime of the object
struct DeviceScope {
    int prev = -1, rc = JPEGX_OK;
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; rc = fail(JPEGX_E_HIP, "hipGetDevice failed"); return; }
        if (device != prev && hipSetDevice(device) != hipSuccess) {
            (void)hipGetLastError();          // the refused index must not linger as the thread's "last error"
            prev = -1;
            rc = fail(JPEGX_E_INVALID, "no such device");
        }
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace

int jpegx_forward_fused_on(int device, const float *d_in, int H, int W, ptrdiff_t pitch, int mode, double param,
                           unsigned flags, int16_t *d_out, jpegx_stream_t stream)
{
    DeviceScope scope(device);
    return scope.rc ? scope.rc : jpegx_forward_fused(d_in, H, W, pitch, mode, param, flags, d_out, stream);
}

int jpegx_inverse_fused_on(int device, const int16_t *d_in, int H, int W, int mode, double param, unsigned flags,
                           void *d_out, ptrdiff_t out_pitch, int out_type, jpegx_stream_t stream)
{
    DeviceScope scope(device);
    return scope.rc ? scope.rc : jpegx_inverse_fused(d_in, H, W, mode, param, flags, d_out, out_pitch, out_type, stream);
}

int jpegx_device_name(int device, char *buf, size_t buflen)
{
    if (!buf || buflen == 0) return fail(JPEGX_E_INVALID, "null name buffer");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return JPEGX_OK;
}

int jpegx_device_synchronize(void) { HIP_TRY(hipDeviceSynchronize()); return JPEGX_OK; }

int jpegx_malloc(void **dptr, size_t bytes)
{
    if (!dptr) return fail(JPEGX_E_INVALID, "null pointer");
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 1));
    return JPEGX_OK;
}
int jpegx_free(void *dptr) { HIP_TRY(hipFree(dptr)); return JPEGX_OK; }
int jpegx_memset(void *dptr, int value, size_t bytes, jpegx_stream_t stream)
{
    HIP_TRY(hipMemsetAsync(dptr, value, bytes, (hipStream_t)stream));
    return JPEGX_OK;
}
int jpegx_memcpy_h2d(void *dst, const void *src, size_t bytes, jpegx_stream_t stream)
{
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return JPEGX_OK;
}
int jpegx_memcpy_d2h(void *dst, const void *src, size_t bytes, jpegx_stream_t stream)
{
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return JPEGX_OK;
}
int jpegx_memcpy_d2d(void *dst, const void *src, size_t bytes, jpegx_stream_t stream)
{
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return JPEGX_OK;
}

int jpegx_stream_create(jpegx_stream_t *stream)
{
    if (!stream) return fail(JPEGX_E_INVALID, "null pointer");
    hipStream_t s;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (jpegx_stream_t)s;
    return JPEGX_OK;
}
int jpegx_stream_destroy(jpegx_stream_t stream) { HIP_TRY(hipStreamDestroy((hipStream_t)stream)); return JPEGX_OK; }
int jpegx_stream_synchronize(jpegx_stream_t stream) { HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); return JPEGX_OK; }
int jpegx_event_create(jpegx_event_t *event)
{
    if (!event) return fail(JPEGX_E_INVALID, "null pointer");
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    *event = (jpegx_event_t)e;
    return JPEGX_OK;
}
int jpegx_event_destroy(jpegx_event_t event) { HIP_TRY(hipEventDestroy((hipEvent_t)event)); return JPEGX_OK; }
int jpegx_event_record(jpegx_event_t event, jpegx_stream_t stream)
{
    HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return JPEGX_OK;
}
int jpegx_stream_wait_event(jpegx_stream_t stream, jpegx_event_t event)
{
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return JPEGX_OK;
}
int jpegx_event_synchronize(jpegx_event_t event) { HIP_TRY(hipEventSynchronize((hipEvent_t)event)); return JPEGX_OK; }
int jpegx_event_elapsed_ms(jpegx_event_t start, jpegx_event_t stop, float *ms)
{
    if (!ms) return fail(JPEGX_E_INVALID, "null pointer");
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return JPEGX_OK;
}

int jpegx_set_debug_counters(unsigned long long *d_counters)
{
    g_counters = d_counters;
    return JPEGX_OK;
}
}  // extern "C"
