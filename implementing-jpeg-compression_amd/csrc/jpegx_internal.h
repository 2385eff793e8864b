// jpegx_internal.h -- host-side helpers shared by the translation units of libjpegx.so: the
// thread-local error string, argument validation, quantiser parameter tables, RAII for the
// synchronous host-pointer conveniences.  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/jpegx.h"
#include "jpegx_device.h"

namespace jpegx_detail {
extern thread_local char g_err[512];                    // jpegx_runtime.hip
extern thread_local unsigned long long *g_counters;     // jpegx_set_debug_counters
}  // namespace jpegx_detail

namespace {
using jpegx_detail::g_counters;
using jpegx_detail::g_err;

int fail(int code, const char *fmt, const char *detail = "")
{
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            (void)hipGetLastError(); /* reported here: must not linger as the thread's last error */ \
            snprintf(g_err, sizeof(g_err), "%s failed: %s", #expr, hipGetErrorString(e_)); \
            return JPEGX_E_HIP;                                                          \
        }                                                                                \
    } while (0)

int check_plane(const void *in, const void *out, int H, int W, ptrdiff_t pitch, int align_elems)
{
    if (in == nullptr || out == nullptr) return fail(JPEGX_E_INVALID, "null device pointer");
    if (H <= 0 || W <= 0 || (H % 8) != 0 || (W % 8) != 0)
        return fail(JPEGX_E_INVALID, "plane height and width must be positive multiples of 8");
    if (pitch < W) return fail(JPEGX_E_INVALID, "pitch smaller than width");
    if (align_elems > 1 && (pitch % align_elems) != 0)
        return fail(JPEGX_E_INVALID, "pitch must keep rows 16-byte aligned");
    if ((long long)(H / 8) * (long long)(W / 8) > 0x7FFFFFC0LL)
        return fail(JPEGX_E_INVALID, "more than 2^31 blocks in one launch");
    return JPEGX_OK;
}

// Launch geometry of the XCD-private order (xcd_private_wg): default for launches of 2^22 blocks and more
// (16 planes 4096^2) in runs of 128 strips; JPEGX_F_TUNE_XCD_CONTIG forces it, JPEGX_F_TUNE_NO_XCD_CONTIG
// forbids it, bits 20..24 of the flags pick another run length (31 = one run per XCD).  Returns the tune bits
// for QuantParams (0 = natural order) and rounds the grid up to whole runs.
int xcd_order_setup(unsigned flags, int nblk, dim3 *grid)
{
    const bool on = (flags & JPEGX_F_TUNE_XCD_CONTIG) || (nblk >= (1 << 22) && !(flags & JPEGX_F_TUNE_NO_XCD_CONTIG));
    if (!on) return 0;
    int logr = (int)((flags >> 20) & 31u);
    if (logr == 0) logr = 7;
    if (logr >= 31) {
        *grid = dim3(((grid->x + 7) / 8) * 8);
    } else {
        const unsigned run8 = 8u << logr;
        *grid = dim3(((grid->x + run8 - 1) / run8) * run8);
    }
    return 2 | (logr << 8);
}

int aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int is_pow2_float(float f)
{
    int e;
    return f > 0.f && frexpf(f, &e) == 0.5f;
}

// forward table: fp32 reciprocals (fast tier only; the exact tier uses float64 1.0/q or a true division)
int fill_forward_params(int mode, double param, QuantParams *qp)
{
    qp->mode = mode;
    qp->param = param;
    qp->tune = 0;
    switch (mode) {
    case JPEGX_Q_NONE:
        for (int n = 0; n < 64; ++n) qp->rq32[n] = 1.0f;
        return JPEGX_OK;
    case JPEGX_Q_DISCARD: {
        if (!(param >= 0.0) || param != (double)(int)param) return fail(JPEGX_E_INVALID, "discard: keep must be a non-negative integer");
        const int keep = (int)param;
        for (int n = 0; n < 64; ++n) qp->rq32[n] = ((n >> 3) < keep && (n & 7) < keep) ? 1.0f : 0.0f;
        return JPEGX_OK;
    }
    case JPEGX_Q_DIVIDE:
        if (!(param != 0.0) || !(fabs(param) <= 1e30)) return fail(JPEGX_E_INVALID, "divide: divisor must be finite and non-zero");
        for (int n = 0; n < 64; ++n) qp->rq32[n] = (float)(1.0 / param);
        return JPEGX_OK;
    case JPEGX_Q_QTABLE:
        for (int n = 0; n < 64; ++n) qp->rq32[n] = (float)(1.0 / (double)kQT.v[n]);
        return JPEGX_OK;
    default:
        return fail(JPEGX_E_INVALID, "unknown quantiser mode");
    }
}

// The fused forward kernels run the scaled (AAN) transform: coefficient (k, l) comes out multiplied by g_k g_l
// (jpegx_math.h), so their multipliers are the quantiser's reciprocals divided by that -- formed in double and rounded
// to fp32 once, like the plain reciprocals.  Everything that reasons about the quantiser itself (kernel choice, DC
// exactness, the int16 range) looks at the unscaled table from fill_forward_params.
void scale_for_aan(QuantParams *qp)
{
    static const double g[8] = {JPEGX_AAN_G};
    double rq[64];
    switch (qp->mode) {
    case JPEGX_Q_QTABLE:
        for (int n = 0; n < 64; ++n) rq[n] = 1.0 / (double)kQT.v[n];
        break;
    case JPEGX_Q_DIVIDE:
        for (int n = 0; n < 64; ++n) rq[n] = 1.0 / qp->param;
        break;
    default:
        for (int n = 0; n < 64; ++n) rq[n] = (double)qp->rq32[n];      // 1 or 0: exact
    }
    for (int n = 0; n < 64; ++n) qp->rq32[n] = (float)(rq[n] / (g[n >> 3] * g[n & 7]));
}

// inverse table: fp32 multipliers of Quantizer.restore
int fill_inverse_params(int mode, double param, QuantParams *qp)
{
    qp->mode = mode;
    qp->param = param;
    qp->tune = 0;
    switch (mode) {
    case JPEGX_Q_NONE:
    case JPEGX_Q_DISCARD:
        for (int n = 0; n < 64; ++n) qp->rq32[n] = 1.0f;
        return JPEGX_OK;
    case JPEGX_Q_DIVIDE:
        if (!(fabs(param) <= 1e30)) return fail(JPEGX_E_INVALID, "divide: divisor must be finite");
        for (int n = 0; n < 64; ++n) qp->rq32[n] = (float)param;
        return JPEGX_OK;
    case JPEGX_Q_QTABLE:
        for (int n = 0; n < 64; ++n) qp->rq32[n] = (float)kQT.v[n];
        return JPEGX_OK;
    default:
        return fail(JPEGX_E_INVALID, "unknown quantiser mode");
    }
}

extern "C" int jpegx_internal_pool_acquire(size_t in_bytes, size_t out_bytes, void **d_in, void **d_out, void **stream);
extern "C" void jpegx_internal_pool_release(void);

// generic "copy in, run, copy out" helper on the device's pooled stream and buffers (jpegx_hostpipe.cpp):
// no hipMalloc / hipStreamCreate per call once the pool has grown to the working size
template <typename F>
int host_roundtrip(const void *h_in, size_t in_bytes, void *h_out, size_t out_bytes, F &&run)
{
    if (!h_in || !h_out) return fail(JPEGX_E_INVALID, "null host pointer");
    void *din = nullptr, *dout = nullptr, *st = nullptr;
    int rc = jpegx_internal_pool_acquire(in_bytes, out_bytes, &din, &dout, &st);
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(din, h_in, in_bytes, hipMemcpyHostToDevice, (hipStream_t)st);
    if (e == hipSuccess) {
        rc = run(din, dout, (jpegx_stream_t)st);
        if (rc == JPEGX_OK) e = hipMemcpyAsync(h_out, dout, out_bytes, hipMemcpyDeviceToHost, (hipStream_t)st);
    }
    const hipError_t e2 = hipStreamSynchronize((hipStream_t)st);
    jpegx_internal_pool_release();
    if (rc) return rc;
    if (e != hipSuccess || e2 != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "host round trip failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
        return JPEGX_E_HIP;
    }
    return JPEGX_OK;
}

}  // namespace
