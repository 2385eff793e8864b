// jpegx_hostpipe.cpp -- the host-pointer side of libjpegx.so done natively: per-device pools of
// device buffers, a stream and pinned staging memory (grow-only, reused across calls instead of a
// hipMalloc + hipStreamCreate per call), and the whole compress_band job for one plane
// (range check + narrowing of wide integer bands, upload, fused forward, device entropy stage,
// download straight into the caller's bytes) in two C calls.
//
// Replaces, for 8-bit bands with transform 'DCT' / dct_size 8, the loop of pipeline/__init__.py:71-76
// of the reference (steps 1..8 on one band; step 0 padding stays with the caller).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/jpegx.h"
#include "jpegx_entropy_decode.h"

extern "C" void jpegx_internal_set_error(const char *msg);

namespace {

int fail(int code, const char *msg)
{
    jpegx_internal_set_error(msg);
    return code;
}

#define HP_TRY(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            (void)hipGetLastError(); /* reported here: must not linger as the thread's last error */ \
            char buf_[400];                                                                 \
            snprintf(buf_, sizeof(buf_), "%s failed: %s", #expr, hipGetErrorString(e_));    \
            jpegx_internal_set_error(buf_);                                                 \
            return JPEGX_E_HIP;                                                             \
        }                                                                                   \
    } while (0)

// a grow-only allocation (device or pinned host)
struct Span {
    void *p = nullptr;
    size_t cap = 0;
    bool pinned = false;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return JPEGX_OK;
        if (p) {
            if (pinned) (void)hipHostFree(p); else (void)hipFree(p);
            p = nullptr;
            cap = 0;
        }
        const size_t want = bytes + bytes / 8 + 4096;          // head room: similar bands follow
        if (pinned) HP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault)); else HP_TRY(hipMalloc(&p, want));
        cap = want;
        return JPEGX_OK;
    }
};

constexpr int MAX_DEVICES = 16;

struct DevicePool {
    std::mutex mu;                // one host job at a time per device
    hipStream_t stream = nullptr;
    Span d_in, d_zz, d_ws, d_out, d_tmp;
    Span h_in{nullptr, 0, true}, h_out{nullptr, 0, true};
    // state between jpegx_host_compress_begin and _finish (the pool stays locked in between)
    bool open = false;
    size_t out_bytes = 0;
    std::thread::id owner;
};

DevicePool g_pool[MAX_DEVICES];

int current_pool(DevicePool **pool)
{
    int dev = 0;
    HP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) return fail(JPEGX_E_UNSUPPORTED, "device index beyond the pool table");
    *pool = &g_pool[dev];
    return JPEGX_OK;
}

// wide integers -> bytes, checking 0..255 on the way, split over a few host threads (a 4096^2 int64 band is
// 128 MiB: one core needs ~13 ms for it, eight need ~2)
template <typename T>
bool narrow_rows(const T *src, ptrdiff_t src_pitch, int H, int W, uint8_t *dst, ptrdiff_t dst_pitch)
{
    const unsigned hw = std::thread::hardware_concurrency();
    const int nthreads = (int)(((size_t)H * W < (1u << 20)) ? 1 : (hw >= 8 ? 8 : (hw ? hw : 1)));
    std::atomic<bool> ok{true};
    auto work = [&](int y0, int y1) {
        bool good = true;
        for (int y = y0; y < y1; ++y) {
            const T *s = src + (size_t)y * src_pitch;
            uint8_t *d = dst + (size_t)y * dst_pitch;
            T seen = 0;
            for (int x = 0; x < W; ++x) {
                seen |= s[x];
                d[x] = (uint8_t)s[x];
            }
            if (seen & ~(T)0xFF) good = false;               // a negative value or one above 255 in this row
        }
        if (!good) ok = false;
    };
    if (nthreads == 1) {
        work(0, H);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t) th.emplace_back(work, (int)((long long)H * t / nthreads), (int)((long long)H * (t + 1) / nthreads));
        for (auto &t : th) t.join();
    }
    return ok;
}

}  // namespace

extern "C" {

int jpegx_host_compress_begin(const void *h_plane, int elem_size, int H, int W, ptrdiff_t pitch, int bs, int mode,
                              double param, size_t *nbytes)
{
    if (!h_plane || !nbytes) return fail(JPEGX_E_INVALID, "null pointer");
    if (bs < 1 || bs > 255) return fail(JPEGX_E_UNSUPPORTED, "host_compress supports block_size 1..255");
    const bool fused_pool = bs == 1 || bs == 2 || bs == 4;   // uint8 kernels with the mean folded in; else pool to float64 first
    if (H <= 0 || W <= 0 || (H % 8) || (W % 8)) return fail(JPEGX_E_INVALID, "plane height and width (after pooling) must be positive multiples of 8");
    if (elem_size != 1 && elem_size != 4 && elem_size != 8) return fail(JPEGX_E_UNSUPPORTED, "host_compress takes uint8, int32 or int64 samples");
    const int HH = H * bs, WW = W * bs;
    if (pitch < WW) return fail(JPEGX_E_INVALID, "pitch smaller than the row");
    if (fused_pool && (WW % 16) != 0) return fail(JPEGX_E_UNSUPPORTED, "host_compress needs rows of a multiple of 16 samples");
    DevicePool *pool = nullptr;
    int rc = current_pool(&pool);
    if (rc) return rc;
    pool->mu.lock();
    auto bail = [&](int code) { pool->mu.unlock(); return code; };
    if (!pool->stream && hipStreamCreateWithFlags(&pool->stream, hipStreamNonBlocking) != hipSuccess)
        return bail(fail(JPEGX_E_HIP, "hipStreamCreate failed"));
    const size_t in_bytes = (size_t)HH * WW;
    const long long nblocks = (long long)(H / 8) * (W / 8);
    if ((rc = pool->d_in.ensure(in_bytes)) || (rc = pool->d_zz.ensure((size_t)nblocks * 128)) ||
        (rc = pool->d_ws.ensure(jpegx_entropy_workspace_bytes(nblocks))))
        return bail(rc);
    hipStream_t st = pool->stream;
    const uint8_t *src8 = static_cast<const uint8_t *>(h_plane);
    ptrdiff_t src_pitch = pitch;
    if (elem_size != 1) {
        if ((rc = pool->h_in.ensure(in_bytes))) return bail(rc);
        const bool ok = elem_size == 8
            ? narrow_rows(static_cast<const int64_t *>(h_plane), pitch, HH, WW, static_cast<uint8_t *>(pool->h_in.p), WW)
            : narrow_rows(static_cast<const int32_t *>(h_plane), pitch, HH, WW, static_cast<uint8_t *>(pool->h_in.p), WW);
        if (!ok) return bail(fail(JPEGX_E_UNSUPPORTED, "samples outside 0..255: not an 8-bit band"));
        src8 = static_cast<const uint8_t *>(pool->h_in.p);
        src_pitch = WW;
    }
    hipError_t e = (src_pitch == WW)
        ? hipMemcpyAsync(pool->d_in.p, src8, in_bytes, hipMemcpyHostToDevice, st)
        : hipMemcpy2DAsync(pool->d_in.p, WW, src8, (size_t)src_pitch, WW, HH, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return bail(fail(JPEGX_E_HIP, "host to device copy failed"));
    if (fused_pool) {
        rc = jpegx_forward_fused_u8(static_cast<const uint8_t *>(pool->d_in.p), H, W, WW, bs, mode, param, 0,
                                    static_cast<int16_t *>(pool->d_zz.p), st);
    } else {
        // any other block_size: SubSampling on the device in float64 (exact sum, one division), then the
        // all-float64 fused forward -- k/9, k/25 ... are not fp32 numbers
        if ((rc = pool->d_tmp.ensure((size_t)H * W * 8))) return bail(rc);
        rc = jpegx_mean_pool_f64(pool->d_in.p, 1, H, W, WW, bs, static_cast<double *>(pool->d_tmp.p), W, st);
        if (!rc)
            rc = jpegx_forward_fused_f64(static_cast<const double *>(pool->d_tmp.p), H, W, W, mode, param, 0,
                                         static_cast<int16_t *>(pool->d_zz.p), st);
    }
    if (rc || (rc = jpegx_entropy_sizes(static_cast<const int16_t *>(pool->d_zz.p), nblocks, pool->d_ws.p, st)))
        return bail(rc);
    unsigned long long total = 0;
    if ((rc = jpegx_entropy_total(pool->d_ws.p, &total, st))) return bail(rc);      // synchronises
    if ((rc = pool->d_out.ensure(total ? (size_t)total : 1))) return bail(rc);
    if ((rc = jpegx_entropy_emit(static_cast<const int16_t *>(pool->d_zz.p), nblocks, pool->d_ws.p,
                                 static_cast<uint8_t *>(pool->d_out.p), st)))
        return bail(rc);
    pool->open = true;
    pool->out_bytes = (size_t)total;
    pool->owner = std::this_thread::get_id();
    *nbytes = (size_t)total;
    return JPEGX_OK;                       // the pool stays locked until _finish / _abort
}

int jpegx_host_compress_finish(uint8_t *h_out)
{
    DevicePool *pool = nullptr;
    int rc = current_pool(&pool);
    if (rc) return rc;
    if (!pool->open || pool->owner != std::this_thread::get_id()) return fail(JPEGX_E_INVALID, "no open compress job on this thread and device");
    hipError_t e = hipSuccess;
    if (pool->out_bytes) {
        if (!h_out) e = hipErrorInvalidValue;
        if (e == hipSuccess) e = hipMemcpyAsync(h_out, pool->d_out.p, pool->out_bytes, hipMemcpyDeviceToHost, pool->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(pool->stream);
    pool->open = false;
    pool->mu.unlock();
    if (e != hipSuccess) return fail(JPEGX_E_HIP, "device to host copy failed");
    return JPEGX_OK;
}

int jpegx_host_compress_abort(void)
{
    DevicePool *pool = nullptr;
    int rc = current_pool(&pool);
    if (rc) return rc;
    if (pool->open && pool->owner == std::this_thread::get_id()) {
        (void)hipStreamSynchronize(pool->stream);
        pool->open = false;
        pool->mu.unlock();
    }
    return JPEGX_OK;
}

namespace {
// bytes already on the device (pool->d_in, padded) -> int16 stream in pool->d_zz; the caller holds the pool
int decode_on_device(DevicePool *pool, size_t nbytes, long long nblocks)
{
    hipStream_t st = pool->stream;
    int rc;
    if ((rc = pool->d_ws.ensure(jpegx_decode::phase1_bytes(nbytes))) || (rc = pool->d_zz.ensure((size_t)nblocks * 128))) return rc;
    jpegx_decode::enqueue_phase1(static_cast<const uint8_t *>(pool->d_in.p), nbytes, pool->d_ws.p, st);
    unsigned head[4] = {0, 0, 0, 0};
    HP_TRY(hipMemcpyAsync(head, pool->d_ws.p, 16, hipMemcpyDeviceToHost, st));
    HP_TRY(hipStreamSynchronize(st));
    const unsigned ncand = head[0];
    if (ncand == 0 || (long long)ncand < nblocks) return fail(JPEGX_E_INVALID, "entropy stream holds fewer blocks than the plane has");
    if ((rc = pool->d_tmp.ensure(jpegx_decode::phase2_bytes(ncand, nblocks)))) return rc;
    jpegx_decode::enqueue_phase2(static_cast<const uint8_t *>(pool->d_in.p), nbytes, nblocks, pool->d_ws.p, ncand, pool->d_tmp.p,
                                 static_cast<int16_t *>(pool->d_zz.p), st);
    HP_TRY(hipGetLastError());
    return JPEGX_OK;
}

int decode_status(DevicePool *pool)     // after the stream has been synchronised
{
    unsigned head[4] = {0, 0, 0, 0};
    HP_TRY(hipMemcpy(head, pool->d_ws.p, 16, hipMemcpyDeviceToHost));
    if (head[1] != 0) return fail(JPEGX_E_INVALID, "entropy stream is not a sequence of well-formed blocks (device decoder)");
    return JPEGX_OK;
}
}  // namespace

// Inverse of jpegx_host_compress_*: the whole decompress_band job for one plane (pipeline/__init__.py:79-88 for
// transform 'DCT', dct_size 8): bytes up, entropy decoding ON THE DEVICE (jpegx_entropy_decode.hip), fused
// inverse with clamp and SubSampling.invert, uint8 samples down.  h_out: [H*bs][out_pitch] bytes.
static int decompress_plane_locked(DevicePool *pool, const uint8_t *h_bytes, size_t nbytes, int H, int W, int bs, int mode,
                                   double param, uint8_t *h_out, ptrdiff_t out_pitch)
{
    if (!h_bytes || !h_out) return fail(JPEGX_E_INVALID, "null host pointer");
    if (H <= 0 || W <= 0 || (H % 8) || (W % 8)) return fail(JPEGX_E_INVALID, "plane height and width must be positive multiples of 8");
    if (bs != 1 && bs != 2 && bs != 4) return fail(JPEGX_E_UNSUPPORTED, "host_decompress supports block_size 1, 2 and 4");
    if (nbytes == 0 || nbytes >= 0xFFFFFFF0ull) return fail(JPEGX_E_INVALID, "entropy stream empty or beyond 4 GiB");
    if (out_pitch < (ptrdiff_t)W * bs || (out_pitch % (bs == 1 ? 8 : 16)) != 0) return fail(JPEGX_E_INVALID, "output pitch too small or misaligned");
    const long long nblocks = (long long)(H / 8) * (W / 8);
    int rc;
    if (!pool->stream && hipStreamCreateWithFlags(&pool->stream, hipStreamNonBlocking) != hipSuccess) return fail(JPEGX_E_HIP, "hipStreamCreate failed");
    hipStream_t st = pool->stream;
    const size_t out_bytes = (size_t)H * bs * out_pitch;
    if ((rc = pool->d_in.ensure(nbytes + 16)) || (rc = pool->d_out.ensure(out_bytes))) return rc;
    HP_TRY(hipMemsetAsync(static_cast<uint8_t *>(pool->d_in.p) + (nbytes & ~(size_t)3), 0, 16 + (nbytes & 3), st));   // zero tail (whole dwords)
    HP_TRY(hipMemcpyAsync(pool->d_in.p, h_bytes, nbytes, hipMemcpyHostToDevice, st));
    if ((rc = decode_on_device(pool, nbytes, nblocks))) return rc;
    if ((rc = jpegx_inverse_fused_u8_inflated(static_cast<const int16_t *>(pool->d_zz.p), H, W, mode, param, 0, bs,
                                              static_cast<uint8_t *>(pool->d_out.p), out_pitch, st)))
        return rc;
    HP_TRY(hipMemcpyAsync(h_out, pool->d_out.p, out_bytes, hipMemcpyDeviceToHost, st));
    HP_TRY(hipStreamSynchronize(st));
    return decode_status(pool);
}

int jpegx_host_decompress_plane(const uint8_t *h_bytes, size_t nbytes, int H, int W, int bs, int mode, double param,
                                uint8_t *h_out, ptrdiff_t out_pitch)
{
    DevicePool *pool = nullptr;
    int rc = current_pool(&pool);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(pool->mu);
    return decompress_plane_locked(pool, h_bytes, nbytes, H, W, bs, mode, param, h_out, out_pitch);
}

// The same, handing back what the reference's decompress_band returns: a [rows][cols] int64 array (the band
// cropped to its configured size).  The uint8 samples come down into pinned staging memory and are widened
// by a few host threads -- NumPy's astype(int) on a 4096 x 4096 band costs more than the device pipeline.
int jpegx_host_decompress_plane_i64(const uint8_t *h_bytes, size_t nbytes, int H, int W, int bs, int mode, double param,
                                    int64_t *h_out, int rows, int cols)
{
    if (!h_out || rows <= 0 || cols <= 0 || H <= 0 || W <= 0 || bs <= 0 || rows > (long long)H * bs || cols > (long long)W * bs)
        return fail(JPEGX_E_INVALID, "bad output shape");
    const ptrdiff_t pitch = ((ptrdiff_t)W * bs + 15) / 16 * 16;
    const size_t stage_bytes = (size_t)H * bs * pitch;
    DevicePool *pool = nullptr;
    int rc = current_pool(&pool);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(pool->mu);          // held to the end: the staging span belongs to this job
    if ((rc = pool->h_out.ensure(stage_bytes))) return rc;
    uint8_t *stage = static_cast<uint8_t *>(pool->h_out.p);
    if ((rc = decompress_plane_locked(pool, h_bytes, nbytes, H, W, bs, mode, param, stage, pitch))) return rc;
    const unsigned hw = std::thread::hardware_concurrency();
    const int nthreads = ((size_t)rows * cols < (1u << 20)) ? 1 : (hw >= 8 ? 8 : (hw ? (int)hw : 1));
    auto work = [&](int y0, int y1) {
        for (int y = y0; y < y1; ++y) {
            const uint8_t *srow = stage + (size_t)y * pitch;
            int64_t *drow = h_out + (size_t)y * cols;
            for (int x = 0; x < cols; ++x) drow[x] = srow[x];
        }
    };
    if (nthreads == 1) {
        work(0, rows);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t) th.emplace_back(work, (int)((long long)rows * t / nthreads), (int)((long long)rows * (t + 1) / nthreads));
        for (auto &t : th) t.join();
    }
    return JPEGX_OK;
}

// bytes -> int16 [nblocks][64] on the device, host arrays in and out (what jpegx_host_entropy_decode does on the CPU)
int jpegx_host_entropy_decode_gpu(const uint8_t *h_bytes, size_t nbytes, long long nblocks, int16_t *h_zz)
{
    if (!h_bytes || !h_zz) return fail(JPEGX_E_INVALID, "null host pointer");
    if (nblocks <= 0 || nblocks > 0x7FFFFFC0LL) return fail(JPEGX_E_INVALID, "block count must be in 1 .. 2^31-64");
    if (nbytes == 0 || nbytes >= 0xFFFFFFF0ull) return fail(JPEGX_E_INVALID, "entropy stream empty or beyond 4 GiB");
    DevicePool *pool = nullptr;
    int rc = current_pool(&pool);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(pool->mu);
    if (!pool->stream && hipStreamCreateWithFlags(&pool->stream, hipStreamNonBlocking) != hipSuccess) return fail(JPEGX_E_HIP, "hipStreamCreate failed");
    hipStream_t st = pool->stream;
    if ((rc = pool->d_in.ensure(nbytes + 16))) return rc;
    HP_TRY(hipMemsetAsync(static_cast<uint8_t *>(pool->d_in.p) + (nbytes & ~(size_t)3), 0, 16 + (nbytes & 3), st));
    HP_TRY(hipMemcpyAsync(pool->d_in.p, h_bytes, nbytes, hipMemcpyHostToDevice, st));
    if ((rc = decode_on_device(pool, nbytes, nblocks))) return rc;
    HP_TRY(hipMemcpyAsync(h_zz, pool->d_zz.p, (size_t)nblocks * 128, hipMemcpyDeviceToHost, st));
    HP_TRY(hipStreamSynchronize(st));
    return decode_status(pool);
}

// used by host_roundtrip (jpegx_internal.h): the synchronous host-pointer conveniences borrow the pool's
// stream and its input / output device spans for the duration of one call; not part of the public ABI
int jpegx_internal_pool_acquire(size_t in_bytes, size_t out_bytes, void **d_in, void **d_out, void **stream)
{
    DevicePool *pool = nullptr;
    int rc = current_pool(&pool);
    if (rc) return rc;
    pool->mu.lock();
    if (!pool->stream && hipStreamCreateWithFlags(&pool->stream, hipStreamNonBlocking) != hipSuccess) {
        pool->mu.unlock();
        return fail(JPEGX_E_HIP, "hipStreamCreate failed");
    }
    if ((rc = pool->d_in.ensure(in_bytes ? in_bytes : 1)) || (rc = pool->d_out.ensure(out_bytes ? out_bytes : 1))) {
        pool->mu.unlock();
        return rc;
    }
    *d_in = pool->d_in.p;
    *d_out = pool->d_out.p;
    *stream = pool->stream;
    return JPEGX_OK;
}

void jpegx_internal_pool_release(void)
{
    DevicePool *pool = nullptr;
    if (current_pool(&pool) == JPEGX_OK) pool->mu.unlock();
}

// release everything the pools hold on the current device (tests; long-lived processes that are done)
int jpegx_host_pool_release(void)
{
    DevicePool *pool = nullptr;
    int rc = current_pool(&pool);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(pool->mu);
    for (Span *s : {&pool->d_in, &pool->d_zz, &pool->d_ws, &pool->d_out, &pool->d_tmp, &pool->h_in, &pool->h_out}) {
        if (s->p) { if (s->pinned) (void)hipHostFree(s->p); else (void)hipFree(s->p); }
        s->p = nullptr;
        s->cap = 0;
    }
    if (pool->stream) { (void)hipStreamDestroy(pool->stream); pool->stream = nullptr; }
    return JPEGX_OK;
}

}  // extern "C"
