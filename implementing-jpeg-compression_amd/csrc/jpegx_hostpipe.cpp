// jpegx_hostpipe.cpp -- the host-pointer side of libjpegx.so done natively: per-device pools of
// device buffers, streams and pinned staging memory (grow-only, reused across calls instead of a
// hipMalloc + hipStreamCreate per call), the whole compress_band / decompress_band job for one plane
// (range check + narrowing of wide integer bands, upload, fused forward, device entropy stage,
// download straight into the caller's bytes) and, round 3, the whole-IMAGE jobs: the bands of one
// picture through one lock on two alternating streams, so that the upload and transform of band k + 1
// overlap the entropy stage and the download of band k.
//
// Replaces, for 8-bit bands with transform 'DCT' / dct_size 8, the loops of pipeline/__init__.py:71-76,
// 79-88 (one band) and :102-124 (Jpeg.compress / Jpeg.decompress: Y, Cb, Cr one after another) of the
// reference; step 0 padding stays with the caller.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/mman.h>

#include <atomic>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/jpegx.h"
#include "jpegx_entropy_decode.h"

extern "C" void jpegx_internal_set_error(const char *msg);
// the uint8 forward kernels sizing their own blocks for the entropy stage (jpegx_forward.hip, jpegx_entropy.hip)
extern "C" int jpegx_internal_forward_u8_sized(const uint8_t *d_in, int H, int W, ptrdiff_t pitch, int bs, int mode, double param, unsigned flags,
                                               int16_t *d_out, unsigned *block_bytes, unsigned *wave_bytes, unsigned *half_info, jpegx_stream_t stream);
extern "C" void jpegx_internal_entropy_views(void *d_workspace, long long nblocks, unsigned **block_bytes, unsigned **wave_bytes, unsigned **half_info);
extern "C" int jpegx_internal_entropy_emit2(const int16_t *d_zz, long long nblocks, const void *d_workspace, uint8_t *d_out, jpegx_stream_t stream);
extern "C" int jpegx_internal_entropy_scan(long long nblocks, void *d_workspace, jpegx_stream_t stream);

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
    void release()
    {
        if (p) { if (pinned) (void)hipHostFree(p); else (void)hipFree(p); }
        p = nullptr;
        cap = 0;
    }
};

constexpr int MAX_DEVICES = 16;
constexpr int MAX_BANDS = JPEGX_MAX_IMAGE_BANDS;

// the device-side working set of one band
struct BandSlot {
    Span d_in, d_zz, d_ws, d_out, d_tmp;
    Span d_seg, d_seg_state;      // the segmented entropy decoder's scratch, and its state: that stays clean from call to call
    void *seg_clean = nullptr;    // the allocation that has been cleared
    size_t seg_clean_cap = 0;
    unsigned seg_calls = 0;
    int seg_parity = -1;          // status block of the last decode (-1: the general scheme ran, status in d_ws)
    bool sized = false;           // compress: the forward kernel sized the blocks itself (half_info is there)
    // decoder, first try: candidates that cannot start a block of non-negative samples are dropped (jpegx_entropy_decode.hip).
    // A stream with blocks that do start otherwise (DC 0 beside non-zero AC: very dark content) misses there and takes the
    // second try; the filter then stays off for this working set's next calls, so that such content pays once in a while
    unsigned filter_pause = 0;
    bool last_filter = false;
};

// One job context: a stream set and grow-only buffers.  A device has POOL_CONTEXTS of them, so that jobs of several host
// threads overlap on the device (one's upload under another's kernels and download) instead of queueing behind one lock;
// a single-threaded caller only ever meets the first (the others stay empty).
struct DevicePool {
    std::mutex mu;                // one host job at a time per context
    hipStream_t stream = nullptr; // single-band jobs and the host-pointer conveniences
    hipStream_t aux[2] = {nullptr, nullptr};   // image jobs: bands alternate between these two
    hipEvent_t ev[MAX_BANDS] = {};
    hipEvent_t ev_x = nullptr;    // image jobs from packed pixels: the planes are there
    BandSlot slot[MAX_BANDS];     // slot 0 doubles as the single-band working set
    Span d_packed;                // image jobs: the pixel-interleaved picture before it goes down
    Span h_in{nullptr, 0, true}, h_out{nullptr, 0, true}, h_head{nullptr, 0, true};
    // state between jpegx_host_compress_begin and _finish (the pool stays locked in between)
    bool open = false;
    size_t out_bytes = 0;
};

constexpr int POOL_CONTEXTS = 4;
DevicePool g_pool[MAX_DEVICES][POOL_CONTEXTS];

// The pool this thread holds across C calls (an open compress job) or inside one (a borrowed working set).  Every
// pooled entry asks here first: a thread that already holds the pool gets JPEGX_E_INVALID instead of locking the
// non-recursive mutex a second time, and finish / abort / release find THEIR pool whatever the thread's current
// device has become in the meantime.
thread_local DevicePool *t_held = nullptr;
thread_local int t_held_device = -1;

// lock a job context of the current device for this thread (released by unlock_pool): the first one that is free, or --
// all busy -- the one this thread's id hashes to
int lock_pool(DevicePool **out)
{
    int dev = 0;
    HP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) return fail(JPEGX_E_UNSUPPORTED, "device index beyond the pool table");
    if (t_held != nullptr)
        return fail(JPEGX_E_INVALID, t_held->open ? "a compress job is open on this thread: finish or abort it first"
                                                   : "this thread already holds a device pool");
    DevicePool *pool = nullptr;
    for (int k = 0; k < POOL_CONTEXTS && !pool; ++k)
        if (g_pool[dev][k].mu.try_lock()) pool = &g_pool[dev][k];
    if (!pool) {
        pool = &g_pool[dev][std::hash<std::thread::id>()(std::this_thread::get_id()) % POOL_CONTEXTS];
        pool->mu.lock();
    }
    t_held = pool;
    t_held_device = dev;
    *out = pool;
    return JPEGX_OK;
}

void unlock_pool(DevicePool *pool)
{
    if (t_held == pool) {
        t_held = nullptr;
        t_held_device = -1;
        pool->mu.unlock();
    }
}

struct PoolLock {     // scope form
    DevicePool *pool = nullptr;
    int rc;
    PoolLock() { rc = lock_pool(&pool); }
    ~PoolLock() { if (pool) unlock_pool(pool); }
};

int ensure_streams(DevicePool *pool, bool image)
{
    if (!pool->stream) HP_TRY(hipStreamCreateWithFlags(&pool->stream, hipStreamNonBlocking));
    if (!pool->ev[0]) HP_TRY(hipEventCreateWithFlags(&pool->ev[0], hipEventDisableTiming));      // band jobs wait on it for the stream's size
    if (image) {
        for (int i = 0; i < 2; ++i)
            if (!pool->aux[i]) HP_TRY(hipStreamCreateWithFlags(&pool->aux[i], hipStreamNonBlocking));
        for (int i = 0; i < MAX_BANDS; ++i)
            if (!pool->ev[i]) HP_TRY(hipEventCreateWithFlags(&pool->ev[i], hipEventDisableTiming));
        if (!pool->ev_x) HP_TRY(hipEventCreateWithFlags(&pool->ev_x, hipEventDisableTiming));
    }
    return JPEGX_OK;
}

// Wide integers -> bytes in the pinned staging area, checking 0..255 on the way, and up to the device -- in strips, each
// uploaded as soon as it is narrowed: the threads walk the strips together (every thread its share of the rows of strip
// k, then of strip k + 1), the calling thread waits for a strip's last share and enqueues its copy, so only the last
// strip's copy is not hidden behind the narrowing of the next.  What the narrowing costs is reading the wide array:
// 128 MiB for a 4096^2 int64 band, ~13 ms on one core (JPEGX_NARROW_THREADS, default 8).
template <typename T>
int narrow_and_upload(const T *src, ptrdiff_t src_pitch, int H, int W, uint8_t *stage, void *d_dst, hipStream_t st)
{
    const unsigned hw = std::thread::hardware_concurrency();
    static const int want = [] { const char *e = getenv("JPEGX_NARROW_THREADS"); return e && *e ? atoi(e) : 8; }();      // 8 / 16 / 32: 1.26-1.57 / 1.38-1.99 / 1.33-1.94 ms per int64 band (starting the threads costs more than sixteen gain)
    const int nthreads = ((size_t)H * W < (1u << 20)) ? 1 : ((int)hw >= want ? (want > 0 ? want : 1) : (hw ? (int)hw : 1));
    const int nstrips = nthreads == 1 ? 1 : 8;
    std::atomic<bool> ok{true};
    std::vector<std::atomic<int>> done(nstrips);
    for (auto &d : done) d.store(0, std::memory_order_relaxed);
    auto rows_of = [&](int k) { return (int)((long long)H * k / nstrips); };
    auto work = [&](int t) {
        bool good = true;
        for (int k = 0; k < nstrips; ++k) {
            const int r0 = rows_of(k), r1 = rows_of(k + 1);
            const int y0 = r0 + (int)((long long)(r1 - r0) * t / nthreads), y1 = r0 + (int)((long long)(r1 - r0) * (t + 1) / nthreads);
            for (int y = y0; y < y1; ++y) {
                const T *sp = src + (size_t)y * src_pitch;
                uint8_t *d = stage + (size_t)y * W;
                T seen = 0;
                for (int x = 0; x < W; ++x) {
                    seen |= sp[x];
                    d[x] = (uint8_t)sp[x];
                }
                if (seen & ~(T)0xFF) good = false;           // a negative value or one above 255 in this row
            }
            done[k].fetch_add(1, std::memory_order_release);
        }
        if (!good) ok = false;
    };
    std::vector<std::thread> th;
    hipError_t e = hipSuccess;
    if (nthreads == 1) work(0);                               // a small band: here and now
    else for (int t = 0; t < nthreads; ++t) th.emplace_back(work, t);
    for (int k = 0; k < nstrips; ++k) {                       // the calling thread: wait for the strip, enqueue its copy
        while (done[k].load(std::memory_order_acquire) < nthreads) __builtin_ia32_pause();
        const int r0 = rows_of(k), r1 = rows_of(k + 1);
        if (e == hipSuccess && r1 > r0)
            e = hipMemcpyAsync(static_cast<uint8_t *>(d_dst) + (size_t)r0 * W, stage + (size_t)r0 * W, (size_t)(r1 - r0) * W, hipMemcpyHostToDevice, st);
    }
    for (auto &t : th) t.join();
    if (!ok) {
        (void)hipStreamSynchronize(st);                       // the strips already on their way read the staging area: let them finish
        return fail(JPEGX_E_UNSUPPORTED, "samples outside 0..255: not an 8-bit band");
    }
    if (e != hipSuccess) return fail(JPEGX_E_HIP, "host to device copy failed");
    return JPEGX_OK;
}

// Touch every page of a (usually fresh) result buffer with a few host threads: the kernel hands out zeroed pages one
// fault at a time, and left to the copy that lands in them this costs more than the copy (the 7-10 ms outliers of
// decompress_band in round 2's profiles were exactly that: the first touch of a 128 MiB NumPy result).
void prefault(uint8_t *p, size_t n)
{
    if (n < (4u << 20)) return;
    // What a fresh result buffer costs is the operating system's, and it depends on the page size (microbench/pagefault.cpp
    // on the GPU box, profiles/r03_pagefault.txt): 72 MB of 4 KiB pages 8.7 ms touched by one thread, 4.4-5.8 ms by eight
    // (the faults of one address space serialise on its lock), 0.7 ms as 2 MiB pages touched by eight threads in
    // interleaved 2 MiB grains.  NumPy asks for huge pages itself (its 128 MiB arrays: 64 faults); a Python bytes object
    // comes from malloc -> mmap without that advice, so it is given here for the aligned interior of the destination
    // (transparent_hugepage is in `madvise` mode on these hosts; where it is `never` the call changes nothing).
    // JPEGX_PREFAULT: touching threads (default 8; 0 = leave the faults to the copy; -1 MADV_POPULATE_WRITE, -2 with huge
    // pages); JPEGX_PREFAULT_HUGE=0: no advice (A/B, profiles/r03_prefault_ab.txt).
    static const int mode = [] { const char *e = getenv("JPEGX_PREFAULT"); return e && *e ? atoi(e) : 8; }();
    static const bool advise = [] { const char *e = getenv("JPEGX_PREFAULT_HUGE"); return !(e && *e == '0'); }();
    if (mode == 0) return;
    const uintptr_t lo = (reinterpret_cast<uintptr_t>(p) + 4095) & ~(uintptr_t)4095, hi = (reinterpret_cast<uintptr_t>(p) + n) & ~(uintptr_t)4095;
    if (hi <= lo) return;
    {
        // memory the allocator hands back warm (malloc recycles blocks of up to 32 MiB; a caller's reused buffer) needs
        // none of this: eight probes of the page tables, and if every probed page is there the threads are not started
        bool warm = true;
        for (int k = 0; k < 8 && warm; ++k) {
            unsigned char vec = 0;
            const uintptr_t a = lo + (((hi - lo) / 4096) * (uintptr_t)k / 8) * 4096;
            warm = mincore(reinterpret_cast<void *>(a), 4096, &vec) == 0 && (vec & 1u);
        }
        if (warm) return;
    }
    const uintptr_t huge = (uintptr_t)2 << 20;
    const uintptr_t hlo = (lo + huge - 1) & ~(huge - 1), hhi = hi & ~(huge - 1);
    if ((advise || mode == -2) && hhi > hlo) (void)madvise(reinterpret_cast<void *>(hlo), hhi - hlo, MADV_HUGEPAGE);
    if (mode < 0) {
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23      /* Linux 5.14 */
#endif
        if (madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_POPULATE_WRITE) == 0) return;
    }
    const int nthreads = mode > 0 ? mode : 1;
    // thread t takes the 2 MiB-aligned grains t, t + nthreads, ...: one fault per grain where huge pages are granted
    auto work = [&](int t) {
        for (uintptr_t g = (lo & ~(huge - 1)) + (uintptr_t)t * huge; g < hi; g += huge * (uintptr_t)nthreads) {
            const uintptr_t a = g < lo ? lo : g, b = g + huge < hi ? g + huge : hi;
            for (uintptr_t o = a; o < b; o += 4096) { volatile uint8_t *q = reinterpret_cast<volatile uint8_t *>(o); *q = 0; }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &t : th) t.join();
}

// the same on a helper thread for the duration of a scope: the pages are touched while the device works
struct BackgroundTouch {
    std::thread t;
    BackgroundTouch(void *p, size_t n) : t(prefault, static_cast<uint8_t *>(p), n) {}
    ~BackgroundTouch() { if (t.joinable()) t.join(); }
    void wait() { if (t.joinable()) t.join(); }
};

int check_compress_shape(const void *h_plane, int elem_size, int H, int W, ptrdiff_t pitch, int bs)
{
    if (!h_plane) return fail(JPEGX_E_INVALID, "null pointer");
    if (bs < 1 || bs > 255) return fail(JPEGX_E_UNSUPPORTED, "host_compress supports block_size 1..255");
    if (H <= 0 || W <= 0 || (H % 8) || (W % 8)) return fail(JPEGX_E_INVALID, "plane height and width (after pooling) must be positive multiples of 8");
    if ((long long)(H / 8) * (W / 8) > 0x7FFFFFC0LL) return fail(JPEGX_E_INVALID, "more than 2^31 blocks in one plane");
    if (elem_size != 1 && elem_size != 4 && elem_size != 8) return fail(JPEGX_E_UNSUPPORTED, "host_compress takes uint8, int32 or int64 samples");
    if (pitch < (ptrdiff_t)W * bs) return fail(JPEGX_E_INVALID, "pitch smaller than the row");
    const bool fused_pool = bs == 1 || bs == 2 || bs == 4;
    if (fused_pool && ((W * bs) % 16) != 0) return fail(JPEGX_E_UNSUPPORTED, "host_compress needs rows of a multiple of 16 samples");
    return JPEGX_OK;
}

// Steps 1 + 4..7 of one band enqueued on `st`: upload (narrowing wide integers through `stage`, pinned), fused
// forward (uint8 kernels for block_size 1, 2, 4; mean-pool + all-float64 forward otherwise), sizes + scans.
// Afterwards the 16-byte head of slot.d_ws holds the total byte count and the error flag.
int enqueue_front(DevicePool *pool, BandSlot &slot, uint8_t *stage, const void *h_plane, int elem_size, int H, int W, ptrdiff_t pitch,
                  int bs, int mode, double param, hipStream_t st)
{
    const int HH = H * bs, WW = W * bs;
    const size_t in_bytes = (size_t)HH * WW;
    const long long nblocks = (long long)(H / 8) * (W / 8);
    const bool fused_pool = bs == 1 || bs == 2 || bs == 4;   // uint8 kernels with the mean folded in; else pool to float64 first
    int rc;
    if ((rc = slot.d_in.ensure(in_bytes)) || (rc = slot.d_zz.ensure((size_t)nblocks * 128)) ||
        (rc = slot.d_ws.ensure(jpegx_entropy_workspace_bytes(nblocks))))
        return rc;
    const uint8_t *src8 = static_cast<const uint8_t *>(h_plane);
    ptrdiff_t src_pitch = pitch;
    if (h_plane == nullptr) {
        // the plane is in slot.d_in already (image jobs from packed pixels: de-interleaved on the device)
    } else if (elem_size != 1) {
        // wide integers: narrowed into the pinned staging area strip by strip, every strip on its way to the device while
        // the next is narrowed
        rc = elem_size == 8 ? narrow_and_upload(static_cast<const int64_t *>(h_plane), pitch, HH, WW, stage, slot.d_in.p, st)
                            : narrow_and_upload(static_cast<const int32_t *>(h_plane), pitch, HH, WW, stage, slot.d_in.p, st);
        if (rc) return rc;
    } else {
        hipError_t e = (src_pitch == WW)
            ? hipMemcpyAsync(slot.d_in.p, src8, in_bytes, hipMemcpyHostToDevice, st)
            : hipMemcpy2DAsync(slot.d_in.p, WW, src8, (size_t)src_pitch, WW, HH, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return fail(JPEGX_E_HIP, "host to device copy failed");
    }
    if (fused_pool) {
        // the forward kernel sizes its blocks from the registers (RunLengthEncoding's bit counts): no second pass over the
        // stream, and the scan that follows writes the workspace's head itself -- two launches, no memset
        unsigned *block_bytes = nullptr, *wave_bytes = nullptr, *half_info = nullptr;
        jpegx_internal_entropy_views(slot.d_ws.p, nblocks, &block_bytes, &wave_bytes, &half_info);
        rc = jpegx_internal_forward_u8_sized(static_cast<const uint8_t *>(slot.d_in.p), H, W, WW, bs, mode, param, 0,
                                             static_cast<int16_t *>(slot.d_zz.p), block_bytes, wave_bytes, half_info, st);
        if (rc) return rc;
        slot.sized = true;                 // the emitter may go with two lanes per block
        return jpegx_internal_entropy_scan(nblocks, slot.d_ws.p, st);
    } else {
        // any other block_size: SubSampling on the device in float64 (exact sum, one division), then the
        // all-float64 fused forward -- k/9, k/25 ... are not fp32 numbers
        if ((rc = slot.d_tmp.ensure((size_t)H * W * 8))) return rc;
        rc = jpegx_mean_pool_f64(slot.d_in.p, 1, H, W, WW, bs, static_cast<double *>(slot.d_tmp.p), W, st);
        if (!rc)
            rc = jpegx_forward_fused_f64(static_cast<const double *>(slot.d_tmp.p), H, W, W, mode, param, 0,
                                         static_cast<int16_t *>(slot.d_zz.p), st);
    }
    if (rc) return rc;
    slot.sized = false;
    return jpegx_entropy_sizes(static_cast<const int16_t *>(slot.d_zz.p), nblocks, slot.d_ws.p, st);
}

// the emit launch that fits how the band's sizes were made
int enqueue_emit(BandSlot &slot, long long nblocks, uint8_t *d_out, hipStream_t st)
{
    return slot.sized ? jpegx_internal_entropy_emit2(static_cast<const int16_t *>(slot.d_zz.p), nblocks, slot.d_ws.p, d_out, st)
                      : jpegx_entropy_emit(static_cast<const int16_t *>(slot.d_zz.p), nblocks, slot.d_ws.p, d_out, st);
}

// JPEGX_TRACE=1: time stamps of the image jobs' host-side steps on stderr (where does a job's wall time go)
struct Trace {
    bool on;
    double t0;
    static double now()
    {
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return (double)ts.tv_sec * 1e3 + 1e-6 * (double)ts.tv_nsec;
    }
    Trace()
    {
        const char *e = getenv("JPEGX_TRACE");
        on = e && *e && *e != '0';
        t0 = on ? now() : 0.0;
    }
    void mark(const char *what, int k = -1) const
    {
        if (on) fprintf(stderr, "[jpegx trace +%.3f ms] %s%s%c\n", now() - t0, what, k >= 0 ? " band " : "", k >= 0 ? (char)('0' + k) : ' ');
    }
};

int head_verdict(const unsigned long long *head, unsigned long long *total)
{
    *total = head[0];
    if ((unsigned)(head[1] & 0xFFFFFFFFull) != 0)
        return fail(JPEGX_E_INVALID, "BadRleCodeError: an amplitude needs more than 15 bits (|a| > 16383)");
    return JPEGX_OK;
}

}  // namespace

extern "C" {

int jpegx_host_compress_begin(const void *h_plane, int elem_size, int H, int W, ptrdiff_t pitch, int bs, int mode,
                              double param, size_t *nbytes)
{
    if (!nbytes) return fail(JPEGX_E_INVALID, "null pointer");
    int rc = check_compress_shape(h_plane, elem_size, H, W, pitch, bs);
    if (rc) return rc;
    DevicePool *pool = nullptr;
    if ((rc = lock_pool(&pool))) return rc;
    auto bail = [&](int code) { unlock_pool(pool); return code; };
    if ((rc = ensure_streams(pool, false))) return bail(rc);
    if (elem_size != 1 && (rc = pool->h_in.ensure((size_t)H * bs * W * bs))) return bail(rc);
    BandSlot &slot = pool->slot[0];
    hipStream_t st = pool->stream;
    // The emitter goes into the stream BEHIND the size read-back without waiting for it: the device does not idle while the
    // host learns the byte count (a 16-byte copy, a wake-up and a launch: ~25 us), and the emitter refuses by itself when
    // the sizes pass flagged an amplitude.  Its destination is therefore sized for the worst case (185 bytes per block).
    const long long nblocks = (long long)(H / 8) * (W / 8);
    if ((rc = slot.d_out.ensure((size_t)nblocks * 188 + 64)) || (rc = pool->h_head.ensure(16 * MAX_BANDS))) return bail(rc);
    if ((rc = enqueue_front(pool, slot, static_cast<uint8_t *>(pool->h_in.p), h_plane, elem_size, H, W, pitch, bs, mode, param, st)))
        return bail(rc);
    unsigned long long *head = static_cast<unsigned long long *>(pool->h_head.p);
    if (hipMemcpyAsync(head, slot.d_ws.p, 16, hipMemcpyDeviceToHost, st) != hipSuccess || hipEventRecord(pool->ev[0], st) != hipSuccess) {
        (void)hipStreamSynchronize(st);
        return bail(fail(JPEGX_E_HIP, "device to host copy failed"));
    }
    rc = enqueue_emit(slot, nblocks, static_cast<uint8_t *>(slot.d_out.p), st);
    if (hipEventSynchronize(pool->ev[0]) != hipSuccess) { (void)hipStreamSynchronize(st); return bail(fail(JPEGX_E_HIP, "hipEventSynchronize failed")); }
    if (rc) { (void)hipStreamSynchronize(st); return bail(rc); }
    unsigned long long total = 0;
    if ((rc = head_verdict(head, &total))) { (void)hipStreamSynchronize(st); return bail(rc); }
    pool->open = true;
    pool->out_bytes = (size_t)total;
    *nbytes = (size_t)total;
    return JPEGX_OK;                       // the pool stays locked (and t_held set) until _finish / _abort
}

int jpegx_host_compress_finish(uint8_t *h_out)
{
    DevicePool *pool = t_held;             // this thread's open job, whatever its current device is by now
    if (!pool || !pool->open) return fail(JPEGX_E_INVALID, "no open compress job on this thread");
    hipError_t e = hipSuccess;
    if (pool->out_bytes) {
        if (!h_out) e = hipErrorInvalidValue;
        if (e == hipSuccess) prefault(h_out, pool->out_bytes);      // a fresh bytes object: its pages first (the emit kernel runs meanwhile)
        if (e == hipSuccess) e = hipMemcpyAsync(h_out, pool->slot[0].d_out.p, pool->out_bytes, hipMemcpyDeviceToHost, pool->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(pool->stream);
    pool->open = false;
    unlock_pool(pool);
    if (e != hipSuccess) return fail(JPEGX_E_HIP, "device to host copy failed");
    return JPEGX_OK;
}

int jpegx_host_compress_abort(void)
{
    DevicePool *pool = t_held;
    if (pool && pool->open) {
        (void)hipStreamSynchronize(pool->stream);
        pool->open = false;
        unlock_pool(pool);
    }
    return JPEGX_OK;
}

// ---- whole image, forward (pipeline/__init__.py:102-110 + file_format.generate_data, file_format.py:86-93) --------
// Band k runs on stream k % 2: upload -> fused forward -> sizes / scans -> 16-byte head down to pinned memory ->
// event.  Once every band's byte count is known the caller is asked ONCE for the destination of the whole result
// (`alloc`, e.g. a fresh Python bytes object): [prefix][u32 LE count of band 0][band 0][count][band 1] ... -- with
// the container header as prefix that IS the reference's file, no concatenation on the host afterwards.  The
// destination's pages are touched by a few host threads (fresh memory: the kernel hands out zeroed pages one fault
// at a time, which costs more than the copy itself) while the emit kernels run, then every band's bytes are
// copied from the device straight to their place.
static int compress_image_impl(const void *const *h_planes, const uint8_t *h_packed, int nbands, int elem_size, int H, int W, ptrdiff_t pitch, int bs,
                               int mode, double param, const void *prefix, size_t prefix_len, int length_prefixes,
                               jpegx_alloc_fn alloc, void *user, size_t *nbytes)
{
    if ((!h_planes && !h_packed) || !alloc || !nbytes || (prefix_len && !prefix)) return fail(JPEGX_E_INVALID, "null pointer");
    if (nbands < 1 || nbands > MAX_BANDS) return fail(JPEGX_E_INVALID, "compress_image takes 1..JPEGX_MAX_IMAGE_BANDS bands");
    int rc;
    if (h_packed) {
        if ((rc = check_compress_shape(h_packed, 1, H, W, (ptrdiff_t)W * bs, bs))) return rc;
        if (pitch < (ptrdiff_t)W * bs * nbands) return fail(JPEGX_E_INVALID, "packed pitch smaller than a row of pixels");
        if ((long long)H * bs > 65535) return fail(JPEGX_E_UNSUPPORTED, "packed pixels: more than 65535 rows");
    } else {
        for (int k = 0; k < nbands; ++k)
            if ((rc = check_compress_shape(h_planes[k], elem_size, H, W, pitch, bs))) return rc;
    }
    PoolLock lock;
    if (lock.rc) return lock.rc;
    DevicePool *pool = lock.pool;
    if ((rc = ensure_streams(pool, true))) return rc;
    const size_t in_bytes = (size_t)H * bs * W * bs;
    const long long nblocks = (long long)(H / 8) * (W / 8);
    if ((rc = pool->h_head.ensure(16 * MAX_BANDS))) return rc;
    if (!h_packed && elem_size != 1 && (rc = pool->h_in.ensure(in_bytes * nbands))) return rc;
    unsigned long long *heads = static_cast<unsigned long long *>(pool->h_head.p);
    auto drain = [&]() { (void)hipStreamSynchronize(pool->aux[0]); (void)hipStreamSynchronize(pool->aux[1]); };
    const Trace tr;
    tr.mark("compress_image: pool ready");
    if (h_packed) {
        // [rows][cols][nbands] pixels (what np.asarray(image) gives: half the host time of image.split() + one array per
        // band): one upload, the planes made on the device, then the bands as below without their own uploads
        const int rows = H * bs, cols = W * bs;
        const size_t row_bytes = (size_t)cols * nbands;
        if ((rc = pool->d_packed.ensure((size_t)rows * row_bytes))) return rc;
        void *planes[MAX_BANDS] = {};
        for (int k = 0; k < nbands; ++k) {
            if ((rc = pool->slot[k].d_in.ensure(in_bytes))) return rc;
            planes[k] = pool->slot[k].d_in.p;
        }
        hipError_t e = ((size_t)pitch == row_bytes)
            ? hipMemcpyAsync(pool->d_packed.p, h_packed, (size_t)rows * row_bytes, hipMemcpyHostToDevice, pool->aux[0])
            : hipMemcpy2DAsync(pool->d_packed.p, row_bytes, h_packed, (size_t)pitch, row_bytes, (size_t)rows, hipMemcpyHostToDevice, pool->aux[0]);
        if (e != hipSuccess) { drain(); return fail(JPEGX_E_HIP, "host to device copy failed"); }
        if ((rc = jpegx_deinterleave_u8(static_cast<const uint8_t *>(pool->d_packed.p), (ptrdiff_t)row_bytes, nbands, rows, cols, planes, cols, pool->aux[0]))) { drain(); return rc; }
        if (hipEventRecord(pool->ev_x, pool->aux[0]) != hipSuccess || hipStreamWaitEvent(pool->aux[1], pool->ev_x, 0) != hipSuccess) {
            drain();
            return fail(JPEGX_E_HIP, "event between the image job's streams failed");
        }
        tr.mark("pixels enqueued");
    }
    for (int k = 0; k < nbands; ++k) {
        hipStream_t st = pool->aux[k & 1];
        uint8_t *stage = (!h_packed && elem_size != 1) ? static_cast<uint8_t *>(pool->h_in.p) + in_bytes * k : nullptr;
        rc = enqueue_front(pool, pool->slot[k], stage, h_packed ? nullptr : h_planes[k], h_packed ? 1 : elem_size, H, W, pitch, bs, mode, param, st);
        if (!rc && hipMemcpyAsync(heads + 2 * k, pool->slot[k].d_ws.p, 16, hipMemcpyDeviceToHost, st) != hipSuccess)
            rc = fail(JPEGX_E_HIP, "device to host copy failed");
        if (!rc && hipEventRecord(pool->ev[k], st) != hipSuccess) rc = fail(JPEGX_E_HIP, "hipEventRecord failed");
        if (rc) { drain(); return rc; }
        tr.mark("front enqueued", k);
    }
    size_t total_all = prefix_len;
    size_t offset[MAX_BANDS] = {};
    for (int k = 0; k < nbands; ++k) {
        if (hipEventSynchronize(pool->ev[k]) != hipSuccess) { drain(); return fail(JPEGX_E_HIP, "hipEventSynchronize failed"); }
        unsigned long long total = 0;
        if ((rc = head_verdict(heads + 2 * k, &total))) { drain(); return rc; }
        if (length_prefixes && total > 0xFFFFFFFFull) { drain(); return fail(JPEGX_E_INVALID, "a band's stream does not fit the container's 32-bit length field"); }
        nbytes[k] = (size_t)total;
        offset[k] = total_all + (length_prefixes ? 4 : 0);
        total_all = offset[k] + (size_t)total;
        // the emit kernel needs only the device-side offsets: enqueue it now, the destination comes later
        BandSlot &slot = pool->slot[k];
        if ((rc = slot.d_out.ensure(total ? (size_t)total : 1)) ||
            (rc = enqueue_emit(slot, nblocks, static_cast<uint8_t *>(slot.d_out.p), pool->aux[k & 1]))) {
            drain();
            return rc;
        }
    }
    tr.mark("sizes known, emits enqueued");
    uint8_t *dst = static_cast<uint8_t *>(alloc(user, total_all));
    if (!dst && total_all) { drain(); return fail(JPEGX_E_INVALID, "the allocator returned no destination"); }
    tr.mark("destination allocated");
    prefault(dst, total_all);
    tr.mark("destination touched");
    if (prefix_len) memcpy(dst, prefix, prefix_len);
    for (int k = 0; k < nbands; ++k) {
        if (length_prefixes) {
            const uint32_t n32 = (uint32_t)nbytes[k];
            const uint8_t le[4] = {(uint8_t)n32, (uint8_t)(n32 >> 8), (uint8_t)(n32 >> 16), (uint8_t)(n32 >> 24)};   // struct '<L'
            memcpy(dst + offset[k] - 4, le, 4);
        }
        if (nbytes[k] && hipMemcpyAsync(dst + offset[k], pool->slot[k].d_out.p, nbytes[k], hipMemcpyDeviceToHost, pool->aux[k & 1]) != hipSuccess) {
            drain();
            return fail(JPEGX_E_HIP, "device to host copy failed");
        }
        tr.mark("download enqueued", k);
    }
    HP_TRY(hipStreamSynchronize(pool->aux[0]));
    HP_TRY(hipStreamSynchronize(pool->aux[1]));
    tr.mark("compress_image: done");
    return JPEGX_OK;
}

int jpegx_host_compress_image(const void *const *h_planes, int nbands, int elem_size, int H, int W, ptrdiff_t pitch, int bs,
                              int mode, double param, const void *prefix, size_t prefix_len, int length_prefixes,
                              jpegx_alloc_fn alloc, void *user, size_t *nbytes)
{
    if (!h_planes) return fail(JPEGX_E_INVALID, "null pointer");
    return compress_image_impl(h_planes, nullptr, nbands, elem_size, H, W, pitch, bs, mode, param, prefix, prefix_len, length_prefixes, alloc, user, nbytes);
}

// the same from pixel-interleaved samples, [H * bs][W * bs][nbands] uint8 with rows `pitch` bytes apart (np.asarray(image))
int jpegx_host_compress_image_packed(const uint8_t *h_pixels, int nbands, int H, int W, ptrdiff_t pitch, int bs,
                                     int mode, double param, const void *prefix, size_t prefix_len, int length_prefixes,
                                     jpegx_alloc_fn alloc, void *user, size_t *nbytes)
{
    if (!h_pixels) return fail(JPEGX_E_INVALID, "null pointer");
    return compress_image_impl(nullptr, h_pixels, nbands, 1, H, W, pitch, bs, mode, param, prefix, prefix_len, length_prefixes, alloc, user, nbytes);
}

}  // extern "C"

namespace {
// bytes already on the device (slot.d_in, padded) -> int16 stream in slot.d_zz; the caller holds the pool.
// level 0: the segmented scheme (jpegx_entropy_decode.hip, round 3) -- three launches whose workspace depends on the
// stream's length only, no host round trip; decode_status afterwards may answer DECODE_RETRY_GENERAL ("next level") for a
// stream whose densest stretch overflows a segment's tables.  level 1: the same with 256-byte segments.  level 2: the
// pointer-jumping scheme over the whole stream, which reads the candidate count back in the middle.
constexpr int DECODE_RETRY_GENERAL = 1;

thread_local int t_last_decode_level = -1;      // which scheme took the last stream on this thread (tests)

int decode_on_device(BandSlot &slot, size_t nbytes, long long nblocks, hipStream_t st, int level)
{
    int rc;
    if ((rc = slot.d_zz.ensure((size_t)nblocks * 128))) return rc;
    const char *force = getenv("JPEGX_DECODE_GENERAL");    // tests / A-B runs: the general scheme from the start
    if (force && *force && *force != '0') level = 2;
    if (level == 1 && jpegx_decode::seg_plan(nbytes, nblocks, 0).seg == 256) level = 2;      // the first try had the smallest segments already
    const jpegx_decode::SegPlan plan = jpegx_decode::seg_plan(nbytes, nblocks, level, (level == 0 && slot.filter_pause > 0) ? 0 : -1);
    if (level == 0 && slot.filter_pause > 0) --slot.filter_pause;
    slot.last_filter = plan.filter;
    t_last_decode_level = level;
    if (level < 2 && plan.ok) {
        if ((rc = slot.d_seg.ensure(plan.ws_bytes)) || (rc = slot.d_seg_state.ensure(plan.state_bytes))) return rc;
        bool fresh = slot.d_seg_state.p != slot.seg_clean || slot.d_seg_state.cap != slot.seg_clean_cap;
        if (const char *ff = getenv("JPEGX_DECODE_FRESH")) fresh = fresh || (*ff && *ff != '0');      // tests: clear the state on every call
        slot.seg_clean = slot.d_seg_state.p;
        slot.seg_clean_cap = slot.d_seg_state.cap;
        slot.seg_parity = (int)(slot.seg_calls++ & 1u);
        jpegx_decode::enqueue_segmented(static_cast<const uint8_t *>(slot.d_in.p), nbytes, nblocks, plan, slot.d_seg_state.p, slot.d_seg_state.cap, fresh,
                                        slot.seg_parity, slot.d_seg.p, static_cast<int16_t *>(slot.d_zz.p), st);
        HP_TRY(hipGetLastError());
        return JPEGX_OK;
    }
    slot.seg_parity = -1;
    if ((rc = slot.d_ws.ensure(jpegx_decode::phase1_bytes(nbytes)))) return rc;
    jpegx_decode::enqueue_phase1(static_cast<const uint8_t *>(slot.d_in.p), nbytes, slot.d_ws.p, st);
    unsigned head[4] = {0, 0, 0, 0};
    HP_TRY(hipMemcpyAsync(head, slot.d_ws.p, 16, hipMemcpyDeviceToHost, st));
    HP_TRY(hipStreamSynchronize(st));
    const unsigned ncand = head[0];
    if (ncand == 0 || (long long)ncand < nblocks) return fail(JPEGX_E_INVALID, "entropy stream holds fewer blocks than the plane has");
    if ((rc = slot.d_tmp.ensure(jpegx_decode::phase2_bytes(ncand, nblocks)))) return rc;
    jpegx_decode::enqueue_phase2(static_cast<const uint8_t *>(slot.d_in.p), nbytes, nblocks, slot.d_ws.p, ncand, slot.d_tmp.p,
                                 static_cast<int16_t *>(slot.d_zz.p), st);
    HP_TRY(hipGetLastError());
    return JPEGX_OK;
}

// after the stream has been synchronised: JPEGX_OK, an error, or DECODE_RETRY_GENERAL
int decode_status(BandSlot &slot)
{
    unsigned head[16] = {0};
    const bool seg = slot.seg_parity >= 0;
    const unsigned char *src = seg ? static_cast<const unsigned char *>(slot.d_seg_state.p) + 64 * slot.seg_parity : static_cast<const unsigned char *>(slot.d_ws.p);
    HP_TRY(hipMemcpy(head, src, 64, hipMemcpyDeviceToHost));
    if (const char *dump = getenv("JPEGX_DECODE_STATS")) {     // a -DJPEGX_DECODE_STATS build leaves per-segment time stamps in its scratch
        if (seg) {
            std::vector<unsigned char> raw(slot.d_seg.cap);
            HP_TRY(hipMemcpy(raw.data(), slot.d_seg.p, raw.size(), hipMemcpyDeviceToHost));
            if (FILE *f = fopen(dump, "wb")) { fwrite(raw.data(), 1, raw.size(), f); fclose(f); }
        }
    }
    if (seg && head[2] != 0) {
        if ((head[2] & 4u) && slot.last_filter) slot.filter_pause = 64;      // the candidate filter missed a block start: without it for a while
        return DECODE_RETRY_GENERAL;
    }
    if (head[1] != 0) return fail(JPEGX_E_INVALID, "entropy stream is not a sequence of well-formed blocks (device decoder)");
    return JPEGX_OK;
}

int check_decompress_shape(const uint8_t *h_bytes, size_t nbytes, int H, int W, int bs)
{
    if (!h_bytes) return fail(JPEGX_E_INVALID, "null host pointer");
    if (H <= 0 || W <= 0 || (H % 8) || (W % 8)) return fail(JPEGX_E_INVALID, "plane height and width must be positive multiples of 8");
    if ((long long)(H / 8) * (W / 8) > 0x7FFFFFC0LL) return fail(JPEGX_E_INVALID, "block count must be in 1 .. 2^31-64");
    if (bs < 1 || bs > 255) return fail(JPEGX_E_UNSUPPORTED, "host_decompress supports block_size 1..255");
    if (nbytes == 0 || nbytes >= 0xFFFFFFF0ull) return fail(JPEGX_E_INVALID, "entropy stream empty or beyond 4 GiB");
    return JPEGX_OK;
}

// upload + device entropy decoding + fused inverse (clamp, SubSampling.invert) of one band into slot.d_out
// ([H*bs][dev_pitch] bytes), all on `st`
int enqueue_back(BandSlot &slot, const uint8_t *h_bytes, size_t nbytes, int H, int W, int bs, int mode, double param,
                 ptrdiff_t dev_pitch, hipStream_t st, int level)
{
    const long long nblocks = (long long)(H / 8) * (W / 8);
    int rc;
    if ((rc = slot.d_in.ensure(nbytes + 16)) || (rc = slot.d_out.ensure((size_t)H * bs * dev_pitch))) return rc;
    HP_TRY(hipMemsetAsync(static_cast<uint8_t *>(slot.d_in.p) + (nbytes & ~(size_t)3), 0, 16 + (nbytes & 3), st));   // zero tail (whole dwords)
    HP_TRY(hipMemcpyAsync(slot.d_in.p, h_bytes, nbytes, hipMemcpyHostToDevice, st));
    if ((rc = decode_on_device(slot, nbytes, nblocks, st, level))) return rc;
    return jpegx_inverse_fused_u8_inflated(static_cast<const int16_t *>(slot.d_zz.p), H, W, mode, param, 0, bs,
                                           static_cast<uint8_t *>(slot.d_out.p), dev_pitch, st);
}
}  // namespace

extern "C" {

// Inverse of jpegx_host_compress_*: the whole decompress_band job for one plane (pipeline/__init__.py:79-88 for
// transform 'DCT', dct_size 8): bytes up, entropy decoding ON THE DEVICE (jpegx_entropy_decode.hip), fused
// inverse with clamp and SubSampling.invert (any block_size), uint8 samples down.  h_out: [H*bs][out_pitch] bytes.
static int decompress_plane_locked(DevicePool *pool, const uint8_t *h_bytes, size_t nbytes, int H, int W, int bs, int mode,
                                   double param, uint8_t *h_out, ptrdiff_t out_pitch, bool fresh_out)
{
    int rc = check_decompress_shape(h_bytes, nbytes, H, W, bs);
    if (rc) return rc;
    if (!h_out) return fail(JPEGX_E_INVALID, "null host pointer");
    if (out_pitch < (ptrdiff_t)W * bs || (out_pitch % ((bs == 2 || bs == 4) ? 16 : 8)) != 0)
        return fail(JPEGX_E_INVALID, "output pitch too small or misaligned");
    if ((rc = ensure_streams(pool, false))) return rc;
    hipStream_t st = pool->stream;
    BandSlot &slot = pool->slot[0];
    BackgroundTouch touch(h_out, fresh_out ? (size_t)H * bs * out_pitch : 0);       // a fresh result array: fault its pages in meanwhile
    for (int level = 0; level < 3; ++level) {      // planned segments, 256-byte segments, the whole-stream scheme
        if ((rc = enqueue_back(slot, h_bytes, nbytes, H, W, bs, mode, param, out_pitch, st, level))) return rc;
        touch.wait();
        HP_TRY(hipMemcpyAsync(h_out, slot.d_out.p, (size_t)H * bs * out_pitch, hipMemcpyDeviceToHost, st));
        HP_TRY(hipStreamSynchronize(st));
        if ((rc = decode_status(slot)) != DECODE_RETRY_GENERAL) return rc;
    }
    return fail(JPEGX_E_INVALID, "device decoder: no scheme took the stream");
}

int jpegx_host_decompress_plane(const uint8_t *h_bytes, size_t nbytes, int H, int W, int bs, int mode, double param,
                                uint8_t *h_out, ptrdiff_t out_pitch)
{
    PoolLock lock;
    if (lock.rc) return lock.rc;
    return decompress_plane_locked(lock.pool, h_bytes, nbytes, H, W, bs, mode, param, h_out, out_pitch, true);
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
    PoolLock lock;                                         // held to the end: the staging span belongs to this job
    if (lock.rc) return lock.rc;
    DevicePool *pool = lock.pool;
    int rc;
    if ((rc = pool->h_out.ensure(stage_bytes))) return rc;
    uint8_t *stage = static_cast<uint8_t *>(pool->h_out.p);
    BackgroundTouch touch(h_out, (size_t)rows * cols * sizeof(int64_t));   // 128 MiB for a 4096 x 4096 band, usually never touched before
    if ((rc = decompress_plane_locked(pool, h_bytes, nbytes, H, W, bs, mode, param, stage, pitch, false))) return rc;
    touch.wait();
    const unsigned hw = std::thread::hardware_concurrency();
    static const int want = [] { const char *e = getenv("JPEGX_WIDEN_THREADS"); return e && *e ? atoi(e) : 8; }();      // A/B (8: 1.8-2.4 ms, 16: 1.7-2.6 ms per 4096^2 band: no difference)
    const int nthreads = ((size_t)rows * cols < (1u << 20)) ? 1 : ((int)hw >= want ? want : (hw ? (int)hw : 1));
    // non-temporal stores: the array is written once, 8 bytes per sample, and is eight times the size of what is read --
    // ordinary stores would first READ every line of it for ownership (twice the memory traffic)
    auto work = [&](int y0, int y1) {
        for (int y = y0; y < y1; ++y) {
            const uint8_t *srow = stage + (size_t)y * pitch;
            int64_t *drow = h_out + (size_t)y * cols;
            for (int x = 0; x < cols; ++x) __builtin_nontemporal_store((int64_t)srow[x], drow + x);
        }
        __builtin_ia32_sfence();
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

// ---- whole image, inverse (pipeline/__init__.py:112-124) ---------------------------------------------------------
// The bands of one picture on two alternating streams; the samples come back either as `nbands` planes stacked
// behind each other ([band][H*bs][out_pitch], interleave = 0) or as the pixel-interleaved array PIL wants
// ([H*bs][W*bs][nbands] with rows `out_pitch` bytes apart, interleave = 1: np.dstack on the host costs more than the
// whole device pipeline), cropped to rows x cols.
int jpegx_host_decompress_image(const uint8_t *const *h_bytes, const size_t *nbytes, int nbands, int H, int W, int bs, int mode,
                                double param, uint8_t *h_out, ptrdiff_t out_pitch, int rows, int cols, int interleave)
{
    if (!h_bytes || !nbytes || !h_out) return fail(JPEGX_E_INVALID, "null pointer");
    if (nbands < 1 || nbands > MAX_BANDS) return fail(JPEGX_E_INVALID, "decompress_image takes 1..JPEGX_MAX_IMAGE_BANDS bands");
    int rc;
    for (int k = 0; k < nbands; ++k)
        if ((rc = check_decompress_shape(h_bytes[k], nbytes[k], H, W, bs))) return rc;
    if (rows <= 0 || cols <= 0 || rows > (long long)H * bs || cols > (long long)W * bs) return fail(JPEGX_E_INVALID, "bad output shape");
    if (out_pitch < (ptrdiff_t)cols * (interleave ? nbands : 1)) return fail(JPEGX_E_INVALID, "output pitch smaller than the row");
    const ptrdiff_t dev_pitch = ((ptrdiff_t)W * bs + 15) / 16 * 16;
    PoolLock lock;
    if (lock.rc) return lock.rc;
    DevicePool *pool = lock.pool;
    if ((rc = ensure_streams(pool, true))) return rc;
    const size_t packed_pitch = (size_t)cols * nbands;
    if (interleave && (rc = pool->d_packed.ensure((size_t)rows * packed_pitch))) return rc;      // before anything is enqueued
    // the result array is usually fresh memory: touch its pages on a helper thread while the bands are uploaded and decoded
    const size_t out_span = interleave ? (size_t)rows * out_pitch : (size_t)nbands * rows * out_pitch;
    BackgroundTouch touch(h_out, out_span);
    auto drain = [&]() { (void)hipStreamSynchronize(pool->aux[0]); (void)hipStreamSynchronize(pool->aux[1]); };
    int level[MAX_BANDS] = {};                              // per band: planned segments, 256-byte segments, the whole-stream scheme
    bool done[MAX_BANDS] = {};
    auto copy_down = [&](int j) -> int {                    // band j's samples to their place in the result, on the band's stream
        if (hipMemcpy2DAsync(h_out + (size_t)j * rows * out_pitch, (size_t)out_pitch, pool->slot[j].d_out.p, (size_t)dev_pitch,
                             (size_t)cols, (size_t)rows, hipMemcpyDeviceToHost, pool->aux[j & 1]) != hipSuccess)
            return fail(JPEGX_E_HIP, "device to host copy failed");
        return JPEGX_OK;
    };
    for (int attempt = 0; attempt < 3; ++attempt) {
        int held = -1;                                       // a band whose copy down is not enqueued yet
        for (int k = 0; k < nbands; ++k) {
            hipStream_t st = pool->aux[k & 1];
            if (done[k]) {                                   // this band is done: only the packing below waits for it again
                if (interleave && hipEventRecord(pool->ev[k], st) != hipSuccess) { drain(); return fail(JPEGX_E_HIP, "hipEventRecord failed"); }
                continue;
            }
            if ((rc = enqueue_back(pool->slot[k], h_bytes[k], nbytes[k], H, W, bs, mode, param, dev_pitch, st, level[k]))) { drain(); return rc; }
            if (!interleave) {
                // No copy may be ENQUEUED while the helper thread still writes its zeros into the result's pages (a copy
                // that landed first would lose one byte per page).  The first band's copy is therefore held back until
                // the second band's work is in its own stream: the wait then costs nothing the device could notice.
                if (k == 0 && nbands > 1 && !done[1]) { held = 0; continue; }
                touch.wait();
                if (held >= 0 && (rc = copy_down(held))) { drain(); return rc; }
                held = -1;
                if ((rc = copy_down(k))) { drain(); return rc; }
            } else if (hipEventRecord(pool->ev[k], st) != hipSuccess) {
                drain();
                return fail(JPEGX_E_HIP, "hipEventRecord failed");
            }
        }
        if (interleave) {
            // the packing kernel runs on stream 0 behind every band
            hipStream_t st = pool->aux[0];
            const void *planes[MAX_BANDS] = {};
            for (int k = 0; k < nbands; ++k) {
                planes[k] = pool->slot[k].d_out.p;
                if (k != 0 && hipStreamWaitEvent(st, pool->ev[k], 0) != hipSuccess) { drain(); return fail(JPEGX_E_HIP, "hipStreamWaitEvent failed"); }
            }
            uint8_t *packed = static_cast<uint8_t *>(pool->d_packed.p);
            if ((rc = jpegx_interleave_u8(planes, nbands, rows, cols, dev_pitch, packed, (ptrdiff_t)packed_pitch, st))) { drain(); return rc; }
            touch.wait();                                    // the helper thread's zeros first, then the copy (see above)
            if (hipMemcpy2DAsync(h_out, (size_t)out_pitch, packed, packed_pitch, packed_pitch, (size_t)rows, hipMemcpyDeviceToHost, st) != hipSuccess) {
                drain();
                return fail(JPEGX_E_HIP, "device to host copy failed");
            }
        }
        HP_TRY(hipStreamSynchronize(pool->aux[0]));
        HP_TRY(hipStreamSynchronize(pool->aux[1]));
        bool again = false;
        for (int k = 0; k < nbands; ++k) {
            if (done[k]) continue;
            rc = decode_status(pool->slot[k]);
            if (rc == DECODE_RETRY_GENERAL) { ++level[k]; again = true; }
            else if (rc) return rc;
            else done[k] = true;
        }
        if (!again) return JPEGX_OK;
    }
    return fail(JPEGX_E_INVALID, "device decoder: no scheme took the stream");
}

extern "C" int jpegx_internal_last_decode_level(void) { return t_last_decode_level; }

// ---- the device decoder on the caller's device buffers (include/jpegx.h) -------------------------------------------
// workspace: [state of the larger plan, rounded up to 256 bytes][scratch of the larger plan]; the state is cleared by
// every call (the pooled jobs above keep theirs clean from call to call instead: one fill launch less)
namespace {
size_t decode_state_span(size_t nbytes, long long nblocks)
{
    size_t m = 0;
    for (int level = 0; level < 2; ++level)
        for (int filter = -1; filter <= 0; ++filter) {
            const jpegx_decode::SegPlan p = jpegx_decode::seg_plan(nbytes, nblocks, level, filter);
            if (p.state_bytes > m) m = p.state_bytes;
        }
    return (m + 255) & ~(size_t)255;
}
}  // namespace

extern "C" size_t jpegx_entropy_decode_workspace_bytes(size_t nbytes, long long nblocks)
{
    if (nbytes == 0 || nblocks <= 0) return 0;
    size_t scratch = 0;
    for (int level = 0; level < 2; ++level)
        for (int filter = -1; filter <= 0; ++filter) {
            const jpegx_decode::SegPlan p = jpegx_decode::seg_plan(nbytes, nblocks, level, filter);
            if (p.ws_bytes > scratch) scratch = p.ws_bytes;
        }
    return decode_state_span(nbytes, nblocks) + scratch + 256;
}

extern "C" int jpegx_entropy_decode(const uint8_t *d_bytes, size_t nbytes, long long nblocks, void *d_workspace, int16_t *d_zz,
                                    int level, jpegx_stream_t stream)
{
    if (!d_bytes || !d_workspace || !d_zz) return fail(JPEGX_E_INVALID, "null device pointer");
    if (nblocks <= 0 || nblocks > 0x7FFFFFC0LL) return fail(JPEGX_E_INVALID, "block count must be in 1 .. 2^31-64");
    if (nbytes == 0 || nbytes >= 0xFFFFFFF0ull) return fail(JPEGX_E_INVALID, "entropy stream empty or beyond 4 GiB");
    if (level < 0 || level > 1) return fail(JPEGX_E_UNSUPPORTED, "levels 0 and 1 run on caller buffers; the whole-stream scheme is jpegx_host_entropy_decode_gpu's");
    if ((reinterpret_cast<uintptr_t>(d_workspace) & 255u) != 0) return fail(JPEGX_E_INVALID, "workspace must be 256-byte aligned");
    const jpegx_decode::SegPlan plan = jpegx_decode::seg_plan(nbytes, nblocks, level);
    if (!plan.ok) return fail(JPEGX_E_UNSUPPORTED, "stream too long for the segmented decoder");
    const size_t state_span = decode_state_span(nbytes, nblocks);
    unsigned char *w = static_cast<unsigned char *>(d_workspace);
    jpegx_decode::enqueue_segmented(d_bytes, nbytes, nblocks, plan, w, state_span, true, 0, w + state_span, d_zz, (hipStream_t)stream);
    HP_TRY(hipGetLastError());
    return JPEGX_OK;
}

extern "C" int jpegx_entropy_decode_status(const void *d_workspace, jpegx_stream_t stream)
{
    if (!d_workspace) return fail(JPEGX_E_INVALID, "null device pointer");
    unsigned head[16] = {0};
    HP_TRY(hipMemcpyAsync(head, d_workspace, 64, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (head[2] != 0) return 1;
    if (head[1] != 0) return fail(JPEGX_E_INVALID, "entropy stream is not a sequence of well-formed blocks (device decoder)");
    return JPEGX_OK;
}

// bytes -> int16 [nblocks][64] on the device, host arrays in and out (what jpegx_host_entropy_decode does on the CPU)
int jpegx_host_entropy_decode_gpu(const uint8_t *h_bytes, size_t nbytes, long long nblocks, int16_t *h_zz)
{
    if (!h_bytes || !h_zz) return fail(JPEGX_E_INVALID, "null host pointer");
    if (nblocks <= 0 || nblocks > 0x7FFFFFC0LL) return fail(JPEGX_E_INVALID, "block count must be in 1 .. 2^31-64");
    if (nbytes == 0 || nbytes >= 0xFFFFFFF0ull) return fail(JPEGX_E_INVALID, "entropy stream empty or beyond 4 GiB");
    PoolLock lock;
    if (lock.rc) return lock.rc;
    DevicePool *pool = lock.pool;
    int rc;
    if ((rc = ensure_streams(pool, false))) return rc;
    hipStream_t st = pool->stream;
    BandSlot &slot = pool->slot[0];
    if ((rc = slot.d_in.ensure(nbytes + 16))) return rc;
    HP_TRY(hipMemsetAsync(static_cast<uint8_t *>(slot.d_in.p) + (nbytes & ~(size_t)3), 0, 16 + (nbytes & 3), st));
    HP_TRY(hipMemcpyAsync(slot.d_in.p, h_bytes, nbytes, hipMemcpyHostToDevice, st));
    for (int level = 0; level < 3; ++level) {
        if ((rc = decode_on_device(slot, nbytes, nblocks, st, level))) return rc;
        HP_TRY(hipMemcpyAsync(h_zz, slot.d_zz.p, (size_t)nblocks * 128, hipMemcpyDeviceToHost, st));
        HP_TRY(hipStreamSynchronize(st));
        if ((rc = decode_status(slot)) != DECODE_RETRY_GENERAL) return rc;
    }
    return fail(JPEGX_E_INVALID, "device decoder: no scheme took the stream");
}

// used by host_roundtrip (jpegx_internal.h): the synchronous host-pointer conveniences borrow the pool's
// stream and its input / output device spans for the duration of one call; not part of the public ABI
int jpegx_internal_pool_acquire(size_t in_bytes, size_t out_bytes, void **d_in, void **d_out, void **stream)
{
    DevicePool *pool = nullptr;
    int rc = lock_pool(&pool);
    if (rc) return rc;
    if ((rc = ensure_streams(pool, false)) || (rc = pool->slot[0].d_in.ensure(in_bytes ? in_bytes : 1)) ||
        (rc = pool->slot[0].d_out.ensure(out_bytes ? out_bytes : 1))) {
        unlock_pool(pool);
        return rc;
    }
    *d_in = pool->slot[0].d_in.p;
    *d_out = pool->slot[0].d_out.p;
    *stream = pool->stream;
    return JPEGX_OK;
}

void jpegx_internal_pool_release(void)
{
    if (t_held && !t_held->open) unlock_pool(t_held);      // the pool this thread borrowed, not "the current device's"
}

// release everything the pools hold on the current device (tests; long-lived processes that are done)
int jpegx_host_pool_release(void)
{
    int dev = 0;
    HP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) return fail(JPEGX_E_UNSUPPORTED, "device index beyond the pool table");
    if (t_held != nullptr)
        return fail(JPEGX_E_INVALID, t_held->open ? "a compress job is open on this thread: finish or abort it first"
                                                   : "this thread already holds a device pool");
    for (DevicePool &ctx : g_pool[dev]) {
        std::lock_guard<std::mutex> guard(ctx.mu);           // waits for a job of another thread to finish
        DevicePool *pool = &ctx;
        for (BandSlot &b : pool->slot) {
            for (Span *s : {&b.d_in, &b.d_zz, &b.d_ws, &b.d_out, &b.d_tmp, &b.d_seg, &b.d_seg_state}) s->release();
            b.seg_clean = nullptr;
            b.seg_clean_cap = 0;
            b.filter_pause = 0;
            b.last_filter = false;
        }
        for (Span *s : {&pool->d_packed, &pool->h_in, &pool->h_out, &pool->h_head}) s->release();
        if (pool->stream) { (void)hipStreamDestroy(pool->stream); pool->stream = nullptr; }
        for (hipStream_t &s : pool->aux)
            if (s) { (void)hipStreamDestroy(s); s = nullptr; }
        for (hipEvent_t &e : pool->ev)
            if (e) { (void)hipEventDestroy(e); e = nullptr; }
        if (pool->ev_x) { (void)hipEventDestroy(pool->ev_x); pool->ev_x = nullptr; }
    }
    return JPEGX_OK;
}

}  // extern "C"
