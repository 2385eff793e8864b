// jpegx_entropy_decode.h -- internal interface between the device entropy decoder's kernels
// (jpegx_entropy_decode.hip) and the host orchestration that owns the buffers (jpegx_hostpipe.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace jpegx_decode {
int levels_for(long long nblocks);
size_t phase1_bytes(size_t nbytes);                       // workspace of the candidate count
size_t phase2_bytes(size_t ncand, long long nblocks);     // workspace of everything after it
// d_bytes: dword aligned, at least 16 zero bytes readable behind the stream
void enqueue_phase1(const uint8_t *d_bytes, size_t nbytes, void *d_ws1, hipStream_t st);
void enqueue_phase2(const uint8_t *d_bytes, size_t nbytes, long long nblocks, void *d_ws1, unsigned ncand, void *d_ws2,
                    int16_t *d_zz, hipStream_t st);

// ---- the segmented scheme (round 3): workspace and launch count depend on the stream's LENGTH only, so the whole
// decode is enqueued without a host round trip.  Afterwards head[1] != 0: refused; head[2] != 0: a stream the
// segment tables do not fit (handed back: run phase 1 / phase 2 above).
struct SegPlan {
    int seg, cmax, levels;      // bytes per segment, candidates a segment's tables hold, doubling levels
    unsigned nseg;
    size_t ws_bytes;
    bool ok;                    // false: stream too long for this scheme (> 64 K segments)
};
SegPlan seg_plan(size_t nbytes, long long nblocks);
void enqueue_segmented(const uint8_t *d_bytes, size_t nbytes, long long nblocks, const SegPlan &plan, void *d_ws, int16_t *d_zz, hipStream_t st);
}  // namespace jpegx_decode
