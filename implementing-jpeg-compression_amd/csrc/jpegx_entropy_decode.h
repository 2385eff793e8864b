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
    int span_cap;               // LDS bytes the block decoder has for the bytes of a tile's 64 blocks
    unsigned nseg;
    size_t ws_bytes, state_bytes;   // scratch; the persistent state (status blocks + exit words)
    bool ok;                    // false: stream too long for this scheme
    bool filter;                // first try: only candidates whose first byte can start a block
};
// level 0: segments sized from the stream's average block length; level 1: the smallest segments (256 bytes), the second try
// for a stream whose local density overflowed a segment's tables (flat regions in a busy picture)
// filter: 0 = keep every candidate also in the first try (a caller that has just seen the filter miss), -1 = the default
SegPlan seg_plan(size_t nbytes, long long nblocks, int level = 0, int filter = -1);
// d_state: state_cap >= plan.state_bytes bytes that only this scheme touches -- fresh: never used before (it is cleared whole, once;
// afterwards every call leaves it clean); parity alternates from call to call on one d_state (the call's status words are at
// d_state + 64 * parity).  d_ws: plan.ws_bytes of scratch.
void enqueue_segmented(const uint8_t *d_bytes, size_t nbytes, long long nblocks, const SegPlan &plan, void *d_state, size_t state_cap, bool fresh, int parity,
                       void *d_ws, int16_t *d_zz, hipStream_t st);
}  // namespace jpegx_decode
