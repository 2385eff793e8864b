// jpegx_entropy.hip -- the entropy stage on the GPU: run-length coding of the
// zigzag stream and bit packing, i.e. RunLengthEncoding.execute (pipeline/run_length_encoding.py:
// 47-64, 14-32) + RleBytestream.execute (pipeline/rle_byte_stream.py:48-59) with util.RunLengthCode
// (util.py:134-221) and util.BitEncoder (util.py:115-131) of the reference:
//   a non-zero value a after `run` zeros -> floor(run/15) zero-chain codes 1111 0000 (FIFTEEN zeros
//   each), then 4-bit run % 15, 4-bit size = bit_length(|a|) + 1, sign bit ('1' iff a > 0) and the
//   bit_length(|a|) magnitude bits; every block ends with EOB = 8 zero bits and is zero-padded to a
//   byte boundary, so blocks are independent byte strings that are simply concatenated.
//
// Three launches (all enqueue-only):
//   k_rle_sizes   lane-per-block (same LDS-DMA tile staging as the inverse kernel): bytes per block,
//                 plus the wave's total by a cross-lane sum;
//   k_scan_waves  one workgroup: exclusive scan of the wave totals -> 64-bit byte offset per wave;
//   k_rle_emit    lane-per-block again: in-wave exclusive scan of the block sizes, then every lane
//                 packs its block MSB-first through a 64-bit accumulator and writes its bytes.
// Blocks are independent, so this shards over GPUs exactly like the transform (jpegx/multigpu.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/jpegx.h"

extern "C" void jpegx_internal_set_error(const char *msg);  // jpegx_kernels.hip (thread-local string)

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE_BYTES = 64 * 128;
__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// workspace layout (bytes): [0,8) total, [8,12) error flag, [16, 16+8(nw+1)) wave offsets (u64),
// then wave totals (u32 x nw, 8-byte aligned), then block sizes (u32 x nblocks)
struct Workspace {
    unsigned long long *total;
    unsigned *error;
    unsigned long long *wave_off;
    unsigned *wave_bytes;
    unsigned *block_bytes;
};

__host__ __device__ inline size_t align8(size_t v) { return (v + 7) & ~(size_t)7; }

__host__ __device__ inline Workspace carve(void *ws, long long nblocks)
{
    const long long nw = (nblocks + 63) / 64;
    unsigned char *p = reinterpret_cast<unsigned char *>(ws);
    Workspace w;
    w.total = reinterpret_cast<unsigned long long *>(p);
    w.error = reinterpret_cast<unsigned *>(p + 8);
    w.wave_off = reinterpret_cast<unsigned long long *>(p + 16);
    size_t off = 16 + (size_t)(nw + 1) * 8;
    w.wave_bytes = reinterpret_cast<unsigned *>(p + off);
    off = align8(off + (size_t)nw * 4);
    w.block_bytes = reinterpret_cast<unsigned *>(p + off);
    return w;
}

size_t workspace_bytes(long long nblocks)
{
    const long long nw = (nblocks + 63) / 64;
    return align8(16 + (size_t)(nw + 1) * 8 + (size_t)nw * 4) + align8((size_t)nblocks * 4);
}

// the wave's 64 x 128 B of coefficients -> swizzled LDS tile (LDS-DMA, as in k_inverse_fused)
__device__ __forceinline__ void stage_tile(const int16_t *__restrict__ zz, int g0, int nblk, int lane, unsigned char *lds)
{
    const int row0 = lane >> 3, c = (lane & 7) ^ (lane >> 3);
    const int last = nblk - 1 - g0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = min(i * 8 + row0, last);
        const unsigned char *src = reinterpret_cast<const unsigned char *>(zz) + (size_t)(g0 + row) * 128 + c * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(lds + i * 1024), 16, 0, 2);
    }
    __syncthreads();
}

__device__ __forceinline__ void load_block(const unsigned char *lds, int lane, unsigned (&w)[32])
{
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const u32x4 q = *reinterpret_cast<const u32x4 *>(lds + tile_off(lane, c));
        w[c * 4 + 0] = q.x; w[c * 4 + 1] = q.y; w[c * 4 + 2] = q.z; w[c * 4 + 3] = q.w;
    }
}

__device__ __forceinline__ int coef(const unsigned (&w)[32], int p)
{
    return (p & 1) ? ((int)w[p >> 1] >> 16) : (int)(short)(w[p >> 1] & 0xFFFFu);
}

// bits of one block; sets bad when an amplitude needs more than 15 bits (util.py:140-149 BadRleCodeError)
__device__ __forceinline__ unsigned block_bits(const unsigned (&w)[32], bool &bad)
{
    unsigned bits = 8;  // EOB
    int prev = -1;
#pragma unroll
    for (int p = 0; p < 64; ++p) {
        const int q = coef(w, p);
        if (q != 0) {
            const int run = p - prev - 1;
            const int chains = (run >= 60) ? 4 : (run >= 45) ? 3 : (run >= 30) ? 2 : (run >= 15) ? 1 : 0;
            const unsigned mag = (unsigned)(q < 0 ? -q : q);
            const int bl = 32 - __clz((int)mag);
            bad |= bl > 14;
            bits += 8u * chains + 8u + (unsigned)bl + 1u;
            prev = p;
        }
    }
    return bits;
}

__global__ __launch_bounds__(64) void k_rle_sizes(const int16_t *__restrict__ zz, int nblk, void *ws)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[TILE_BYTES];
    const Workspace W = carve(ws, nblk);
    const int lane = threadIdx.x, g0 = blockIdx.x * 64, g = g0 + lane;
    stage_tile(zz, g0, nblk, lane, lds);
    unsigned w[32];
    load_block(lds, lane, w);
    bool bad = false;
    unsigned bytes = (block_bits(w, bad) + 7u) >> 3;
    if (g >= nblk) { bytes = 0; bad = false; }
    if (g < nblk) W.block_bytes[g] = bytes;
    if (__any(bad) && lane == 0) atomicOr(W.error, 1u);
    unsigned sum = bytes;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
    if (lane == 0) W.wave_bytes[blockIdx.x] = sum;
}

// exclusive scan of the per-wave totals by ONE workgroup of 1024 threads (chunked, carry in LDS)
__global__ __launch_bounds__(1024) void k_scan_waves(int nwaves, void *ws, int nblk)
{
    __shared__ unsigned long long part[1024];
    const Workspace W = carve(ws, nblk);
    const int t = threadIdx.x;
    const int per = (nwaves + 1023) / 1024;
    const int lo = min(t * per, nwaves), hi = min(lo + per, nwaves);
    unsigned long long s = 0;
    for (int i = lo; i < hi; ++i) s += W.wave_bytes[i];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                 // Hillis-Steele inclusive scan of the chunk sums
        const unsigned long long v = (t >= d) ? part[t - d] : 0ull;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned long long run = part[t] - s;                // exclusive prefix of this thread's chunk
    for (int i = lo; i < hi; ++i) {
        W.wave_off[i] = run;
        run += W.wave_bytes[i];
    }
    if (t == 1023) {
        W.wave_off[nwaves] = part[1023];
        *W.total = part[1023];
    }
}

__global__ __launch_bounds__(64) void k_rle_emit(const int16_t *__restrict__ zz, int nblk, const void *ws,
                                                 unsigned char *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[TILE_BYTES];
    const Workspace W = carve(const_cast<void *>(ws), nblk);
    const int lane = threadIdx.x, g0 = blockIdx.x * 64, g = g0 + lane;
    stage_tile(zz, g0, nblk, lane, lds);
    unsigned w[32];
    load_block(lds, lane, w);
    const unsigned mine = (g < nblk) ? W.block_bytes[g] : 0u;
    unsigned incl = mine;                                 // in-wave inclusive scan of the block sizes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    if (g >= nblk) return;
    unsigned char *p = out + W.wave_off[blockIdx.x] + (incl - mine);

    unsigned long long acc = 0;                           // MSB-first bit accumulator
    int nb = 0;
    auto put = [&](unsigned v, int n) {
        acc = (acc << n) | v;
        nb += n;
        while (nb >= 8) {
            *p++ = (unsigned char)(acc >> (nb - 8));
            nb -= 8;
        }
    };
    int prev = -1;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const int q = coef(w, i);
        if (q != 0) {
            int run = i - prev - 1;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (run >= 15) { put(0xF0u, 8); run -= 15; }          // (15, 0, 0): fifteen zeros
            const unsigned mag = (unsigned)(q < 0 ? -q : q);
            const int bl = 32 - __clz((int)mag);
            put(((unsigned)run << 4) | (unsigned)(bl + 1), 8);         // 4-bit run, 4-bit size
            put(((q > 0 ? 1u : 0u) << bl) | mag, bl + 1);              // sign + magnitude
            prev = i;
        }
    }
    put(0u, 8);                                                       // EOB
    if (nb) *p++ = (unsigned char)(acc << (8 - nb));                   // zero-pad to the byte boundary
}

int fail(int code, const char *msg)
{
    jpegx_internal_set_error(msg);
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            char buf_[400];                                                                \
            snprintf(buf_, sizeof(buf_), "%s failed: %s", #expr, hipGetErrorString(e_));   \
            return fail(JPEGX_E_HIP, buf_);                                                \
        }                                                                                  \
    } while (0)

int check_args(const void *zz, long long nblocks, const void *ws)
{
    if (!zz || !ws) return fail(JPEGX_E_INVALID, "null device pointer");
    if (nblocks <= 0 || nblocks > 0x7FFFFFC0LL) return fail(JPEGX_E_INVALID, "block count must be in 1 .. 2^31-64");
    if ((reinterpret_cast<uintptr_t>(zz) & 15u) || (reinterpret_cast<uintptr_t>(ws) & 15u))
        return fail(JPEGX_E_INVALID, "stream and workspace must be 16-byte aligned");
    return JPEGX_OK;
}

}  // namespace

extern "C" {

size_t jpegx_entropy_workspace_bytes(long long nblocks) { return nblocks > 0 ? workspace_bytes(nblocks) : 0; }

int jpegx_entropy_sizes(const int16_t *d_zz, long long nblocks, void *d_workspace, jpegx_stream_t stream)
{
    int rc = check_args(d_zz, nblocks, d_workspace);
    if (rc) return rc;
    const int nblk = (int)nblocks, nw = (nblk + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(d_workspace, 0, 16, st));
    hipLaunchKernelGGL(k_rle_sizes, dim3(nw), dim3(64), 0, st, d_zz, nblk, d_workspace);
    hipLaunchKernelGGL(k_scan_waves, dim3(1), dim3(1024), 0, st, nw, d_workspace, nblk);
    HIP_TRY(hipGetLastError());
    return JPEGX_OK;
}

int jpegx_entropy_total(const void *d_workspace, unsigned long long *h_total, jpegx_stream_t stream)
{
    if (!d_workspace || !h_total) return fail(JPEGX_E_INVALID, "null pointer");
    unsigned long long head[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(head, d_workspace, 16, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    *h_total = head[0];
    if ((unsigned)(head[1] & 0xFFFFFFFFull) != 0)
        return fail(JPEGX_E_INVALID, "BadRleCodeError: an amplitude needs more than 15 bits (|a| > 16383)");
    return JPEGX_OK;
}

int jpegx_entropy_emit(const int16_t *d_zz, long long nblocks, const void *d_workspace, uint8_t *d_out,
                       jpegx_stream_t stream)
{
    int rc = check_args(d_zz, nblocks, d_workspace);
    if (rc) return rc;
    if (!d_out) return fail(JPEGX_E_INVALID, "null output pointer");
    const int nblk = (int)nblocks;
    hipLaunchKernelGGL(k_rle_emit, dim3((nblk + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_zz, nblk, d_workspace, d_out);
    HIP_TRY(hipGetLastError());
    return JPEGX_OK;
}

int jpegx_host_entropy_encode(const int16_t *h_zz, long long nblocks, uint8_t *h_out, size_t cap, size_t *nbytes)
{
    if (!h_zz || !nbytes) return fail(JPEGX_E_INVALID, "null host pointer");
    if (nblocks <= 0 || nblocks > 0x7FFFFFC0LL) return fail(JPEGX_E_INVALID, "block count must be in 1 .. 2^31-64");
    void *dzz = nullptr, *dws = nullptr, *dout = nullptr;
    hipStream_t st = nullptr;
    int rc = JPEGX_OK;
    unsigned long long total = 0;
    do {
        if (hipMalloc(&dzz, (size_t)nblocks * 128) != hipSuccess || hipMalloc(&dws, workspace_bytes(nblocks)) != hipSuccess ||
            hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { rc = fail(JPEGX_E_HIP, "device allocation failed"); break; }
        if (hipMemcpyAsync(dzz, h_zz, (size_t)nblocks * 128, hipMemcpyHostToDevice, st) != hipSuccess) { rc = fail(JPEGX_E_HIP, "H2D copy failed"); break; }
        if ((rc = jpegx_entropy_sizes((const int16_t *)dzz, nblocks, dws, st))) break;
        if ((rc = jpegx_entropy_total(dws, &total, st))) break;
        *nbytes = (size_t)total;
        if (!h_out) break;                                   // size query only
        if (cap < total) { rc = fail(JPEGX_E_INVALID, "output buffer too small"); break; }
        if (hipMalloc(&dout, total ? total : 1) != hipSuccess) { rc = fail(JPEGX_E_HIP, "device allocation failed"); break; }
        if ((rc = jpegx_entropy_emit((const int16_t *)dzz, nblocks, dws, (uint8_t *)dout, st))) break;
        if (hipMemcpyAsync(h_out, dout, total, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { rc = fail(JPEGX_E_HIP, "D2H copy failed"); break; }
    } while (0);
    if (dzz) (void)hipFree(dzz);
    if (dws) (void)hipFree(dws);
    if (dout) (void)hipFree(dout);
    if (st) (void)hipStreamDestroy(st);
    return rc;
}

}  // extern "C"
