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
//   k_rle_sizes   lane-per-block (same LDS-DMA tile staging as the inverse kernel): a 64-bit
//                 non-zero mask per block, then a walk over the set bits only; bytes per block
//                 plus the wave's total by a cross-lane sum;
//   k_scan_level1/2  exclusive scan of the wave totals (chunked) -> byte offset of every wave;
//   k_rle_emit    lane-per-block again: in-wave exclusive scan of the block sizes, every lane packs
//                 its block MSB-first through a 64-bit accumulator into an LDS staging area laid
//                 out at the destination's offset modulo 16, and the wave copies the contiguous
//                 span out with aligned 16-byte stores (bytes only at the two ends).
// Blocks are independent, so this shards over GPUs exactly like the transform (jpegx/multigpu.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/jpegx.h"

extern "C" void jpegx_internal_set_error(const char *msg);  // jpegx_runtime.hip (thread-local string)

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE_BYTES = 64 * 128;
__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// workspace layout (bytes): [0,8) total, [8,12) error flag, [16, ...) 64-bit byte offset of every
// scan chunk (SCAN_CHUNK waves), then per wave its total and its 32-bit offset inside the chunk
// (both 16-byte aligned arrays), then block sizes (u32 x nblocks)
constexpr int SCAN_CHUNK = 4096;   // waves per level-1 scan workgroup (1024 threads x 4)

struct Workspace {
    unsigned long long *total;
    unsigned *error;
    unsigned long long *chunk_off;   // [nchunks + 1]
    unsigned *wave_bytes;            // [nw rounded up to SCAN_CHUNK]
    unsigned *wave_off;              // [nw rounded up to SCAN_CHUNK], offset inside the wave's chunk
    unsigned *block_bytes;           // [nblocks]
    unsigned *half_info;             // [nblocks] bits of the codes of coefficients 0..31 | (1 + last non-zero among them) << 12 (forward kernels that size their own blocks)
};

__host__ __device__ inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

__host__ __device__ inline Workspace carve(void *ws, long long nblocks)
{
    const long long nw = (nblocks + 63) / 64;
    const long long nchunks = (nw + SCAN_CHUNK - 1) / SCAN_CHUNK;
    unsigned char *p = reinterpret_cast<unsigned char *>(ws);
    Workspace w;
    w.total = reinterpret_cast<unsigned long long *>(p);
    w.error = reinterpret_cast<unsigned *>(p + 8);
    w.chunk_off = reinterpret_cast<unsigned long long *>(p + 16);
    size_t off = align16(16 + (size_t)(nchunks + 1) * 8);
    w.wave_bytes = reinterpret_cast<unsigned *>(p + off);
    off += (size_t)nchunks * SCAN_CHUNK * 4;
    w.wave_off = reinterpret_cast<unsigned *>(p + off);
    off += (size_t)nchunks * SCAN_CHUNK * 4;
    w.block_bytes = reinterpret_cast<unsigned *>(p + off);
    off += align16((size_t)nblocks * 4);
    w.half_info = reinterpret_cast<unsigned *>(p + off);
    return w;
}

size_t workspace_bytes(long long nblocks)
{
    const long long nw = (nblocks + 63) / 64;
    const long long nchunks = (nw + SCAN_CHUNK - 1) / SCAN_CHUNK;
    return align16(16 + (size_t)(nchunks + 1) * 8) + 2 * (size_t)nchunks * SCAN_CHUNK * 4 + 2 * align16((size_t)nblocks * 4);
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

__device__ __forceinline__ int bit_length(unsigned v) { return 32 - __clz((int)v); }   // 0 for v == 0

// Number of zero-chain codes of a block from its non-zero mask M (bit p = coefficient p != 0):
// sum over the non-zeros of floor(run_before / 15) = #{run >= 15} + #{run >= 30} + #{>= 45} + #{>= 60}.
// R_k has bit p set when positions p-k+1..p are all zero; a non-zero at p has run >= k iff bit p-1
// of R_k is set (shifts bring in zeros at the bottom, so short prefixes never qualify).
__device__ __forceinline__ unsigned chain_count(unsigned long long M)
{
    const unsigned long long z = ~M;
    const unsigned long long r2 = z & (z << 1), r4 = r2 & (r2 << 2), r8 = r4 & (r4 << 4);
    const unsigned long long r15 = r8 & (r8 << 7);
    const unsigned long long r30 = r15 & (r15 << 15);
    const unsigned long long r45 = r30 & (r15 << 30);
    const unsigned long long r60 = r30 & (r30 << 30);
    return __popcll((r15 << 1) & M) + __popcll((r30 << 1) & M) + __popcll((r45 << 1) & M) + __popcll((r60 << 1) & M);
}

__global__ __launch_bounds__(64) void k_rle_sizes(const int16_t *__restrict__ zz, int nblk, void *ws)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[TILE_BYTES];
    const Workspace W = carve(ws, nblk);
    const int lane = threadIdx.x, g0 = blockIdx.x * 64, g = g0 + lane;
    stage_tile(zz, g0, nblk, lane, lds);
    unsigned w[32];
    load_block(lds, lane, w);
    // branch-free: bits = EOB + sum over non-zeros (4 + 4 + 1 + bit_length) + 8 per zero chain
    unsigned sum_bl = 0, max_bl = 0, mlo = 0, mhi = 0;
#pragma unroll
    for (int p = 0; p < 64; ++p) {
        const int q = coef(w, p);
        const unsigned bl = (unsigned)bit_length((unsigned)(q < 0 ? -q : q));
        sum_bl += bl;
        max_bl = max(max_bl, bl);
        const unsigned bit = (q != 0) ? (1u << (p & 31)) : 0u;
        if (p < 32) mlo |= bit; else mhi |= bit;
    }
    const unsigned long long M = ((unsigned long long)mhi << 32) | mlo;
    const unsigned bits = 8u + sum_bl + 9u * (unsigned)__popcll(M) + 8u * chain_count(M);
    bool bad = max_bl > 14;  // an amplitude beyond 15 bits (util.py:140-149 BadRleCodeError)
    unsigned bytes = (bits + 7u) >> 3;
    if (g >= nblk) { bytes = 0; bad = false; }
    if (g < nblk) W.block_bytes[g] = bytes;
    if (__any(bad) && lane == 0) atomicOr(W.error, 1u);
    unsigned sum = bytes;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
    if (lane == 0) W.wave_bytes[blockIdx.x] = sum;
}

// Two-level exclusive scan of the per-wave totals.  Level 1: one workgroup per SCAN_CHUNK waves,
// 16 bytes per thread (coalesced), wave scan by __shfl_up + a 16-entry LDS carry.  Level 2: one
// wave walks the chunk totals (<= 128 for 2^31 blocks).
__global__ __launch_bounds__(1024) void k_scan_level1(int nwaves, void *ws, int nblk)
{
    __shared__ unsigned carry[16];
    const Workspace W = carve(ws, nblk);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int i0 = blockIdx.x * SCAN_CHUNK + t * 4;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (i0 + 3 < nwaves) {
        v = *reinterpret_cast<const u32x4 *>(W.wave_bytes + i0);
    } else {
        if (i0 + 0 < nwaves) v.x = W.wave_bytes[i0 + 0];
        if (i0 + 1 < nwaves) v.y = W.wave_bytes[i0 + 1];
        if (i0 + 2 < nwaves) v.z = W.wave_bytes[i0 + 2];
    }
    // bit 31 of a wave's total: an amplitude beyond 15 bits there (set by the forward kernels that size their own blocks)
    const bool bad = ((v.x | v.y | v.z | v.w) & 0x80000000u) != 0u;
    v.x &= 0x7FFFFFFFu; v.y &= 0x7FFFFFFFu; v.z &= 0x7FFFFFFFu; v.w &= 0x7FFFFFFFu;
    if (__any(bad) && lane == 0) atomicOr(W.error, 1u);
    const unsigned s = v.x + v.y + v.z + v.w;
    unsigned incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned u = __shfl_up(incl, d);
        if (lane >= d) incl += u;
    }
    if (lane == 63) carry[wv] = incl;
    __syncthreads();
    unsigned base = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) base += (k < wv) ? carry[k] : 0u;
    const unsigned e = base + incl - s;                   // exclusive prefix of this thread's 4 waves
    *reinterpret_cast<u32x4 *>(W.wave_off + i0) = u32x4{e, e + v.x, e + v.x + v.y, e + v.x + v.y + v.z};
    if (t == 1023) W.chunk_off[blockIdx.x] = (unsigned long long)(base + incl);   // chunk total, scanned next
}

// The two levels in one launch when the waves fit one chunk (a 4096 x 4096 band: exactly), for the path whose sizes come
// from the forward kernel: offsets, total and the error flag are WRITTEN here (bit 31 of the waves' totals), so the
// workspace's head needs no clearing beforehand.
__global__ __launch_bounds__(1024) void k_scan_one_chunk(int nwaves, void *ws, int nblk)
{
    __shared__ unsigned carry[16];
    __shared__ unsigned bad_s[16];
    const Workspace W = carve(ws, nblk);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int i0 = t * 4;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (i0 + 3 < nwaves) {
        v = *reinterpret_cast<const u32x4 *>(W.wave_bytes + i0);
    } else {
        if (i0 + 0 < nwaves) v.x = W.wave_bytes[i0 + 0];
        if (i0 + 1 < nwaves) v.y = W.wave_bytes[i0 + 1];
        if (i0 + 2 < nwaves) v.z = W.wave_bytes[i0 + 2];
    }
    const bool bad = ((v.x | v.y | v.z | v.w) & 0x80000000u) != 0u;
    v.x &= 0x7FFFFFFFu; v.y &= 0x7FFFFFFFu; v.z &= 0x7FFFFFFFu; v.w &= 0x7FFFFFFFu;
    const unsigned s = v.x + v.y + v.z + v.w;
    unsigned incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned u = __shfl_up(incl, d);
        if (lane >= d) incl += u;
    }
    if (lane == 63) carry[wv] = incl;
    if (lane == 0) bad_s[wv] = __any(bad) ? 1u : 0u;
    __syncthreads();
    unsigned base = 0, anybad = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { base += (k < wv) ? carry[k] : 0u; anybad |= bad_s[k]; }
    const unsigned e = base + incl - s;
    *reinterpret_cast<u32x4 *>(W.wave_off + i0) = u32x4{e, e + v.x, e + v.x + v.y, e + v.x + v.y + v.z};
    if (t == 1023) {
        const unsigned long long total = (unsigned long long)(base + incl);
        W.chunk_off[0] = 0;
        W.chunk_off[1] = total;
        *W.total = total;
        *W.error = anybad;
    }
}

__global__ __launch_bounds__(64) void k_scan_level2(int nchunks, void *ws, int nblk)
{
    const Workspace W = carve(ws, nblk);
    if (threadIdx.x != 0) return;
    unsigned long long run = 0;
    for (int c = 0; c < nchunks; ++c) {
        const unsigned long long t = W.chunk_off[c];
        W.chunk_off[c] = run;
        run += t;
    }
    W.chunk_off[nchunks] = run;
    *W.total = run;
}

// Worst case per block: 63 non-zeros x (8 + 15) bits + DC 23 + EOB 8 < 1500 bits = 188 bytes.
constexpr int EMIT_STAGE_BYTES = 64 * 192 + 32;

__global__ __launch_bounds__(64) void k_rle_emit(const int16_t *__restrict__ zz, int nblk, const void *ws,
                                                 unsigned char *__restrict__ out)
{
    // [tile 8 KiB | bit staging]: the wave's bit string is assembled in LDS as big-endian 32-bit
    // words (ds_or_b32 at each code's bit offset -- no per-byte loops), laid out at the same offset
    // modulo 16 as the global destination, then byte-swapped and copied out with aligned 16-byte
    // stores (single bytes only at the two ends, which neighbouring waves share).
    // (the coefficient tile is dead once the block sits in registers, so the staging area reuses it)
    __shared__ __attribute__((aligned(16))) unsigned char lds[EMIT_STAGE_BYTES];
    unsigned *stage = reinterpret_cast<unsigned *>(lds);
    const Workspace W = carve(const_cast<void *>(ws), nblk);
    // the sizes pass flagged an amplitude beyond 15 bits: such a block can exceed the staging area (and
    // the reference raises BadRleCodeError for it), so nothing is emitted -- whether or not the caller
    // looked at jpegx_entropy_total's return code
    if (*W.error != 0) return;
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
    const unsigned total = __shfl(incl, 63);
    unsigned char *gdst = out + W.chunk_off[blockIdx.x / SCAN_CHUNK] + W.wave_off[blockIdx.x];   // wave-uniform
    const unsigned skew = (unsigned)(reinterpret_cast<uintptr_t>(gdst) & 15u);
    const unsigned end = skew + total;                    // bytes of staging in use

    __syncthreads();                                      // every lane has its block in registers
    for (unsigned c = lane * 16u; c < end + 16u; c += 64u * 16u)
        *reinterpret_cast<u32x4 *>(reinterpret_cast<unsigned char *>(stage) + c) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();

    if (g < nblk) {
        unsigned o = 8u * (skew + incl - mine);           // bit offset of the next code
        auto put = [&](unsigned code, int n) {            // n <= 24 bits, MSB-first at bit offset o
            const unsigned long long v = (unsigned long long)code << (64 - n - (int)(o & 31u));
            atomicOr(&stage[o >> 5], (unsigned)(v >> 32));
            const unsigned lo = (unsigned)v;
            if (lo) atomicOr(&stage[(o >> 5) + 1], lo);
            o += (unsigned)n;
        };
        int prev = -1;
#pragma unroll
        for (int p = 0; p < 64; ++p) {
            const int q = coef(w, p);
            if (q != 0) {
                int run = p - prev - 1;
                while (run >= 15) { put(0xF0u, 8); run -= 15; }           // (15, 0, 0): fifteen zeros
                const unsigned mag = (unsigned)(q < 0 ? -q : q);
                const int bl = bit_length(mag);
                const unsigned hdr = ((unsigned)run << 4) | (unsigned)(bl + 1);   // 4-bit run, 4-bit size
                put((hdr << (bl + 1)) | ((q > 0 ? 1u : 0u) << bl) | mag, 9 + bl);  // + sign + magnitude
                prev = p;
            }
        }
        // EOB (8 zero bits) and the zero padding to the byte boundary are already there
    }
    __syncthreads();

    unsigned char *gbase = gdst - skew;                   // 16-byte aligned
    for (unsigned c = lane * 16u; c < end; c += 64u * 16u) {
        u32x4 t = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const unsigned char *>(stage) + c);
        t.x = __builtin_bswap32(t.x); t.y = __builtin_bswap32(t.y);
        t.z = __builtin_bswap32(t.z); t.w = __builtin_bswap32(t.w);
        if (c >= skew && c + 16u <= end) {
            __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(gbase + c));
        } else {
            const unsigned wd[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (unsigned k = 0; k < 16u; ++k)
                if (c + k >= skew && c + k < end) gbase[c + k] = (unsigned char)(wd[k >> 2] >> (8u * (k & 3u)));
        }
    }
}

// k_rle_emit with TWO lanes per block (workgroup of two waves: wave 0 emits the codes of coefficients 0..31, wave 1 those
// of 32..63, from the bit offset and the last non-zero the forward kernel left in half_info).  A single 4096 x 4096
// band is 4096 such workgroups -- a handful per CU -- and the time of this kernel is then the length of one lane's
// dependent walk: 32 steps instead of 64.
__global__ __launch_bounds__(128) void k_rle_emit2(const int16_t *__restrict__ zz, int nblk, const void *ws,
                                                   unsigned char *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[EMIT_STAGE_BYTES];
    unsigned *stage = reinterpret_cast<unsigned *>(lds);
    const Workspace W = carve(const_cast<void *>(ws), nblk);
    if (*W.error != 0) return;                           // an amplitude beyond 15 bits: nothing is emitted (see k_rle_emit)
    const int t = threadIdx.x, lane = t & 63, half = t >> 6, g0 = blockIdx.x * 64, g = g0 + lane;
    {
        // the wave's 64 x 128 B of coefficients -> swizzled LDS tile, four LDS-DMA pieces per wave
        const int row0 = lane >> 3, c = (lane & 7) ^ (lane >> 3);
        const int last = nblk - 1 - g0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = half * 4 + i, row = min(piece * 8 + row0, last);
            const unsigned char *src = reinterpret_cast<const unsigned char *>(zz) + (size_t)(g0 + row) * 128 + c * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(lds + piece * 1024), 16, 0, 2);
        }
    }
    const unsigned mine = (g < nblk) ? W.block_bytes[g] : 0u;
    const unsigned info = (g < nblk) ? W.half_info[g] : 0u;
    unsigned char *gdst = out + W.chunk_off[blockIdx.x / SCAN_CHUNK] + W.wave_off[blockIdx.x];   // workgroup-uniform
    __syncthreads();
    unsigned w[16];                                      // the lane's half of the block: coefficients 32 half .. 32 half + 31
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const u32x4 q = *reinterpret_cast<const u32x4 *>(lds + tile_off(lane, half * 4 + c));
        w[c * 4 + 0] = q.x; w[c * 4 + 1] = q.y; w[c * 4 + 2] = q.z; w[c * 4 + 3] = q.w;
    }
    unsigned incl = mine;                                 // in-wave inclusive scan of the block sizes (both waves: same numbers)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    const unsigned total = __shfl(incl, 63);
    const unsigned skew = (unsigned)(reinterpret_cast<uintptr_t>(gdst) & 15u);
    const unsigned end = skew + total;                    // bytes of staging in use
    __syncthreads();                                      // every lane has its half block in registers
    for (unsigned c = t * 16u; c < end + 16u; c += 128u * 16u)
        *reinterpret_cast<u32x4 *>(reinterpret_cast<unsigned char *>(stage) + c) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();

    if (g < nblk) {
        unsigned o = 8u * (skew + incl - mine) + (half ? (info & 0xFFFu) : 0u);   // bit offset of the lane's first code
        auto put = [&](unsigned code, int n) {            // n <= 24 bits, MSB-first at bit offset o
            const unsigned long long v = (unsigned long long)code << (64 - n - (int)(o & 31u));
            atomicOr(&stage[o >> 5], (unsigned)(v >> 32));
            const unsigned lo = (unsigned)v;
            if (lo) atomicOr(&stage[(o >> 5) + 1], lo);
            o += (unsigned)n;
        };
        int prev = half ? (int)(info >> 12) - 1 - 32 : -1;            // last non-zero in front, relative to the lane's first coefficient
#pragma unroll
        for (int p = 0; p < 32; ++p) {
            const int q = (p & 1) ? ((int)w[p >> 1] >> 16) : (int)(short)(w[p >> 1] & 0xFFFFu);
            if (q != 0) {
                int run = p - prev - 1;
                while (run >= 15) { put(0xF0u, 8); run -= 15; }           // (15, 0, 0): fifteen zeros
                const unsigned mag = (unsigned)(q < 0 ? -q : q);
                const int bl = bit_length(mag);
                const unsigned hdr = ((unsigned)run << 4) | (unsigned)(bl + 1);   // 4-bit run, 4-bit size
                put((hdr << (bl + 1)) | ((q > 0 ? 1u : 0u) << bl) | mag, 9 + bl);  // + sign + magnitude
                prev = p;
            }
        }
        // EOB (8 zero bits) and the zero padding to the byte boundary are already there
    }
    __syncthreads();

    unsigned char *gbase = gdst - skew;                   // 16-byte aligned
    for (unsigned c = t * 16u; c < end; c += 128u * 16u) {
        u32x4 x = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const unsigned char *>(stage) + c);
        x.x = __builtin_bswap32(x.x); x.y = __builtin_bswap32(x.y);
        x.z = __builtin_bswap32(x.z); x.w = __builtin_bswap32(x.w);
        if (c >= skew && c + 16u <= end) {
            __builtin_nontemporal_store(x, reinterpret_cast<u32x4 *>(gbase + c));
        } else {
            const unsigned wd[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (unsigned k = 0; k < 16u; ++k)
                if (c + k >= skew && c + k < end) gbase[c + k] = (unsigned char)(wd[k >> 2] >> (8u * (k & 3u)));
        }
    }
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
            (void)hipGetLastError(); /* reported here: must not linger as the thread's last error */ \
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
    const int nchunks = (nw + SCAN_CHUNK - 1) / SCAN_CHUNK;
    hipLaunchKernelGGL(k_rle_sizes, dim3(nw), dim3(64), 0, st, d_zz, nblk, d_workspace);
    hipLaunchKernelGGL(k_scan_level1, dim3(nchunks), dim3(1024), 0, st, nw, d_workspace, nblk);
    hipLaunchKernelGGL(k_scan_level2, dim3(1), dim3(64), 0, st, nchunks, d_workspace, nblk);
    HIP_TRY(hipGetLastError());
    return JPEGX_OK;
}

// internal: where the forward kernels that size their own blocks put the sizes, and the scans behind them
void jpegx_internal_entropy_views(void *d_workspace, long long nblocks, unsigned **block_bytes, unsigned **wave_bytes, unsigned **half_info)
{
    const Workspace W = carve(d_workspace, nblocks);
    *block_bytes = W.block_bytes;
    *wave_bytes = W.wave_bytes;
    *half_info = W.half_info;
}

// the emitter for streams whose forward kernel left half_info behind: two lanes per block
int jpegx_internal_entropy_emit2(const int16_t *d_zz, long long nblocks, const void *d_workspace, uint8_t *d_out, jpegx_stream_t stream)
{
    int rc = check_args(d_zz, nblocks, d_workspace);
    if (rc) return rc;
    if (!d_out) return fail(JPEGX_E_INVALID, "null output pointer");
    const int nblk = (int)nblocks;
    hipLaunchKernelGGL(k_rle_emit2, dim3((nblk + 63) / 64), dim3(128), 0, (hipStream_t)stream, d_zz, nblk, d_workspace, d_out);
    HIP_TRY(hipGetLastError());
    return JPEGX_OK;
}

int jpegx_internal_entropy_scan(long long nblocks, void *d_workspace, jpegx_stream_t stream)
{
    if (!d_workspace || nblocks <= 0 || nblocks > 0x7FFFFFC0LL) return fail(JPEGX_E_INVALID, "bad scan arguments");
    const int nblk = (int)nblocks, nw = (nblk + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
    const int nchunks = (nw + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (nchunks == 1) {
        hipLaunchKernelGGL(k_scan_one_chunk, dim3(1), dim3(1024), 0, st, nw, d_workspace, nblk);
    } else {
        HIP_TRY(hipMemsetAsync(d_workspace, 0, 16, st));
        hipLaunchKernelGGL(k_scan_level1, dim3(nchunks), dim3(1024), 0, st, nw, d_workspace, nblk);
        hipLaunchKernelGGL(k_scan_level2, dim3(1), dim3(64), 0, st, nchunks, d_workspace, nblk);
    }
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

int jpegx_entropy_block_sizes(const void *d_workspace, long long nblocks, uint32_t *h_sizes, jpegx_stream_t stream)
{
    if (!d_workspace || !h_sizes) return fail(JPEGX_E_INVALID, "null pointer");
    if (nblocks <= 0 || nblocks > 0x7FFFFFC0LL) return fail(JPEGX_E_INVALID, "block count must be in 1 .. 2^31-64");
    const Workspace W = carve(const_cast<void *>(d_workspace), nblocks);
    HIP_TRY(hipMemcpyAsync(h_sizes, W.block_bytes, (size_t)nblocks * 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
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
