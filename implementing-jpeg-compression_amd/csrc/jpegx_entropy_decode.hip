// jpegx_entropy_decode.hip -- the entropy stage INVERTED on the GPU: RleBytestream.invert
// (pipeline/rle_byte_stream.py:61-88) + RunLengthEncoding.invert (pipeline/run_length_encoding.py:66-97)
// for dct_size 8, bytes -> int16 [nblocks][64].
//
// The byte stream carries no index: where a block starts is only known once the block before it has been
// parsed, which is why the reference (and libjpegx's host parser, jpegx_host.cpp) walk it sequentially.
// What the format does guarantee is that every block ends with a zero byte -- the 8-bit end marker plus the
// zero padding to the next byte boundary always leave the last byte of a block 0x00 -- so a block can only
// start at position 0 or right behind a 0x00 byte.  That turns the sequential walk into parallel work.  Two schemes
// live in this file: the segmented one of round 3 (second half: three launches, what the host pipeline runs) and round
// 2's whole-stream scheme, kept as the fallback for streams the segment tables do not fit, which works like this:
//   1. candidates  every position 0 or behind a 0x00 byte, compacted in stream order (count + scan + scatter);
//   2. parse       one thread per candidate parses ONE block from there and records at which candidate the
//                  next block would start (false candidates -- zero bytes inside a block's amplitude bits --
//                  parse garbage; they are simply never reached);
//   3. jump tables J_k[c] = the candidate 4^k blocks behind candidate c (radix-4 pointer jumping, log4(nblocks) passes);
//   4. starts      block i starts at the candidate reached from position 0 by following the base-4 digits of i;
//   5. decode      lane per block, values into an LDS tile, coalesced 1 KiB stores into the zigzag stream.
// Steps 1-4 read the stream twice and touch one to two candidates per block; the result equals the host parser's.
// One difference on DAMAGED input: the host parser skips the padding bits unread, so a stream whose padding has
// been tampered with still parses there, while here the block behind it is not found (its start is not behind a
// 0x00 byte) and the stream is refused; the callers (decompress_band) then take the host parser.  The encoder
// always writes zero padding (rle_byte_stream.py:55-56).  tests/test_gpu_entropy.py fuzzes both decoders.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/jpegx.h"
#include "jpegx_entropy_decode.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned NIL = 0xFFFFFFFFu;
constexpr int CHUNK = 4096;              // bytes per workgroup of 256 threads in the candidate passes
__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// number of candidate positions contributed by bytes [q0, q0+16): position q+1 for every zero byte q,
// plus position 0 for the very first thread
__device__ __forceinline__ unsigned zero_mask16(const unsigned char *__restrict__ bytes, size_t q0, size_t nbytes)
{
    unsigned m = 0;
    if (q0 + 16 <= nbytes) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(bytes + q0);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 16; ++k) m |= (((w[k >> 2] >> (8 * (k & 3))) & 0xFFu) == 0u ? 1u : 0u) << k;
    } else {
        for (int k = 0; k < 16; ++k)
            if (q0 + k < nbytes && bytes[q0 + k] == 0) m |= 1u << k;
    }
    return m;
}

__global__ __launch_bounds__(256) void k_dec_count(const unsigned char *__restrict__ bytes, size_t nbytes, unsigned *__restrict__ chunk_count)
{
    __shared__ unsigned part[4];
    const size_t q0 = (size_t)blockIdx.x * CHUNK + (size_t)threadIdx.x * 16;
    unsigned c = q0 < nbytes ? __popc(zero_mask16(bytes, q0, nbytes)) : 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) c += 1;                 // position 0
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) chunk_count[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// exclusive scan of the chunk counts by ONE workgroup (a 96 MB stream has 24 K chunks); head[0] = total
__global__ __launch_bounds__(1024) void k_dec_scan(unsigned *__restrict__ chunk_count, int nchunks, unsigned *__restrict__ head)
{
    __shared__ unsigned carry[16];
    __shared__ unsigned base_s;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) base_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < nchunks; i0 += 1024) {
        const int i = i0 + t;
        const unsigned v = i < nchunks ? chunk_count[i] : 0u;
        unsigned incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned u = __shfl_up(incl, d);
            if (lane >= d) incl += u;
        }
        if (lane == 63) carry[wv] = incl;
        __syncthreads();
        unsigned base = base_s;
#pragma unroll
        for (int k = 0; k < 16; ++k) base += (k < wv) ? carry[k] : 0u;
        if (i < nchunks) chunk_count[i] = base + incl - v;          // exclusive prefix, in place
        __syncthreads();
        if (t == 1023) base_s = base + incl;
        __syncthreads();
    }
    if (t == 0) head[0] = base_s;
}

__global__ __launch_bounds__(256) void k_dec_scatter(const unsigned char *__restrict__ bytes, size_t nbytes,
                                                     const unsigned *__restrict__ chunk_off, unsigned *__restrict__ cand_pos)
{
    __shared__ unsigned part[4];
    const size_t q0 = (size_t)blockIdx.x * CHUNK + (size_t)threadIdx.x * 16;
    const unsigned m = q0 < nbytes ? zero_mask16(bytes, q0, nbytes) : 0u;
    const bool first = blockIdx.x == 0 && threadIdx.x == 0;
    const unsigned c = __popc(m) + (first ? 1u : 0u);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned u = __shfl_up(incl, d);
        if (lane >= d) incl += u;
    }
    if (lane == 63) part[wv] = incl;
    __syncthreads();
    unsigned o = chunk_off[blockIdx.x] + incl - c;
    for (int k = 0; k < wv; ++k) o += part[k];
    if (first) cand_pos[o++] = 0u;
    unsigned mm = m;
    while (mm) {
        const int k = __ffs((int)mm) - 1;
        mm &= mm - 1;
        cand_pos[o++] = (unsigned)(q0 + k + 1);
    }
}

// parse one block starting at byte `p` (p <= nbytes): returns the byte position behind it, or NIL if what is
// there is not a block (illegal code, more than 64 coefficients, or the stream ends inside it).  With `tile` != null
// the coefficients are written into the lane's row of the LDS tile (which must be zero).
// These loops are ALU-bound (every lane walks its own block: ~40 dependent steps on noise), so the step is kept
// short: 32-bit positions relative to the block's first dword, the stream seen through two big-endian dword
// registers and one v_alignbit_b32, a new dword shifted in whenever the position crosses a dword boundary (a code
// is at most 23 bits: at most once per code) and requested one step ahead.  The buffer is dword aligned and at
// least 16 readable bytes longer than the stream.
template <bool WRITE>
__device__ __forceinline__ unsigned parse_block(const unsigned *__restrict__ words, unsigned long long nbits, unsigned p,
                                                unsigned char *tile, int row, int linear_stride = 0)   // 0: the swizzled tile; else bytes per plain row
{
    const unsigned long long left = nbits - (unsigned long long)p * 8u;
    const unsigned *wp = words + (p >> 2);
    const unsigned first = (p & 3u) * 8u;                                  // the block's first bit, counted from wp[0]
    const unsigned end = first + (left > 0x7FFFFF00ull ? 0x7FFFFF00u : (unsigned)left);   // the stream's last bit + 1, same origin
    unsigned pos = first, wi = 0;
    unsigned hi = __builtin_bswap32(wp[0]), lo = __builtin_bswap32(wp[1]), ahead = wp[2];
    unsigned n = 0, ret = NIL;
    // one exit test per step and everything else by selects: with an early return at every check the compiler
    // spends half of the loop's instructions moving registers between the divergent paths
    for (int it = 0; it < 66; ++it) {
        const unsigned sh = pos & 31u;
        const unsigned fun = __builtin_amdgcn_alignbit(hi, lo, 32u - sh);   // (hi:lo) >> (32 - sh); sh = 0 needs hi itself
        const unsigned w = sh ? fun : hi;
        const unsigned run = w >> 28, size = (w >> 24) & 15u;
        const bool zero = size == 0;
        const bool eob = (w >> 24) == 0;                                    // end marker, then the zero padding
        const unsigned nn = n + (zero ? 15u : run);                         // a chain code is FIFTEEN zeros (util.py:134-154)
        // not a block: the stream ends inside the code, a zero size with a run other than 0 / 15, a size of 1 (a sign bit
        // without amplitude bits: the reference's decode_signed fails on it, rle_byte_stream.py:35-42), a 65th coefficient
        const bool bad = (pos + 8 + size > end) | (zero & (run != 15u) & !eob) | (size == 1u) | (!eob & (nn > (zero ? 64u : 63u)));
        if (WRITE && !zero && !bad) {
            const unsigned bits = (w << 8) >> (32 - size);
            const unsigned mag = bits & ((1u << (size - 1)) - 1u);
            const int amp = (bits >> (size - 1)) ? (int)mag : -(int)mag;                    // sign bit '1' = positive
            unsigned char *at = linear_stride ? tile + row * linear_stride + nn * 2u : tile + tile_off(row, (int)(nn >> 3)) + (nn & 7u) * 2;
            *reinterpret_cast<int16_t *>(at) = (int16_t)amp;
        }
        if (bad | eob) {
            ret = bad ? NIL : p + ((pos - first + 8 + 7) >> 3);
            break;
        }
        n = nn + (zero ? 0u : 1u);
        pos += 8 + size;
        if ((pos >> 5) != wi) {                                            // crossed into the next dword: shift it in
            ++wi;
            hi = lo;
            lo = __builtin_bswap32(ahead);
            ahead = wp[wi + 2];                                            // pos <= end: at most 11 bytes into the slack
        }
    }
    return ret;
}

__global__ __launch_bounds__(256) void k_dec_parse(const unsigned *__restrict__ words, size_t nbytes, const unsigned *__restrict__ cand_pos,
                                                   unsigned ncand, unsigned *__restrict__ J0)
{
    const unsigned c = blockIdx.x * 256u + threadIdx.x;
    if (c >= ncand) return;
    const unsigned p = cand_pos[c];
    unsigned nxt = NIL;
    if (p < nbytes) {
        const unsigned e = parse_block<false>(words, (unsigned long long)nbytes * 8u, p, nullptr, 0);
        if (e != NIL) {
            // the position behind a well-formed block is itself a candidate (its last byte is 0x00): find its index
            unsigned lo = c + 1, hi = ncand;
            while (lo < hi) {
                const unsigned mid = (lo + hi) >> 1;
                if (cand_pos[mid] < e) lo = mid + 1; else hi = mid;
            }
            if (lo < ncand && cand_pos[lo] == e) nxt = lo;
        }
    }
    J0[c] = nxt;
}

// J_{k+1}[c] = the candidate 4^(k+1) blocks behind c = four steps of J_k (radix-4 pointer jumping: half as many
// passes -- and launches, which is what a 4096 x 4096 band's decode is made of -- as doubling)
__global__ __launch_bounds__(256) void k_dec_jump(const unsigned *__restrict__ Jk, unsigned *__restrict__ Jk1, unsigned ncand)
{
    const unsigned c = blockIdx.x * 256u + threadIdx.x;
    if (c >= ncand) return;
    unsigned a = Jk[c];
#pragma unroll
    for (int step = 0; step < 3; ++step) a = (a == NIL) ? NIL : Jk[a];
    Jk1[c] = a;
}

__global__ __launch_bounds__(256) void k_dec_starts(const unsigned *__restrict__ J, int levels, unsigned ncand,
                                                    const unsigned *__restrict__ cand_pos, size_t nbytes, unsigned nblocks,
                                                    unsigned *__restrict__ start_pos, unsigned *__restrict__ head)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nblocks) return;
    unsigned cur = 0;                                   // candidate 0 = position 0
    for (int k = 0; k < levels && cur != NIL; ++k) {    // follow the base-4 digits of i
        const unsigned digit = (i >> (2 * k)) & 3u;
        for (unsigned d = 0; d < digit && cur != NIL; ++d) cur = J[(size_t)k * ncand + cur];
    }
    const unsigned p = cur == NIL ? NIL : cand_pos[cur];
    if (p == NIL || p >= nbytes) atomicOr(&head[1], 1u);            // fewer than nblocks blocks in the stream
    start_pos[i] = p;
}

__device__ __forceinline__ void dec_blocks_body(const unsigned *__restrict__ words, size_t nbytes, const unsigned *__restrict__ start_pos,
                                                int nblk, int16_t *__restrict__ out, unsigned *__restrict__ head)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 128];
    const int lane = threadIdx.x, g0 = blockIdx.x * 64, g = g0 + lane;
#pragma unroll
    for (int c = 0; c < 8; ++c) *reinterpret_cast<u32x4 *>(lds + lane * 128 + c * 16) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    if (g < nblk) {
        const unsigned p = start_pos[g];
        const unsigned e = p == NIL ? NIL : parse_block<true>(words, (unsigned long long)nbytes * 8u, p, lds, lane);
        if (e == NIL) atomicOr(&head[1], 2u);
        // the last block must end where the stream ends: bytes (or whole blocks) behind it make the reference fail
        // in its reshape (run_length_encoding.py:77-79), so they are refused here too
        else if (g == nblk - 1 && e != (unsigned)nbytes) atomicOr(&head[1], 4u);
    }
    __syncthreads();
    unsigned char *dst = reinterpret_cast<unsigned char *>(out) + (size_t)g0 * 128;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = i * 8 + (lane >> 3), c = lane & 7;
        const u32x4 q = *reinterpret_cast<const u32x4 *>(lds + tile_off(row, c));
        if (g0 + row < nblk) *reinterpret_cast<u32x4 *>(dst + (size_t)row * 128 + c * 16) = q;
    }
}

__global__ __launch_bounds__(64) void k_dec_blocks(const unsigned *__restrict__ words, size_t nbytes, const unsigned *__restrict__ start_pos,
                                                   int nblk, int16_t *__restrict__ out, unsigned *__restrict__ head)
{
    dec_blocks_body(words, nbytes, start_pos, nblk, out, head);
}

// ================================================================================================
// Round 3: block starts by SEGMENTS, blocks decoded from LDS by four lanes each -- three launches, no host round trip.
//
// The pointer-jumping scheme above works on the whole stream at once: candidate compaction (3 launches), one
// parse per candidate with a binary search in global memory for the candidate behind it, log4(blocks) jump
// passes (8 launches for a 4096 x 4096 band), a host read-back of the candidate count in the middle, and a
// parse of every block straight from global memory (one dependent load per dword of the block: latency bound).
// What it does not use: a block is at most 185 bytes long, so everything about a stretch of the stream can be
// resolved LOCALLY, in LDS, except which of a handful of candidates the chain of true blocks enters the stretch
// at -- and that is 16 bits the segment in front can hand over.
//
// k_seg_starts, one wave per segment of `seg` bytes (about 46 candidates: one parse pass of the wave's 64 lanes):
//   1. the segment (+ 16 bytes in front, + 208 behind: the longest block and the bit reader's look-ahead) into LDS
//      as big-endian dwords; candidates = position 0 and every position behind a 0x00 byte, compacted in order;
//   2. ONE parse per candidate from LDS, three codes per step, leaving way marks (where the walk stands after 15, 30,
//      45 codes); J[0][c] = the candidate the block at c ends at | EXIT + position in the next segment | END |
//      INVALID; binary lifting in LDS: J[k][c] = the candidate 2^k blocks on;
//   3. candidates within the first 192 bytes are the only places the chain can enter at.  Chains merge within a
//      block or two, so for almost every segment all entries agree on the exit: that exit is PUBLISHED AT ONCE
//      (one 4-byte agent-scope store), before the segment knows its own entry;
//   4. the segment reads the exit of the segment in front (a relaxed agent-scope poll; the only wait in the scheme,
//      for a neighbour that started at the same moment), which names its entry, hence its number of blocks;
//   5. block r of the segment is the candidate r steps from the entry: byte position + way marks -> prov[s][r],
//      the count -> C[s].  Nothing here needs the block's index in the plane.
// k_seg_scan, one workgroup: exclusive scan of the counts (first block index of every segment), the segment every
// 64th block lies in, and the check that the stream holds the plane's number of blocks.
// k_dec_blocks_lds, one workgroup of four waves per 64 blocks: the blocks' bytes (one contiguous span of the stream)
// into LDS with 16-byte loads, wave p decodes every block from its p-th way mark to the next into the block's row
// of an LDS tile, and the tile leaves as 1 KiB stores.
// Built and measured on the way (profiles/r03_decode_designs.txt): ONE launch with a decoupled look-back for the
// block indices and the blocks decoded speculatively while they are parsed (the tile quadruples the LDS per wave,
// 9 waves per CU then sit in the look-back's waits, half of the decoding is for false candidates: 125-190 us); an
// atomic ticket instead of the workgroup index (one word hands out ~88 tickets per microsecond: +45 us); the block
// indices by a last-arriver per group of 64 segments (its scatter is the kernel's tail: +30 us); grids cut to what
// the chip holds at once, every wave walking through several segments (no gain: 47 against 46 us, and the loop's
// scalar registers cost three waves per CU).
// The stream is read twice; what reaches global memory is 16 bytes per block and 8 per segment.  Every spin is
// bounded; a wave that gives up poisons the segments behind it and the host takes the scheme above (also for a
// segment with more candidates than the tables hold).  Everything the chain is made of is derived from the exit in
// front, so the early publication can only ever be wrong for a stream that is refused anyway (its entry turns out
// INVALID).
// ================================================================================================
namespace seg {

typedef __attribute__((address_space(1))) unsigned gu32;
#define JPEGX_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int REACH = 192;                 // a block is at most 185 bytes: the chain enters a segment within its first REACH bytes
constexpr int ROW = 144;                   // bytes per tile row (128 + 16: rows stay 16-byte aligned, the bank pattern of a column of rows spreads)
constexpr int FRONT = 16, TAIL = 208;      // window bytes in front of / behind the segment
constexpr unsigned INVALID = 0xFFFFu, END = 0xFFFEu, AFTER = 0xFFFDu, POISON = 0xFFFCu, EXIT = 0x8000u;   // EXIT | position in the NEXT segment
constexpr unsigned TAG = 0x10000u;         // set in every published exit word (0 = not there yet)
constexpr unsigned SPIN_LIMIT = 1u << 18;  // polls (each behind an s_sleep) before a wave gives the stream back
__device__ __forceinline__ bool is_exit(unsigned x) { return (x & EXIT) != 0 && x < POISON; }

struct Ws {
    unsigned *head;              // [1] error bits, [2] fallback bits, [3] blocks found
    unsigned *next_head;         // the next call's status block (the block decoder leaves it zero)
    unsigned *E;                 // [nseg] TAG | exit code; zero between calls (the block decoder sees to that)
    unsigned *C;                 // [nseg] the segment's blocks
    unsigned *F;                 // [nseg + 1] index of the segment's first block (k_seg_scan)
    unsigned *wave_seg;          // [ceil(nblocks / 64)] the segment that holds block 64 w (k_seg_scan)
    u32x4 *prov;                 // [nseg][cmax] the segment's blocks: byte position, three way marks (bit offset << 8 | coefficients so far, 0 = none)
    unsigned long long *trace;   // -DJPEGX_DECODE_STATS builds: [nseg][16] time stamps
};

__host__ __device__ inline size_t up16(size_t v) { return (v + 15) & ~(size_t)15; }

// state: two status blocks (alternate calls) and the exit words -- zero between calls, in an allocation of their own (the
// other arrays' places depend on the segment count: a short stream's arrays would lie in a long stream's exit words);
// scratch: everything that is written before it is read
__host__ __device__ inline Ws carve(void *state, void *scratch, unsigned nseg, int cmax, long long nblocks, int parity, size_t *scratch_bytes = nullptr,
                                    size_t *state_bytes = nullptr)
{
    unsigned char *q = static_cast<unsigned char *>(state), *p = static_cast<unsigned char *>(scratch);
    Ws w;
    w.head = reinterpret_cast<unsigned *>(q + 64 * parity);
    w.next_head = reinterpret_cast<unsigned *>(q + 64 * (1 - parity));
    w.E = reinterpret_cast<unsigned *>(q + 128);
    if (state_bytes) *state_bytes = 128 + up16((size_t)nseg * 4);
    size_t o = 0;
    w.C = reinterpret_cast<unsigned *>(p + o); o += up16((size_t)nseg * 4 + 64);      // read in 16-byte pieces
    w.F = reinterpret_cast<unsigned *>(p + o); o += up16(((size_t)nseg + 1) * 4);
    w.wave_seg = reinterpret_cast<unsigned *>(p + o); o += up16((size_t)((nblocks + 63) / 64) * 4);
    w.prov = reinterpret_cast<u32x4 *>(p + o); o += up16((size_t)nseg * cmax * 16);
    w.trace = reinterpret_cast<unsigned long long *>(p + o);
#ifdef JPEGX_DECODE_STATS
    o += 64 + (size_t)nseg * 128;
#endif
    if (scratch_bytes) *scratch_bytes = o;
    return w;
}

__host__ __device__ inline int levels_of(int cmax)          // 2^levels = cmax: chains inside a segment are shorter than that
{
    int l = 0;
    while ((1 << l) < cmax) ++l;
    return l;
}

__host__ __device__ inline size_t dec_lds_bytes(int span_cap) { return (size_t)span_cap + 64 * ROW + 1024 + 288; }

__host__ __device__ inline size_t lds_bytes(int seg, int cmax)
{
    return (size_t)(FRONT + seg + TAIL) + 64 * 12 + 256 + (size_t)cmax * (4 + levels_of(cmax));   // lane records; advance table; cpos, J0: 16 bit; the upper levels: bytes
}

constexpr int MARK_STEPS = 5;               // steps of three codes between two way marks

// by how much a code advances the coefficient counter, from its header byte: run + 1 for an amplitude code, 15 for the
// chain code 1111 0000 (FIFTEEN zeros, util.py:134-154), 128 -- beyond any block -- for everything a block cannot go
// on with: the end marker, a zero size with another run, size 1 (a sign without amplitude bits: the reference's
// decode_signed fails on it, rle_byte_stream.py:35-42).  Lane l fills entries 4l .. 4l + 3.
__device__ __forceinline__ void fill_advance_table(unsigned char *tab, int lane)
{
    unsigned packed = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned h = (unsigned)lane * 4u + j, size = h & 15u, run = h >> 4;
        const unsigned inc = size >= 2u ? run + 1u : (h == 0xF0u ? 15u : 128u);
        packed |= inc << (8 * j);
    }
    reinterpret_cast<unsigned *>(tab)[lane] = packed;
}

// Where the block at bit position q of a window of big-endian dwords ends (BYTE position), or NIL if what is there is
// not a block -- no coefficients wanted.  Every lane walks its own candidate, 63 dependent codes on noise, and a wave's
// time in this loop is the latency of a step times the steps, whatever else runs on the CU.  So a step takes THREE codes:
// the 64 bits at q (three dwords) hold the first header and -- a code is at most 23 bits long -- the two behind it; all
// go through the table above, and ONE comparison each (a 65th coefficient) ends the walk for every reason there is.  A
// lane that has stopped keeps its q, pointing at the code that stopped it; what that was is looked at once, behind the
// loop.  The window holds zeros behind the stream's end, so a walk that runs over it meets an end marker there and is
// refused by position.
__device__ __forceinline__ unsigned parse_tab(const unsigned *sw, const unsigned char *tab, const unsigned q0, unsigned end_bits, bool live,
                                              unsigned (&mark)[3])
{
    unsigned n = 0, q = q0;
    auto step = [&]() {
        const unsigned wi = q >> 5, sh = q & 31u;
        const unsigned d0 = sw[wi], d1 = sw[wi + 1], d2 = sw[wi + 2];
        const unsigned w = (unsigned)(((((unsigned long long)d0 << 32) | d1) << sh) >> 32);     // bits q .. q + 31
        const unsigned x = (unsigned)(((((unsigned long long)d1 << 32) | d2) << sh) >> 32);     // bits q + 32 .. q + 63
        const unsigned h1 = w >> 24, a1 = 8u + (h1 & 15u);
        const unsigned v = (unsigned)(((((unsigned long long)w << 32) | x) << a1) >> 32);       // bits from the second code on (a1 <= 23)
        const unsigned h2 = v >> 24, a2 = 8u + (h2 & 15u);
        const unsigned h3 = (v << a2) >> 24, a3 = 8u + (h3 & 15u);                              // a2 <= 23: the third header lies inside v
        const unsigned n1 = n + tab[h1], n2 = n1 + tab[h2], n3 = n2 + tab[h3];
        const bool go1 = live && n1 <= 64u, go2 = go1 && n2 <= 64u;
        live = go2 && n3 <= 64u;
        q += (go1 ? a1 : 0u) + (go2 ? a2 : 0u) + (live ? a3 : 0u);
        n = n3;
    };
    // Way marks for the block decoder: where the walk stands after 15, 30 and 45 codes (bit offset into the block and
    // coefficients so far), if it goes on from there -- four lanes can then share a block.  One loop per stretch between
    // marks, so that no loop carries the marks through its iterations (as ONE loop with the marks assigned inside, every
    // step paid six register moves for them).
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        for (int it = 0; it < MARK_STEPS && __ballot(live) != 0ull; ++it) step();
        mark[k] = live ? ((q - q0) << 8) | n : 0u;
    }
    while (__ballot(live) != 0ull) step();
    const unsigned wi = q >> 5;
    const unsigned w = (unsigned)(((((unsigned long long)sw[wi] << 32) | sw[wi + 1]) << (q & 31u)) >> 32);
    return ((w >> 24) == 0u && q + 8u <= end_bits) ? (q + 15u) >> 3 : NIL;   // an end marker (+ the zero padding) inside the stream
}

// The walk WITH the coefficients, into `row` (64 int16, zero beforehand), for the block decoder: from bit position q with n
// coefficients behind it, up to the way mark q_stop (or to the end marker).  k_seg_starts has walked the same codes and
// found them a block, so nothing is checked here but what keeps a damaged workspace from doing harm: the coefficient
// index is masked to the row and the steps are counted.  Two codes per step, from the 64 bits at q (three dwords);
// stores go by address select (`dummy` when the code carries no amplitude or the lane has stopped), not under a branch.
__device__ __forceinline__ unsigned parse_coefficients(const unsigned *sw, unsigned q, unsigned n, const unsigned q_stop, unsigned char *row,
                                                       unsigned char *dummy, bool live)
{
    auto amplitude = [](unsigned w, unsigned size) -> int {
        const unsigned mag = __builtin_amdgcn_ubfe(w, 24u - size, size - 1u);
        return (w & 0x00800000u) ? (int)mag : -(int)mag;                     // sign bit '1' = positive
    };
    for (int steps = 0; steps < 40 && __any(live); ++steps) {                // a block has at most 64 codes + the end marker
        const unsigned wi = q >> 5, sh = q & 31u;
        const unsigned d0 = sw[wi], d1 = sw[wi + 1], d2 = sw[wi + 2];
        const unsigned w1 = (unsigned)(((((unsigned long long)d0 << 32) | d1) << sh) >> 32);
        const unsigned x1 = (unsigned)(((((unsigned long long)d1 << 32) | d2) << sh) >> 32);
        const unsigned h1 = w1 >> 24, s1 = h1 & 15u, q1 = q + 8u + s1;
        const unsigned w2 = (unsigned)(((((unsigned long long)w1 << 32) | x1) << (8u + s1)) >> 32);
        const unsigned h2 = w2 >> 24, s2 = h2 & 15u, q2 = q1 + 8u + s2;
        // an amplitude code advances the counter by run + 1, the chain code (size 0, run 15) by fifteen, the end marker stops
        const unsigned n1 = n + (h1 >> 4) + (s1 >= 2u ? 1u : 0u), n2 = n1 + (h2 >> 4) + (s2 >= 2u ? 1u : 0u);
        const bool go1 = live && h1 != 0u;
        const bool go2 = go1 && q1 < q_stop && h2 != 0u;
        *reinterpret_cast<int16_t *>((go1 && s1 >= 2u) ? row + ((n1 - 1u) & 63u) * 2u : dummy) = (int16_t)amplitude(w1, s1);
        *reinterpret_cast<int16_t *>((go2 && s2 >= 2u) ? row + ((n2 - 1u) & 63u) * 2u : dummy) = (int16_t)amplitude(w2, s2);
        q = go2 ? q2 : (go1 ? q1 : q);
        n = go2 ? n2 : n1;
        live = go2 && q < q_stop;
    }
    return q;       // the code the walk stopped in front of: the way mark it was told, or the end marker
}

// workgroup i (it runs on XCD i % 8) -> segment: the XCDs take turns in runs of 32 segments
constexpr int RUN_LOG = 5;
__device__ __forceinline__ unsigned xcd_run_segment(unsigned i)
{
    const unsigned j = i >> 3, x = i & 7u;
    return ((((j >> RUN_LOG) << 3) + x) << RUN_LOG) + (j & ((1u << RUN_LOG) - 1u));
}

// 16-byte pieces [first, first + n) of the stream -> LDS as big-endian dwords, up to PER pieces per thread, all loads in flight at once
template <int PER, int THREADS = 64>
__device__ __forceinline__ void stage_pieces(const unsigned char *__restrict__ bytes, size_t nbytes, long long first_off, int npieces,
                                             unsigned char *sm, int lane)
{
    const size_t readable = (nbytes + 16) & ~(size_t)3;                        // the buffer carries 16 bytes behind the stream
    u32x4 v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = lane + THREADS * k;
        const long long off = first_off + 16LL * i;
        v[k] = u32x4{0u, 0u, 0u, 0u};
        if (i < npieces) {
            if (off >= 0 && (size_t)off + 16 <= readable) {
                v[k] = *reinterpret_cast<const u32x4 *>(bytes + off);
            } else {
                unsigned t[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const long long o = off + 4 * j;
                    if (o >= 0 && (size_t)o + 4 <= readable) t[j] = *reinterpret_cast<const unsigned *>(bytes + o);
                }
                v[k] = u32x4{t[0], t[1], t[2], t[3]};
            }
        }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = lane + THREADS * k;
        if (i < npieces) {
            u32x4 b = v[k];
            b.x = __builtin_bswap32(b.x); b.y = __builtin_bswap32(b.y); b.z = __builtin_bswap32(b.z); b.w = __builtin_bswap32(b.w);
            *reinterpret_cast<u32x4 *>(sm + 16 * i) = b;
        }
    }
}

__global__ __launch_bounds__(64) void k_seg_starts(const unsigned char *__restrict__ bytes, size_t nbytes, int seg, int cmax, unsigned nseg,
                                                   void *state, void *ws, long long nblocks, int parity, int filter)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const Ws W = carve(state, ws, nseg, cmax, nblocks, parity);
    gu32 *gE = (gu32 *)W.E;
    gu32 *ghead = (gu32 *)W.head;
    const int lane = threadIdx.x;
    // Segment = workgroup index.  A wave waits only for lower-numbered segments; the workgroups of a grid are handed to
    // the compute units in increasing order (each XCD its share), so the lowest-numbered unfinished one is always
    // running and the waits end.  HIP does not promise that order: should it ever not hold, the bounded spins below
    // run out, the waves poison what is behind them and the host takes the general scheme -- slower, never wrong.
    // (An atomic ticket would make the order a fact, but one word hands out only ~88 tickets per microsecond:
    // 130 us for the 11 000 segments of a 4096 x 4096 noise band, several times the rest of the kernel.)
    // The XCDs take turns in runs of 32 waves (workgroup i runs on XCD i % 8 and each XCD starts its share in order): a
    // wave and the one in front are then neighbours in ONE XCD's order, started back to back, except at a run's first
    // wave.  With the natural numbering neighbours sit on different XCDs, and whenever those are a round apart a wave
    // waits a whole round for the exit in front, holding its LDS (measured: waits of 20-40 us, +40 %).
    const unsigned s = xcd_run_segment(blockIdx.x);                            // the grid is padded to whole runs
    if (s >= nseg) return;
#ifdef JPEGX_DECODE_STATS
    unsigned long long *trace = W.trace + 8 + (size_t)s * 16; // time stamps (100 MHz): slot 0 = start, k + 1 = end of phase k
    if (lane == 0) trace[0] = wall_clock64();
    if (lane == 0 && s == 0) { unsigned *info = reinterpret_cast<unsigned *>(W.trace); info[4] = nseg; info[5] = (unsigned)seg; info[6] = (unsigned)cmax; info[7] = (unsigned)nblocks; }
#define JPEGX_PHASE(k) do { if (lane == 0) trace[(k) + 1] = wall_clock64(); } while (0)
#else
#define JPEGX_PHASE(k) do { } while (0)
#endif
    const int win = FRONT + seg + TAIL;
    const int levels = levels_of(cmax);
    unsigned *sw = reinterpret_cast<unsigned *>(sm);
    struct LaneRec { unsigned lo, hi, front; };                                // per lane: candidate mask (64 bit), candidates in front
    LaneRec *rec = reinterpret_cast<LaneRec *>(sm + win);
    unsigned char *tab = reinterpret_cast<unsigned char *>(rec + 64);
    unsigned short *cpos = reinterpret_cast<unsigned short *>(tab + 256);
    unsigned short *J = cpos + cmax;                                           // J0[c]: the candidate the block at c ends at, or a terminal code
    unsigned char *Jk = reinterpret_cast<unsigned char *>(J + cmax) - cmax;    // Jk[k * cmax + c], k >= 1: the candidate 2^k blocks on, 0xFF: the chain ends before

    // ---- 1. the window
    const size_t base = (size_t)s * seg;
    stage_pieces<5>(bytes, nbytes, (long long)base - FRONT, win / 16, sm, lane);     // (16 + 4096 + 208) / 16 / 64 < 5 pieces per lane
    fill_advance_table(tab, lane);
    __syncthreads();
    JPEGX_PHASE(0);

    // candidates: position 0 of the stream, and every position behind a 0x00 byte
    const int stretch = seg >> 6;                                              // bytes per lane: a multiple of 4, at most 64
    const unsigned inv = ((1u << 20) + (unsigned)stretch - 1u) / (unsigned)stretch;
    const int p0 = lane * stretch;
    unsigned long long m;
    {
        const unsigned *swl = reinterpret_cast<const unsigned *>(sm + FRONT + p0);
        unsigned long long z = 0, lk = 0;                                      // bit j: byte p0 + j is zero / has a zero high nibble
        for (int k = 0; k < (stretch >> 2); ++k) {
            const unsigned x = swl[k];
            const unsigned t = (~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu)) >> 7;   // bits 24, 16, 8, 0 <-> bytes 0..3
            z |= (unsigned long long)(((t * 0x08040201u) >> 24) & 15u) << (4 * k);
            const unsigned y = x & 0xF0F0F0F0u;
            const unsigned t2 = (~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu)) >> 7;
            lk |= (unsigned long long)(((t2 * 0x08040201u) >> 24) & 15u) << (4 * k);
        }
        const unsigned top = (unsigned)(z >> (stretch - 1)) & 1u;
        const unsigned up = (unsigned)__shfl_up((int)top, 1);
        const unsigned carry0 = base == 0 ? 1u : ((sw[FRONT / 4 - 1] & 0xFFu) == 0u ? 1u : 0u);
        m = (z << 1) | (lane ? up : carry0);
        const long long left = (long long)nbytes - (long long)base - p0;       // positions of this lane inside the stream
        const int nvalid = left <= 0 ? 0 : (left >= stretch ? stretch : (int)left);
        m &= nvalid >= 64 ? ~0ull : ((1ull << nvalid) - 1ull);
        if (filter) {
            // First try only: a block of this codec that holds anything starts with a header whose run nibble is zero
            // (samples are non-negative: a non-zero coefficient means a non-zero DC), an empty one is the byte 0x00 -- so a
            // true start has a first byte below 0x10, and most false candidates (a zero byte inside a block's bits) do
            // not.  Dropping the others halves the candidates of a busy stream; positions up to REACH stay all (the chain
            // from the segment in front may enter at any of them).  A chain that lands on a dropped position reads as a
            // chain that leaves the candidates: with the filter on that hands the stream to the second try (all
            // candidates, 256-byte segments), it does not refuse it.
            const int nkeep = (int)REACH + 1 - p0;
            const unsigned long long keep = nkeep <= 0 ? 0ull : (nkeep >= 64 ? ~0ull : ((1ull << nkeep) - 1ull));
            m &= lk | keep;
        }
    }
    const unsigned mine = (unsigned)__popcll(m);
    unsigned incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned u = (unsigned)__shfl_up((int)incl, d);
        if (lane >= d) incl += u;
    }
    const unsigned total = (unsigned)__shfl((int)incl, 63);
    rec[lane] = LaneRec{(unsigned)m, (unsigned)(m >> 32), incl - mine};
    const bool overflow = total > (unsigned)cmax;
    if (!overflow) {
        unsigned at = incl - mine;
        unsigned long long mm = m;
        while (mm) {
            const int i = __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            cpos[at++] = (unsigned short)(p0 + i);
        }
    }
    __syncthreads();
    // candidates in front of position p, and whether p is one itself
    auto rank_of = [&](unsigned p, bool &is_cand) -> unsigned {
        const unsigned L = (p * inv) >> 20;
        const unsigned bit = p - L * (unsigned)stretch;
        const LaneRec r = rec[L];
        const unsigned long long mk = ((unsigned long long)r.hi << 32) | r.lo;
        is_cand = ((mk >> bit) & 1ull) != 0;
        return r.front + (unsigned)__popcll(mk & ((1ull << bit) - 1ull));
    };
    JPEGX_PHASE(1);

    bool poisoned = overflow;
    if (overflow && lane == 0) __hip_atomic_fetch_or(ghead + 2, 1u, JPEGX_RLX_AGENT);
    const size_t win_end = nbytes - base + FRONT;                              // the stream's end, from the window's start
    const unsigned end_bits = (unsigned)((win_end < (size_t)(win - 16) ? win_end : (size_t)(win - 16)) * 8u);
    unsigned nent = 0;
    bool konst = false;
    int lv = 1;                                                                // levels in use: 2^(lv-1) >= the longest chain (or the tables' depth)
    while ((1u << (lv - 1)) < total && lv < levels) ++lv;
    // chain from candidate c: where it leaves the segment and after how many blocks (binary lifting, top level down)
    auto chase = [&](unsigned c, unsigned &blocks) -> unsigned {
        unsigned cur = c, cnt = 0;
        for (int k = lv - 1; k >= 1; --k) {
            const unsigned nxt = Jk[k * cmax + cur];
            if (nxt != 0xFFu) { cur = nxt; cnt += 1u << k; }
        }
        const unsigned nxt0 = J[cur];
        if (nxt0 < EXIT) { cur = nxt0; cnt += 1u; }
        blocks = cnt + 1u;
        return J[cur];
    };
    unsigned my_exit_if_entry = INVALID, my_blocks_if_entry = 0;              // lane j: entry j's exit and blocks
    unsigned marks[2][3] = {{0u, 0u, 0u}, {0u, 0u, 0u}};                       // lane l: the way marks of candidates l and 64 + l
    if (!poisoned) {
        // ---- 2. one parse per candidate
        for (unsigned c0 = 0; c0 < total; c0 += 64) {
            const unsigned c = c0 + lane;
            const bool have = c < total;
            const unsigned p = have ? cpos[c] : 0u;
            unsigned mk[3];
            const unsigned e = parse_tab(sw, tab, (FRONT + p) * 8u, end_bits, have, mk);
            // candidates beyond the 128th go without way marks: one lane decodes such a block alone
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (c0 == 0) marks[0][k] = mk[k];
                if (c0 == 64) marks[1][k] = mk[k];
            }
            unsigned code = INVALID;
            if (e != NIL) {
                const unsigned er = e - FRONT;
                if (base + er == nbytes) code = END;
                else if (er >= (unsigned)seg) code = EXIT | (er - (unsigned)seg);
                else {
                    bool is_cand;
                    const unsigned i = rank_of(er, is_cand);
                    code = is_cand ? i : INVALID;
                }
            }
            if (have) J[c] = (unsigned short)code;
        }
        __syncthreads();
        JPEGX_PHASE(2);
        // J[k][c] = the candidate 2^k blocks behind c, or the terminal code the chain meets before that
        for (int k = 1; k < lv; ++k) {
            for (unsigned c = lane; c < total; c += 64) {
                unsigned two = 0xFFu;
                if (k == 1) {
                    const unsigned a = J[c];
                    if (a < EXIT) { const unsigned b = J[a]; if (b < EXIT) two = b; }
                } else {
                    const unsigned a = Jk[(k - 1) * cmax + c];
                    if (a != 0xFFu) two = Jk[(k - 1) * cmax + a];
                }
                Jk[k * cmax + c] = (unsigned char)two;
            }
            __syncthreads();
        }
        // ---- 3. the entries; if they all agree on the exit, the segment behind can go on at once
        bool dummy_b;
        nent = total ? rank_of(REACH + 1, dummy_b) : 0u;
        if ((unsigned)lane < nent) my_exit_if_entry = chase((unsigned)lane, my_blocks_if_entry);
        if (nent >= 1 && nent <= 64) {
            const unsigned x = my_exit_if_entry;
            const unsigned long long valid = __ballot(x != INVALID);
            if (valid) {
                const unsigned x0 = (unsigned)__shfl((int)x, __ffsll((long long)valid) - 1);
                konst = __ballot(x != INVALID && x != x0) == 0;
                if (konst && lane == 0) __hip_atomic_store(gE + s, TAG | x0, JPEGX_RLX_AGENT);
            }
        }
    }
    JPEGX_PHASE(3);

    // ---- 4. the entry (from the exit in front) and the number of blocks
    unsigned entry = INVALID, myexit = INVALID, count = 0, err = 0;
    if (!poisoned) {
        if (s == 0) {
            entry = total ? 0u : INVALID;                                      // position 0 is candidate 0 of segment 0
        } else {
            unsigned x = 0;
            unsigned dbg_spins = 0;
            for (unsigned spins = 0;;) {
                x = __hip_atomic_load(gE + (s - 1), JPEGX_RLX_AGENT);
                if (x != 0) break;
                ++dbg_spins;
                if (++spins > SPIN_LIMIT) { poisoned = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            x &= 0xFFFFu;
            (void)dbg_spins;
#ifdef JPEGX_DECODE_STATS
            if (lane == 0) trace[12] = dbg_spins;
#endif
            if (poisoned || x == POISON) poisoned = true;
            else if (x == END || x == AFTER) entry = AFTER;
            else if (is_exit(x)) {
                const unsigned rel = x & 0x7FFFu;
                bool is_cand = false;
                const unsigned i = rel < (unsigned)seg ? rank_of(rel, is_cand) : 0u;
                entry = is_cand ? i : INVALID;
            }
        }
    }
    JPEGX_PHASE(4);
    if (!poisoned) {
        if (entry == AFTER) myexit = AFTER;
        else if (entry == INVALID) err |= filter ? 0x100u : 2u;
        else {
            if (entry < 64u && entry < nent) {                                 // chased already, by lane `entry`
                myexit = (unsigned)__shfl((int)my_exit_if_entry, (int)entry);
                count = (unsigned)__shfl((int)my_blocks_if_entry, (int)entry);
            } else {
                myexit = chase(entry, count);
            }
            if (myexit == INVALID) { err |= filter ? 0x100u : 2u; count = 0; }
        }
        if ((err & 0xFFu) && lane == 0) __hip_atomic_fetch_or(ghead + 1, err & 0xFFu, JPEGX_RLX_AGENT);
        if ((err & 0x100u) && lane == 0) __hip_atomic_fetch_or(ghead + 2, 4u, JPEGX_RLX_AGENT);      // filtered candidates: the next try decides
        // block r of the segment is the candidate r steps from the entry: its byte position and its way marks (held by
        // the lane that parsed it)
        for (unsigned r0 = 0; r0 < count; r0 += 64) {
            const unsigned r = r0 + lane;
            unsigned cur = entry;
            if (r < count) {
                if (r & 1u) cur = J[cur];
                for (int k = 1; k < lv; ++k)
                    if ((r >> k) & 1u) cur = Jk[k * cmax + cur];
            }
            unsigned mk[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const unsigned a = (unsigned)__shfl((int)marks[0][k], (int)(cur & 63u)), b = (unsigned)__shfl((int)marks[1][k], (int)(cur & 63u));
                mk[k] = cur < 64u ? a : (cur < 128u ? b : 0u);
            }
            if (r < count) W.prov[(size_t)s * cmax + r] = u32x4{(unsigned)base + cpos[cur], mk[0], mk[1], mk[2]};
        }
    } else if (lane == 0) {
        __hip_atomic_fetch_or(ghead + 2, 2u, JPEGX_RLX_AGENT);
    }
    if (lane == 0) {
        if (poisoned) __hip_atomic_store(gE + s, TAG | POISON, JPEGX_RLX_AGENT);
        else if (!konst || myexit == INVALID) __hip_atomic_store(gE + s, TAG | myexit, JPEGX_RLX_AGENT);
        W.C[s] = count;                 // where these blocks go in the plane's order is k_seg_scan's business: nobody waits for it here
    }
#ifdef JPEGX_DECODE_STATS
    if (lane == 0) { trace[15] = total; trace[13] = konst ? 1 : 0; trace[14] = count; }
#endif
    JPEGX_PHASE(5);
}

// One workgroup: index of every segment's first block (exclusive scan of the counts), for every wave of the block
// decoder the segment that holds its first block, and the check that the stream holds the plane's number of blocks.  Chunks of 16 K segments: 16 consecutive counts per thread, all loads in flight at once
// (a dependent load per count made this kernel 11 us for 11 000 segments).
__global__ __launch_bounds__(1024) void k_seg_scan(unsigned nseg, int cmax, void *state, void *ws, long long nblocks, int parity)
{
    __shared__ unsigned carry[16];
    __shared__ unsigned base_s;
    const Ws W = carve(state, ws, nseg, cmax, nblocks, parity);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const unsigned nwaves = (unsigned)((nblocks + 63) / 64);
    if (t == 0) base_s = 0;
    __syncthreads();
    for (unsigned c0 = 0; c0 < nseg; c0 += 16384u) {
        const unsigned i0 = c0 + (unsigned)t * 16u;
        unsigned c[16];
        if (i0 + 16u <= nseg) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(W.C + i0 + 4 * k);
                c[4 * k] = v.x; c[4 * k + 1] = v.y; c[4 * k + 2] = v.z; c[4 * k + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) c[k] = i0 + k < nseg ? W.C[i0 + k] : 0u;
        }
        unsigned sum = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum += c[k];
        unsigned incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned u = (unsigned)__shfl_up((int)incl, d);
            if (lane >= d) incl += u;
        }
        if (lane == 63) carry[wv] = incl;
        __syncthreads();
        unsigned run = base_s + incl - sum;
        for (int k = 0; k < wv; ++k) run += carry[k];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const unsigned i = i0 + k;
            if (i < nseg) {
                W.F[i] = run;
                for (unsigned m = (run + 63u) & ~63u; m < run + c[k]; m += 64)      // at most a few: a segment holds about 46 candidates
                    if ((m >> 6) < nwaves) W.wave_seg[m >> 6] = i;
            }
            run += c[k];
        }
        __syncthreads();
        if (t == 1023) base_s = run;
        __syncthreads();
    }
    if (t == 0) {
        const unsigned run = base_s;
        W.F[nseg] = run;
        W.head[3] = run;
        unsigned err = 0;
        if ((long long)run < nblocks) err |= 1u;           // fewer blocks than the plane has
        if ((long long)run > nblocks) err |= 4u;           // more: bytes behind the plane's last block
        if (err) atomicOr(&W.head[1], err);
    }
}

// FOUR lanes per block, 64 blocks per workgroup of four waves, from LDS.  Where a block starts: the workgroup's first
// block lies in segment wave_seg[w]; the first-block indices of that segment and the 64 behind it go into LDS, a lane
// finds its block's segment there (the last one that starts at or in front of the block) and reads what k_seg_starts
// left: the byte position and up to three way marks.  Wave p takes every block from its p-th mark (wave 0: from the
// start) to the next one, or to the block's end: a quarter of the dependent steps.  The blocks' bytes are one
// contiguous span of the stream, [start of the first, start of the block behind the last); `span_cap` bytes of LDS
// hold it (the host sizes that from the stream's average, with room to spare); a workgroup whose span is longer --
// at most 64 x 185 bytes -- reads its blocks from global memory like the general scheme's kernel.
__global__ __launch_bounds__(256) void k_dec_blocks_lds(const unsigned char *__restrict__ bytes, size_t nbytes, unsigned nseg, int cmax, void *state,
                                                        const void *ws, int nblk, int span_cap, int16_t *__restrict__ out, int parity)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const Ws W = carve(state, const_cast<void *>(ws), nseg, cmax, nblk, parity);
    // house-keeping for the NEXT call on this workspace, so that no call starts with a memset launch (4.5 us): the exit
    // words go back to zero (k_seg_starts is done with them) and the other status block is cleared
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < nseg; i += gridDim.x * 256u) W.E[i] = 0u;
    if (blockIdx.x == 0 && threadIdx.x < 16) W.next_head[threadIdx.x] = 0u;
    // the status words and the workgroup's segment in ONE round trip (the segment index is only meaningful when the status is clean)
    const unsigned s0 = W.wave_seg[blockIdx.x], refused = W.head[1], handed_on = W.head[2];
    if (refused != 0 || handed_on != 0) return;          // refused, or handed to the general scheme: the block starts are not there
    const int t = threadIdx.x, lane = t & 63, part = t >> 6;
    const int tile_idx = blockIdx.x, g0 = tile_idx * 64, g = g0 + lane;
#ifdef JPEGX_DECODE_STATS
    unsigned long long *trace3 = W.trace + 8 + (size_t)tile_idx * 16;
    if (t == 0 && (unsigned)tile_idx < nseg) trace3[7] = wall_clock64();
#define JPEGX_PHASE3(k) do { if (t == 0 && (unsigned)tile_idx < nseg) trace3[k] = wall_clock64(); } while (0)
#else
#define JPEGX_PHASE3(k) do { } while (0)
#endif
    unsigned char *tile = sm + span_cap;
    unsigned char *dummy = tile + 64 * ROW + t * 4;
    unsigned *Fs = reinterpret_cast<unsigned *>(tile + 64 * ROW + 1024);       // [65] first-block indices from segment s0 on, [66] [67]: the span
    for (int i = t; i < 64 * ROW / 16; i += 256) *reinterpret_cast<u32x4 *>(tile + i * 16) = u32x4{0u, 0u, 0u, 0u};
    if (t <= 64) Fs[t] = s0 + t <= nseg ? W.F[s0 + t] : 0xFFFFFFFFu;
    __syncthreads();
    auto entry_of = [&](unsigned blk) -> u32x4 {          // what k_seg_starts left for block blk (>= the workgroup's first)
        unsigned lo = 0, hi = 65;                           // the last of the 65 loaded segments that starts at or in front of blk
        while (hi - lo > 1) {
            const unsigned mid = (lo + hi) >> 1;
            if (Fs[mid] <= blk) lo = mid; else hi = mid;
        }
        unsigned sgm = s0 + lo, first = Fs[lo];
        if (lo == 64) {                                     // more than 64 segments for 64 blocks (empty ones in between): search on in global memory
            unsigned a = s0 + 64, b = nseg;                 // F[a] <= blk < F[nseg] = blocks in the stream
            while (b - a > 1) {
                const unsigned mid = a + ((b - a) >> 1);
                if (W.F[mid] <= blk) a = mid; else b = mid;
            }
            sgm = a; first = W.F[a];
        }
        return W.prov[(size_t)sgm * cmax + (blk - first)];
    };
    const bool have = g < nblk;
    const u32x4 ent = entry_of(have ? (unsigned)g : (unsigned)g0);
    const unsigned p = ent.x;
    if (t == 0) Fs[66] = p;
    if (t == 64) Fs[67] = g0 + 64 < nblk ? entry_of((unsigned)g0 + 64u).x : (unsigned)nbytes;   // the block behind the last: wave 1 has time for it
    __syncthreads();
    const unsigned p_lo = Fs[66], p_hi = Fs[67];
    unsigned p_next = (unsigned)__shfl_down((int)p, 1);    // where the lane's block must end: the next block's start
    if (lane == 63 || g + 1 >= nblk) p_next = g + 1 >= nblk ? (unsigned)nbytes : p_hi;
    const unsigned w0 = p_lo & ~15u;                       // the window's first byte
    JPEGX_PHASE3(8);
    bool ok = true;
    if (p_hi >= p_lo && p_hi - w0 + 40 <= (unsigned)span_cap && p >= p_lo && p <= p_hi) {
        stage_pieces<3, 256>(bytes, nbytes, (long long)w0, (int)((p_hi - w0 + 24 + 15) / 16), sm, t);   // + the bit reader's look-ahead; 64 x 185 / 16 / 256 < 3
        __syncthreads();
        JPEGX_PHASE3(9);
        const unsigned marks[3] = {ent.y, ent.z, ent.w};
        const unsigned q0 = (p - w0) * 8u;
        const unsigned from = part == 0 ? 0u : marks[part - 1];                 // bit offset << 8 | coefficients so far; 0: the block ends before
        const unsigned to = part < 3 ? marks[part] : 0u;
        const bool mine = have && (part == 0 || from != 0u);
        const unsigned q_stop = to != 0u ? q0 + (to >> 8) : 0xFFFFFFFFu;
        const unsigned q = parse_coefficients(reinterpret_cast<const unsigned *>(sm), q0 + (from >> 8), from & 0xFFu, q_stop, tile + lane * ROW, dummy, mine);
        if (mine) {
            if (to != 0u) ok = q == q_stop;                                    // the walk met the next way mark
            else {
                // ... or the block's end: an end marker, the zero padding, and the next block (the stream's end behind the last:
                // bytes or whole blocks behind it make the reference fail in its reshape, run_length_encoding.py:77-79)
                const unsigned *sw = reinterpret_cast<const unsigned *>(sm);
                const unsigned wi = q >> 5;
                const unsigned w = (unsigned)(((((unsigned long long)sw[wi] << 32) | sw[wi + 1]) << (q & 31u)) >> 32);
                ok = (w >> 24) == 0u && w0 + ((q + 15u) >> 3) == p_next;
            }
        }
    } else {
        __syncthreads();
        JPEGX_PHASE3(9);
        if (have && part == 0) {
            const unsigned e = parse_block<true>(reinterpret_cast<const unsigned *>(bytes), (unsigned long long)nbytes * 8u, p, tile, lane, ROW);
            ok = e != NIL && e == p_next;
        }
    }
    if (__any(!ok) && lane == 0) atomicOr(&W.head[1], 2u);
    __syncthreads();
    JPEGX_PHASE3(10);
    unsigned char *dst = reinterpret_cast<unsigned char *>(out) + (size_t)g0 * 128;
    for (int i = t; i < 512; i += 256) {
        const int row = i >> 3, ch = i & 7;
        const u32x4 q = *reinterpret_cast<const u32x4 *>(tile + row * ROW + ch * 16);
        if (g0 + row < nblk) *reinterpret_cast<u32x4 *>(dst + (size_t)row * 128 + ch * 16) = q;
    }
    JPEGX_PHASE3(11);
}

}  // namespace seg

}  // namespace

namespace jpegx_decode {

int levels_for(long long nblocks)          // base-4 digits of the largest block index
{
    int l = 1;
    while ((1ll << (2 * l)) < nblocks) ++l;
    return l;
}

// ---- segmented scheme (round 3) ----------------------------------------------------------------------------------
SegPlan seg_plan(size_t nbytes, long long nblocks, int level, int filter)
{
    SegPlan p;
    // One wave per segment and one parse per candidate: the segment should hold about 46 candidates, so that a 65th
    // (a second pass through the parse loop for a handful of lanes) stays rare.  A candidate is a block start or a
    // zero byte inside a block's bits; measured on 4096 x 4096 bands, one byte in 81 (noise, 91 bytes per block) to one
    // in 150 (smooth, 28 bytes per block) is such a zero -- far more than the 1/256 of random bytes, because a run of 0
    // makes the header's high nibble zero.  Segments are multiples of 256 bytes (4 bytes per lane), 256 bytes to 4 KiB.
    const double per_block = (double)(nbytes ? nbytes : 1) / (double)(nblocks > 0 ? nblocks : 1);
    // first try: candidates whose first byte cannot start a block are dropped (k_seg_starts) -- what is left of the false
    // ones is about one byte in 900 (measured on noise), against one in 81-150 without the filter
    const char *nfl = getenv("JPEGX_DECODE_NOFILTER");
    p.filter = level == 0 && filter != 0 && !(nfl && *nfl && *nfl != '0');
    const double per_byte = 1.0 / per_block + (p.filter ? 1.0 / 900.0 : 1.0 / 90.0);
    int k = (int)(46.0 / per_byte / 256.0 + 0.5);
    const char *force = getenv("JPEGX_DECODE_SEG");                          // A/B runs
    if (force && *force) k = atoi(force) / 256;
    // second try: a segment of the first overflowed its tables -- a stretch of the stream far denser in blocks than the
    // average (flat regions of a busy picture: three bytes per block).  256-byte segments hold at most 128 candidates
    // unless blocks are single bytes; that costs waves (four times as many for a stream of 20-byte blocks), not a host
    // round trip and the whole-stream scheme
    if (level >= 1) k = 1;
    k = k < 1 ? 1 : (k > 16 ? 16 : k);
    if (level == 0 && !(force && *force)) {
        // One round of waves if the stream allows it: k_seg_starts takes as many wave lives as it has rounds (its waves run in
        // step), so a segment size whose waves all fit the chip at once -- with a margin: 831 waves on an XCD's 832 slots
        // measured as two rounds -- beats a neighbouring size that needs a few more.  Waves per CU from the wave's LDS
        // (allocated in steps taken as 1280 bytes, the coarser of what the measurements allow), at most 32.
        static const int ncu = [] { int v = 0, dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256; return v; }();
        const int cmax_l0 = p.filter ? 96 : 128;
        auto margin = [&](int kk) {
            const size_t lds = (seg::lds_bytes(256 * kk, cmax_l0) + 1279) / 1280 * 1280;
            const size_t per_cu = 163840 / lds > 32 ? 32 : 163840 / lds;
            const double slots = (double)ncu * (double)per_cu, waves = (double)((nbytes + 256 * (size_t)kk - 1) / (256 * (size_t)kk));
            return 1.0 - waves / slots;
        };
        if (margin(k) < 0.03) {
            int best = k;
            double best_margin = 0.03;
            for (int kk : {k + 1, k - 1, k + 2, k - 2})
                if (kk >= 1 && kk <= 16 && per_byte * 256.0 * kk <= 56.0 && margin(kk) > best_margin) { best = kk; best_margin = margin(kk); }
            k = best;
        }
    }
    p.seg = 256 * k;
    // candidates a segment's tables hold: about twice the average.  96 with the filter (the wave's LDS then stays at 6 KiB
    // for 3840-byte segments: 26 waves per CU, so that the 6 204 waves of a 4096 x 4096 noise band run as ONE round)
    p.cmax = p.filter ? 96 : 128;
    const char *fc = getenv("JPEGX_DECODE_CMAX");
    if (fc && *fc) { p.cmax = atoi(fc); p.cmax = p.cmax < 32 ? 32 : (p.cmax > 128 ? 128 : p.cmax); }    // the upper table levels hold bytes: below 255
    p.levels = seg::levels_of(p.cmax);
    p.nseg = (unsigned)((nbytes + p.seg - 1) / p.seg);
    // the block decoder's LDS span: 64 average blocks and half again (never more than 64 blocks can be: 64 x 185 bytes)
    size_t span = (size_t)(per_block * 64.0 * 1.5) + 64;
    span = (span + 15) & ~(size_t)15;
    if (span < 1024) span = 1024;
    if (span > 64 * 185 + 48) span = 64 * 185 + 48;
    p.span_cap = (int)span;
    p.ok = nbytes > 0 && (size_t)p.nseg * p.cmax * 16 <= ((size_t)1 << 32) && nblocks > 0;
    seg::carve(nullptr, nullptr, p.nseg ? p.nseg : 1, p.cmax, nblocks > 0 ? nblocks : 1, 0, &p.ws_bytes, &p.state_bytes);
    return p;
}

void enqueue_segmented(const uint8_t *d_bytes, size_t nbytes, long long nblocks, const SegPlan &p, void *d_state, size_t state_cap, bool fresh, int parity,
                       void *d_ws, int16_t *d_zz, hipStream_t st)
{
    // the state (status blocks, exit words) is zero when a call starts: a fresh allocation is cleared once, whole; after
    // that every call's block decoder clears what the call used
    if (fresh) (void)hipMemsetAsync(d_state, 0, state_cap, st);
#ifdef JPEGX_DECODE_STATS
    (void)hipMemsetAsync(d_ws, 0, p.ws_bytes, st);
#endif
    const unsigned run8 = 8u << seg::RUN_LOG;                // the grid covers whole rounds of the XCDs' turns; the surplus returns at once
    hipLaunchKernelGGL(seg::k_seg_starts, dim3((p.nseg + run8 - 1) / run8 * run8), dim3(64), seg::lds_bytes(p.seg, p.cmax), st, d_bytes, nbytes, p.seg, p.cmax,
                       p.nseg, d_state, d_ws, nblocks, parity, p.filter ? 1 : 0);
    hipLaunchKernelGGL(seg::k_seg_scan, dim3(1), dim3(1024), 0, st, p.nseg, p.cmax, d_state, d_ws, nblocks, parity);
    hipLaunchKernelGGL(seg::k_dec_blocks_lds, dim3((unsigned)((nblocks + 63) / 64)), dim3(256), seg::dec_lds_bytes(p.span_cap), st, d_bytes, nbytes,
                       p.nseg, p.cmax, d_state, d_ws, (int)nblocks, p.span_cap, d_zz, parity);
}

size_t phase1_bytes(size_t nbytes) { return 64 + ((nbytes + CHUNK - 1) / CHUNK + 1) * 4; }   // the status words are read back as one 64-byte piece

size_t phase2_bytes(size_t ncand, long long nblocks)
{
    return ((size_t)ncand * 4 + 15) / 16 * 16 + (size_t)levels_for(nblocks) * ncand * 4 + (size_t)nblocks * 4 + 64;
}

// phase 1: candidates are counted; head[0] = their number (read it back, then size phase 2)
void enqueue_phase1(const uint8_t *d_bytes, size_t nbytes, void *d_ws1, hipStream_t st)
{
    unsigned *head = static_cast<unsigned *>(d_ws1);
    unsigned *chunk = head + 4;
    const int nchunks = (int)((nbytes + CHUNK - 1) / CHUNK);
    (void)hipMemsetAsync(head, 0, 16, st);
    hipLaunchKernelGGL(k_dec_count, dim3(nchunks), dim3(256), 0, st, d_bytes, nbytes, chunk);
    hipLaunchKernelGGL(k_dec_scan, dim3(1), dim3(1024), 0, st, chunk, nchunks, head);
}

// phase 2: everything else; head[1] != 0 afterwards means the stream does not hold nblocks well-formed blocks
void enqueue_phase2(const uint8_t *d_bytes, size_t nbytes, long long nblocks, void *d_ws1, unsigned ncand, void *d_ws2,
                    int16_t *d_zz, hipStream_t st)
{
    unsigned *head = static_cast<unsigned *>(d_ws1);
    unsigned *chunk = head + 4;
    unsigned *cand_pos = static_cast<unsigned *>(d_ws2);
    unsigned *J = cand_pos + ((size_t)ncand * 4 + 15) / 16 * 4;
    const int levels = levels_for(nblocks);
    unsigned *start_pos = J + (size_t)levels * ncand;
    const unsigned *words = reinterpret_cast<const unsigned *>(d_bytes);
    const int nchunks = (int)((nbytes + CHUNK - 1) / CHUNK);
    const unsigned gc = (ncand + 255) / 256;
    hipLaunchKernelGGL(k_dec_scatter, dim3(nchunks), dim3(256), 0, st, d_bytes, nbytes, chunk, cand_pos);
    hipLaunchKernelGGL(k_dec_parse, dim3(gc), dim3(256), 0, st, words, nbytes, cand_pos, ncand, J);
    for (int k = 0; k + 1 < levels; ++k)
        hipLaunchKernelGGL(k_dec_jump, dim3(gc), dim3(256), 0, st, J + (size_t)k * ncand, J + (size_t)(k + 1) * ncand, ncand);
    hipLaunchKernelGGL(k_dec_starts, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, st, J, levels, ncand, cand_pos, nbytes,
                       (unsigned)nblocks, start_pos, head);
    hipLaunchKernelGGL(k_dec_blocks, dim3((unsigned)((nblocks + 63) / 64)), dim3(64), 0, st, words, nbytes, start_pos, (int)nblocks, d_zz, head);
}

}  // namespace jpegx_decode
