// jpegx_entropy_decode.hip -- the entropy stage INVERTED on the GPU: RleBytestream.invert
// (pipeline/rle_byte_stream.py:61-88) + RunLengthEncoding.invert (pipeline/run_length_encoding.py:66-97)
// for dct_size 8, bytes -> int16 [nblocks][64].
//
// The byte stream carries no index: where a block starts is only known once the block before it has been
// parsed, which is why the reference (and libjpegx's host parser, jpegx_host.cpp) walk it sequentially.
// What the format does guarantee is that every block ends with a zero byte -- the 8-bit end marker plus the
// zero padding to the next byte boundary always leave the last byte of a block 0x00 -- so a block can only
// start at position 0 or right behind a 0x00 byte.  That turns the sequential walk into parallel work:
//   1. candidates  every position 0 or behind a 0x00 byte, compacted in stream order (count + scan + scatter);
//   2. parse       one thread per candidate parses ONE block from there and records at which candidate the
//                  next block would start (false candidates -- zero bytes inside a block's amplitude bits --
//                  parse garbage; they are simply never reached);
//   3. jump tables J_k[c] = the candidate 4^k blocks behind candidate c (radix-4 pointer jumping, log4(nblocks) passes);
//   4. starts      block i starts at the candidate reached from position 0 by following the base-4 digits of i;
//   5. decode      lane per block, values into an LDS tile, coalesced 1 KiB stores into the zigzag stream.
// Steps 1-4 read the stream twice and touch ~1.4 candidates per block; the result equals the host parser's.
// One difference on DAMAGED input: the host parser skips the padding bits unread, so a stream whose padding has
// been tampered with still parses there, while here the block behind it is not found (its start is not behind a
// 0x00 byte) and the stream is refused; the callers (decompress_band) then take the host parser.  The encoder
// always writes zero padding (rle_byte_stream.py:55-56).  tests/test_gpu_entropy.py fuzzes both decoders.
#include <hip/hip_runtime.h>
#include <stdint.h>

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
                                                unsigned char *tile, int row)
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
            *reinterpret_cast<int16_t *>(tile + tile_off(row, (int)(nn >> 3)) + (nn & 7u) * 2) = (int16_t)amp;
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

// the same behind the segmented scheme: nothing to decode when that refused the stream or handed it back
__global__ __launch_bounds__(64) void k_dec_blocks_guarded(const unsigned *__restrict__ words, size_t nbytes, const unsigned *__restrict__ start_pos,
                                                           int nblk, int16_t *__restrict__ out, unsigned *__restrict__ head)
{
    if (head[1] != 0 || head[2] != 0) return;
    dec_blocks_body(words, nbytes, start_pos, nblk, out, head);
}

// ================================================================================================
// Round 3: block starts by SEGMENTS, five launches instead of fourteen and no host round trip.
//
// The pointer-jumping scheme above works on the whole stream at once: candidate compaction (3 launches), one
// parse per candidate with a binary search in global memory for the candidate behind it, log4(blocks) jump
// passes (8 launches for a 4096 x 4096 band, each a few microseconds of work) and a host read-back of the
// candidate count in the middle.  What it does not use: a block is at most 185 bytes long, so everything about a
// stretch of the stream can be resolved LOCALLY, in LDS, except which of a handful of candidates the chain of true
// blocks enters the stretch at.
//   k_seg_parse   one workgroup per segment of `seg` bytes: the segment (+ 256 bytes) into LDS, its candidates
//                 compacted in order, one parse per candidate from LDS (the candidate behind it found by a binary
//                 search in LDS), pointer doubling IN LDS.  For every candidate in the first 192 bytes -- the only
//                 places the chain can enter at -- where the chain leaves the segment and how many blocks it passes.
//   k_seg_entries which entry each segment is really entered at.  Chains merge within a block or two, so for almost
//                 every segment all entries agree on the exit (a "constant" segment) and the entry of the next one
//                 is known without looking further back; a thread walks back to the nearest constant segment
//                 (bounded) and forward again.
//   k_seg_scan    block index of every segment's first block (scan), and the proof: the chain from position 0,
//                 segment by segment, is exactly the entries chosen (so the shortcut above can only cost time, never
//                 correctness), ends at the stream's last byte and holds the plane's number of blocks.
//   k_seg_starts  the segment's doubling tables again (from the persisted first level), block i's start position.
//   k_dec_blocks  as before.
// A stream this layout does not fit (more candidates in a segment than its tables hold: a quarter of its bytes
// zero; more than 64 K segments) raises a flag and the host falls back to the scheme above.
// ================================================================================================
namespace seg {

constexpr int THREADS = 128;              // two waves per segment: its ~60-200 candidates fill one or two, more would idle
constexpr int REACH = 192;                 // a block is at most 185 bytes: the chain enters a segment within its first REACH bytes
constexpr int EMAX = REACH + 1;
constexpr unsigned INVALID = 0xFFFFu, END = 0xFFFEu, EXIT = 0x8000u;   // EXIT | position relative to the NEXT segment
constexpr unsigned AFTER = 0xFFFDu;        // entry of a segment behind the stream's last block (a short tail segment)
__device__ __forceinline__ bool is_exit(unsigned x) { return (x & EXIT) != 0 && x < AFTER; }
constexpr int WALK_LIMIT = 64;

struct Ws {
    unsigned *head;        // [1] error bits, [2] fallback bits, [3] blocks found
    unsigned *ncand, *nent, *konst, *entry, *count, *first;    // per segment
    unsigned short *pos, *j0;                                   // [nseg][cmax]
    unsigned short *ent_exit, *ent_cnt;                         // [nseg][EMAX]
    unsigned *start_pos;                                        // [nblocks]
};

__host__ __device__ inline size_t up16(size_t v) { return (v + 15) & ~(size_t)15; }

__host__ __device__ inline Ws carve(void *base, unsigned nseg, int cmax, long long nblocks, size_t *total = nullptr)
{
    unsigned char *p = static_cast<unsigned char *>(base);
    size_t o = 16;
    Ws w;
    w.head = reinterpret_cast<unsigned *>(p);
    unsigned **per[] = {&w.ncand, &w.nent, &w.konst, &w.entry, &w.count, &w.first};
    for (unsigned **q : per) { *q = reinterpret_cast<unsigned *>(p + o); o += up16((size_t)nseg * 4); }
    w.pos = reinterpret_cast<unsigned short *>(p + o); o += up16((size_t)nseg * cmax * 2);
    w.j0 = reinterpret_cast<unsigned short *>(p + o); o += up16((size_t)nseg * cmax * 2);
    w.ent_exit = reinterpret_cast<unsigned short *>(p + o); o += up16((size_t)nseg * EMAX * 2);
    w.ent_cnt = reinterpret_cast<unsigned short *>(p + o); o += up16((size_t)nseg * EMAX * 2);
    w.start_pos = reinterpret_cast<unsigned *>(p + o); o += up16((size_t)nblocks * 4);
    if (total) *total = o;
    return w;
}

// workgroup-wide exclusive scan of one value per thread; returns the exclusive prefix, *total = sum
__device__ __forceinline__ unsigned block_scan(unsigned v, unsigned *s_part, unsigned *total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned u = __shfl_up(incl, d);
        if (lane >= d) incl += u;
    }
    if (lane == 63) s_part[wv] = incl;
    __syncthreads();
    unsigned base = 0;
    for (int k = 0; k < wv; ++k) base += s_part[k];
    unsigned sum = 0;
    for (int k = 0; k < THREADS / 64; ++k) sum += s_part[k];
    *total = sum;
    __syncthreads();
    return base + incl - v;
}

// index of position `e` in the sorted list pos[lo, hi), or -1
__device__ __forceinline__ int find_pos(const unsigned short *pos, int lo, int hi, unsigned e)
{
    const int end = hi;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (pos[mid] < e) lo = mid + 1; else hi = mid;
    }
    return (lo < end && pos[lo] == e) ? lo : -1;
}

// J[k][c] = the candidate 2^k blocks behind c, or the terminal code the chain meets before that
__device__ __forceinline__ void double_up(unsigned short *J, int cmax, int levels, int n)
{
    for (int k = 1; k < levels; ++k) {
        const unsigned short *prev = J + (size_t)(k - 1) * cmax;
        unsigned short *cur = J + (size_t)k * cmax;
        for (int c = threadIdx.x; c < n; c += THREADS) {
            const unsigned a = prev[c];
            cur[c] = (unsigned short)(a < EXIT ? prev[a] : a);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(THREADS) void k_seg_parse(const unsigned char *__restrict__ bytes, size_t nbytes, int seg, int cmax, int levels,
                                                       unsigned nseg, void *ws, long long nblocks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    __shared__ unsigned s_part[4];
    __shared__ unsigned short s_exit[EMAX];
    const Ws W = carve(ws, nseg, cmax, nblocks);
    const int wwords = (seg + 264) / 4;                    // window: bytes [base - 4, base + seg + 260)
    unsigned *sw = reinterpret_cast<unsigned *>(sm);
    unsigned short *cpos = reinterpret_cast<unsigned short *>(sm + (size_t)wwords * 4);
    unsigned short *J = cpos + cmax;
    const unsigned s = blockIdx.x;
    const int t = threadIdx.x;
    const size_t base = (size_t)s * seg;
    const size_t readable = (nbytes + 16) & ~(size_t)3;     // the buffer carries at least 16 zero bytes behind the stream
    for (int i = t; i < wwords; i += THREADS) {
        const long long off = (long long)base - 4 + 4LL * i;
        unsigned v = 0;
        if (off >= 0 && (size_t)off + 4 <= readable) v = *reinterpret_cast<const unsigned *>(bytes + off);
        sw[i] = v;
    }
    __syncthreads();
    // candidates in stream order: position 0, and every position behind a 0x00 byte
    const unsigned char *sb = sm + 4;                        // sb[p] = byte at position base + p; sb[-1] is readable
    const int per = seg / THREADS;                           // <= 64 (seg <= 8192)
    unsigned long long mask = 0;
    for (int i = 0; i < per; ++i) {
        const int p = t * per + i;
        const bool c = base + p < nbytes && (base + p == 0 || sb[p - 1] == 0);
        mask |= (unsigned long long)c << i;
    }
    unsigned total = 0;
    unsigned at = block_scan((unsigned)__popcll(mask), s_part, &total);
    if (total > (unsigned)cmax || total == 0) {
        // more zero bytes than the tables hold: the general scheme decides.  None at all: only the tail of the last
        // block can look like that (k_seg_scan checks that nothing tries to enter here)
        if (t == 0) {
            if (total) atomicOr(&W.head[2], 1u);
            W.ncand[s] = 0; W.nent[s] = 0; W.konst[s] = INVALID;
        }
        return;
    }
    while (mask) {
        const int i = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        cpos[at++] = (unsigned short)(t * per + i);
    }
    __syncthreads();
    // one parse per candidate, from LDS
    const unsigned long long win_bits = (unsigned long long)(nbytes - base + 4) * 8u;
    for (unsigned c = t; c < total; c += THREADS) {
        const unsigned p = cpos[c];
        const unsigned e = parse_block<false>(sw, win_bits, p + 4, nullptr, 0);
        unsigned code = INVALID;
        if (e != NIL) {
            const unsigned er = e - 4;
            if (base + er == nbytes) code = END;
            else if (er >= (unsigned)seg) code = EXIT | (er - (unsigned)seg);
            else {
                const int i = find_pos(cpos, (int)c + 1, (int)total, er);
                code = i < 0 ? INVALID : (unsigned)i;
            }
        }
        J[c] = (unsigned short)code;
    }
    __syncthreads();
    for (unsigned c = t; c < total; c += THREADS) {          // the first level and the positions stay for k_seg_starts
        W.pos[(size_t)s * cmax + c] = cpos[c];
        W.j0[(size_t)s * cmax + c] = J[c];
    }
    double_up(J, cmax, levels, (int)total);
    // where the chain leaves, and after how many blocks, from every candidate it can enter at
    unsigned nent = 0;
    {
        int lo = 0, hi = (int)total;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cpos[mid] <= REACH) lo = mid + 1; else hi = mid; }
        nent = (unsigned)lo;
    }
    for (unsigned j = t; j < nent; j += THREADS) {
        unsigned cur = j, cnt = 0;
        for (int k = levels - 1; k >= 0; --k) {
            const unsigned n = J[(size_t)k * cmax + cur];
            if (n < EXIT) { cur = n; cnt += 1u << k; }
        }
        const unsigned fin = J[cur];
        s_exit[j] = (unsigned short)fin;
        W.ent_exit[(size_t)s * EMAX + j] = (unsigned short)fin;
        W.ent_cnt[(size_t)s * EMAX + j] = (unsigned short)(cnt + 1);
    }
    __syncthreads();
    if (t == 0) {
        unsigned k = INVALID;                                // the common exit of all entries that have one, else INVALID
        bool same = true;
        for (unsigned j = 0; j < nent; ++j) {
            const unsigned x = s_exit[j];
            if (x == INVALID) continue;
            if (k == INVALID) k = x; else if (k != x) same = false;
        }
        W.ncand[s] = total;
        W.nent[s] = nent;
        W.konst[s] = same ? k : INVALID;
    }
}

// entry (candidate index) of segment s given the position `rel` the chain arrives at, or -1
__device__ __forceinline__ int entry_of(const Ws &W, unsigned s, int cmax, unsigned rel)
{
    return find_pos(W.pos + (size_t)s * cmax, 0, (int)W.nent[s], rel);
}

__global__ __launch_bounds__(THREADS) void k_seg_entries(unsigned nseg, int cmax, void *ws, long long nblocks)
{
    const Ws W = carve(ws, nseg, cmax, nblocks);
    const unsigned s = blockIdx.x * THREADS + threadIdx.x;
    if (s >= nseg) return;
    unsigned entry = INVALID, count = 0;
    if (s == 0) {
        entry = W.nent[0] > 0 ? 0u : INVALID;               // position 0 is candidate 0 of segment 0
    } else {
        // nearest segment below whose exit does not depend on its entry (or segment 0, whose entry is known)
        unsigned tq = s - 1;
        int steps = 0;
        while (tq > 0 && W.konst[tq] == INVALID && steps < WALK_LIMIT) { --tq; ++steps; }
        unsigned x;                                           // what the chain leaves segment tq with
        if (W.konst[tq] != INVALID) x = W.konst[tq];
        else if (tq == 0 && W.nent[0] > 0) x = W.ent_exit[0];
        else { atomicOr(&W.head[2], 4u); x = INVALID; }
        for (unsigned u = tq + 1; u <= s && x != INVALID; ++u) {
            if (x == END || x == AFTER) { x = AFTER; if (u == s) entry = AFTER; continue; }
            const int j = is_exit(x) ? entry_of(W, u, cmax, x & 0x7FFFu) : -1;
            if (j < 0) break;
            if (u == s) entry = (unsigned)j; else x = W.ent_exit[(size_t)u * EMAX + j];
        }
    }
    if (entry != INVALID && entry != AFTER) count = W.ent_cnt[(size_t)s * EMAX + entry];
    W.entry[s] = entry;
    W.count[s] = count;
}

// one workgroup: first block index of every segment (the proof that the entries chosen ARE the chain from 0 is spread
// over the workgroups of k_seg_starts: every segment checks its own link to the next)
__global__ __launch_bounds__(1024) void k_seg_scan(unsigned nseg, int cmax, void *ws, long long nblocks, size_t nbytes, int seg)
{
    __shared__ unsigned carry[16];
    __shared__ unsigned base_s, bad_s;
    const Ws W = carve(ws, nseg, cmax, nblocks);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) { base_s = 0; bad_s = 0; }
    __syncthreads();
    for (unsigned i0 = 0; i0 < nseg; i0 += 1024) {
        const unsigned i = i0 + t;
        unsigned v = 0;
        bool bad = false;
        if (i < nseg) {
            v = W.count[i];
            bad = W.entry[i] == INVALID;                       // the links between the entries are checked by k_seg_starts
        }
        if (bad) atomicOr(&bad_s, 1u);
        unsigned incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned u = __shfl_up(incl, d);
            if (lane >= d) incl += u;
        }
        if (lane == 63) carry[wv] = incl;
        __syncthreads();
        unsigned base = base_s;
        for (int k = 0; k < wv; ++k) base += carry[k];
        if (i < nseg) W.first[i] = base + incl - v;
        __syncthreads();
        if (t == 1023) base_s = base + incl;
        __syncthreads();
    }
    if (t == 0) {
        W.head[3] = base_s;
        unsigned err = 0;
        if (bad_s) err |= 2u;                                  // the chain breaks (a malformed block) or does not end at the last byte
        if ((long long)base_s < nblocks) err |= 1u;            // fewer blocks than the plane has
        if ((long long)base_s > nblocks) err |= 4u;            // more: bytes behind the plane's last block
        if (err && !(W.head[2] & 7u)) atomicOr(&W.head[1], err);
    }
}

__global__ __launch_bounds__(THREADS) void k_seg_starts(int seg, int cmax, int levels, unsigned nseg, void *ws, long long nblocks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const Ws W = carve(ws, nseg, cmax, nblocks);
    if (W.head[1] != 0 || W.head[2] != 0) return;           // refused or handed to the general scheme: nothing to write
    unsigned short *cpos = reinterpret_cast<unsigned short *>(sm);
    unsigned short *J = cpos + cmax;
    const unsigned s = blockIdx.x;
    const int t = threadIdx.x;
    const unsigned n = W.ncand[s], entry = W.entry[s], count = W.count[s], first = W.first[s];
    if (t == 0 && entry != INVALID && entry != AFTER) {
        // the link to the next segment: what the chain leaves this segment with is where the next one is entered, or the
        // stream's last byte (then nothing may follow).  All links + entry 0 of segment 0 = the chain from position 0.
        const unsigned x = W.ent_exit[(size_t)s * EMAX + entry];
        bool bad;
        if (x == END) bad = s + 1 < nseg && W.entry[s + 1] != AFTER;
        else if (s + 1 < nseg) bad = !is_exit(x) || entry_of(W, s + 1, cmax, x & 0x7FFFu) != (int)W.entry[s + 1];
        else bad = true;
        if (bad) atomicOr(&W.head[1], 2u);
    }
    if (count == 0 || entry == INVALID || entry == AFTER) return;
    for (unsigned c = t; c < n; c += THREADS) {
        cpos[c] = W.pos[(size_t)s * cmax + c];
        J[c] = W.j0[(size_t)s * cmax + c];
    }
    __syncthreads();
    double_up(J, cmax, levels, (int)n);
    for (unsigned r = t; r < count; r += THREADS) {
        unsigned cur = entry;
        for (int k = 0; k < levels; ++k)
            if ((r >> k) & 1u) cur = J[(size_t)k * cmax + cur];
        if ((long long)(first + r) < nblocks) W.start_pos[first + r] = (unsigned)((size_t)s * seg) + cpos[cur];
    }
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
SegPlan seg_plan(size_t nbytes, long long nblocks)
{
    SegPlan p;
    // about four thousand segments for a large stream, segments of 4 KiB at least: typical candidates per segment
    // (one per block + a third again) stay far below the tables' capacity of a quarter of the bytes
    int sg = 4096;
    while (sg < 8192 && nbytes / (size_t)sg > 16384) sg *= 2;
    p.seg = sg;
    // table capacity: twice the candidates an average segment holds (one per block and about a third again from zero
    // bytes inside amplitudes), as a power of two between 128 and a quarter of the segment's bytes -- the tables are
    // what limits how many segments a CU works on at a time
    const double per_seg = 1.4 * (double)sg * (double)(nblocks > 0 ? nblocks : 1) / (double)(nbytes ? nbytes : 1);
    int cm = 128;
    while (cm < sg / 4 && (double)cm < 2.0 * per_seg) cm *= 2;
    p.cmax = cm;
    p.levels = 1;
    while ((1 << (p.levels - 1)) < p.cmax) ++p.levels;
    p.nseg = (unsigned)((nbytes + sg - 1) / sg);
    p.ok = nbytes > 0 && p.nseg <= 262144u && nblocks > 0;
    seg::carve(nullptr, p.nseg ? p.nseg : 1, p.cmax, nblocks > 0 ? nblocks : 1, &p.ws_bytes);
    return p;
}

void enqueue_segmented(const uint8_t *d_bytes, size_t nbytes, long long nblocks, const SegPlan &p, void *d_ws, int16_t *d_zz, hipStream_t st)
{
    (void)hipMemsetAsync(d_ws, 0, 16, st);
    const size_t lds_parse = (size_t)((p.seg + 264) / 4) * 4 + (size_t)p.cmax * 2 * (1 + p.levels);
    const size_t lds_starts = (size_t)p.cmax * 2 * (1 + p.levels);
    hipLaunchKernelGGL(seg::k_seg_parse, dim3(p.nseg), dim3(seg::THREADS), lds_parse, st, d_bytes, nbytes, p.seg, p.cmax, p.levels, p.nseg, d_ws, nblocks);
    hipLaunchKernelGGL(seg::k_seg_entries, dim3((p.nseg + seg::THREADS - 1) / seg::THREADS), dim3(seg::THREADS), 0, st, p.nseg, p.cmax, d_ws, nblocks);
    hipLaunchKernelGGL(seg::k_seg_scan, dim3(1), dim3(1024), 0, st, p.nseg, p.cmax, d_ws, nblocks, nbytes, p.seg);
    hipLaunchKernelGGL(seg::k_seg_starts, dim3(p.nseg), dim3(seg::THREADS), lds_starts, st, p.seg, p.cmax, p.levels, p.nseg, d_ws, nblocks);
    const seg::Ws W = seg::carve(d_ws, p.nseg, p.cmax, nblocks);
    hipLaunchKernelGGL(k_dec_blocks_guarded, dim3((unsigned)((nblocks + 63) / 64)), dim3(64), 0, st, reinterpret_cast<const unsigned *>(d_bytes), nbytes,
                       W.start_pos, (int)nblocks, d_zz, W.head);
}

size_t phase1_bytes(size_t nbytes) { return 16 + ((nbytes + CHUNK - 1) / CHUNK + 1) * 4; }

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
