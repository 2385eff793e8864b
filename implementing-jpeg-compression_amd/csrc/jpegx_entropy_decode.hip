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

__global__ __launch_bounds__(64) void k_dec_blocks(const unsigned *__restrict__ words, size_t nbytes, const unsigned *__restrict__ start_pos,
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

}  // namespace

namespace jpegx_decode {

int levels_for(long long nblocks)          // base-4 digits of the largest block index
{
    int l = 1;
    while ((1ll << (2 * l)) < nblocks) ++l;
    return l;
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
