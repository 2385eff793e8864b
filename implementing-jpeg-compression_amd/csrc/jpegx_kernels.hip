// jpegx_kernels.hip -- gfx950 (MI355X, CDNA4) kernels + C ABI of libjpegx.so.
//
// Work decomposition of the fused kernels ("lane-per-block, strip-per-wave"):
//   * a wavefront (64 lanes) owns 64 consecutive 8x8 blocks of the block-row-major block
//     order (pipeline/base.py:58-66: y outer, x inner); lane b holds ALL 64 samples of block
//     g0+b in registers, so both 1-D DCT passes, the quantiser and the zigzag permutation
//     are register-only -- no cross-lane traffic, the zigzag is a compile-time renaming and
//     the quantiser constants sit in SGPRs;
//   * the wave's 64 output blocks are one contiguous 8 KiB span of the zigzag stream: each
//     lane parks its 128 B in an XOR-swizzled LDS tile and the wave writes the tile back
//     with 1 KiB-per-instruction fully coalesced stores;
//   * exact tier: a lane whose block has a coefficient within the fp32 error bound of a
//     rounding boundary raises a flag; the wave then recomputes each flagged block
//     cooperatively in float64 in the reference's operation order (lane = one coefficient,
//     two LDS exchanges) and patches the tile.  See jpegx_math.h / DESIGN.md.
//
// Built with: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (explicit fma only).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/jpegx.h"
#include "jpegx_math.h"

// ------------------------------------------------------------------------------------------------
// constant tables (reference data, include/jpegx_tables.inc)
// ------------------------------------------------------------------------------------------------
namespace {

struct D64 { double v[64]; };
struct I64 { int v[64]; };

constexpr I64 make_zz() { return I64{{JPEGX_TABLE_ZIGZAG8}}; }
constexpr I64 make_zzinv()
{
    I64 z = make_zz(), r{};
    for (int p = 0; p < 64; ++p) r.v[z.v[p]] = p;
    return r;
}
constexpr I64 make_qt() { return I64{{JPEGX_TABLE_QTABLE}}; }
constexpr D64 make_rq64()
{
    I64 q = make_qt();
    D64 r{};
    for (int n = 0; n < 64; ++n) r.v[n] = 1.0 / (double)q.v[n];  // quantizers.py:49 "1.0 / q"
    return r;
}

constexpr I64 kZZ = make_zz();        // zigzag position p -> natural index n = i*8+j
constexpr I64 kQT = make_qt();

__device__ const double c_dct[64] = {JPEGX_TABLE_DCT_MATRIX};     // C[k][n]
__device__ const double c_cn[64] = {JPEGX_TABLE_DCT_NORMALIZED};  // Cn[k][n]
__device__ const double c_dinv[8] = {JPEGX_TABLE_NORM_DIAG};
__device__ const D64 c_rq64 = make_rq64();
__device__ const I64 c_qt = make_qt();
__device__ const I64 c_zz = make_zz();
__device__ const I64 c_zzinv = make_zzinv();

// by-value kernel parameters of the fused kernels (land in SGPRs through the kernarg segment)
struct QuantParams {
    float rq32[64];  // forward: fp32 reciprocal per natural index (0 = discarded coefficient)
                     // inverse: fp32 multiplier per natural index
    double param;    // keep / divisor
    int mode;
    int tune;        // bit0: skip the exact tier (timing experiments only -- results are then NOT bit-exact)
};

// LDS tile of one wave: 64 rows (blocks) x 128 B, 16-B chunks XOR-swizzled by the row so that
// both the per-lane row writes and the linear read-out are bank-conflict free.
__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// 16-byte global accesses with a selectable cache policy.  NT = nontemporal ("nt" bit): the
// planes and the coefficient stream are touched exactly once, so they should not displace
// each other in L2/MALL; on MI355X a 2:1 read:write stream runs ~10 % faster with nt
// (microbench/membench.hip: 5.7 -> 6.3 TB/s).
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <bool NT> __device__ __forceinline__ f32x4 ld_f32x4(const float *p)
{
    return NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p)) : *reinterpret_cast<const f32x4 *>(p);
}
template <bool NT> __device__ __forceinline__ u32x4 ld_u32x4(const void *p)
{
    return NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p)) : *reinterpret_cast<const u32x4 *>(p);
}
template <bool NT> __device__ __forceinline__ void st_u32x4(void *p, u32x4 v)
{
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p)); else *reinterpret_cast<u32x4 *>(p) = v;
}
template <bool NT> __device__ __forceinline__ void st_f32x4(float *p, f32x4 v)
{
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p)); else *reinterpret_cast<f32x4 *>(p) = v;
}
template <bool NT> __device__ __forceinline__ void st_u32x2(void *p, u32x2 v)
{
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x2 *>(p)); else *reinterpret_cast<u32x2 *>(p) = v;
}

constexpr int TILE_BYTES = 64 * 128;
constexpr int SCRATCH_DOUBLES = 128;  // sA[64] + sM[64]
constexpr int LDS_BYTES = TILE_BYTES + SCRATCH_DOUBLES * 8;

// ------------------------------------------------------------------------------------------------
// float64 exact tier, cooperative: the wave computes ONE block, lane = one matrix element.
// ------------------------------------------------------------------------------------------------

// forward: lane = i*8+j passes A[i][j]; returns Y[k][l] for lane = k*8+l.
// transforms.py:46-58 (rows then columns) in the reference's dgemv order (jpegx_dot8_ref).
__device__ __forceinline__ double coop_fwd_exact(double a_own, double *sA, double *sM, int lane)
{
    const int hi = lane >> 3, lo = lane & 7;
    sA[lane] = a_own;
    __syncthreads();
    const double m = jpegx_dot8_ref(&c_dct[lo * 8], &sA[hi * 8], 1);  // M[i=hi][l=lo]
    sM[lo * 8 + hi] = m;                                              // column l contiguous over i
    __syncthreads();
    const double y = jpegx_dot8_ref(&c_dct[hi * 8], &sM[lo * 8], 1);  // Y[k=hi][l=lo]
    __syncthreads();
    return y;
}

// inverse: lane = k*8+j passes Z[k][j]; returns x[i][j] for lane = i*8+j (float, not rounded).
// transforms.py:60-69 (columns then rows), transform_1d_inverse order (jpegx_idot8_ref).
__device__ __forceinline__ double coop_inv_exact(double z_own, double *sA, double *sM, int lane)
{
    const int hi = lane >> 3, lo = lane & 7;
    double w[8];
    sA[lo * 8 + hi] = c_dinv[hi] * z_own;  // u[k] of column j, stored [j][k]
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = c_cn[k * 8 + hi];
    const double m = jpegx_idot8_ref(w, &sA[lo * 8], 1);  // m[i=hi][j=lo]
    sM[hi * 8 + lo] = c_dinv[lo] * m;                     // u[k=lo] of row i
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = c_cn[k * 8 + lo];
    const double y = jpegx_idot8_ref(w, &sM[hi * 8], 1);  // x[i=hi][j=lo]
    __syncthreads();
    return y;
}

// ------------------------------------------------------------------------------------------------
// pieces shared by the forward kernels
// ------------------------------------------------------------------------------------------------

// Quantise the 64 coefficients of v (natural order) in zigzag order and pack them as int16 pairs
// (pipeline/quantization.py:8-18 + pipeline/zigzag_order.py:85-99; the zigzag is a compile-time
// renaming).  Returns the worst rounding margin max(|t - rint(t)| + E / q): the block is safe iff
// it stays below 1/2.  PIXEL: values provably fit int16, no saturation needed.
template <bool PIXEL, bool DC_EXACT>
__device__ __forceinline__ float quantise_zigzag_pack(const float (&v)[64], const QuantParams &prm, float E,
                                                      unsigned (&pk)[32])
{
    float worst = 0.f;
#pragma unroll
    for (int p = 0; p < 64; p += 2) {
        int q[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = kZZ.v[p + h];
            const float rq = prm.rq32[n];
            const float t = v[n] * rq;
            const float r = rintf(t);
            if (!(DC_EXACT && n == 0)) worst = fmaxf(worst, fmaf(E, fabsf(rq), fabsf(t - r)));
            q[h] = (int)r;
        }
        if (PIXEL) {
            pk[p >> 1] = ((unsigned)q[0] & 0xFFFFu) | ((unsigned)q[1] << 16);
        } else {
            const int a = min(max(q[0], -32768), 32767), b = min(max(q[1], -32768), 32767);
            pk[p >> 1] = ((unsigned)a & 0xFFFFu) | ((unsigned)b << 16);
        }
    }
    return worst;
}

// The wave's 64 x 128 B output tile -> 8 coalesced 1 KiB stores into the zigzag stream.
template <bool NT>
__device__ __forceinline__ void store_tile(const unsigned char *tile, int16_t *out, int g0, int nblk, int lane)
{
    unsigned char *dst = reinterpret_cast<unsigned char *>(out) + (size_t)g0 * 128;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = i * 8 + (lane >> 3), c = lane & 7;
        const u32x4 q = *reinterpret_cast<const u32x4 *>(tile + tile_off(row, c));
        if (g0 + row < nblk) st_u32x4<NT>(dst + (size_t)row * 128 + c * 16, q);
    }
}

// exact-tier census (jpegx_set_debug_counters): [0] += flagged blocks, [1] += blocks of this wave
__device__ __forceinline__ void census(unsigned long long *counters, unsigned long long flagged, int remaining, int lane)
{
    if (counters != nullptr && lane == 0) {
        atomicAdd(&counters[0], (unsigned long long)__popcll(flagged));
        atomicAdd(&counters[1], (unsigned long long)min(64, remaining));
    }
}

// ------------------------------------------------------------------------------------------------
// fused forward: DCT + quantise + zigzag.  VAR bit0 = PIXEL_INPUT, bit1 = DC exact.
// BS = mean-pool factor of the fused SubSampling prologue (1 = none).
// ------------------------------------------------------------------------------------------------
template <int VAR, int BS, bool NT, int STAGED>
__global__ __launch_bounds__(64) void k_forward_fused(const float *__restrict__ in, size_t pitch, int wb,
                                                      int nblk, QuantParams prm, int16_t *__restrict__ out,
                                                      unsigned long long *counters)
{
    constexpr bool PIXEL = (VAR & 1) != 0;
    constexpr bool DC_EXACT = (VAR & 2) != 0;
    constexpr int STAGE_BYTES = STAGED * 2 * BS * 1024;                 // staging buffer (0 if not staged)
    constexpr int FRONT = STAGE_BYTES > TILE_BYTES ? STAGE_BYTES : TILE_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[FRONT + SCRATCH_DOUBLES * 8];
    double *sA = reinterpret_cast<double *>(lds + FRONT);
    double *sM = sA + 64;

    const int lane = threadIdx.x;
    const int g0 = blockIdx.x * 64;
    const int g = g0 + lane;
    const bool valid = g < nblk;
    const int gc = valid ? g : nblk - 1;
    const int by = gc / wb, bx = gc - by * wb;
    const float *src = in + ((size_t)by * 8 * BS) * pitch + (size_t)bx * 8 * BS;

    float v[64];
    if (BS == 1) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float *row = src + (size_t)r * pitch;
            const f32x4 lo = ld_f32x4<NT>(row), hi = ld_f32x4<NT>(row + 4);
            v[r * 8 + 0] = lo.x; v[r * 8 + 1] = lo.y; v[r * 8 + 2] = lo.z; v[r * 8 + 3] = lo.w;
            v[r * 8 + 4] = hi.x; v[r * 8 + 5] = hi.y; v[r * 8 + 6] = hi.z; v[r * 8 + 7] = hi.w;
        }
    } else if (STAGED) {
        // SubSampling.execute fused (pipeline/subsampling.py:9-11), input staged through LDS: the
        // wave's 64 blocks cover 8*BS input rows of 64 * 32*BS bytes; they are streamed in phases of
        // 8 KiB (BS=2: two input rows = one output row; BS=4: one input row) by LDS-DMA -- whole
        // 128-B lines, nontemporal -- and every lane folds its 2*BS chunks per row into the 8 pooled
        // samples of the output row.  Chunk q of block b sits at slot 2BS*b + (q ^ g(b)),
        // g(b) = (b >> (BS == 2 ? 2 : 1)) & (2BS - 1): conflict-free ds_read_b128 at a 32*BS-byte
        // lane stride; the permutation is applied on the DMA source address.
        constexpr int CPB = 2 * BS;                 // 16-B chunks per block and input row
        constexpr int RPP = STAGED;                 // input rows per phase (RPP * CPB KiB of LDS)
        constexpr int GSH = (BS == 2) ? 2 : 1;
        // per-lane 32-bit byte offsets from the wave's first block (keeps the DMA addresses in the
        // "scalar base + vector offset" form: the row advance is scalar arithmetic)
        const int by0 = g0 / wb, bx0 = g0 - by0 * wb;
        const unsigned char *base0 = reinterpret_cast<const unsigned char *>(in + ((size_t)by0 * 8 * BS) * pitch + (size_t)bx0 * 8 * BS);
        unsigned off[CPB];
#pragma unroll
        for (int j = 0; j < CPB; ++j) {
            const int slot = 64 * j + lane;
            const int b = slot / CPB, sl = slot % CPB;
            const int q = sl ^ ((b >> GSH) & (CPB - 1));
            const int gb = min(g0 + b, nblk - 1);
            const int byb = gb / wb, bxb = gb - byb * wb;
            off[j] = (unsigned)(((size_t)(byb - by0) * 8 * BS * pitch + ((size_t)bxb - bx0) * 8 * BS + q * 4) * 4);
        }
        const int gq = (lane >> GSH) & (CPB - 1);
        float acc[8];
#pragma unroll
        for (int ph = 0; ph < 8 * BS / RPP; ++ph) {
            __syncthreads();                        // previous phase's LDS reads are done
#pragma unroll
            for (int rr = 0; rr < RPP; ++rr) {
                const unsigned char *rowbase = base0 + (size_t)(ph * RPP + rr) * pitch * 4;
#pragma unroll
                for (int j = 0; j < CPB; ++j)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(rowbase + off[j]),
                        (__attribute__((address_space(3))) void *)(lds + (rr * CPB + j) * 1024), 16, 0, NT ? 2 : 0);
            }
            __syncthreads();                        // drains vmcnt: the phase has landed
#pragma unroll
            for (int rr = 0; rr < RPP; ++rr) {
                const int ir = ph * RPP + rr, r = ir / BS, a = ir % BS;
                if (a == 0) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
                }
#pragma unroll
                for (int q = 0; q < CPB; ++q) {
                    const f32x4 t = *reinterpret_cast<const f32x4 *>(lds + rr * CPB * 1024 + ((CPB * lane + (q ^ gq)) << 4));
                    const float e[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) acc[(q * 4 + s2) / BS] += e[s2];
                }
                if (a == BS - 1) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        v[r * 8 + c] = acc[c] * (1.0f / (BS * BS));
                        // pin the pooled value here: without it hipcc sinks all the adds below the last
                        // phase and keeps every raw chunk live (178 VGPRs, 2 waves/SIMD)
                        asm volatile("" : "+v"(v[r * 8 + c]) : : "memory");
                    }
                }
            }
        }
        __syncthreads();                            // the staging buffer becomes the output tile
    } else {
        // SubSampling.execute fused (pipeline/subsampling.py:9-11): BS x BS mean, exact in fp32
        // for 8-bit samples (sum < 2^24, 1/BS^2 a power of two).
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float acc[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = 0.f;
#pragma unroll
            for (int a = 0; a < BS; ++a) {
                const float *row = src + (size_t)(r * BS + a) * pitch;
#pragma unroll
                for (int q = 0; q < 2 * BS; ++q) {
                    const f32x4 t = ld_f32x4<NT>(row + 4 * q);
                    const float e[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[(q * 4 + s) / BS] += e[s];
                }
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                v[r * 8 + c] = acc[c] * (1.0f / (BS * BS));
                asm volatile("" : "+v"(v[r * 8 + c]) : : "memory");   // fold now, do not keep raw samples live
            }
        }
    }

    float S = 0.f;
    if (!PIXEL) {
#pragma unroll
        for (int n = 0; n < 64; ++n) S += fabsf(v[n]);
    }
    jpegx_dct8x8_f32(v);
    if (PIXEL) S = v[0];  // non-negative samples: sum|x| == DC, exact
    // generic pooled input: the fp32 tile sums are themselves rounded (<= BS^2 u each)
    const float E = jpegx_fwd_err_bound(S) * ((PIXEL || BS == 1) ? 1.0f : 1.0f + (BS * BS) / 16.0f);

    // quantise in zigzag order, pack pairs, track the worst rounding margin
    unsigned pk[32];
    const float worst = quantise_zigzag_pack<PIXEL, DC_EXACT>(v, prm, E, pk);

    // park the lane's 128 B in the swizzled tile
#pragma unroll
    for (int c = 0; c < 8; ++c)
        *reinterpret_cast<uint4 *>(lds + tile_off(lane, c)) =
            make_uint4(pk[c * 4 + 0], pk[c * 4 + 1], pk[c * 4 + 2], pk[c * 4 + 3]);

    // exact tier for blocks that sit within the error bound of a rounding boundary
    unsigned long long flagged = __ballot(valid && !(worst < 0.5f));
    census(counters, flagged, nblk - g0, lane);
    if (prm.tune & 1) flagged = 0;
    __syncthreads();
    while (flagged) {
        const int b = __ffsll((long long)flagged) - 1;
        flagged &= flagged - 1;
        const int gb = g0 + b;
        const int byb = gb / wb, bxb = gb - byb * wb;
        const int i = lane >> 3, j = lane & 7;
        const float *p = in + ((size_t)(byb * 8 + i) * BS) * pitch + (size_t)(bxb * 8 + j) * BS;
        double a;
        if (BS == 1) {
            a = (double)p[0];
        } else {
            double s = 0.0;  // np.mean: float64 sum then one division (subsampling.py:11)
#pragma unroll
            for (int u = 0; u < BS; ++u)
#pragma unroll
                for (int w = 0; w < BS; ++w) s += (double)p[(size_t)u * pitch + w];
            a = s / (double)(BS * BS);
        }
        const double y = coop_fwd_exact(a, sA, sM, lane);
        const double r = jpegx_quant_ref(y, lane, prm.mode, prm.param, c_rq64.v);
        const int pz = c_zzinv.v[lane];
        *reinterpret_cast<int16_t *>(lds + tile_off(b, pz >> 3) + (pz & 7) * 2) = (int16_t)jpegx_clamp_i16(r);
    }
    __syncthreads();

    // coalesced write-back: 8 x 1 KiB per wave
    store_tile<NT>(lds, out, g0, nblk, lane);
}

// ------------------------------------------------------------------------------------------------
// fused forward, LDS-staged input (the default kernel).  The wave's 64 blocks -- when W/8 is a
// multiple of 64, one 8-row x 2 KiB strip of the plane -- are brought in by LDS-DMA
// (global_load_lds_dwordx4: 16 pieces of 1 KiB, whole-128-B-line requests, no VGPR staging)
// and each lane then picks its own block out of LDS.  Compared with per-lane global loads
// (k_forward_fused: 16 B at a 32 B lane stride, every line touched by two instructions) this
// reads each line exactly once, which is what lets the nontemporal policy pay off:
// 5.6 -> 6.5 TB/s on MI355X (profiles/).  16-B chunks of a row are
// stored at position c ^ f(c >> 1), f(b) = bit2(b) ^ bit3(b), which makes the per-lane
// ds_read_b128 (lane stride 32 B) bank-conflict free; the permutation is applied on the DMA
// SOURCE address because the DMA's LDS destination is always base + lane * 16.
// After the compute the dead strip is reused as the output tile.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int strip_swz(int chunk) { return chunk ^ (((chunk >> 3) ^ (chunk >> 4)) & 1); }

constexpr int STRIP_BYTES = 8 * 2048;
constexpr int STRIP_LDS_BYTES = STRIP_BYTES + SCRATCH_DOUBLES * 8 + 128;

template <int VAR, bool NT>
__global__ __launch_bounds__(64) void k_forward_fused_strip(const float *__restrict__ in, size_t pitch, int wb,
                                                            int nblk, QuantParams prm, int16_t *__restrict__ out,
                                                            unsigned long long *counters)
{
    constexpr bool PIXEL = (VAR & 1) != 0;
    constexpr bool DC_EXACT = (VAR & 2) != 0;
    __shared__ __attribute__((aligned(16))) unsigned char lds[STRIP_LDS_BYTES];
    double *sA = reinterpret_cast<double *>(lds + STRIP_BYTES);
    double *sM = sA + 64;
    int16_t *sP = reinterpret_cast<int16_t *>(lds + STRIP_BYTES + SCRATCH_DOUBLES * 8);

    const int lane = threadIdx.x;
    const int g0 = blockIdx.x * 64;
    const bool valid = g0 + lane < nblk;

    // 16 DMA pieces of 1 KiB: piece (r, j) fills LDS bytes [r*2048 + j*1024, +1024); lane l of
    // piece j fills chunk slot p = 64 j + l of the row, which holds chunk c = strip_swz(p) =
    // half (c & 1) of block c >> 1.  Blocks past the end of the plane re-read the last block.
    {
        const float *src[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = strip_swz(64 * j + lane);
            const int gb = min(g0 + (c >> 1), nblk - 1);
            const int by = gb / wb, bx = gb - by * wb;
            src[j] = in + (size_t)by * 8 * pitch + (size_t)bx * 8 + (c & 1) * 4;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[0] + (size_t)r * pitch),
                                             (__attribute__((address_space(3))) void *)(lds + r * 2048), 16, 0, NT ? 2 : 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[1] + (size_t)r * pitch),
                                             (__attribute__((address_space(3))) void *)(lds + r * 2048 + 1024), 16, 0, NT ? 2 : 0);
        }
    }
    __syncthreads();  // drains vmcnt: the strip has landed

    float v[64];
    {
        const int f = ((lane >> 2) ^ (lane >> 3)) & 1;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const f32x4 lo = *reinterpret_cast<const f32x4 *>(lds + r * 2048 + ((2 * lane + f) << 4));
            const f32x4 hi = *reinterpret_cast<const f32x4 *>(lds + r * 2048 + ((2 * lane + (f ^ 1)) << 4));
            v[r * 8 + 0] = lo.x; v[r * 8 + 1] = lo.y; v[r * 8 + 2] = lo.z; v[r * 8 + 3] = lo.w;
            v[r * 8 + 4] = hi.x; v[r * 8 + 5] = hi.y; v[r * 8 + 6] = hi.z; v[r * 8 + 7] = hi.w;
        }
    }

    float S = 0.f;
    if (!PIXEL) {
#pragma unroll
        for (int n = 0; n < 64; ++n) S += fabsf(v[n]);
    }
    jpegx_dct8x8_f32(v);
    if (PIXEL) S = v[0];
    const float E = jpegx_fwd_err_bound(S);

    unsigned pk[32];
    const float worst = quantise_zigzag_pack<PIXEL, DC_EXACT>(v, prm, E, pk);

    unsigned long long flagged = __ballot(valid && !(worst < 0.5f));
    census(counters, flagged, nblk - g0, lane);
    if (prm.tune & 1) flagged = 0;
    while (flagged) {   // exact tier, inputs re-read from the strip still resident in LDS
        const int b = __ffsll((long long)flagged) - 1;
        flagged &= flagged - 1;
        const int i = lane >> 3, j = lane & 7;
        const int fb = ((b >> 2) ^ (b >> 3)) & 1;
        const float x = *reinterpret_cast<const float *>(lds + i * 2048 + ((2 * b + ((j >> 2) ^ fb)) << 4) + (j & 3) * 4);
        const double y = coop_fwd_exact((double)x, sA, sM, lane);
        const double r = jpegx_quant_ref(y, lane, prm.mode, prm.param, c_rq64.v);
        sP[c_zzinv.v[lane]] = (int16_t)jpegx_clamp_i16(r);
        __syncthreads();
        if (lane == b) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const u32x4 t = *reinterpret_cast<const u32x4 *>(reinterpret_cast<unsigned char *>(sP) + c * 16);
                pk[c * 4 + 0] = t.x; pk[c * 4 + 1] = t.y; pk[c * 4 + 2] = t.z; pk[c * 4 + 3] = t.w;
            }
        }
        __syncthreads();
    }

    // the strip is dead: reuse its first 8 KiB as the swizzled output tile
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 8; ++c)
        *reinterpret_cast<u32x4 *>(lds + tile_off(lane, c)) = u32x4{pk[c * 4 + 0], pk[c * 4 + 1], pk[c * 4 + 2], pk[c * 4 + 3]};
    __syncthreads();
    store_tile<NT>(lds, out, g0, nblk, lane);
}

// ------------------------------------------------------------------------------------------------
// fused forward on uint8 planes (the form in which image bands actually arrive: util.band_to_array,
// util.py:110-112).  64 B (BS=1) or 256 B (BS=2, SubSampling 2x2 mean fused) are read per block
// instead of 256 B / 1 KiB of fp32, which matters twice: the kernel's HBM traffic drops to 192 B per
// block, and a host caller ships 4x fewer bytes over PCIe.  Same structure as k_forward_fused_strip:
// LDS-DMA of the wave's rows, lane-per-block compute, exact tier fed from LDS, tile write-out.
// BS=1 needs W % 16 == 0 (a 16-byte DMA chunk holds the rows of two adjacent blocks).
// ------------------------------------------------------------------------------------------------
template <bool DC_EXACT, int BS, bool NT>
__global__ __launch_bounds__(64) void k_forward_fused_u8(const unsigned char *__restrict__ in, size_t pitch, int wb,
                                                         int nblk, QuantParams prm, int16_t *__restrict__ out,
                                                         unsigned long long *counters)
{
    constexpr int ROWS = 8 * BS;                       // input rows of the wave's blocks
    constexpr int ROW_BYTES = 64 * 8 * BS;             // bytes of one input row in LDS (64 blocks)
    constexpr int IN_BYTES = ROWS * ROW_BYTES;         // 4 KiB (BS=1) / 16 KiB (BS=2)
    constexpr int FRONT = IN_BYTES > TILE_BYTES ? IN_BYTES : TILE_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[FRONT + SCRATCH_DOUBLES * 8 + 128];
    double *sA = reinterpret_cast<double *>(lds + FRONT);
    double *sM = sA + 64;
    int16_t *sP = reinterpret_cast<int16_t *>(lds + FRONT + SCRATCH_DOUBLES * 8);

    const int lane = threadIdx.x;
    const int g0 = blockIdx.x * 64;
    const bool valid = g0 + lane < nblk;

    if (BS == 1) {
        // piece k = rows 2k, 2k+1; lane l -> row 2k + l/32, 16-byte chunk l%32 = blocks 2c, 2c+1
        const int c = lane & 31;
        const int gb = min(g0 + 2 * c, nblk - 2);           // even block index inside the plane (W/8 is even)
        const int by = gb / wb, bx = gb - by * wb;
        const unsigned char *src = in + ((size_t)by * 8 + (lane >> 5)) * pitch + (size_t)bx * 8;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)(2 * k) * pitch),
                                             (__attribute__((address_space(3))) void *)(lds + k * 1024), 16, 0, NT ? 2 : 0);
    } else {
        // one piece per input row: lane l <-> block l (16 bytes of the row)
        const int gb = min(g0 + lane, nblk - 1);
        const int by = gb / wb, bx = gb - by * wb;
        const unsigned char *src = in + ((size_t)by * 16) * pitch + (size_t)bx * 16;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)r * pitch),
                                             (__attribute__((address_space(3))) void *)(lds + r * 1024), 16, 0, NT ? 2 : 0);
    }
    __syncthreads();

    float v[64];
    if (BS == 1) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint2 t = *reinterpret_cast<const uint2 *>(lds + r * 512 + lane * 8);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v[r * 8 + c] = (float)((t.x >> (8 * c)) & 0xFFu);
                v[r * 8 + 4 + c] = (float)((t.y >> (8 * c)) & 0xFFu);
            }
        }
    } else {
        // SubSampling.execute fused (pipeline/subsampling.py:9-11): integer 2x2 sums, then * 1/4 (exact)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const u32x4 a = *reinterpret_cast<const u32x4 *>(lds + (2 * r) * 1024 + lane * 16);
            const u32x4 b = *reinterpret_cast<const u32x4 *>(lds + (2 * r + 1) * 1024 + lane * 16);
            const unsigned wa[4] = {a.x, a.y, a.z, a.w}, wb2[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const unsigned x = wa[c >> 1] >> (16 * (c & 1)), y = wb2[c >> 1] >> (16 * (c & 1));
                const unsigned sum = (x & 0xFFu) + ((x >> 8) & 0xFFu) + (y & 0xFFu) + ((y >> 8) & 0xFFu);
                v[r * 8 + c] = (float)sum * 0.25f;
            }
        }
    }

    jpegx_dct8x8_f32(v);
    const float E = jpegx_fwd_err_bound(v[0]);            // pixel input: sum|x| == DC, exact
    unsigned pk[32];
    const float worst = quantise_zigzag_pack<true, DC_EXACT>(v, prm, E, pk);

    unsigned long long flagged = __ballot(valid && !(worst < 0.5f));
    census(counters, flagged, nblk - g0, lane);
    if (prm.tune & 1) flagged = 0;
    while (flagged) {   // exact tier, samples re-read from the rows still resident in LDS
        const int b = __ffsll((long long)flagged) - 1;
        flagged &= flagged - 1;
        const int i = lane >> 3, j = lane & 7;
        double a;
        if (BS == 1) {
            a = (double)lds[i * 512 + b * 8 + j];
        } else {
            const unsigned char *p0 = lds + (2 * i) * 1024 + b * 16 + 2 * j;
            a = ((double)p0[0] + (double)p0[1] + (double)p0[1024] + (double)p0[1025]) / 4.0;   // np.mean
        }
        const double y = coop_fwd_exact(a, sA, sM, lane);
        const double r = jpegx_quant_ref(y, lane, prm.mode, prm.param, c_rq64.v);
        sP[c_zzinv.v[lane]] = (int16_t)jpegx_clamp_i16(r);
        __syncthreads();
        if (lane == b) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const u32x4 t = *reinterpret_cast<const u32x4 *>(reinterpret_cast<unsigned char *>(sP) + c * 16);
                pk[c * 4 + 0] = t.x; pk[c * 4 + 1] = t.y; pk[c * 4 + 2] = t.z; pk[c * 4 + 3] = t.w;
            }
        }
        __syncthreads();
    }

    __syncthreads();    // the input rows are dead: reuse the front of LDS as the swizzled output tile
#pragma unroll
    for (int c = 0; c < 8; ++c)
        *reinterpret_cast<u32x4 *>(lds + tile_off(lane, c)) = u32x4{pk[c * 4 + 0], pk[c * 4 + 1], pk[c * 4 + 2], pk[c * 4 + 3]};
    __syncthreads();
    store_tile<NT>(lds, out, g0, nblk, lane);
}

// ------------------------------------------------------------------------------------------------
// fused forward, ONE WAVEFRONT PER BLOCK (the layout sketched in BASELINE.json's north_star):
// lane = one coefficient, the strip staged in LDS, both 1-D passes as 8 per-lane FMAs fed by
// ds_bpermute (__shfl) from the 8 lanes of the row / column, quantise + zigzag scatter into the
// LDS tile.  Kept as a selectable variant (JPEGX_F_TUNE_WAVE_PER_BLOCK) so that the choice of
// the lane-per-block kernel above rests on a measurement (profiles/r01_ab_wave_per_block.txt),
// not on an estimate: it needs 16 cross-lane fetches + 16 FMAs per BLOCK where lane-per-block
// spends ~9 VALU instructions per block and no cross-lane traffic.  Results are identical.
// The fp32 dots here are plain 8-term FMA chains (<= 9 roundings per pass), so the error
// bound is scaled by 1.5 (24 u S) to stay rigorous.
// ------------------------------------------------------------------------------------------------
__device__ const float c_dct32[64] = {JPEGX_TABLE_DCT_MATRIX};   // fp32 roundings of C[k][n]

constexpr int WPB_LDS_BYTES = STRIP_BYTES + TILE_BYTES + SCRATCH_DOUBLES * 8;

template <int VAR, bool NT>
__global__ __launch_bounds__(64) void k_forward_fused_wpb(const float *__restrict__ in, size_t pitch, int wb,
                                                          int nblk, QuantParams prm, int16_t *__restrict__ out,
                                                          unsigned long long *counters)
{
    constexpr bool PIXEL = (VAR & 1) != 0;
    constexpr bool DC_EXACT = (VAR & 2) != 0;
    __shared__ __attribute__((aligned(16))) unsigned char lds[WPB_LDS_BYTES];
    unsigned char *tile = lds + STRIP_BYTES;
    double *sA = reinterpret_cast<double *>(lds + STRIP_BYTES + TILE_BYTES);
    double *sM = sA + 64;

    const int lane = threadIdx.x;
    const int g0 = blockIdx.x * 64;
    const int nvalid = min(64, nblk - g0);
    {
        const float *src[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = strip_swz(64 * j + lane);
            const int gb = min(g0 + (c >> 1), nblk - 1);
            const int by = gb / wb, bx = gb - by * wb;
            src[j] = in + (size_t)by * 8 * pitch + (size_t)bx * 8 + (c & 1) * 4;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[0] + (size_t)r * pitch),
                                             (__attribute__((address_space(3))) void *)(lds + r * 2048), 16, 0, NT ? 2 : 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[1] + (size_t)r * pitch),
                                             (__attribute__((address_space(3))) void *)(lds + r * 2048 + 1024), 16, 0, NT ? 2 : 0);
        }
    }
    const int hi = lane >> 3, lo = lane & 7;
    float crow[8], ccol[8];                     // C[l][0..7] for the row pass, C[k][0..7] for the column pass
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        crow[n] = c_dct32[lo * 8 + n];
        ccol[n] = c_dct32[hi * 8 + n];
    }
    float rq = 0.f;
#pragma unroll
    for (int n = 0; n < 64; ++n) rq = (lane == n) ? prm.rq32[n] : rq;   // kernarg SGPRs -> this lane's reciprocal
    const int pz = c_zzinv.v[lane];
    __syncthreads();                            // the strip has landed

    unsigned nexact = 0;
    for (int b = 0; b < nvalid; ++b) {
        const int fb = ((b >> 2) ^ (b >> 3)) & 1;
        const float x = *reinterpret_cast<const float *>(lds + hi * 2048 + ((2 * b + ((lo >> 2) ^ fb)) << 4) + (lo & 3) * 4);
        float m = 0.f;                          // row pass: lane (i, l) = sum_n C[l][n] x[i][n]
#pragma unroll
        for (int n = 0; n < 8; ++n) m = fmaf(crow[n], __shfl(x, (lane & 56) | n), m);
        float y = 0.f;                          // column pass: lane (k, l) = sum_i C[k][i] m[i][l]
#pragma unroll
        for (int i = 0; i < 8; ++i) y = fmaf(ccol[i], __shfl(m, (i << 3) | lo), y);
        float S;
        if (PIXEL) {
            S = __shfl(y, 0);                   // DC = sum of the (non-negative) samples
        } else {
            S = fabsf(x);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) S += __shfl_xor(S, d);
        }
        const float E = 1.5f * jpegx_fwd_err_bound(S);
        const float t = y * rq;
        float r = rintf(t);
        bool unsafe = !(fmaf(E, fabsf(rq), fabsf(t - r)) < 0.5f);
        if (DC_EXACT && lane == 0) unsafe = false;
        bool any = __any(unsafe) != 0;
        if (prm.tune & 1) any = false;
        if (any) {                              // exact tier: already in the one-wave-per-block layout
            const double yd = coop_fwd_exact((double)x, sA, sM, lane);
            r = (float)jpegx_clamp_i16(jpegx_quant_ref(yd, lane, prm.mode, prm.param, c_rq64.v));
            ++nexact;
        }
        const int q = PIXEL ? (int)r : min(max((int)r, -32768), 32767);
        *reinterpret_cast<int16_t *>(tile + tile_off(b, pz >> 3) + (pz & 7) * 2) = (int16_t)q;
    }
    if (counters != nullptr && lane == 0) {
        atomicAdd(&counters[0], (unsigned long long)nexact);
        atomicAdd(&counters[1], (unsigned long long)nvalid);
    }
    __syncthreads();
    unsigned char *dst = reinterpret_cast<unsigned char *>(out) + (size_t)g0 * 128;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = i * 8 + (lane >> 3), c = lane & 7;
        const u32x4 q4 = *reinterpret_cast<const u32x4 *>(tile + tile_off(row, c));
        if (row < nvalid) st_u32x4<NT>(dst + (size_t)row * 128 + c * 16, q4);
    }
}

// ------------------------------------------------------------------------------------------------
// fused inverse: un-zigzag + dequantise + IDCT + round (+ clamp).  OUT: 0 f32, 1 i16, 2 u8.
// ------------------------------------------------------------------------------------------------
template <int OUT, bool NT>
__global__ __launch_bounds__(64) void k_inverse_fused(const int16_t *__restrict__ in, int wb, int nblk,
                                                      QuantParams prm, int clamp, int inflate,
                                                      void *__restrict__ outv, size_t opitch,
                                                      unsigned long long *counters)
{
    // f32 output is staged through an 8-row x 2 KiB strip (coalesced 1 KiB stores); the narrower
    // i16 / u8 rows are already contiguous per store instruction and go out directly.
    constexpr int LDSB = (OUT == 0) ? STRIP_LDS_BYTES : LDS_BYTES;
    constexpr int SCR = (OUT == 0) ? STRIP_BYTES : TILE_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDSB];
    double *sA = reinterpret_cast<double *>(lds + SCR);
    double *sM = sA + 64;
    float *sP = reinterpret_cast<float *>(sA);  // 64 patched samples (reuses sA after the exchange)

    const int lane = threadIdx.x;
    const int g0 = blockIdx.x * 64;
    const int g = g0 + lane;
    const bool valid = g < nblk;

    // the wave's 8 KiB of coefficients -> swizzled LDS tile by LDS-DMA: piece i fills tile rows
    // 8i..8i+7; lane l lands in (row 8i + l/8, slot l%8), which must hold chunk slot ^ (row & 7).
    {
        const int row0 = lane >> 3, c = (lane & 7) ^ (lane >> 3);
        const int last = nblk - 1 - g0;   // rows past the end re-read the last block
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = min(i * 8 + row0, last);
            const unsigned char *src = reinterpret_cast<const unsigned char *>(in) + (size_t)(g0 + row) * 128 + c * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(lds + i * 1024), 16, 0, NT ? 2 : 0);
        }
    }
    __syncthreads();

    float v[64];
    float Sac = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint4 q = *reinterpret_cast<const uint4 *>(lds + tile_off(lane, c));
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int p = c * 8 + s * 2 + h;
                const int n = kZZ.v[p];
                const int z = h ? ((int)w[s] >> 16) : (int)(short)(w[s] & 0xFFFFu);
                const float d = (float)z * prm.rq32[n];  // quantizers.py:8-9,30-31,51-53
                v[n] = d;
                if (n != 0) Sac += fabsf(d);
            }
        }
    }
    const float E = jpegx_inv_err_bound(fabsf(v[0]), Sac);
    jpegx_idct8x8_f32(v);

    float worst = 0.f;
#pragma unroll
    for (int n = 0; n < 64; ++n) {
        const float r = rintf(v[n]);  // np.round of basis_change.py:43
        worst = fmaxf(worst, fabsf(v[n] - r));
        v[n] = r;
    }
    unsigned long long flagged = __ballot(valid && !(worst + E < 0.5f));
    census(counters, flagged, nblk - g0, lane);
    if (prm.tune & 1) flagged = 0;
    while (flagged) {
        const int b = __ffsll((long long)flagged) - 1;
        flagged &= flagged - 1;
        const int pz = c_zzinv.v[lane];
        const int z = *reinterpret_cast<const int16_t *>(lds + tile_off(b, pz >> 3) + (pz & 7) * 2);
        const double zd = jpegx_restore_ref((double)z, lane, prm.mode, prm.param, c_qt.v);
        const double y = coop_inv_exact(zd, sA, sM, lane);
        sP[lane] = (float)rint(y);  // exact: |y| is far below 2^24 for int16 coefficients
        __syncthreads();
        if (lane == b) {
#pragma unroll
            for (int n = 0; n < 64; n += 4) {
                const float4 t = *reinterpret_cast<const float4 *>(&sP[n]);
                v[n] = t.x; v[n + 1] = t.y; v[n + 2] = t.z; v[n + 3] = t.w;
            }
        }
        __syncthreads();
    }

    if (OUT == 0) {
        // park the lane's 8 x 32 B in the strip layout of the forward kernel (chunk c at slot
        // strip_swz(c)), then 16 coalesced 1 KiB stores; the coefficient tile is dead by now.
        __syncthreads();
        const int f = ((lane >> 2) ^ (lane >> 3)) & 1;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float x[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) x[c] = clamp ? fminf(fmaxf(v[r * 8 + c], 0.f), 255.f) : v[r * 8 + c];
            *reinterpret_cast<f32x4 *>(lds + r * 2048 + ((2 * lane + f) << 4)) = f32x4{x[0], x[1], x[2], x[3]};
            *reinterpret_cast<f32x4 *>(lds + r * 2048 + ((2 * lane + (f ^ 1)) << 4)) = f32x4{x[4], x[5], x[6], x[7]};
        }
        __syncthreads();
        float *dstp[2];
        bool ok[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = strip_swz(64 * j + lane);
            const int gb = g0 + (c >> 1);
            ok[j] = gb < nblk;
            const int gbc = ok[j] ? gb : nblk - 1;
            const int by = gbc / wb, bx = gbc - by * wb;
            dstp[j] = reinterpret_cast<float *>(outv) + (size_t)by * 8 * opitch + (size_t)bx * 8 + (c & 1) * 4;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 t = *reinterpret_cast<const f32x4 *>(lds + r * 2048 + j * 1024 + lane * 16);
                if (ok[j]) st_f32x4<NT>(dstp[j] + (size_t)r * opitch, t);
            }
        }
        return;
    }
    if (!valid) return;
    const int by = g / wb, bx = g - by * wb;
    if (OUT == 1) {
        int16_t *o = reinterpret_cast<int16_t *>(outv) + (size_t)by * 8 * opitch + (size_t)bx * 8;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            unsigned w[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float lo = v[r * 8 + 2 * c], hi = v[r * 8 + 2 * c + 1];
                if (clamp) { lo = fminf(fmaxf(lo, 0.f), 255.f); hi = fminf(fmaxf(hi, 0.f), 255.f); }
                const int a = min(max((int)lo, -32768), 32767), b2 = min(max((int)hi, -32768), 32767);
                w[c] = ((unsigned)a & 0xFFFFu) | ((unsigned)b2 << 16);
            }
            st_u32x4<NT>(o + (size_t)r * opitch, u32x4{w[0], w[1], w[2], w[3]});
        }
    } else {
        // uint8 rows, optionally with SubSampling.invert fused (util.inflate, util.py:6-14): every
        // sample is replicated inflate x inflate times, the output plane is [H*inflate][W*inflate].
        unsigned char *o = reinterpret_cast<unsigned char *>(outv) + (size_t)by * 8 * inflate * opitch + (size_t)bx * 8 * inflate;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            unsigned u[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) u[c] = (unsigned)fminf(fmaxf(v[r * 8 + c], 0.f), 255.f);
            if (inflate == 1) {
                st_u32x2<NT>(o + (size_t)r * opitch, u32x2{u[0] | (u[1] << 8) | (u[2] << 16) | (u[3] << 24),
                                                            u[4] | (u[5] << 8) | (u[6] << 16) | (u[7] << 24)});
            } else if (inflate == 2) {
                u32x4 w;
                w.x = (u[0] * 0x0101u) | ((u[1] * 0x0101u) << 16);
                w.y = (u[2] * 0x0101u) | ((u[3] * 0x0101u) << 16);
                w.z = (u[4] * 0x0101u) | ((u[5] * 0x0101u) << 16);
                w.w = (u[6] * 0x0101u) | ((u[7] * 0x0101u) << 16);
                st_u32x4<NT>(o + (size_t)(2 * r) * opitch, w);
                st_u32x4<NT>(o + (size_t)(2 * r + 1) * opitch, w);
            } else {
                const u32x4 w0 = {u[0] * 0x01010101u, u[1] * 0x01010101u, u[2] * 0x01010101u, u[3] * 0x01010101u};
                const u32x4 w1 = {u[4] * 0x01010101u, u[5] * 0x01010101u, u[6] * 0x01010101u, u[7] * 0x01010101u};
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    st_u32x4<NT>(o + (size_t)(4 * r + a) * opitch, w0);
                    st_u32x4<NT>(o + (size_t)(4 * r + a) * opitch + 16, w1);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// unfused fp32 stage kernels (lane-per-block, natural layout in and out)
// ------------------------------------------------------------------------------------------------
template <bool INVERSE>
__global__ __launch_bounds__(64) void k_dct8x8_f32(const float *__restrict__ in, size_t pitch, int wb, int nblk,
                                                   float *__restrict__ out, size_t opitch)
{
    const int g = blockIdx.x * 64 + threadIdx.x;
    if (g >= nblk) return;
    const int by = g / wb, bx = g - by * wb;
    const float *src = in + (size_t)by * 8 * pitch + (size_t)bx * 8;
    float v[64];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float4 *row = reinterpret_cast<const float4 *>(src + (size_t)r * pitch);
        const float4 lo = row[0], hi = row[1];
        v[r * 8 + 0] = lo.x; v[r * 8 + 1] = lo.y; v[r * 8 + 2] = lo.z; v[r * 8 + 3] = lo.w;
        v[r * 8 + 4] = hi.x; v[r * 8 + 5] = hi.y; v[r * 8 + 6] = hi.z; v[r * 8 + 7] = hi.w;
    }
    if (INVERSE) jpegx_idct8x8_f32(v); else jpegx_dct8x8_f32(v);
    float *dst = out + (size_t)by * 8 * opitch + (size_t)bx * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        float4 *row = reinterpret_cast<float4 *>(dst + (size_t)r * opitch);
        row[0] = make_float4(v[r * 8 + 0], v[r * 8 + 1], v[r * 8 + 2], v[r * 8 + 3]);
        row[1] = make_float4(v[r * 8 + 4], v[r * 8 + 5], v[r * 8 + 6], v[r * 8 + 7]);
    }
}

// ------------------------------------------------------------------------------------------------
// exact float64 stage kernels (bit-identical to the reference's float64 arrays)
// ------------------------------------------------------------------------------------------------
constexpr int F64_BLOCKS_PER_WAVE = 8;

template <bool INVERSE>
__global__ __launch_bounds__(64) void k_dct8x8_f64(const double *__restrict__ in, size_t pitch, int wb, int nblk,
                                                   double *__restrict__ out, size_t opitch, int do_round)
{
    __shared__ __attribute__((aligned(16))) double s[SCRATCH_DOUBLES];
    const int lane = threadIdx.x, i = lane >> 3, j = lane & 7;
    const int first = blockIdx.x * F64_BLOCKS_PER_WAVE;
    for (int t = 0; t < F64_BLOCKS_PER_WAVE; ++t) {
        const int g = first + t;
        if (g >= nblk) break;  // wave-uniform
        const int by = g / wb, bx = g - by * wb;
        const double a = in[(size_t)(by * 8 + i) * pitch + (size_t)bx * 8 + j];
        double y = INVERSE ? coop_inv_exact(a, s, s + 64, lane) : coop_fwd_exact(a, s, s + 64, lane);
        if (INVERSE && do_round) y = rint(y);
        out[(size_t)(by * 8 + i) * opitch + (size_t)bx * 8 + j] = y;
    }
}

template <bool RESTORE>
__global__ void k_quant_f64(const double *__restrict__ in, size_t pitch, int H, int W, int mode, double param,
                            double *__restrict__ out, size_t opitch)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)H * W) return;
    const int y = (int)(t / W), x = (int)(t - (size_t)y * W);
    const int n = (y & 7) * 8 + (x & 7);
    const double a = in[(size_t)y * pitch + x];
    out[(size_t)y * opitch + x] = RESTORE ? jpegx_restore_ref(a, n, mode, param, c_qt.v)
                                          : jpegx_quant_ref(a, n, mode, param, c_rq64.v);
}

// zigzag gather / scatter of fixed-size elements; one thread per element of the stream
template <typename T, bool INVERSE>
__global__ void k_zigzag(const T *__restrict__ in, size_t pitch, int wb, size_t nelem, T *__restrict__ out)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nelem) return;
    const size_t blk = t >> 6;
    const int p = (int)(t & 63);
    const int n = c_zz.v[p];
    const size_t by = blk / wb, bx = blk - by * wb;
    const size_t nat = (by * 8 + (n >> 3)) * pitch + bx * 8 + (n & 7);
    if (INVERSE) out[nat] = in[t]; else out[t] = in[nat];
}

__global__ void k_generate_plane(float *__restrict__ out, size_t pitch, int H, int W, int kind, uint32_t pseed,
                                 uint32_t row0)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per 4 pixels
    const int w4 = W >> 2;
    if (t >= (size_t)H * w4) return;
    const uint32_t y = (uint32_t)(t / w4), x = (uint32_t)(t - (size_t)y * w4) * 4;
    float4 v;
    v.x = (float)jpegx_synth_pixel(kind, pseed, (uint32_t)W, y + row0, x + 0);
    v.y = (float)jpegx_synth_pixel(kind, pseed, (uint32_t)W, y + row0, x + 1);
    v.z = (float)jpegx_synth_pixel(kind, pseed, (uint32_t)W, y + row0, x + 2);
    v.w = (float)jpegx_synth_pixel(kind, pseed, (uint32_t)W, y + row0, x + 3);
    *reinterpret_cast<float4 *>(out + (size_t)y * pitch + x) = v;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
thread_local char g_err[512] = "";
thread_local unsigned long long *g_counters = nullptr;

int fail(int code, const char *fmt, const char *detail = "")
{
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
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

template <int BS, bool NT, int STAGED = 0>
int launch_forward(const float *d_in, int H, int W, ptrdiff_t pitch, const QuantParams &qp, unsigned flags,
                   int16_t *d_out, hipStream_t st)
{
    const int wb = W / 8, nblk = (H / 8) * wb;
    const dim3 grid((nblk + 63) / 64), block(64);
    const bool pixel = (flags & JPEGX_F_PIXEL_INPUT) != 0;
    // DC is an exact integer multiple of 2^-8 and rq[0] a power of two -> DC/q needs no tie check
    const bool dc_exact = pixel && is_pow2_float(qp.rq32[0]) &&
                          (qp.mode != JPEGX_Q_DIVIDE || (double)qp.rq32[0] * qp.param == 1.0);
    if (BS == 1 && (flags & JPEGX_F_TUNE_WAVE_PER_BLOCK)) {
        if (dc_exact)
            hipLaunchKernelGGL((k_forward_fused_wpb<3, NT>), grid, block, 0, st, d_in, (size_t)pitch, wb, nblk, qp, d_out, g_counters);
        else if (pixel)
            hipLaunchKernelGGL((k_forward_fused_wpb<1, NT>), grid, block, 0, st, d_in, (size_t)pitch, wb, nblk, qp, d_out, g_counters);
        else
            hipLaunchKernelGGL((k_forward_fused_wpb<0, NT>), grid, block, 0, st, d_in, (size_t)pitch, wb, nblk, qp, d_out, g_counters);
    } else if (BS == 1 && !(flags & JPEGX_F_TUNE_NO_STRIP)) {
        if (dc_exact)
            hipLaunchKernelGGL((k_forward_fused_strip<3, NT>), grid, block, 0, st, d_in, (size_t)pitch, wb, nblk, qp, d_out, g_counters);
        else if (pixel)
            hipLaunchKernelGGL((k_forward_fused_strip<1, NT>), grid, block, 0, st, d_in, (size_t)pitch, wb, nblk, qp, d_out, g_counters);
        else
            hipLaunchKernelGGL((k_forward_fused_strip<0, NT>), grid, block, 0, st, d_in, (size_t)pitch, wb, nblk, qp, d_out, g_counters);
    } else if (dc_exact)
        hipLaunchKernelGGL((k_forward_fused<3, BS, NT, STAGED>), grid, block, 0, st, d_in, (size_t)pitch, wb, nblk, qp, d_out, g_counters);
    else if (pixel)
        hipLaunchKernelGGL((k_forward_fused<1, BS, NT, STAGED>), grid, block, 0, st, d_in, (size_t)pitch, wb, nblk, qp, d_out, g_counters);
    else
        hipLaunchKernelGGL((k_forward_fused<0, BS, NT, STAGED>), grid, block, 0, st, d_in, (size_t)pitch, wb, nblk, qp, d_out, g_counters);
    HIP_TRY(hipGetLastError());
    return JPEGX_OK;
}

template <bool INVERSE>
int zigzag_common(const void *d_in, int H, int W, ptrdiff_t pitch, int elem_size, void *d_out, jpegx_stream_t stream)
{
    int rc = check_plane(d_in, d_out, H, W, pitch, 1);
    if (rc) return rc;
    const int wb = W / 8;
    const size_t n = (size_t)H * W;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    switch (elem_size) {
    case 2: hipLaunchKernelGGL((k_zigzag<uint16_t, INVERSE>), grid, block, 0, st, (const uint16_t *)d_in, (size_t)pitch, wb, n, (uint16_t *)d_out); break;
    case 4: hipLaunchKernelGGL((k_zigzag<uint32_t, INVERSE>), grid, block, 0, st, (const uint32_t *)d_in, (size_t)pitch, wb, n, (uint32_t *)d_out); break;
    case 8: hipLaunchKernelGGL((k_zigzag<uint64_t, INVERSE>), grid, block, 0, st, (const uint64_t *)d_in, (size_t)pitch, wb, n, (uint64_t *)d_out); break;
    case 16: hipLaunchKernelGGL((k_zigzag<ulonglong2, INVERSE>), grid, block, 0, st, (const ulonglong2 *)d_in, (size_t)pitch, wb, n, (ulonglong2 *)d_out); break;
    default: return fail(JPEGX_E_INVALID, "zigzag: element size must be 2, 4, 8 or 16 bytes");
    }
    HIP_TRY(hipGetLastError());
    return JPEGX_OK;
}


struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { HIP_TRY(hipMalloc(&p, bytes ? bytes : 1)); return JPEGX_OK; }
};
struct Stream {
    hipStream_t s = nullptr;
    ~Stream() { if (s) (void)hipStreamDestroy(s); }
    int create() { HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); return JPEGX_OK; }
};

// generic "copy in, run, copy out" helper
template <typename F>
int host_roundtrip(const void *h_in, size_t in_bytes, void *h_out, size_t out_bytes, F &&run)
{
    if (!h_in || !h_out) return fail(JPEGX_E_INVALID, "null host pointer");
    DevBuf din, dout;
    Stream st;
    int rc;
    if ((rc = din.alloc(in_bytes)) || (rc = dout.alloc(out_bytes)) || (rc = st.create())) return rc;
    HIP_TRY(hipMemcpyAsync(din.p, h_in, in_bytes, hipMemcpyHostToDevice, st.s));
    if ((rc = run(din.p, dout.p, (jpegx_stream_t)st.s))) return rc;
    HIP_TRY(hipMemcpyAsync(h_out, dout.p, out_bytes, hipMemcpyDeviceToHost, st.s));
    HIP_TRY(hipStreamSynchronize(st.s));
    return JPEGX_OK;
}

}  // namespace

extern "C" {

// used by the other translation units of libjpegx.so (jpegx_entropy.hip); not part of the public ABI
void jpegx_internal_set_error(const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg ? msg : ""); }

const char *jpegx_last_error(void) { return g_err; }
int jpegx_version(void) { return JPEGX_VERSION; }

int jpegx_init(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return fail(JPEGX_E_NODEVICE, "no usable HIP device");
    if (device < 0 || device >= n) return fail(JPEGX_E_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(nullptr));   // forces context creation
    return JPEGX_OK;
}

int jpegx_shutdown(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return JPEGX_OK;
    HIP_TRY(hipDeviceSynchronize());
    return JPEGX_OK;
}

int jpegx_device_count(int *count)
{
    if (!count) return fail(JPEGX_E_INVALID, "null count pointer");
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        *count = 0;
        return fail(JPEGX_E_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    return JPEGX_OK;
}

int jpegx_set_device(int device) { HIP_TRY(hipSetDevice(device)); return JPEGX_OK; }
int jpegx_get_device(int *device)
{
    if (!device) return fail(JPEGX_E_INVALID, "null device pointer");
    HIP_TRY(hipGetDevice(device));
    return JPEGX_OK;
}

int jpegx_device_name(int device, char *buf, size_t buflen)
{
    if (!buf || buflen == 0) return fail(JPEGX_E_INVALID, "null name buffer");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return JPEGX_OK;
}

int jpegx_device_synchronize(void) { HIP_TRY(hipDeviceSynchronize()); return JPEGX_OK; }

int jpegx_malloc(void **dptr, size_t bytes)
{
    if (!dptr) return fail(JPEGX_E_INVALID, "null pointer");
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 1));
    return JPEGX_OK;
}
int jpegx_free(void *dptr) { HIP_TRY(hipFree(dptr)); return JPEGX_OK; }
int jpegx_memset(void *dptr, int value, size_t bytes, jpegx_stream_t stream)
{
    HIP_TRY(hipMemsetAsync(dptr, value, bytes, (hipStream_t)stream));
    return JPEGX_OK;
}
int jpegx_memcpy_h2d(void *dst, const void *src, size_t bytes, jpegx_stream_t stream)
{
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return JPEGX_OK;
}
int jpegx_memcpy_d2h(void *dst, const void *src, size_t bytes, jpegx_stream_t stream)
{
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return JPEGX_OK;
}
int jpegx_memcpy_d2d(void *dst, const void *src, size_t bytes, jpegx_stream_t stream)
{
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return JPEGX_OK;
}

int jpegx_stream_create(jpegx_stream_t *stream)
{
    if (!stream) return fail(JPEGX_E_INVALID, "null pointer");
    hipStream_t s;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (jpegx_stream_t)s;
    return JPEGX_OK;
}
int jpegx_stream_destroy(jpegx_stream_t stream) { HIP_TRY(hipStreamDestroy((hipStream_t)stream)); return JPEGX_OK; }
int jpegx_stream_synchronize(jpegx_stream_t stream) { HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); return JPEGX_OK; }
int jpegx_event_create(jpegx_event_t *event)
{
    if (!event) return fail(JPEGX_E_INVALID, "null pointer");
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    *event = (jpegx_event_t)e;
    return JPEGX_OK;
}
int jpegx_event_destroy(jpegx_event_t event) { HIP_TRY(hipEventDestroy((hipEvent_t)event)); return JPEGX_OK; }
int jpegx_event_record(jpegx_event_t event, jpegx_stream_t stream)
{
    HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return JPEGX_OK;
}
int jpegx_event_synchronize(jpegx_event_t event) { HIP_TRY(hipEventSynchronize((hipEvent_t)event)); return JPEGX_OK; }
int jpegx_event_elapsed_ms(jpegx_event_t start, jpegx_event_t stop, float *ms)
{
    if (!ms) return fail(JPEGX_E_INVALID, "null pointer");
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return JPEGX_OK;
}

int jpegx_set_debug_counters(unsigned long long *d_counters)
{
    g_counters = d_counters;
    return JPEGX_OK;
}

int jpegx_generate_plane(float *d_plane, int H, int W, ptrdiff_t pitch, int kind, uint32_t seed, uint32_t plane,
                         int row0, jpegx_stream_t stream)
{
    if (!d_plane) return fail(JPEGX_E_INVALID, "null device pointer");
    if (H <= 0 || W <= 0 || (W % 4) != 0 || pitch < W || (pitch % 4) != 0 || !aligned16(d_plane))
        return fail(JPEGX_E_INVALID, "generate_plane: W and pitch must be multiples of 4, base 16-byte aligned");
    if (kind != 0 && kind != 1) return fail(JPEGX_E_INVALID, "generate_plane: unknown kind");
    const uint32_t pseed = jpegx_hash32(seed + plane * 0x9E3779B9u);
    const size_t n = (size_t)H * (W / 4);
    hipLaunchKernelGGL(k_generate_plane, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_plane,
                       (size_t)pitch, H, W, kind, pseed, (uint32_t)row0);
    HIP_TRY(hipGetLastError());
    return JPEGX_OK;
}

int jpegx_forward_fused_pooled(const float *d_in, int H, int W, ptrdiff_t pitch, int bs, int mode, double param,
                               unsigned flags, int16_t *d_out, jpegx_stream_t stream)
{
    if (bs != 1 && bs != 2 && bs != 4) return fail(JPEGX_E_UNSUPPORTED, "fused mean-pool supports block_size 1, 2 and 4");
    int rc = check_plane(d_in, d_out, H, W, pitch / bs, 1);
    if (rc) return rc;
    if (pitch < (ptrdiff_t)W * bs || (pitch % 4) != 0 || !aligned16(d_in) || !aligned16(d_out))
        return fail(JPEGX_E_INVALID, "forward: pitch must be a multiple of 4 floats and >= W*bs; pointers 16-byte aligned");
    QuantParams qp;
    rc = fill_forward_params(mode, param, &qp);
    if (rc) return rc;
    if (flags & JPEGX_F_TUNE_SKIP_EXACT) qp.tune |= 1;
    hipStream_t st = (hipStream_t)stream;
    // Pooled input: staged through LDS by default (whole-line nontemporal DMA).  The per-lane
    // variant reads partial lines per instruction (16 B at a 32*bs-byte lane stride); nontemporal
    // loads then refetch every line and lose 30-60 % (profiles/r01_ab_pooled.txt), so that variant
    // always uses the default cache policy.
    if (bs > 1 && !(flags & JPEGX_F_TUNE_NO_STRIP)) {
        const bool nt = !(flags & JPEGX_F_TUNE_NO_NT);
        const int rpp = (flags & JPEGX_F_TUNE_POOL_ROWS_LO) ? 1 : ((flags & JPEGX_F_TUNE_POOL_ROWS_HI) ? 4 : 2);   // experiment: rows per phase
#define JPEGX_LF(BSV, RPPV) (nt ? launch_forward<BSV, true, RPPV>(d_in, H, W, pitch, qp, flags, d_out, st) \
                                : launch_forward<BSV, false, RPPV>(d_in, H, W, pitch, qp, flags, d_out, st))
        if (bs == 2) return rpp == 1 ? JPEGX_LF(2, 1) : (rpp == 4 ? JPEGX_LF(2, 4) : JPEGX_LF(2, 2));
        return rpp == 4 ? JPEGX_LF(4, 2) : JPEGX_LF(4, 1);
#undef JPEGX_LF
    }
    if (bs == 2) return launch_forward<2, false>(d_in, H, W, pitch, qp, flags, d_out, st);
    if (bs == 4) return launch_forward<4, false>(d_in, H, W, pitch, qp, flags, d_out, st);
    if (flags & JPEGX_F_TUNE_NO_NT) return launch_forward<1, false>(d_in, H, W, pitch, qp, flags, d_out, st);
    return launch_forward<1, true>(d_in, H, W, pitch, qp, flags, d_out, st);
}

int jpegx_forward_fused(const float *d_in, int H, int W, ptrdiff_t pitch, int mode, double param, unsigned flags,
                        int16_t *d_out, jpegx_stream_t stream)
{
    return jpegx_forward_fused_pooled(d_in, H, W, pitch, 1, mode, param, flags, d_out, stream);
}

int jpegx_forward_fused_u8(const uint8_t *d_in, int H, int W, ptrdiff_t pitch, int bs, int mode, double param,
                           unsigned flags, int16_t *d_out, jpegx_stream_t stream)
{
    if (bs != 1 && bs != 2) return fail(JPEGX_E_UNSUPPORTED, "uint8 forward supports block_size 1 and 2");
    int rc = check_plane(d_in, d_out, H, W, pitch / bs, 1);
    if (rc) return rc;
    if (pitch < (ptrdiff_t)W * bs || (pitch % 16) != 0 || !aligned16(d_in) || !aligned16(d_out))
        return fail(JPEGX_E_INVALID, "forward_u8: pitch must be a multiple of 16 bytes and >= W*bs; pointers 16-byte aligned");
    if (bs == 1 && (W % 16) != 0) return fail(JPEGX_E_UNSUPPORTED, "forward_u8 with block_size 1 needs W to be a multiple of 16");
    QuantParams qp;
    rc = fill_forward_params(mode, param, &qp);
    if (rc) return rc;
    if (flags & JPEGX_F_TUNE_SKIP_EXACT) qp.tune |= 1;
    const int wb = W / 8, nblk = (H / 8) * wb;
    const dim3 grid((nblk + 63) / 64), block(64);
    hipStream_t st = (hipStream_t)stream;
    const bool dc_exact = is_pow2_float(qp.rq32[0]) && (qp.mode != JPEGX_Q_DIVIDE || (double)qp.rq32[0] * qp.param == 1.0);
    const bool nt = !(flags & JPEGX_F_TUNE_NO_NT);
#define JPEGX_LU8(DC, BSV) \
    do { if (nt) hipLaunchKernelGGL((k_forward_fused_u8<DC, BSV, true>), grid, block, 0, st, d_in, (size_t)pitch, wb, nblk, qp, d_out, g_counters); \
         else hipLaunchKernelGGL((k_forward_fused_u8<DC, BSV, false>), grid, block, 0, st, d_in, (size_t)pitch, wb, nblk, qp, d_out, g_counters); } while (0)
    if (bs == 1) { if (dc_exact) JPEGX_LU8(true, 1); else JPEGX_LU8(false, 1); }
    else { if (dc_exact) JPEGX_LU8(true, 2); else JPEGX_LU8(false, 2); }
#undef JPEGX_LU8
    HIP_TRY(hipGetLastError());
    return JPEGX_OK;
}

static int inverse_common(const int16_t *d_in, int H, int W, int mode, double param, unsigned flags, void *d_out,
                          ptrdiff_t out_pitch, int out_type, int inflate, jpegx_stream_t stream)
{
    int rc = check_plane(d_in, d_out, H, W, out_pitch / inflate, 1);
    if (rc) return rc;
    if (out_pitch < (ptrdiff_t)W * inflate) return fail(JPEGX_E_INVALID, "output pitch smaller than the inflated width");
    const int esz = out_type == JPEGX_OUT_F32 ? 4 : (out_type == JPEGX_OUT_I16 ? 2 : 1);
    if (out_type < 0 || out_type > 2) return fail(JPEGX_E_INVALID, "unknown output type");
    if (((size_t)out_pitch * esz) % ((esz == 1 && inflate == 1) ? 8 : 16) != 0 || !aligned16(d_in) || !aligned16(d_out))
        return fail(JPEGX_E_INVALID, "inverse: output rows must stay 16-byte (plain u8: 8-byte) aligned");
    QuantParams qp;
    rc = fill_inverse_params(mode, param, &qp);
    if (rc) return rc;
    if (flags & JPEGX_F_TUNE_SKIP_EXACT) qp.tune |= 1;
    const int wb = W / 8, nblk = (H / 8) * wb;
    const dim3 grid((nblk + 63) / 64), block(64);
    const int clamp = (flags & JPEGX_F_CLAMP_U8) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    const bool nt = (flags & JPEGX_F_TUNE_NO_NT) == 0;
#define JPEGX_LAUNCH_INV(OUT, NT, CL) \
    hipLaunchKernelGGL((k_inverse_fused<OUT, NT>), grid, block, 0, st, d_in, wb, nblk, qp, CL, inflate, d_out, (size_t)out_pitch, g_counters)
    if (out_type == JPEGX_OUT_F32) { if (nt) JPEGX_LAUNCH_INV(0, true, clamp); else JPEGX_LAUNCH_INV(0, false, clamp); }
    else if (out_type == JPEGX_OUT_I16) { if (nt) JPEGX_LAUNCH_INV(1, true, clamp); else JPEGX_LAUNCH_INV(1, false, clamp); }
    else { if (nt) JPEGX_LAUNCH_INV(2, true, 1); else JPEGX_LAUNCH_INV(2, false, 1); }
#undef JPEGX_LAUNCH_INV
    HIP_TRY(hipGetLastError());
    return JPEGX_OK;
}

int jpegx_inverse_fused(const int16_t *d_in, int H, int W, int mode, double param, unsigned flags, void *d_out,
                        ptrdiff_t out_pitch, int out_type, jpegx_stream_t stream)
{
    return inverse_common(d_in, H, W, mode, param, flags, d_out, out_pitch, out_type, 1, stream);
}

int jpegx_inverse_fused_u8_inflated(const int16_t *d_in, int H, int W, int mode, double param, unsigned flags, int bs,
                                    uint8_t *d_out, ptrdiff_t out_pitch, jpegx_stream_t stream)
{
    if (bs != 1 && bs != 2 && bs != 4) return fail(JPEGX_E_UNSUPPORTED, "fused inflate supports block_size 1, 2 and 4");
    return inverse_common(d_in, H, W, mode, param, flags, d_out, out_pitch, JPEGX_OUT_U8, bs, stream);
}

int jpegx_dct8x8_f32(const float *d_in, int H, int W, ptrdiff_t pitch, float *d_out, ptrdiff_t out_pitch,
                     jpegx_stream_t stream)
{
    int rc = check_plane(d_in, d_out, H, W, pitch, 4);
    if (rc) return rc;
    if (out_pitch < W || (out_pitch % 4) != 0 || !aligned16(d_in) || !aligned16(d_out))
        return fail(JPEGX_E_INVALID, "dct8x8_f32: rows must be 16-byte aligned");
    const int wb = W / 8, nblk = (H / 8) * wb;
    hipLaunchKernelGGL((k_dct8x8_f32<false>), dim3((nblk + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_in, (size_t)pitch,
                       wb, nblk, d_out, (size_t)out_pitch);
    HIP_TRY(hipGetLastError());
    return JPEGX_OK;
}

int jpegx_idct8x8_f32(const float *d_in, int H, int W, ptrdiff_t pitch, float *d_out, ptrdiff_t out_pitch,
                      jpegx_stream_t stream)
{
    int rc = check_plane(d_in, d_out, H, W, pitch, 4);
    if (rc) return rc;
    if (out_pitch < W || (out_pitch % 4) != 0 || !aligned16(d_in) || !aligned16(d_out))
        return fail(JPEGX_E_INVALID, "idct8x8_f32: rows must be 16-byte aligned");
    const int wb = W / 8, nblk = (H / 8) * wb;
    hipLaunchKernelGGL((k_dct8x8_f32<true>), dim3((nblk + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_in, (size_t)pitch,
                       wb, nblk, d_out, (size_t)out_pitch);
    HIP_TRY(hipGetLastError());
    return JPEGX_OK;
}

int jpegx_dct8x8_f64(const double *d_in, int H, int W, ptrdiff_t pitch, double *d_out, ptrdiff_t out_pitch,
                     jpegx_stream_t stream)
{
    int rc = check_plane(d_in, d_out, H, W, pitch, 1);
    if (rc) return rc;
    if (out_pitch < W) return fail(JPEGX_E_INVALID, "output pitch smaller than width");
    const int wb = W / 8, nblk = (H / 8) * wb;
    hipLaunchKernelGGL((k_dct8x8_f64<false>), dim3((nblk + F64_BLOCKS_PER_WAVE - 1) / F64_BLOCKS_PER_WAVE), dim3(64), 0,
                       (hipStream_t)stream, d_in, (size_t)pitch, wb, nblk, d_out, (size_t)out_pitch, 0);
    HIP_TRY(hipGetLastError());
    return JPEGX_OK;
}

int jpegx_idct8x8_f64(const double *d_in, int H, int W, ptrdiff_t pitch, double *d_out, ptrdiff_t out_pitch,
                      int do_round, jpegx_stream_t stream)
{
    int rc = check_plane(d_in, d_out, H, W, pitch, 1);
    if (rc) return rc;
    if (out_pitch < W) return fail(JPEGX_E_INVALID, "output pitch smaller than width");
    const int wb = W / 8, nblk = (H / 8) * wb;
    hipLaunchKernelGGL((k_dct8x8_f64<true>), dim3((nblk + F64_BLOCKS_PER_WAVE - 1) / F64_BLOCKS_PER_WAVE), dim3(64), 0,
                       (hipStream_t)stream, d_in, (size_t)pitch, wb, nblk, d_out, (size_t)out_pitch, do_round);
    HIP_TRY(hipGetLastError());
    return JPEGX_OK;
}

static int quant_f64_common(bool restore, const double *d_in, int H, int W, ptrdiff_t pitch, int mode, double param,
                            double *d_out, ptrdiff_t out_pitch, jpegx_stream_t stream)
{
    int rc = check_plane(d_in, d_out, H, W, pitch, 1);
    if (rc) return rc;
    if (out_pitch < W) return fail(JPEGX_E_INVALID, "output pitch smaller than width");
    QuantParams qp;  // validates mode / param
    rc = restore ? fill_inverse_params(mode, param, &qp) : fill_forward_params(mode, param, &qp);
    if (rc) return rc;
    const size_t n = (size_t)H * W;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (restore)
        hipLaunchKernelGGL((k_quant_f64<true>), grid, block, 0, (hipStream_t)stream, d_in, (size_t)pitch, H, W, mode, param, d_out, (size_t)out_pitch);
    else
        hipLaunchKernelGGL((k_quant_f64<false>), grid, block, 0, (hipStream_t)stream, d_in, (size_t)pitch, H, W, mode, param, d_out, (size_t)out_pitch);
    HIP_TRY(hipGetLastError());
    return JPEGX_OK;
}

int jpegx_quantize_f64(const double *d_in, int H, int W, ptrdiff_t pitch, int mode, double param, double *d_out,
                       ptrdiff_t out_pitch, jpegx_stream_t stream)
{
    return quant_f64_common(false, d_in, H, W, pitch, mode, param, d_out, out_pitch, stream);
}

int jpegx_restore_f64(const double *d_in, int H, int W, ptrdiff_t pitch, int mode, double param, double *d_out,
                      ptrdiff_t out_pitch, jpegx_stream_t stream)
{
    return quant_f64_common(true, d_in, H, W, pitch, mode, param, d_out, out_pitch, stream);
}

int jpegx_zigzag(const void *d_in, int H, int W, ptrdiff_t pitch, int elem_size, void *d_out, jpegx_stream_t stream)
{
    return zigzag_common<false>(d_in, H, W, pitch, elem_size, d_out, stream);
}

int jpegx_unzigzag(const void *d_in, int H, int W, int elem_size, void *d_out, ptrdiff_t out_pitch, jpegx_stream_t stream)
{
    return zigzag_common<true>(d_in, H, W, out_pitch, elem_size, d_out, stream);
}

// ---- synchronous host-pointer conveniences ------------------------------------------------------
int jpegx_host_forward_fused(const float *h_in, int H, int W, ptrdiff_t pitch, int mode, double param, unsigned flags,
                             int16_t *h_out)
{
    if (H <= 0 || W <= 0 || pitch < W) return fail(JPEGX_E_INVALID, "bad plane shape");
    return host_roundtrip(h_in, (size_t)H * pitch * 4, h_out, (size_t)H * W * 2, [&](void *di, void *dout, jpegx_stream_t s) {
        return jpegx_forward_fused((const float *)di, H, W, pitch, mode, param, flags, (int16_t *)dout, s);
    });
}

int jpegx_host_inverse_fused(const int16_t *h_in, int H, int W, int mode, double param, unsigned flags, void *h_out,
                             ptrdiff_t out_pitch, int out_type)
{
    if (H <= 0 || W <= 0 || out_pitch < W || out_type < 0 || out_type > 2) return fail(JPEGX_E_INVALID, "bad plane shape or type");
    const int esz = out_type == JPEGX_OUT_F32 ? 4 : (out_type == JPEGX_OUT_I16 ? 2 : 1);
    return host_roundtrip(h_in, (size_t)H * W * 2, h_out, (size_t)H * out_pitch * esz, [&](void *di, void *dout, jpegx_stream_t s) {
        return jpegx_inverse_fused((const int16_t *)di, H, W, mode, param, flags, dout, out_pitch, out_type, s);
    });
}

int jpegx_host_dct8x8_f64(const double *h_in, int H, int W, double *h_out)
{
    if (H <= 0 || W <= 0) return fail(JPEGX_E_INVALID, "bad plane shape");
    return host_roundtrip(h_in, (size_t)H * W * 8, h_out, (size_t)H * W * 8, [&](void *di, void *dout, jpegx_stream_t s) {
        return jpegx_dct8x8_f64((const double *)di, H, W, W, (double *)dout, W, s);
    });
}

int jpegx_host_idct8x8_f64(const double *h_in, int H, int W, double *h_out, int do_round)
{
    if (H <= 0 || W <= 0) return fail(JPEGX_E_INVALID, "bad plane shape");
    return host_roundtrip(h_in, (size_t)H * W * 8, h_out, (size_t)H * W * 8, [&](void *di, void *dout, jpegx_stream_t s) {
        return jpegx_idct8x8_f64((const double *)di, H, W, W, (double *)dout, W, do_round, s);
    });
}

int jpegx_host_quantize_f64(const double *h_in, int H, int W, int mode, double param, double *h_out)
{
    if (H <= 0 || W <= 0) return fail(JPEGX_E_INVALID, "bad plane shape");
    return host_roundtrip(h_in, (size_t)H * W * 8, h_out, (size_t)H * W * 8, [&](void *di, void *dout, jpegx_stream_t s) {
        return jpegx_quantize_f64((const double *)di, H, W, W, mode, param, (double *)dout, W, s);
    });
}

int jpegx_host_restore_f64(const double *h_in, int H, int W, int mode, double param, double *h_out)
{
    if (H <= 0 || W <= 0) return fail(JPEGX_E_INVALID, "bad plane shape");
    return host_roundtrip(h_in, (size_t)H * W * 8, h_out, (size_t)H * W * 8, [&](void *di, void *dout, jpegx_stream_t s) {
        return jpegx_restore_f64((const double *)di, H, W, W, mode, param, (double *)dout, W, s);
    });
}

int jpegx_host_zigzag(const void *h_in, int H, int W, int elem_size, void *h_out)
{
    if (H <= 0 || W <= 0 || elem_size <= 0) return fail(JPEGX_E_INVALID, "bad plane shape");
    const size_t bytes = (size_t)H * W * elem_size;
    return host_roundtrip(h_in, bytes, h_out, bytes, [&](void *di, void *dout, jpegx_stream_t s) {
        return jpegx_zigzag(di, H, W, W, elem_size, dout, s);
    });
}

int jpegx_host_unzigzag(const void *h_in, int H, int W, int elem_size, void *h_out)
{
    if (H <= 0 || W <= 0 || elem_size <= 0) return fail(JPEGX_E_INVALID, "bad plane shape");
    const size_t bytes = (size_t)H * W * elem_size;
    return host_roundtrip(h_in, bytes, h_out, bytes, [&](void *di, void *dout, jpegx_stream_t s) {
        return jpegx_unzigzag(di, H, W, elem_size, dout, W, s);
    });
}

int jpegx_host_dct8x8_f32(const float *h_in, int H, int W, float *h_out)
{
    if (H <= 0 || W <= 0) return fail(JPEGX_E_INVALID, "bad plane shape");
    return host_roundtrip(h_in, (size_t)H * W * 4, h_out, (size_t)H * W * 4, [&](void *di, void *dout, jpegx_stream_t s) {
        return jpegx_dct8x8_f32((const float *)di, H, W, W, (float *)dout, W, s);
    });
}

int jpegx_host_idct8x8_f32(const float *h_in, int H, int W, float *h_out)
{
    if (H <= 0 || W <= 0) return fail(JPEGX_E_INVALID, "bad plane shape");
    return host_roundtrip(h_in, (size_t)H * W * 4, h_out, (size_t)H * W * 4, [&](void *di, void *dout, jpegx_stream_t s) {
        return jpegx_idct8x8_f32((const float *)di, H, W, W, (float *)dout, W, s);
    });
}

}  // extern "C"
