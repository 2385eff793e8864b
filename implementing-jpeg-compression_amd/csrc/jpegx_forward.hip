// jpegx_forward.hip -- gfx950 (MI355X, CDNA4) fused forward kernels + their C ABI entries (libjpegx.so).
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
// This file: the fused forward kernels (fp32 strip kernel = default, its column-wise-tier form for finer
// quantisers, per-lane / LDS-staged pooled variants, several planes in one grid, uint8 input, the
// all-float64 kernel with eight lanes per block for the finest quantisers and float64 planes, the
// one-wavefront-per-block and lane-per-block-float64 comparison variants) and their C entry points.
// Built with: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize (explicit fma only).
#include "jpegx_internal.h"
#include <string.h>

namespace {

// ------------------------------------------------------------------------------------------------
// fused forward: DCT + quantise + zigzag.  VAR bit0 = PIXEL_INPUT, bit1 = DC exact.
// BS = mean-pool factor of the fused SubSampling prologue (1 = none).
// ------------------------------------------------------------------------------------------------
template <int BS, int STAGED> struct PooledLds {
    static constexpr int STAGE_BYTES = STAGED * 2 * BS * 1024;                 // staging buffer (0 if not staged)
    static constexpr int FRONT = STAGE_BYTES > TILE_BYTES ? STAGE_BYTES : TILE_BYTES;
    // staged variants: behind the cooperative scratch, the exact tier's 1 KiB table (by LDS-DMA) and 128 B
    // through which the owner of a flagged block hands its samples to the wave
    static constexpr int TAB = FRONT + SCRATCH_DOUBLES * 8;
    static constexpr int XBLK = TAB + 1024;
    static constexpr int BYTES = STAGED ? XBLK + 128 : FRONT + SCRATCH_DOUBLES * 8;
};

// body of one workgroup (= one wave = 64 blocks starting at block 64 * wg of this plane); `lds` is the
// workgroup's PooledLds<BS, STAGED>::BYTES of shared memory
template <int VAR, int BS, bool NT, int STAGED>
__device__ __forceinline__ void forward_fused_body(unsigned char *lds, int wg, const float *__restrict__ in, size_t pitch, int wb,
                                                   int nblk, const QuantParams &prm, int16_t *__restrict__ out,
                                                   unsigned long long *counters)
{
    constexpr bool PIXEL = (VAR & 1) != 0;
    constexpr bool DC_EXACT = (VAR & 2) != 0;
    constexpr int FRONT = PooledLds<BS, STAGED>::FRONT;
    double *sA = reinterpret_cast<double *>(lds + FRONT);
    double *sM = sA + 64;

    const int lane = threadIdx.x;
    const int g0 = wg * 64;
    const int g = g0 + lane;
    const bool valid = g < nblk;
    const int gc = valid ? g : nblk - 1;
    const int by = gc / wb, bx = gc - by * wb;
    const float *src = in + ((size_t)by * 8 * BS) * pitch + (size_t)bx * 8 * BS;

    // Staged variants keep the exact tier free of vector memory loads (under the streaming load such a
    // load waits for microseconds, see k_inverse_fused): its tables come in by one LDS-DMA instruction here,
    // and with 8-bit content pooled 2x2 the lane keeps its 64 pooled samples as fp16 (k/4 <= 255 is exact
    // there) in 32 registers instead of re-reading the 16 x 16 input samples of a flagged block from memory.
    constexpr bool TABBED = STAGED > 0;
    constexpr bool STASH = TABBED && PIXEL && BS == 2;
    if (TABBED)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const unsigned char *>(&c_fwd_exact) + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + PooledLds<BS, STAGED>::TAB), 16, 0, 0);
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

    unsigned stash[STASH ? 32 : 1];
    if (STASH) {
        typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int p = 0; p < 32; ++p) {
            const h16x2 h = __builtin_amdgcn_cvt_pkrtz(v[2 * p], v[2 * p + 1]);
            stash[p] = __builtin_bit_cast(unsigned, h);
            // pack HERE: the optimiser would sink the conversions down to their use in the exact tier and
            // keep all 64 raw samples alive next to the DCT's (164 VGPRs instead of 118)
            asm volatile("" : "+v"(stash[p]));
        }
    }
    float S = 0.f;
    if (!PIXEL) {
#pragma unroll
        for (int n = 0; n < 64; ++n) S += fabsf(v[n]);
    }
    jpegx_dct8x8_aan_f32(v);
    if (PIXEL) S = v[0];  // non-negative samples: sum|x| == DC, exact
    // generic pooled input: the fp32 tile sums are themselves rounded (<= BS^2 u each)
    const float E = jpegx_fwd_err_unit(S) * ((PIXEL || BS == 1) ? 1.0f : 1.0f + (BS * BS) / 16.0f);

    // quantise in zigzag order, pack pairs, track the worst rounding margin
    unsigned pk[32];
    const float worst = quantise_zigzag_pack<PIXEL, DC_EXACT>(v, prm, E, pk);

    // park the lane's 128 B in the swizzled tile
#pragma unroll
    for (int c = 0; c < 8; ++c)
        *reinterpret_cast<uint4 *>(lds + tile_off(lane, c)) =
            make_uint4(pk[c * 4 + 0], pk[c * 4 + 1], pk[c * 4 + 2], pk[c * 4 + 3]);

    // exact tier for blocks that sit within the error bound of a rounding boundary
    unsigned long long flagged = __ballot(valid && !(worst < JPEGX_SAFE_HALF));
    census(counters, flagged, nblk - g0, lane);
    if (prm.tune & 1) flagged = 0;
    __syncthreads();
    if (TABBED && flagged) {
        const unsigned char *tab = lds + PooledLds<BS, STAGED>::TAB;
        unsigned char *xblk = lds + PooledLds<BS, STAGED>::XBLK;
        const double *tabC = reinterpret_cast<const double *>(tab);
        const double rq_lane = 1.0 / (double)tab[512 + lane];          // quantizers.py:49 "1.0 / q", same quotient
        const int pz = tab[576 + lane];
        const int i = lane >> 3, j = lane & 7;
        auto fetch = [&](int b) -> double {                             // variants without the fp16 stash
            const int gb = g0 + b;
            const int byb = gb / wb, bxb = gb - byb * wb;
            const float *p = in + ((size_t)(byb * 8 + i) * BS) * pitch + (size_t)(bxb * 8 + j) * BS;
            double s = 0.0;  // np.mean: float64 sum then one division (subsampling.py:11)
#pragma unroll
            for (int u = 0; u < BS; ++u)
#pragma unroll
                for (int w = 0; w < BS; ++w) s += (double)p[(size_t)u * pitch + w];
            return s / (double)(BS * BS);
        };
        double a_next = STASH ? 0.0 : fetch(__ffsll((long long)flagged) - 1);
        while (flagged) {
            const int b = __ffsll((long long)flagged) - 1;
            flagged &= flagged - 1;
            double a;
            if (STASH) {
                if (lane == b) {
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        *reinterpret_cast<u32x4 *>(xblk + c * 16) = u32x4{stash[4 * c], stash[4 * c + 1], stash[4 * c + 2], stash[4 * c + 3]};
                }
                __syncthreads();
                a = (double)(float)*reinterpret_cast<const __fp16 *>(xblk + lane * 2);   // exact: the pooled sample k/4
            } else {
                a = a_next;
                if (flagged) a_next = fetch(__ffsll((long long)flagged) - 1);
            }
            const double y = coop_fwd_exact_tab(a, sA, sM, lane, tabC);
            const double r = jpegx_quant_lane(y, lane, prm.mode, prm.param, rq_lane);
            *reinterpret_cast<int16_t *>(lds + tile_off(b, pz >> 3) + (pz & 7) * 2) = (int16_t)jpegx_clamp_i16(r);
        }
    } else if (flagged) {
        // Every vector load in here waits behind the wave's neighbours' streaming traffic, so: the lane's
        // quantiser entry is fetched once, and the samples of the NEXT flagged block are requested before
        // the cooperative pass of the current one (one memory round trip per wave instead of one per block).
        const double rq_lane = c_rq64.v[lane];
        const int i = lane >> 3, j = lane & 7;
        auto fetch = [&](int b) -> double {
            const int gb = g0 + b;
            const int byb = gb / wb, bxb = gb - byb * wb;
            const float *p = in + ((size_t)(byb * 8 + i) * BS) * pitch + (size_t)(bxb * 8 + j) * BS;
            if (BS == 1) return (double)p[0];
            double s = 0.0;  // np.mean: float64 sum then one division (subsampling.py:11)
#pragma unroll
            for (int u = 0; u < BS; ++u)
#pragma unroll
                for (int w = 0; w < BS; ++w) s += (double)p[(size_t)u * pitch + w];
            return s / (double)(BS * BS);
        };
        double a_next = fetch(__ffsll((long long)flagged) - 1);
        while (flagged) {
            const int b = __ffsll((long long)flagged) - 1;
            flagged &= flagged - 1;
            const double a = a_next;
            if (flagged) a_next = fetch(__ffsll((long long)flagged) - 1);
            const double y = coop_fwd_exact(a, sA, sM, lane);
            const double r = jpegx_quant_lane(y, lane, prm.mode, prm.param, rq_lane);
            const int pz = c_zzinv.v[lane];
            *reinterpret_cast<int16_t *>(lds + tile_off(b, pz >> 3) + (pz & 7) * 2) = (int16_t)jpegx_clamp_i16(r);
        }
    }
    __syncthreads();

    // coalesced write-back: 8 x 1 KiB per wave
    store_tile<NT>(lds, out, g0, nblk, lane);
}

// Register budget (see k_forward_fused_u8 below): the LDS-staged forms hold 10 KiB of LDS (15 waves per CU), so 4
// waves per SIMD is what the registers should allow; 2 x 2 pooling of pixel input carries its 32-register fp16 stash
// and would spill at 3 or 4 (A/B: -DJPEGX_POOLED2_WPE=3), so it keeps the scheduler's own choice.
#ifndef JPEGX_POOLED2_WPE
#define JPEGX_POOLED2_WPE 2
#endif
constexpr int pooled_waves_per_simd(int lds_bytes, bool stash)
{
    int w = (163840 / lds_bytes) / 4;             // what the LDS footprint admits per SIMD
    w = w < 1 ? 1 : (w > 4 ? 4 : w);
    return (stash && w > JPEGX_POOLED2_WPE) ? JPEGX_POOLED2_WPE : w;
}
template <int VAR, int BS, bool NT, int STAGED>
__global__ __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu(STAGED ? pooled_waves_per_simd(PooledLds<BS, STAGED>::BYTES, BS == 2 && (VAR & 1)) : 2)))
void k_forward_fused(const float *__restrict__ in, size_t pitch, int wb,
                                                      int nblk, QuantParams prm, int16_t *__restrict__ out,
                                                      unsigned long long *counters)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[PooledLds<BS, STAGED>::BYTES];
    forward_fused_body<VAR, BS, NT, STAGED>(lds, blockIdx.x, in, pitch, wb, nblk, prm, out, counters);
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
template <int VAR, bool NT>
__device__ __forceinline__ void forward_strip_body(unsigned char *lds, int wg, const float *__restrict__ in, size_t pitch, int wb,
                                                   int nblk, const QuantParams &prm, int16_t *__restrict__ out,
                                                   unsigned long long *counters)
{
    constexpr bool PIXEL = (VAR & 1) != 0;
    constexpr bool DC_EXACT = (VAR & 2) != 0;
    double *sA = reinterpret_cast<double *>(lds + STRIP_BYTES);
    double *sM = sA + 64;
    int16_t *sP = reinterpret_cast<int16_t *>(lds + STRIP_BYTES + SCRATCH_DOUBLES * 8);

    const int lane = threadIdx.x;
    const int g0 = wg * 64;
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
    jpegx_dct8x8_aan_f32(v);
    if (PIXEL) S = v[0];
    const float E = jpegx_fwd_err_unit(S);

    unsigned pk[32];
    const float worst = quantise_zigzag_pack<PIXEL, DC_EXACT>(v, prm, E, pk);

    unsigned long long flagged = __ballot(valid && !(worst < JPEGX_SAFE_HALF));
    census(counters, flagged, nblk - g0, lane);
    if (prm.tune & 1) flagged = 0;
    const double rq_lane = flagged ? c_rq64.v[lane] : 0.0;   // once per wave, not once per flagged block
    const int pz_lane = flagged ? c_zzinv.v[lane] : 0;
    while (flagged) {   // exact tier, inputs re-read from the strip still resident in LDS
        const int b = __ffsll((long long)flagged) - 1;
        flagged &= flagged - 1;
        const int i = lane >> 3, j = lane & 7;
        const int fb = ((b >> 2) ^ (b >> 3)) & 1;
        const float x = *reinterpret_cast<const float *>(lds + i * 2048 + ((2 * b + ((j >> 2) ^ fb)) << 4) + (j & 3) * 4);
        const double y = coop_fwd_exact((double)x, sA, sM, lane);
        const double r = jpegx_quant_lane(y, lane, prm.mode, prm.param, rq_lane);
        sP[pz_lane] = (int16_t)jpegx_clamp_i16(r);
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

template <int VAR, bool NT>
__global__ __launch_bounds__(64) void k_forward_fused_strip(const float *__restrict__ in, size_t pitch, int wb,
                                                            int nblk, QuantParams prm, int16_t *__restrict__ out,
                                                            unsigned long long *counters)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[STRIP_LDS_BYTES];
    int wg = blockIdx.x;
    if (prm.tune & 2) {   // XCD-private order (jpegx_device.h: xcd_private_wg)
        wg = xcd_private_wg(blockIdx.x, (nblk + 63) >> 6, (prm.tune >> 8) & 31);
        if (wg >= ((nblk + 63) >> 6)) return;
    }
    forward_strip_body<VAR, NT>(lds, wg, in, pitch, wb, nblk, prm, out, counters);
}

// ------------------------------------------------------------------------------------------------
// The strip kernel with a COLUMN-WISE exact tier, for quantisers fine enough that many blocks hold an
// unsafe coefficient (divisors of a few units: 14-33 % of the blocks) but not so many that the all-float64
// kernel wins.  Same staging, fast tier and write-out as k_forward_fused_strip; what differs:
//   * the fast tier knows per COLUMN of a block whether one of its coefficients is unsafe
//     (quantise_zigzag_pack_cols), so the unit of exact work is one column (b, l):
//     M[i][l] = C[l] . A[i] for the 8 rows, then Y[k][l] = C[k] . M[:, l] for the 8 frequencies -- 16 dots of
//     length 8 in the reference's order instead of the block's 128;
//   * eight units per pass, lane (s, t) serving unit s: row t in the first half, frequency t in the second;
//   * the tier's tables (FwdExactTab) ride into LDS with the strip by a seventeenth DMA instruction and the
//     samples come from the strip: no vector memory load;
//   * a unit's 8 exact coefficients travel to the block's owner lane through LDS together with their zigzag
//     positions; the owner holds up to two units in registers and writes them over its 128 B of the output
//     tile after parking it (a third unit of the same block first flushes the oldest into the packed registers).
// ------------------------------------------------------------------------------------------------
constexpr int CU_TAB = STRIP_BYTES;                 // FwdExactTab: C (512 B), luminance table, inverse zigzag
constexpr int CU_M = CU_TAB + 1024;                 // 8 slots x 8 doubles
constexpr int CU_M_STRIDE = 64;
constexpr int CU_RES = CU_M + 8 * CU_M_STRIDE;      // 8 slots x 8 x {int16 value, uint16 zigzag position}
constexpr int CU_LDS_BYTES = CU_RES + 8 * 32;       // 18176: nine waves per CU, like the plain strip kernel
static_assert(CU_LDS_BYTES <= 163840 / 9, "nine waves per CU");

// write one exact column (8 x {value, position}) into the packed coefficient registers (rare: third unit of a block)
__device__ __forceinline__ void flush_unit_into_pk(unsigned (&pk)[32], const unsigned (&unit)[8])
{
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned val = unit[k] & 0xFFFFu, pos = unit[k] >> 16;
#pragma unroll
        for (int w = 0; w < 32; ++w) {
            pk[w] = (pos == 2u * w) ? ((pk[w] & 0xFFFF0000u) | val) : pk[w];
            pk[w] = (pos == 2u * w + 1u) ? ((pk[w] & 0x0000FFFFu) | (val << 16)) : pk[w];
        }
    }
}

template <int VAR, bool NT>
__global__ __launch_bounds__(64) void k_forward_fused_strip_cols(const float *__restrict__ in, size_t pitch, int wb,
                                                                 int nblk, QuantParams prm, int16_t *__restrict__ out,
                                                                 unsigned long long *counters)
{
    constexpr bool PIXEL = (VAR & 1) != 0;
    constexpr bool DC_EXACT = (VAR & 2) != 0;
    __shared__ __attribute__((aligned(16))) unsigned char lds[CU_LDS_BYTES];
    const int lane = threadIdx.x;
    int wg = blockIdx.x;
    if (prm.tune & 2) {
        wg = xcd_private_wg(blockIdx.x, (nblk + 63) >> 6, (prm.tune >> 8) & 31);
        if (wg >= ((nblk + 63) >> 6)) return;
    }
    const int g0 = wg * 64;
    const bool valid = g0 + lane < nblk;
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
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const unsigned char *>(&c_fwd_exact) + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + CU_TAB), 16, 0, 0);
    }
    __syncthreads();

    float v[64];
    const int f = ((lane >> 2) ^ (lane >> 3)) & 1;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(lds + r * 2048 + ((2 * lane + f) << 4));
        const f32x4 hi = *reinterpret_cast<const f32x4 *>(lds + r * 2048 + ((2 * lane + (f ^ 1)) << 4));
        v[r * 8 + 0] = lo.x; v[r * 8 + 1] = lo.y; v[r * 8 + 2] = lo.z; v[r * 8 + 3] = lo.w;
        v[r * 8 + 4] = hi.x; v[r * 8 + 5] = hi.y; v[r * 8 + 6] = hi.z; v[r * 8 + 7] = hi.w;
    }
    float S = 0.f;
    if (!PIXEL) {
#pragma unroll
        for (int n = 0; n < 64; ++n) S += fabsf(v[n]);
    }
    jpegx_dct8x8_aan_f32(v);
    if (PIXEL) S = v[0];
    const float E = jpegx_fwd_err_unit(S);
    unsigned pk[32];
    unsigned colmask = quantise_zigzag_pack_cols<PIXEL, DC_EXACT>(v, prm, E, pk);
    if (!valid) colmask = 0;
    const unsigned long long flagged = __ballot(colmask != 0);
    census(counters, flagged, nblk - g0, lane);
    if (prm.tune & 1) colmask = 0;

    // up to two exact columns held by the owner lane: 8 x {int16 value | zigzag position << 16} each
    unsigned held0[8], held1[8];
    int nheld = 0;
    if (flagged && !(prm.tune & 1)) {
        const int s = lane >> 3, t = lane & 7;
        const unsigned char *tab = lds + CU_TAB;
        for (;;) {
            unsigned long long pend = __ballot(colmask != 0);
            if (!pend) break;
            int b = -1, mine = -1;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int bk = pend ? __ffsll((long long)pend) - 1 : -1;
                pend &= pend - 1;
                if (k == s) b = bk;
                if (bk == lane) mine = k;
            }
            const int mycol = __ffs((int)colmask) - 1;
            const int l = __shfl(mycol, b < 0 ? lane : b);       // the unit's column, from its owner
            if (mine >= 0) colmask &= colmask - 1;
            if (b >= 0) {
                // M[t][l] = C[l] . A[t] (row pass, transforms.py:46-58), samples of block b's row t from the strip
                const int fb = ((b >> 2) ^ (b >> 3)) & 1;
                const f32x4 lo = *reinterpret_cast<const f32x4 *>(lds + t * 2048 + ((2 * b + fb) << 4));
                const f32x4 hi = *reinterpret_cast<const f32x4 *>(lds + t * 2048 + ((2 * b + (fb ^ 1)) << 4));
                const double x[8] = {(double)lo.x, (double)lo.y, (double)lo.z, (double)lo.w, (double)hi.x, (double)hi.y, (double)hi.z, (double)hi.w};
                double w[8];
#pragma unroll
                for (int n = 0; n < 8; n += 2) {
                    const double2 q = *reinterpret_cast<const double2 *>(tab + l * 64 + n * 8);
                    w[n] = q.x; w[n + 1] = q.y;
                }
                *reinterpret_cast<double *>(lds + CU_M + s * CU_M_STRIDE + t * 8) = jpegx_dot8_ref(w, x, 1);
            }
            __syncthreads();
            if (b >= 0) {
                // Y[t][l] = C[t] . M[:, l] (column pass), the reference's float64 quantiser, np.round
                double m[8], w[8];
#pragma unroll
                for (int n = 0; n < 8; ++n) m[n] = *reinterpret_cast<const double *>(lds + CU_M + s * CU_M_STRIDE + n * 8);
#pragma unroll
                for (int n = 0; n < 8; n += 2) {
                    const double2 q = *reinterpret_cast<const double2 *>(tab + t * 64 + n * 8);
                    w[n] = q.x; w[n + 1] = q.y;
                }
                const double y = jpegx_dot8_ref(w, m, 1);
                const int n = t * 8 + l;
                const double rq = 1.0 / (double)tab[512 + n];               // quantizers.py:49 "1.0 / q": the same quotient
                const int val = jpegx_clamp_i16(jpegx_quant_lane(y, n, prm.mode, prm.param, rq));
                const unsigned pos = tab[576 + n];
                *reinterpret_cast<unsigned *>(lds + CU_RES + s * 32 + t * 4) = ((unsigned)val & 0xFFFFu) | (pos << 16);
            }
            __syncthreads();
            if (__any(mine >= 0 && nheld == 2)) {               // a third column of one block (rare): make room
                if (mine >= 0 && nheld == 2) {
                    flush_unit_into_pk(pk, held0);
#pragma unroll
                    for (int k = 0; k < 8; ++k) held0[k] = held1[k];
                    nheld = 1;
                }
            }
            if (mine >= 0) {
                const u32x4 a = *reinterpret_cast<const u32x4 *>(lds + CU_RES + mine * 32);
                const u32x4 c = *reinterpret_cast<const u32x4 *>(lds + CU_RES + mine * 32 + 16);
                const unsigned u[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (nheld == 0) held0[k] = u[k]; else held1[k] = u[k];
                }
                ++nheld;
            }
            // the next pass rewrites CU_M only after its own reads of CU_RES ... no hazard: CU_M is read before the
            // second barrier of this pass and rewritten after it; CU_RES is read here and rewritten only after the
            // next pass's first barrier
        }
    }

    // the strip is dead: its first 8 KiB become the swizzled output tile; exact columns go over the lane's row
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 8; ++c)
        *reinterpret_cast<u32x4 *>(lds + tile_off(lane, c)) = u32x4{pk[c * 4 + 0], pk[c * 4 + 1], pk[c * 4 + 2], pk[c * 4 + 3]};
    if (nheld > 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned pos = held0[k] >> 16;
            *reinterpret_cast<int16_t *>(lds + tile_off(lane, pos >> 3) + (pos & 7) * 2) = (int16_t)(held0[k] & 0xFFFFu);
        }
    }
    if (nheld > 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned pos = held1[k] >> 16;
            *reinterpret_cast<int16_t *>(lds + tile_off(lane, pos >> 3) + (pos & 7) * 2) = (int16_t)(held1[k] & 0xFFFFu);
        }
    }
    __syncthreads();
    store_tile<NT>(lds, out, g0, nblk, lane);
}

// ------------------------------------------------------------------------------------------------
// several planes in ONE grid (BASELINE.json configs[2]: Y + Cb + Cr of a 4:2:0 image).  One launch per
// plane leaves the two 4096-workgroup chroma launches as a single generation of long-lived waves with
// nothing to overlap their DMA phases with; as one grid the chroma workgroups are dispatched FIRST
// (they live ~4x longer: 8 LDS-DMA phases) and the Y workgroups fill in behind them, so the chroma
// tail overlaps the Y body.  Every workgroup finds its plane from the by-value descriptor table
// (uniform scalar compare) and runs that plane's body: strip (bs = 1) or LDS-staged mean-pool (bs = 2, 4).
// ------------------------------------------------------------------------------------------------
constexpr int MAX_PLANES = JPEGX_MAX_PLANES;
struct PlaneArgs {
    const float *in;
    int16_t *out;
    size_t pitch;
    int wb, nblk, bs;
    int wg0;            // first workgroup of this plane in the grid
};
struct PlaneTable {
    PlaneArgs p[MAX_PLANES];
    int n;
};

template <int VAR, bool NT>
__global__ __launch_bounds__(64) void k_forward_fused_planes(PlaneTable tab, QuantParams prm, unsigned long long *counters)
{
    constexpr int LDSB = STRIP_LDS_BYTES > PooledLds<4, 1>::BYTES ? STRIP_LDS_BYTES : PooledLds<4, 1>::BYTES;
    static_assert(LDSB >= PooledLds<2, 2>::BYTES, "LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDSB];
    int i = 0;
#pragma unroll
    for (int k = 1; k < MAX_PLANES; ++k)
        if (k < tab.n && (int)blockIdx.x >= tab.p[k].wg0) i = k;
    i = __builtin_amdgcn_readfirstlane(i);
    const PlaneArgs a = tab.p[i];     // uniform index into the kernarg segment: one scalar load
    const int wg = (int)blockIdx.x - a.wg0;
    if (a.bs == 1)
        forward_strip_body<VAR, NT>(lds, wg, a.in, a.pitch, a.wb, a.nblk, prm, a.out, counters);
    else if (a.bs == 2)
        forward_fused_body<VAR, 2, NT, 2>(lds, wg, a.in, a.pitch, a.wb, a.nblk, prm, a.out, counters);
    else
        forward_fused_body<VAR, 4, NT, 1>(lds, wg, a.in, a.pitch, a.wb, a.nblk, prm, a.out, counters);
}

// ------------------------------------------------------------------------------------------------
// fused forward, ALL-FLOAT64 variant.  With quantisers whose step is close to 1 (mode 'none', small
// divisors, wide 'discard' windows) the rounding margin E/q of the fp32 tier is so wide that most
// blocks would fall back to the cooperative exact tier (80 % of noise blocks for 'none': 5 Gblocks/s).
// For those the whole block is computed per lane in float64 in the reference's operation order
// straight away -- 1408 fp64 flops per block, no flags, no second tier -- which is bit-exact by
// construction.  Same LDS-DMA strip staging and tile write-out as the default kernel.  Chosen by
// the launcher from the expected exact-tier share (jpegx_forward_fused_pooled).
// ------------------------------------------------------------------------------------------------
template <bool NT>
__global__ __launch_bounds__(64) void k_forward_fused_strip_f64(const float *__restrict__ in, size_t pitch, int wb,
                                                                int nblk, QuantParams prm, int16_t *__restrict__ out,
                                                                unsigned long long *counters)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[STRIP_BYTES];
    const int lane = threadIdx.x;
    const int g0 = blockIdx.x * 64;
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
    __syncthreads();

    // The block stays in registers as fp32 (64 VGPRs); for each output column l the row pass
    // M[i][l] = C[l] . A[i] (rows first, transforms.py:46-58) is evaluated for the 8 rows, then the
    // column pass Y[k][l] = C[k] . M[:, l], quantised on the fly (quantizers.py:4-49) and packed at
    // its zigzag position.  Keeping only one column of M live (16 VGPRs instead of 128) costs 7 extra
    // float->double conversions per sample but lifts occupancy from 1 to 3 waves per SIMD.
    float a[64];
    const int f = ((lane >> 2) ^ (lane >> 3)) & 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(lds + i * 2048 + ((2 * lane + f) << 4));
        const f32x4 hi = *reinterpret_cast<const f32x4 *>(lds + i * 2048 + ((2 * lane + (f ^ 1)) << 4));
        a[i * 8 + 0] = lo.x; a[i * 8 + 1] = lo.y; a[i * 8 + 2] = lo.z; a[i * 8 + 3] = lo.w;
        a[i * 8 + 4] = hi.x; a[i * 8 + 5] = hi.y; a[i * 8 + 6] = hi.z; a[i * 8 + 7] = hi.w;
    }
    unsigned pk[32];
#pragma unroll
    for (int w = 0; w < 32; ++w) pk[w] = 0u;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        double m[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // opaque to the optimiser: otherwise the 64 conversions are hoisted out of the column loop
            // and all 64 doubles stay live (256 VGPRs, 1 wave per SIMD)
            asm volatile("" : "+v"(a[i * 8 + 0]), "+v"(a[i * 8 + 1]), "+v"(a[i * 8 + 2]), "+v"(a[i * 8 + 3]),
                              "+v"(a[i * 8 + 4]), "+v"(a[i * 8 + 5]), "+v"(a[i * 8 + 6]), "+v"(a[i * 8 + 7]));
            const double x[8] = {(double)a[i * 8 + 0], (double)a[i * 8 + 1], (double)a[i * 8 + 2], (double)a[i * 8 + 3],
                                 (double)a[i * 8 + 4], (double)a[i * 8 + 5], (double)a[i * 8 + 6], (double)a[i * 8 + 7]};
            m[i] = jpegx_dot8_ref(&c_dct[l * 8], x, 1);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int n = k * 8 + l;
            const double y = jpegx_dot8_ref(&c_dct[k * 8], m, 1);
            const int q = jpegx_clamp_i16(jpegx_quant_ref(y, n, prm.mode, prm.param, c_rq64.v));
            constexpr I64 zi = make_zzinv();
            const int p = zi.v[n];
            pk[p >> 1] |= ((unsigned)q & 0xFFFFu) << (16 * (p & 1));
        }
        asm volatile("" : "+v"(pk[0]) : : "memory");   // keep the columns in program order (register pressure)
    }
    census(counters, 0ull, nblk - g0, lane);

    __syncthreads();    // the strip is dead: reuse its first 8 KiB as the swizzled output tile
#pragma unroll
    for (int c = 0; c < 8; ++c)
        *reinterpret_cast<u32x4 *>(lds + tile_off(lane, c)) = u32x4{pk[c * 4 + 0], pk[c * 4 + 1], pk[c * 4 + 2], pk[c * 4 + 3]};
    __syncthreads();
    store_tile<NT>(lds, out, g0, nblk, lane);
}

// ------------------------------------------------------------------------------------------------
// all-float64 forward, EIGHT LANES PER BLOCK.  The lane-per-block form above keeps a whole block per lane
// (~200 VGPRs, two waves per SIMD) and its fp64 pipe idles half the time waiting on itself.  Here a wave
// works on 8 blocks at a time and a lane owns one ROW of its block in the row pass (8 conversions, 8 dot
// products M[i][0..7]), hands the row to the block's other lanes through LDS, and owns one COLUMN in the
// column pass (8 dot products Y[0..7][l], quantised and dropped at their zigzag positions in a 1 KiB LDS
// tile that leaves as one 16-byte store per lane).  Every product and sum is the same float64 operation
// in the same order as before (jpegx_dot8_ref = transforms.py:36-38 through OpenBLAS) -- only which lane
// does it changes -- so the results stay bit-exact by construction, at ~40 VGPRs and 4 waves per SIMD.
// ------------------------------------------------------------------------------------------------
#ifndef JPEGX_F8_BLOCKS
#define JPEGX_F8_BLOCKS 16
#endif
constexpr int F8_BLOCKS = JPEGX_F8_BLOCKS;                     // blocks per workgroup (one wave), in passes of 8
constexpr int F8_ROW = F8_BLOCKS * 32;            // bytes of one sample row of the wave's blocks
constexpr int F8_MSTRIDE = 576;                   // M buffer: 64 doubles per block + 64 B so that blocks spread over the banks
constexpr int F8_STRIP = 8 * F8_ROW;               // fp32 input only

struct ZigzagColumns { unsigned long long v[8]; };     // v[k] byte l = zigzag position of coefficient (k, l)
constexpr ZigzagColumns make_zigzag_columns()
{
    ZigzagColumns z{};
    constexpr I64 zi = make_zzinv();
    for (int k = 0; k < 8; ++k)
        for (int l = 0; l < 8; ++l) z.v[k] |= (unsigned long long)zi.v[k * 8 + l] << (8 * l);
    return z;
}

template <typename T, bool NT>
__global__ __launch_bounds__(64) void k_forward_fused_f64x8(const T *__restrict__ in, size_t pitch, int wb, int nblk,
                                                            QuantParams prm, int16_t *__restrict__ out,
                                                            unsigned long long *counters)
{
    constexpr bool F32IN = sizeof(T) == 4;              // fp32 planes are staged by LDS-DMA; float64 rows are read directly
    constexpr int F8_M = F32IN ? F8_STRIP : 0, F8_TILE = F8_M + 8 * F8_MSTRIDE, F8_LDS = F8_TILE + 1024;
    __shared__ __attribute__((aligned(16))) unsigned char lds[F8_LDS];
    const int lane = threadIdx.x;
    const int g0 = blockIdx.x * F8_BLOCKS;
    if constexpr (F32IN) {
        // one LDS-DMA instruction brings two sample rows (2 x F8_ROW = 64 lanes x 16 B); the 16-byte chunk
        // (block c >> 1, half c & 1) of row r sits at position c ^ (r & 1) of its row, which makes the
        // row-pass reads below conflict free
        constexpr int CPR = 2 * F8_BLOCKS, RPI = 64 / CPR;          // chunks per row, rows per instruction
        static_assert(RPI >= 2 && (RPI & 1) == 0, "two or four rows per LDS-DMA instruction");
        const int pos = lane % CPR, rr = lane / CPR;
        const int c = pos ^ (rr & 1);
        const int gb = min(g0 + (c >> 1), nblk - 1);
        const int by = gb / wb, bx = gb - by * wb;
        const float *src = reinterpret_cast<const float *>(in) + ((size_t)by * 8 + rr) * pitch + (size_t)bx * 8 + (c & 1) * 4;
#pragma unroll
        for (int j = 0; j < 8 / RPI; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)(RPI * j) * pitch),
                                             (__attribute__((address_space(3))) void *)(lds + j * RPI * F8_ROW), 16, 0, NT ? 2 : 0);
    }
    const int hi3 = lane >> 3, lo3 = lane & 7;
    // quantiser table entries of this lane's column (mode 'qtable' only; the other modes are uniform)
    double rq[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) rq[k] = 0.0;
    if (prm.mode == JPEGX_QM_QTABLE) {
#pragma unroll
        for (int k = 0; k < 8; ++k) rq[k] = c_rq64.v[k * 8 + lo3];
    }
    if (counters != nullptr && lane == 0) atomicAdd(&counters[1], (unsigned long long)min(F8_BLOCKS, nblk - g0));
    __syncthreads();

#pragma unroll 1
    for (int t = 0; t < F8_BLOCKS / 8; ++t) {
        {
            // row pass: lane = (row i = hi3, block b = lo3 of this pass)
            const int i = hi3, b = lo3, c2 = 2 * (8 * t + b);
            double x[8];
            if constexpr (F32IN) {
                const unsigned char *row = lds + i * F8_ROW;
                const f32x4 lo = *reinterpret_cast<const f32x4 *>(row + (((c2 + 0) ^ (i & 1)) << 4));
                const f32x4 hi = *reinterpret_cast<const f32x4 *>(row + (((c2 + 1) ^ (i & 1)) << 4));
                x[0] = (double)lo.x; x[1] = (double)lo.y; x[2] = (double)lo.z; x[3] = (double)lo.w;
                x[4] = (double)hi.x; x[5] = (double)hi.y; x[6] = (double)hi.z; x[7] = (double)hi.w;
            } else {
                const int gb = min(g0 + 8 * t + b, nblk - 1);
                const int by = gb / wb, bx = gb - by * wb;
                const double *src = reinterpret_cast<const double *>(in) + ((size_t)by * 8 + i) * pitch + (size_t)bx * 8;
#pragma unroll
                for (int n = 0; n < 8; n += 2) {
                    const double2 t2 = *reinterpret_cast<const double2 *>(src + n);
                    x[n] = t2.x; x[n + 1] = t2.y;
                }
            }
            unsigned char *mb = lds + F8_M + b * F8_MSTRIDE + (i & 1) * 8;
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                const double m = jpegx_dot8_ref(&c_dct[l * 8], x, 1);                        // M[i][l] = C[l] . A[i]
                const int f = (l >> 2) | ((b & 1) << 1);
                *reinterpret_cast<double *>(mb + l * 64 + (((i >> 1) ^ f) << 4)) = m;
            }
        }
        __syncthreads();
        {
            // column pass: lane = (block b = hi3 of this pass, column l = lo3)
            const int b = hi3, l = lo3;
            const int f = (l >> 2) | ((b & 1) << 1);
            const unsigned char *mb = lds + F8_M + b * F8_MSTRIDE + l * 64;
            double m[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double2 t2 = *reinterpret_cast<const double2 *>(mb + ((k ^ f) << 4));
                m[2 * k] = t2.x; m[2 * k + 1] = t2.y;
            }
            unsigned char *tile = lds + F8_TILE + b * 128;
            constexpr ZigzagColumns zc = make_zigzag_columns();
            double y[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) y[k] = jpegx_dot8_ref(&c_dct[k * 8], m, 1);          // Y[k][l] = C[k] . M[:, l]
            // the quantiser (quantizers.py:4-49, = jpegx_quant_lane) once per column, outside the unrolled loops,
            // so that the eight dot products above are one straight-line stretch of independent float64 chains
            if (prm.mode == JPEGX_QM_QTABLE) {
#pragma unroll
                for (int k = 0; k < 8; ++k) y[k] = rint(y[k] * rq[k]);
            } else if (prm.mode == JPEGX_QM_DIVIDE) {
                // a / d (true division, quantizers.py:27-28).  For d = +-2^e both a / d and a * (1 / d) are the
                // correctly rounded value of the same real number, so the multiply is bit-identical and 10x cheaper
                const unsigned long long pb = (unsigned long long)__double_as_longlong(prm.param);
                const int ex = (int)((pb >> 52) & 0x7FFu);
                if ((pb & 0xFFFFFFFFFFFFFull) == 0 && ex > 123 && ex < 1923) {
                    const double r = 1.0 / prm.param;
#pragma unroll
                    for (int k = 0; k < 8; ++k) y[k] = rint(y[k] * r);
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) y[k] = rint(y[k] / prm.param);
                }
            } else if (prm.mode == JPEGX_QM_DISCARD) {
                const int keep = (int)prm.param;
#pragma unroll
                for (int k = 0; k < 8; ++k) y[k] = (k >= keep || l >= keep) ? 0.0 : rint(y[k]);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) y[k] = rint(y[k]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                // jpegx_clamp_i16 with the hardware's saturating conversion (v_cvt_i32_f64 clamps to int32, NaN -> 0)
                // and one integer median instead of two float64 compares and three selects
                int q;
                asm("v_cvt_i32_f64 %0, %1" : "=v"(q) : "v"(y[k]));
                q = min(max(q, -32768), 32767);
                const unsigned p = (unsigned)(zc.v[k] >> (8 * l)) & 63u;
                *reinterpret_cast<int16_t *>(tile + p * 2) = (int16_t)q;
            }
        }
        __syncthreads();
        {
            const u32x4 q = *reinterpret_cast<const u32x4 *>(lds + F8_TILE + lane * 16);
            const int gq = g0 + 8 * t;
            if (gq + hi3 < nblk) st_u32x4<NT>(reinterpret_cast<unsigned char *>(out) + (size_t)gq * 128 + lane * 16, q);
        }
        // the next pass's row pass writes M (read above, before the barrier) and its column pass writes the
        // tile only after the next barrier, which every lane reaches after its tile read: no further barrier
    }
}

// ------------------------------------------------------------------------------------------------
// fused forward on a FLOAT64 plane: what step 4 receives when the samples are not exact in fp32 -- the
// means of SubSampling with block_size 3, 5, 6 ... (pipeline/subsampling.py:9-11: k/9, k/25 ...) or any
// float64 band a caller passes to BasisChange.  Lane-per-block, everything in float64 in the reference's
// operation order (transforms.py:46-58 via jpegx_dot8_ref): row pass in place, then per output column
// the column pass, the reference's own quantiser (quantizers.py:4-49) and the zigzag packing.
// ------------------------------------------------------------------------------------------------
template <bool NT>
__global__ __launch_bounds__(64) void k_forward_fused_f64in(const double *__restrict__ in, size_t pitch, int wb, int nblk,
                                                            QuantParams prm, int16_t *__restrict__ out,
                                                            unsigned long long *counters)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[TILE_BYTES];
    const int lane = threadIdx.x;
    const int g0 = blockIdx.x * 64;
    const int gb = min(g0 + lane, nblk - 1);
    const int by = gb / wb, bx = gb - by * wb;
    const double *src = in + (size_t)by * 8 * pitch + (size_t)bx * 8;
    double a[64];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        double x[8];
#pragma unroll
        for (int n = 0; n < 8; n += 2) {
            const double2 t = *reinterpret_cast<const double2 *>(src + (size_t)i * pitch + n);
            x[n] = t.x; x[n + 1] = t.y;
        }
#pragma unroll
        for (int l = 0; l < 8; ++l) a[i * 8 + l] = jpegx_dot8_ref(&c_dct[l * 8], x, 1);      // M[i][l] = C[l] . A[i]
    }
    unsigned pk[32];
#pragma unroll
    for (int w = 0; w < 32; ++w) pk[w] = 0u;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int n = k * 8 + l;
            const double y = jpegx_dot8_ref(&c_dct[k * 8], &a[l], 8);                          // Y[k][l] = C[k] . M[:, l]
            const int q = jpegx_clamp_i16(jpegx_quant_ref(y, n, prm.mode, prm.param, c_rq64.v));
            constexpr I64 zi = make_zzinv();
            const int p = zi.v[n];
            pk[p >> 1] |= ((unsigned)q & 0xFFFFu) << (16 * (p & 1));
        }
    }
    census(counters, 0ull, nblk - g0, lane);
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
// Register budget: the scheduler trades registers for latency hiding up to what it believes the occupancy
// allows; told nothing it lets these kernels drift past the 128 / 168 registers of 4 / 3 waves per SIMD and
// loses a wave.  BS = 1 (9.3 KiB of LDS): 4 waves per SIMD; BS = 2, 4 (18 KiB): 2 by LDS, so up to 3 is free.
#ifdef JPEGX_U8_WPE   // A/B builds (microbench/build_variant.sh): another budget
#define JPEGX_U8_OCC __attribute__((amdgpu_waves_per_eu(JPEGX_U8_WPE, JPEGX_U8_WPE)))
#else
#define JPEGX_U8_OCC __attribute__((amdgpu_waves_per_eu(BS == 1 ? 5 : 2)))
#endif
// SIZES: the entropy stage's byte count of every block, from the registers (rle_block_bytes): block_bytes[block] and
// the wave's total (bit 31: an amplitude beyond 15 bits in the wave) -> wave_bytes[wave]; k_rle_sizes then has nothing
// left to read the stream for.
template <bool DC_EXACT, int BS, bool NT, bool SIZES = false>
__global__ __launch_bounds__(64) JPEGX_U8_OCC void k_forward_fused_u8(const unsigned char *__restrict__ in, size_t pitch, int wb,
                                                         int nblk, QuantParams prm, int16_t *__restrict__ out,
                                                         unsigned long long *counters, unsigned *__restrict__ block_bytes = nullptr,
                                                         unsigned *__restrict__ wave_bytes = nullptr, unsigned *__restrict__ half_info = nullptr)
{
    constexpr int ROWS = 8 * BS;                       // input rows of the wave's blocks
    constexpr int ROW_BYTES = 64 * 8 * BS;             // bytes of one input row in LDS (64 blocks)
    // 4 KiB (BS=1) / 16 KiB (BS=2) hold all rows at once; BS=4 (64 KiB) is streamed in 4 phases of
    // 8 input rows = 16 KiB, laid out exactly like the fp32 strip (2 KiB rows, strip_swz chunks)
    constexpr int IN_BYTES = (BS == 4) ? STRIP_BYTES : ROWS * ROW_BYTES;
    constexpr int FRONT = IN_BYTES > TILE_BYTES ? IN_BYTES : TILE_BYTES;
    // behind the cooperative scratch and the 128-byte patch area: the exact tier's table (C, the luminance table,
    // the inverse zigzag order -- one LDS-DMA instruction) and, for BS = 4, 128 bytes through which the owner of a
    // flagged block hands its tile sums to the wave: the tier issues NO vector memory load (under the streaming
    // load such a load waits for microseconds; k_inverse_fused has the measurement)
    // (round 2 kept the constant-memory tables for BS = 1 because the extra KiB cost a wave of occupancy; since round 3
    // that variant's LDS is the 8 KiB of its output tile whatever lives inside, so its tier is load-free as well)
    constexpr bool TABBED = true;
    // BS = 1: the input rows fill only the first 4 KiB of the 8 KiB that become the output tile, so the exact tier's
    // scratch and patch area live in the second half (dead before the tile is written): 8 KiB of LDS in all = 20
    // waves per CU, which the 94 registers of this variant allow (5 per SIMD)
    constexpr int SCR = (BS == 1) ? IN_BYTES : FRONT;
    static_assert(BS != 1 || IN_BYTES + SCRATCH_DOUBLES * 8 + 128 + 1024 <= TILE_BYTES, "scratch and table fit behind the rows");
    constexpr int U8_PATCH = SCR + SCRATCH_DOUBLES * 8, U8_TAB = U8_PATCH + 128, U8_XBLK = U8_TAB + (TABBED ? 1024 : 0);
    __shared__ __attribute__((aligned(16))) unsigned char lds[BS == 1 ? TILE_BYTES : U8_XBLK + (BS == 4 ? 128 : 0)];
    double *sA = reinterpret_cast<double *>(lds + SCR);
    double *sM = sA + 64;
    int16_t *sP = reinterpret_cast<int16_t *>(lds + U8_PATCH);

    const int lane = threadIdx.x;
    const int g0 = blockIdx.x * 64;
    const bool valid = g0 + lane < nblk;
    if (TABBED)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const unsigned char *>(&c_fwd_exact) + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds + U8_TAB), 16, 0, 0);

    float v[64];
    unsigned stash[BS == 4 ? 32 : 1];      // BS = 4: the 64 tile sums (<= 4080) as 16-bit pairs, for the exact tier
    if (BS == 4) {
        // SubSampling.execute fused for block_size 4 (the CLI default, compress.py:33): per phase the
        // wave brings in 8 input rows of 2 KiB (lane l of piece j <-> chunk strip_swz(64 j + l) = half
        // (c & 1) of block c >> 1), every lane sums its 4 x 4 byte tiles with integer adds.
        const unsigned char *src[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = strip_swz(64 * j + lane);
            const int gb = min(g0 + (c >> 1), nblk - 1);
            const int by = gb / wb, bx = gb - by * wb;
            src[j] = in + ((size_t)by * 32) * pitch + (size_t)bx * 32 + (c & 1) * 16;
        }
        const int f = ((lane >> 2) ^ (lane >> 3)) & 1;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            __syncthreads();                            // previous phase's LDS reads are done
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const size_t roff = (size_t)(ph * 8 + r) * pitch;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[0] + roff),
                                                 (__attribute__((address_space(3))) void *)(lds + r * 2048), 16, 0, NT ? 2 : 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[1] + roff),
                                                 (__attribute__((address_space(3))) void *)(lds + r * 2048 + 1024), 16, 0, NT ? 2 : 0);
            }
            __syncthreads();                            // the phase has landed
#pragma unroll
            for (int ro = 0; ro < 2; ++ro) {            // two output rows per phase
                unsigned sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int r = ro * 4 + a;
                    const u32x4 lo = *reinterpret_cast<const u32x4 *>(lds + r * 2048 + ((2 * lane + f) << 4));
                    const u32x4 hi = *reinterpret_cast<const u32x4 *>(lds + r * 2048 + ((2 * lane + (f ^ 1)) << 4));
                    const unsigned w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        sum[c] += (w[c] & 0xFFu) + ((w[c] >> 8) & 0xFFu) + ((w[c] >> 16) & 0xFFu) + (w[c] >> 24);
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    v[(ph * 2 + ro) * 8 + c] = (float)sum[c] * 0.0625f;
                    asm volatile("" : "+v"(v[(ph * 2 + ro) * 8 + c]) : : "memory");   // fold now (register pressure)
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    stash[(ph * 2 + ro) * 4 + c] = sum[2 * c] | (sum[2 * c + 1] << 16);
                    asm volatile("" : "+v"(stash[(ph * 2 + ro) * 4 + c]));           // packed here, not at its use
                }
            }
        }
        __syncthreads();
    } else if (BS == 1) {
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

    if (BS == 4) {
        // already pooled above
    } else if (BS == 1) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint2 t = *reinterpret_cast<const uint2 *>(lds + r * 512 + lane * 8);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v[r * 8 + c] = (float)((t.x >> (8 * c)) & 0xFFu);
                v[r * 8 + 4 + c] = (float)((t.y >> (8 * c)) & 0xFFu);
                // one v_cvt_f32_ubyteN each, and opaque from here on: left to itself the compiler moves the first
                // butterfly stage in front of the conversion (v_add/sub_u32_sdwa on the bytes, then v_cvt_f32_i32),
                // which swaps 64 full-rate float adds for 64 SDWA operations that issue at 0.6 of that rate
                // (profiles/r03_valu_rate.txt)
                asm volatile("" : "+v"(v[r * 8 + c]), "+v"(v[r * 8 + 4 + c]));
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

    jpegx_dct8x8_aan_f32(v);
    const float E = jpegx_fwd_err_unit(v[0]);             // pixel input: sum|x| == DC, exact
    unsigned pk[32];
    const float worst = quantise_zigzag_pack<true, DC_EXACT>(v, prm, E, pk);

    unsigned long long flagged = __ballot(valid && !(worst < JPEGX_SAFE_HALF));
    census(counters, flagged, nblk - g0, lane);
    if (prm.tune & 1) flagged = 0;
    if (flagged) {      // exact tier: samples from the rows still resident in LDS (BS = 4: from the owner's tile sums)
        const unsigned char *tab = lds + U8_TAB;
        const double *tabC = reinterpret_cast<const double *>(tab);
        const double rq_lane = TABBED ? 1.0 / (double)tab[512 + lane] : c_rq64.v[lane];      // quantizers.py:49 "1.0 / q", same quotient
        const int pz = TABBED ? (int)tab[576 + lane] : c_zzinv.v[lane];
        const int i = lane >> 3, j = lane & 7;
        while (flagged) {
            const int b = __ffsll((long long)flagged) - 1;
            flagged &= flagged - 1;
            double a;
            if (BS == 1) {
                a = (double)lds[i * 512 + b * 8 + j];
            } else if (BS == 2) {
                const unsigned char *p0 = lds + (2 * i) * 1024 + b * 16 + 2 * j;
                a = ((double)p0[0] + (double)p0[1] + (double)p0[1024] + (double)p0[1025]) / 4.0;   // np.mean
            } else {
                // block_size 4: the rows have left LDS; the owner lane publishes its 64 integer tile sums
                unsigned char *xblk = lds + U8_XBLK;
                if (lane == b) {
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        *reinterpret_cast<u32x4 *>(xblk + c * 16) = u32x4{stash[4 * c], stash[4 * c + 1], stash[4 * c + 2], stash[4 * c + 3]};
                }
                __syncthreads();
                a = (double)*reinterpret_cast<const unsigned short *>(xblk + lane * 2) / 16.0;       // np.mean
            }
            const double y = TABBED ? coop_fwd_exact_tab(a, sA, sM, lane, tabC) : coop_fwd_exact(a, sA, sM, lane);
            const double r = jpegx_quant_lane(y, lane, prm.mode, prm.param, rq_lane);
            sP[pz] = (int16_t)jpegx_clamp_i16(r);
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
    }

    if (SIZES) {
        bool bad;
        unsigned half;
        unsigned bytes = rle_block_bytes(pk, bad, half);
        if (!valid) { bytes = 0; bad = false; }
        if (valid) { block_bytes[g0 + lane] = bytes; half_info[g0 + lane] = half; }
        unsigned sum = bytes;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
        if (lane == 0) wave_bytes[blockIdx.x] = sum | (__any(bad) ? 0x80000000u : 0u);
    }
    if (prm.tune & 4) {
        // A/B (JPEGX_F_TUNE_DIRECT_STORE): every lane stores its own 128 bytes, eight 16-byte pieces at a 128-byte
        // lane stride, default cache policy so that L2 merges the pieces of a line -- no LDS tile, no barriers
        if (valid) {
            unsigned char *dst = reinterpret_cast<unsigned char *>(out) + (size_t)(g0 + lane) * 128;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                st_u32x4<false>(dst + c * 16, u32x4{pk[c * 4 + 0], pk[c * 4 + 1], pk[c * 4 + 2], pk[c * 4 + 3]});
        }
        return;
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

// Expected share of blocks that the fp32 tier would hand to the exact tier, for typical 8-bit content
// (sum|x| ~ 64 * 128) and the average per-coefficient rounding count of pixel input (10.75, jpegx_math.h):
// a coefficient is flagged with probability ~ 2 E / q.  Only steers the choice between kernels that produce
// identical output (measured crossovers, profiles/r02_ab_column_units.txt): whole-block tier below 3 %
// (JPEG table: 2.5 %), column-wise tier up to 40 % (divisors 25 ... 1.5), all-float64 kernel beyond.
double expected_exact_share(const QuantParams &qp)
{
    const double E = 8192.0 * 10.75 * 0x1p-24;
    double keep = 1.0;
    for (int n = 0; n < 64; ++n) {
        const double pflag = 2.0 * E * fabs((double)qp.rq32[n]);
        keep *= pflag < 1.0 ? 1.0 - pflag : 0.0;
    }
    return 1.0 - keep;
}

float max_abs_multiplier(const QuantParams &qp)
{
    float m = 0.f;
    for (int n = 0; n < 64; ++n) m = fmaxf(m, fabsf(qp.rq32[n]));
    return m;
}

template <int BS, bool NT, int STAGED = 0>
int launch_forward(const float *d_in, int H, int W, ptrdiff_t pitch, const QuantParams &qp, unsigned flags,
                   int16_t *d_out, hipStream_t st)
{
    const int wb = W / 8, nblk = (H / 8) * wb;
    const dim3 grid((nblk + 63) / 64), block(64);
    // The PIXEL variants pack without saturating: |coefficient| <= 16320 for 8-bit content, so the
    // quantised value fits int16 only while the multiplier is at most 2 (a divisor below 0.5 goes
    // through the generic, saturating variant; the reference raises BadRleCodeError on such amplitudes
    // and so does the entropy stage behind this kernel).
    const bool pixel = (flags & JPEGX_F_PIXEL_INPUT) != 0 && max_abs_multiplier(qp) <= 2.0f;
    // DC is an exact integer multiple of 2^-8 and rq[0] a power of two -> DC/q needs no tie check
    const bool dc_exact = pixel && is_pow2_float(qp.rq32[0]) &&
                          (qp.mode != JPEGX_Q_DIVIDE || (double)qp.rq32[0] * qp.param == 1.0);
    // the all-float64 kernel costs the same whatever the data (0.35 ms per 2^22 blocks; 0.44 ms when it has to
    // divide by something that is not a power of two): measured crossovers against the column-wise tier
    // (profiles/r02_ab_f64_eight_lanes.txt) at an expected share of 25 % / 40 %
    const bool cheap_f64 = qp.mode != JPEGX_Q_DIVIDE || is_pow2_float((float)qp.param);
    if (BS == 1 && !(flags & (JPEGX_F_TUNE_WAVE_PER_BLOCK | JPEGX_F_TUNE_NO_STRIP | JPEGX_F_TUNE_NO_F64_KERNEL)) &&
        ((flags & JPEGX_F_TUNE_F64_KERNEL) || expected_exact_share(qp) > (cheap_f64 ? 0.25 : 0.40))) {
        if (flags & JPEGX_F_TUNE_F64_LANE_PER_BLOCK)
            hipLaunchKernelGGL((k_forward_fused_strip_f64<NT>), grid, block, 0, st, d_in, (size_t)pitch, wb, nblk, qp, d_out, g_counters);
        else
            hipLaunchKernelGGL((k_forward_fused_f64x8<float, NT>), dim3((nblk + F8_BLOCKS - 1) / F8_BLOCKS), block, 0, st, d_in, (size_t)pitch, wb,
                               nblk, qp, d_out, g_counters);
    } else if (BS == 1 && (flags & JPEGX_F_TUNE_WAVE_PER_BLOCK)) {
        if (dc_exact)
            hipLaunchKernelGGL((k_forward_fused_wpb<3, NT>), grid, block, 0, st, d_in, (size_t)pitch, wb, nblk, qp, d_out, g_counters);
        else if (pixel)
            hipLaunchKernelGGL((k_forward_fused_wpb<1, NT>), grid, block, 0, st, d_in, (size_t)pitch, wb, nblk, qp, d_out, g_counters);
        else
            hipLaunchKernelGGL((k_forward_fused_wpb<0, NT>), grid, block, 0, st, d_in, (size_t)pitch, wb, nblk, qp, d_out, g_counters);
    } else if (BS == 1 && !(flags & JPEGX_F_TUNE_NO_STRIP)) {
        QuantParams q2 = qp;
        dim3 g2 = grid;
        q2.tune |= xcd_order_setup(flags, nblk, &g2);
        const bool cols = (flags & JPEGX_F_TUNE_COLUMN_UNITS) || (expected_exact_share(qp) > 0.03 && !(flags & JPEGX_F_TUNE_NO_COLUMN_UNITS));
        scale_for_aan(&q2);                         // the fast tier works on the scaled (AAN) coefficients
        // column-wise exact tier once more than ~3 % of the blocks are expected to be flagged
        if (cols) {
            if (dc_exact)
                hipLaunchKernelGGL((k_forward_fused_strip_cols<3, NT>), g2, block, 0, st, d_in, (size_t)pitch, wb, nblk, q2, d_out, g_counters);
            else if (pixel)
                hipLaunchKernelGGL((k_forward_fused_strip_cols<1, NT>), g2, block, 0, st, d_in, (size_t)pitch, wb, nblk, q2, d_out, g_counters);
            else
                hipLaunchKernelGGL((k_forward_fused_strip_cols<0, NT>), g2, block, 0, st, d_in, (size_t)pitch, wb, nblk, q2, d_out, g_counters);
            HIP_TRY(hipGetLastError());
            return JPEGX_OK;
        }
        if (dc_exact)
            hipLaunchKernelGGL((k_forward_fused_strip<3, NT>), g2, block, 0, st, d_in, (size_t)pitch, wb, nblk, q2, d_out, g_counters);
        else if (pixel)
            hipLaunchKernelGGL((k_forward_fused_strip<1, NT>), g2, block, 0, st, d_in, (size_t)pitch, wb, nblk, q2, d_out, g_counters);
        else
            hipLaunchKernelGGL((k_forward_fused_strip<0, NT>), g2, block, 0, st, d_in, (size_t)pitch, wb, nblk, q2, d_out, g_counters);
    } else {
        QuantParams qa = qp;
        scale_for_aan(&qa);
        if (dc_exact)
            hipLaunchKernelGGL((k_forward_fused<3, BS, NT, STAGED>), grid, block, 0, st, d_in, (size_t)pitch, wb, nblk, qa, d_out, g_counters);
        else if (pixel)
            hipLaunchKernelGGL((k_forward_fused<1, BS, NT, STAGED>), grid, block, 0, st, d_in, (size_t)pitch, wb, nblk, qa, d_out, g_counters);
        else
            hipLaunchKernelGGL((k_forward_fused<0, BS, NT, STAGED>), grid, block, 0, st, d_in, (size_t)pitch, wb, nblk, qa, d_out, g_counters);
    }
    HIP_TRY(hipGetLastError());
    return JPEGX_OK;
}
}  // namespace

extern "C" {

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

int jpegx_forward_fused_planes(const jpegx_plane_desc *planes, int nplanes, int mode, double param, unsigned flags,
                               jpegx_stream_t stream)
{
    if (!planes || nplanes < 1 || nplanes > MAX_PLANES) return fail(JPEGX_E_INVALID, "forward_planes: 1..JPEGX_MAX_PLANES plane descriptors");
    QuantParams qp;
    int rc = fill_forward_params(mode, param, &qp);
    if (rc) return rc;
    if (flags & JPEGX_F_TUNE_SKIP_EXACT) qp.tune |= 1;
    // pooled planes first: their workgroups live longest, the bs = 1 workgroups fill in behind them
    int order[MAX_PLANES], n = 0;
    for (int pass = 0; pass < 3; ++pass)
        for (int i = 0; i < nplanes; ++i) {
            const int bs = planes[i].bs;
            if (bs != 1 && bs != 2 && bs != 4) return fail(JPEGX_E_UNSUPPORTED, "fused mean-pool supports block_size 1, 2 and 4");
            if ((pass == 0 && bs == 4) || (pass == 1 && bs == 2) || (pass == 2 && bs == 1)) order[n++] = i;
        }
    PlaneTable tab;
    memset(&tab, 0, sizeof(tab));
    long long wg = 0;
    for (int k = 0; k < nplanes; ++k) {
        const jpegx_plane_desc &d = planes[order[k]];
        rc = check_plane(d.d_in, d.d_out, d.H, d.W, d.pitch / d.bs, 1);
        if (rc) return rc;
        if (d.pitch < (ptrdiff_t)d.W * d.bs || (d.pitch % 4) != 0 || !aligned16(d.d_in) || !aligned16(d.d_out))
            return fail(JPEGX_E_INVALID, "forward_planes: pitch must be a multiple of 4 floats and >= W*bs; pointers 16-byte aligned");
        PlaneArgs &a = tab.p[k];
        a.in = d.d_in;
        a.out = d.d_out;
        a.pitch = (size_t)d.pitch;
        a.wb = d.W / 8;
        a.nblk = (d.H / 8) * a.wb;
        a.bs = d.bs;
        a.wg0 = (int)wg;
        wg += (a.nblk + 63) / 64;
        if (wg > 0x7FFFFFFFLL) return fail(JPEGX_E_INVALID, "forward_planes: more than 2^31 workgroups in one launch");
    }
    tab.n = nplanes;
    const bool pixel = (flags & JPEGX_F_PIXEL_INPUT) != 0 && max_abs_multiplier(qp) <= 2.0f;
    const bool dc_exact = pixel && is_pow2_float(qp.rq32[0]) && (qp.mode != JPEGX_Q_DIVIDE || (double)qp.rq32[0] * qp.param == 1.0);
    const bool nt = !(flags & JPEGX_F_TUNE_NO_NT);
    const dim3 grid((unsigned)wg), block(64);
    hipStream_t st = (hipStream_t)stream;
    QuantParams qa = qp;
    scale_for_aan(&qa);
#define JPEGX_LP(VARV) \
    do { if (nt) hipLaunchKernelGGL((k_forward_fused_planes<VARV, true>), grid, block, 0, st, tab, qa, g_counters); \
         else hipLaunchKernelGGL((k_forward_fused_planes<VARV, false>), grid, block, 0, st, tab, qa, g_counters); } while (0)
    if (dc_exact) JPEGX_LP(3); else if (pixel) JPEGX_LP(1); else JPEGX_LP(0);
#undef JPEGX_LP
    HIP_TRY(hipGetLastError());
    return JPEGX_OK;
}

int jpegx_forward_fused_f64(const double *d_in, int H, int W, ptrdiff_t pitch, int mode, double param, unsigned flags,
                            int16_t *d_out, jpegx_stream_t stream)
{
    int rc = check_plane(d_in, d_out, H, W, pitch, 1);
    if (rc) return rc;
    if ((pitch % 2) != 0 || !aligned16(d_in) || !aligned16(d_out))
        return fail(JPEGX_E_INVALID, "forward_f64: pitch must be even (16-byte rows); pointers 16-byte aligned");
    QuantParams qp;
    rc = fill_forward_params(mode, param, &qp);
    if (rc) return rc;
    const int wb = W / 8, nblk = (H / 8) * wb;
    const dim3 grid((nblk + 63) / 64), block(64);
    const dim3 grid8((nblk + F8_BLOCKS - 1) / F8_BLOCKS);
    const bool nt = !(flags & JPEGX_F_TUNE_NO_NT);
    if (flags & JPEGX_F_TUNE_F64_LANE_PER_BLOCK) {
        if (nt) hipLaunchKernelGGL((k_forward_fused_f64in<true>), grid, block, 0, (hipStream_t)stream, d_in, (size_t)pitch, wb, nblk, qp, d_out, g_counters);
        else hipLaunchKernelGGL((k_forward_fused_f64in<false>), grid, block, 0, (hipStream_t)stream, d_in, (size_t)pitch, wb, nblk, qp, d_out, g_counters);
    } else {
        if (nt) hipLaunchKernelGGL((k_forward_fused_f64x8<double, true>), grid8, block, 0, (hipStream_t)stream, d_in, (size_t)pitch, wb, nblk, qp, d_out, g_counters);
        else hipLaunchKernelGGL((k_forward_fused_f64x8<double, false>), grid8, block, 0, (hipStream_t)stream, d_in, (size_t)pitch, wb, nblk, qp, d_out, g_counters);
    }
    HIP_TRY(hipGetLastError());
    return JPEGX_OK;
}

int jpegx_forward_fused(const float *d_in, int H, int W, ptrdiff_t pitch, int mode, double param, unsigned flags,
                        int16_t *d_out, jpegx_stream_t stream)
{
    return jpegx_forward_fused_pooled(d_in, H, W, pitch, 1, mode, param, flags, d_out, stream);
}

static int forward_u8_common(const uint8_t *d_in, int H, int W, ptrdiff_t pitch, int bs, int mode, double param,
                             unsigned flags, int16_t *d_out, unsigned *block_bytes, unsigned *wave_bytes, unsigned *half_info, jpegx_stream_t stream)
{
    if (bs != 1 && bs != 2 && bs != 4) return fail(JPEGX_E_UNSUPPORTED, "uint8 forward supports block_size 1, 2 and 4");
    int rc = check_plane(d_in, d_out, H, W, pitch / bs, 1);
    if (rc) return rc;
    if (pitch < (ptrdiff_t)W * bs || (pitch % 16) != 0 || !aligned16(d_in) || !aligned16(d_out))
        return fail(JPEGX_E_INVALID, "forward_u8: pitch must be a multiple of 16 bytes and >= W*bs; pointers 16-byte aligned");
    if (bs == 1 && (W % 16) != 0) return fail(JPEGX_E_UNSUPPORTED, "forward_u8 with block_size 1 needs W to be a multiple of 16");
    QuantParams qp;
    rc = fill_forward_params(mode, param, &qp);
    if (rc) return rc;
    if (max_abs_multiplier(qp) > 2.0f)   // the uint8 kernels pack without saturating (see launch_forward)
        return fail(JPEGX_E_UNSUPPORTED, "forward_u8: a divisor below 0.5 can overflow int16; use the fp32 entry (it saturates)");
    if (flags & JPEGX_F_TUNE_SKIP_EXACT) qp.tune |= 1;
    if (flags & JPEGX_F_TUNE_DIRECT_STORE) qp.tune |= 4;
    const int wb = W / 8, nblk = (H / 8) * wb;
    const dim3 grid((nblk + 63) / 64), block(64);
    hipStream_t st = (hipStream_t)stream;
    const bool dc_exact = is_pow2_float(qp.rq32[0]) && (qp.mode != JPEGX_Q_DIVIDE || (double)qp.rq32[0] * qp.param == 1.0);
    const bool nt = !(flags & JPEGX_F_TUNE_NO_NT);
    const bool sizes = block_bytes != nullptr;
    QuantParams qa = qp;
    scale_for_aan(&qa);
#define JPEGX_LU8K(DC, BSV, NTV, SZ) \
    hipLaunchKernelGGL((k_forward_fused_u8<DC, BSV, NTV, SZ>), grid, block, 0, st, d_in, (size_t)pitch, wb, nblk, qa, d_out, g_counters, block_bytes, wave_bytes, half_info)
#define JPEGX_LU8(DC, BSV) \
    do { if (sizes) { if (nt) JPEGX_LU8K(DC, BSV, true, true); else JPEGX_LU8K(DC, BSV, false, true); } \
         else { if (nt) JPEGX_LU8K(DC, BSV, true, false); else JPEGX_LU8K(DC, BSV, false, false); } } while (0)
    if (bs == 1) { if (dc_exact) JPEGX_LU8(true, 1); else JPEGX_LU8(false, 1); }
    else if (bs == 2) { if (dc_exact) JPEGX_LU8(true, 2); else JPEGX_LU8(false, 2); }
    else { if (dc_exact) JPEGX_LU8(true, 4); else JPEGX_LU8(false, 4); }
#undef JPEGX_LU8
#undef JPEGX_LU8K
    HIP_TRY(hipGetLastError());
    return JPEGX_OK;
}

int jpegx_forward_fused_u8(const uint8_t *d_in, int H, int W, ptrdiff_t pitch, int bs, int mode, double param,
                           unsigned flags, int16_t *d_out, jpegx_stream_t stream)
{
    return forward_u8_common(d_in, H, W, pitch, bs, mode, param, flags, d_out, nullptr, nullptr, nullptr, stream);
}

// internal (jpegx_internal.h): the same with the entropy stage's block sizes written into its workspace on the way
int jpegx_internal_forward_u8_sized(const uint8_t *d_in, int H, int W, ptrdiff_t pitch, int bs, int mode, double param, unsigned flags,
                                    int16_t *d_out, unsigned *block_bytes, unsigned *wave_bytes, unsigned *half_info, jpegx_stream_t stream)
{
    if (!block_bytes || !wave_bytes || !half_info) return fail(JPEGX_E_INVALID, "null workspace views");
    return forward_u8_common(d_in, H, W, pitch, bs, mode, param, flags, d_out, block_bytes, wave_bytes, half_info, stream);
}
int jpegx_host_forward_fused_f64(const double *h_in, int H, int W, int mode, double param, unsigned flags, int16_t *h_out)
{
    if (H <= 0 || W <= 0) return fail(JPEGX_E_INVALID, "bad plane shape");
    return host_roundtrip(h_in, (size_t)H * W * 8, h_out, (size_t)H * W * 2, [&](void *di, void *dout, jpegx_stream_t s) {
        return jpegx_forward_fused_f64((const double *)di, H, W, W, mode, param, flags, (int16_t *)dout, s);
    });
}
int jpegx_host_forward_fused(const float *h_in, int H, int W, ptrdiff_t pitch, int mode, double param, unsigned flags,
                             int16_t *h_out)
{
    if (H <= 0 || W <= 0 || pitch < W) return fail(JPEGX_E_INVALID, "bad plane shape");
    return host_roundtrip(h_in, (size_t)H * pitch * 4, h_out, (size_t)H * W * 2, [&](void *di, void *dout, jpegx_stream_t s) {
        return jpegx_forward_fused((const float *)di, H, W, pitch, mode, param, flags, (int16_t *)dout, s);
    });
}
}  // extern "C"
