// jpegx_inverse.hip -- fused inverse kernel (un-zigzag + dequantise + IDCT + round [+ clamp]
// [+ up-sampling]) and its C entry points.  Part of libjpegx.so (C ABI: include/jpegx.h).
// Built with: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (explicit fma only).
#include "jpegx_internal.h"

namespace {

// ------------------------------------------------------------------------------------------------
// float64 exact tier of the inverse, by flagged ROWS, eight at a time.
//
// The fast tier knows per output row whether one of its samples lies within the fp32 error bound of a
// rounding boundary (8 row maxima instead of one block maximum: no extra instructions).  A flagged row
// (block b, row i) is a *unit*; sample x[i][j] needs m[i][l] = sum_k Cn[k][i] Dinv[k] Z[k][l] for
// l = 0..7 (one output each of the eight column transforms, transforms.py:60-69) and then
// x[i][j] = sum_l Cn[l][j] Dinv[l] m[i][l]: 8 + 8 dot products of length 8 instead of the 128 of the
// whole block.  Lane (slot s = lane / 8, t = lane % 8) serves the s-th unit of the pass: it computes
// m[i][t], publishes it to the slot's eight lanes through LDS, then computes x[i][t] -- every dot in the
// reference's own operation order (jpegx_idot8_ref), so the rounded sample is the reference's.
// The owner lane keeps its exact row in 8 registers (`fix`) and writes it over the fast-tier row on
// the way out; no register patching by dynamic index.
// Compared with one block per pass and lane = one sample (coop_inv_exact, round 1) this removes the
// serial LDS / table latency chain per flagged BLOCK: on noise-like planes, where ~8 % of the blocks
// are flagged (5 per wave), that chain was the whole distance to the HBM roofline.
//
// LDS behind the coefficient tile: the tier's tables (InvExactTab: the transposed matrix CnT, so that a
// dot's eight weights are one contiguous row, Dinv, zigzag positions and quantiser entries per column),
// brought in by one LDS-DMA instruction together with the tile; the m exchange slots; the result slots.
// No vector memory load anywhere in the tier.
// ------------------------------------------------------------------------------------------------
constexpr int XT_TAB = 0;                     // InvExactTab (1 KiB, by LDS-DMA at kernel start)
constexpr int XT_DINV = 512;
constexpr int XT_PZ = 576;
constexpr int XT_Q = 640;
constexpr int XT_M = 1024;                    // 8 slots x 8 doubles, slot stride 80 B (bank spread)
constexpr int XT_M_STRIDE = 80;
constexpr int XT_X = XT_M + 8 * XT_M_STRIDE;  // 8 slots x 8 floats
constexpr int XCH_BYTES = XT_X + 8 * 32;      // 1920
constexpr int INV_LDS_BYTES = TILE_BYTES + XCH_BYTES;

// fold a pending exact row into the lane's sample registers (row -1 matches nothing)
__device__ __forceinline__ void fold_fix(float (&v)[64], const float (&fix)[8], int fixrow)
{
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < 8; ++c) v[r * 8 + c] = (fixrow == r) ? fix[c] : v[r * 8 + c];
}

__device__ __forceinline__ void inv_exact_rows(unsigned rowmask, const unsigned char *tile, const unsigned char *xch_c,
                                               unsigned char *xch, const QuantParams &prm, float (&v)[64], float (&fix)[8],
                                               int &fixrow, int lane)
{
    const int s = lane >> 3, t = lane & 7;
    // column t's zigzag positions and table entries, one byte per k
    uint2 pzw = *reinterpret_cast<const uint2 *>(xch_c + XT_PZ + t * 8);
    uint2 qw = *reinterpret_cast<const uint2 *>(xch_c + XT_Q + t * 8);
    // Quantizer.restore (quantizers.py:8-9,30-31,51-53) as ONE multiplication z * f: f = the table entry
    // (the product of two integers below 2^22 is exact, so the reference's np.round of it is the identity),
    // the divisor, or 1
    const bool use_table = prm.mode == JPEGX_QM_QTABLE;
    const double f_uniform = prm.mode == JPEGX_QM_DIVIDE ? prm.param : 1.0;
    for (;;) {
        unsigned long long pend = __ballot(rowmask != 0);
        if (!pend) break;
        // table reads stay inside the pass: hoisted out of the loop they would hold 36 VGPRs across it
        int tab = XT_TAB;
        asm volatile("" : "+v"(tab), "+v"(pzw.x), "+v"(pzw.y), "+v"(qw.x), "+v"(qw.y));
        // the next (up to) eight blocks with a pending row, one unit per block and pass
        int b = -1, mine = -1;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int bk = pend ? __ffsll((long long)pend) - 1 : -1;
            pend &= pend - 1;                // 0 stays 0
            if (k == s) b = bk;
            if (bk == lane) mine = k;        // this lane owns the unit of slot k
        }
        const int myrow = __ffs((int)rowmask) - 1;
        const int i = __shfl(myrow, b < 0 ? lane : b);      // the unit's row, from its owner
        if (mine >= 0) rowmask &= rowmask - 1;
        if (b >= 0) {
            // m[i][t] = sum_k Cn[k][i] * (Dinv[k] * Z[k][t]) in the order of jpegx_idot8_ref: the chain over
            // k = 1, 0, 2, 3, then the chain over k = 5, 4, 6, 7, then their sum.  Half by half, with a
            // scheduling fence in between, so that only four (w, u) pairs are live at a time.
            double acc[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double u[4], w[4];
#pragma unroll
                for (int k = 0; k < 4; k += 2) {
                    const double2 q = *reinterpret_cast<const double2 *>(xch_c + tab + i * 64 + (4 * h + k) * 8);
                    const double2 d = *reinterpret_cast<const double2 *>(xch_c + tab + XT_DINV + (4 * h + k) * 8);
                    w[k] = q.x; w[k + 1] = q.y;
                    u[k] = d.x; u[k + 1] = d.y;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int pz = ((h ? pzw.y : pzw.x) >> (8 * k)) & 0xFF;
                    const int qk = ((h ? qw.y : qw.x) >> (8 * k)) & 0xFF;
                    const double z = (double)*reinterpret_cast<const int16_t *>(tile + tile_off(b, pz >> 3) + (pz & 7) * 2);
                    u[k] = u[k] * (z * (use_table ? (double)qk : f_uniform));
                }
                double p = w[1] * u[1];
                p = fma(w[0], u[0], p);
                p = fma(w[2], u[2], p);
                acc[h] = fma(w[3], u[3], p);
                __builtin_amdgcn_sched_barrier(0);
            }
            *reinterpret_cast<double *>(xch + XT_M + s * XT_M_STRIDE + t * 8) = acc[0] + acc[1];
        }
        __syncthreads();
        if (b >= 0) {
            // x[i][t] = sum_l Cn[l][t] * (Dinv[l] * m[i][l]), same order, then np.round (basis_change.py:43)
            double acc[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double u[4], w[4];
#pragma unroll
                for (int l = 0; l < 4; l += 2) {
                    const double2 m = *reinterpret_cast<const double2 *>(xch + XT_M + s * XT_M_STRIDE + (4 * h + l) * 8);
                    const double2 q = *reinterpret_cast<const double2 *>(xch_c + tab + t * 64 + (4 * h + l) * 8);
                    const double2 d = *reinterpret_cast<const double2 *>(xch_c + tab + XT_DINV + (4 * h + l) * 8);
                    u[l] = d.x * m.x; u[l + 1] = d.y * m.y;
                    w[l] = q.x; w[l + 1] = q.y;
                }
                double p = w[1] * u[1];
                p = fma(w[0], u[0], p);
                p = fma(w[2], u[2], p);
                acc[h] = fma(w[3], u[3], p);
                __builtin_amdgcn_sched_barrier(0);
            }
            // exact conversion: |x| is far below 2^24 for int16 coefficients
            *reinterpret_cast<float *>(xch + XT_X + s * 32 + t * 4) = (float)rint(acc[0] + acc[1]);
        }
        __syncthreads();
        // a second row of the same block (rare): fold the pending one into the sample registers.  Uniform
        // branch, per-lane selects inside (row -1 matches nothing), so that v is updated in place.
        const int pending = (mine >= 0) ? fixrow : -1;
        if (__any(pending >= 0)) fold_fix(v, fix, pending);
        if (mine >= 0) {
            const f32x4 lo = *reinterpret_cast<const f32x4 *>(xch + XT_X + mine * 32);
            const f32x4 hi = *reinterpret_cast<const f32x4 *>(xch + XT_X + mine * 32 + 16);
            fix[0] = lo.x; fix[1] = lo.y; fix[2] = lo.z; fix[3] = lo.w;
            fix[4] = hi.x; fix[5] = hi.y; fix[6] = hi.z; fix[7] = hi.w;
            fixrow = myrow;
        }
        // XT_M is rewritten by the next pass only after every lane has passed the barrier that follows its
        // reads above; XT_X is rewritten only after the next pass's first barrier, i.e. after these reads.
    }
}

// one output row of 8 samples (already rounded; clamped here if asked) -> its place in the output plane
// inflate == 0: replication factor `rep` known only at run time (any block_size, util.inflate util.py:6-14)
template <int OUT, bool NT, int inflate>
__device__ __forceinline__ void store_row(void *base, size_t opitch, int r, const float (&x)[8], int clamp, int rep = 1)
{
    if (OUT == 1) {
        unsigned w[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float lo = x[2 * c], hi = x[2 * c + 1];
            if (clamp) { lo = fminf(fmaxf(lo, 0.f), 255.f); hi = fminf(fmaxf(hi, 0.f), 255.f); }
            const int a = min(max((int)lo, -32768), 32767), b2 = min(max((int)hi, -32768), 32767);
            w[c] = ((unsigned)a & 0xFFFFu) | ((unsigned)b2 << 16);
        }
        st_u32x4<NT>(reinterpret_cast<int16_t *>(base) + (size_t)r * opitch, u32x4{w[0], w[1], w[2], w[3]});
    } else {
        // uint8 rows, optionally with SubSampling.invert fused (util.inflate, util.py:6-14): every
        // sample is replicated inflate x inflate times, the output plane is [H*inflate][W*inflate].
        // v_cvt_pk_u8_f32 converts with saturation to 0..255 and drops the byte into its place: one instruction per
        // sample where clamp + conversion + shift/or took three (Normalization.invert, normalization.py:10-14); the
        // samples are integers already (np.round above), so the conversion's own rounding mode does not matter
        unsigned char *o = reinterpret_cast<unsigned char *>(base);
        if (inflate == 0) {
            // The row's 8 bytes sit in (lo, hi); its inflated form is 8 * rep bytes = rep 8-byte pieces, written to
            // `rep` consecutive output rows.  `rep` is wave-uniform, so which sample feeds which output byte is
            // scalar bookkeeping (sample index s, copies emitted c): no division, no dynamic register indexing --
            // byte s comes out of the 64-bit pair with a uniform shift.
            unsigned lo = 0u, hi = 0u;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                lo = __builtin_amdgcn_cvt_pk_u8_f32(x[c], c, lo);
                hi = __builtin_amdgcn_cvt_pk_u8_f32(x[4 + c], c, hi);
            }
            const unsigned long long row8 = ((unsigned long long)hi << 32) | lo;
            int s = 0, c = 0;
            for (int piece = 0; piece < rep; ++piece) {
                unsigned w[2] = {0u, 0u};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned b = (unsigned)(row8 >> (8 * s)) & 0xFFu;
                    w[i >> 2] |= b << (8 * (i & 3));
                    if (++c == rep) { c = 0; ++s; }
                }
                for (int a = 0; a < rep; ++a)
                    st_u32x2<NT>(o + (size_t)(rep * r + a) * opitch + (size_t)piece * 8, u32x2{w[0], w[1]});
            }
        } else if (inflate == 1) {
            unsigned lo = 0u, hi = 0u;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                lo = __builtin_amdgcn_cvt_pk_u8_f32(x[c], c, lo);
                hi = __builtin_amdgcn_cvt_pk_u8_f32(x[4 + c], c, hi);
            }
            st_u32x2<NT>(o + (size_t)r * opitch, u32x2{lo, hi});
        } else if (inflate == 2) {
            unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                w[c >> 1] = __builtin_amdgcn_cvt_pk_u8_f32(x[c], 2 * (c & 1), w[c >> 1]);
                w[c >> 1] = __builtin_amdgcn_cvt_pk_u8_f32(x[c], 2 * (c & 1) + 1, w[c >> 1]);
            }
            const u32x4 q = {w[0], w[1], w[2], w[3]};
            st_u32x4<NT>(o + (size_t)(2 * r) * opitch, q);
            st_u32x4<NT>(o + (size_t)(2 * r + 1) * opitch, q);
        } else {
            unsigned u[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) u[c] = __builtin_amdgcn_cvt_pk_u8_f32(x[c], 0, 0u);
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

// ------------------------------------------------------------------------------------------------
// fused inverse: un-zigzag + dequantise + IDCT + round (+ clamp).  OUT: 0 f32, 1 i16, 2 u8.
// ------------------------------------------------------------------------------------------------
// INF: SubSampling.invert replication factor fused into the uint8 write-out (1 for the other types).
#ifdef JPEGX_INV_WPE   // A/B builds only (microbench/build_variant.sh)
#define JPEGX_INV_OCC __attribute__((amdgpu_waves_per_eu(JPEGX_INV_WPE, JPEGX_INV_WPE)))
#else
#define JPEGX_INV_OCC
#endif
template <int OUT, bool NT, int INF>
__global__ __launch_bounds__(64) JPEGX_INV_OCC void k_inverse_fused(const int16_t *__restrict__ in, int wb, int nblk,
                                                      QuantParams prm, int clamp,
                                                      void *__restrict__ outv, size_t opitch,
                                                      unsigned long long *counters, int rep)
{
    // [0, 8 KiB): the wave's coefficient tile, behind it the exact tier's area.  f32 output is staged
    // through an 8-row x 2 KiB strip (16 KiB, coalesced 1 KiB stores) laid over both once they are dead; the
    // narrower i16 / u8 rows are contiguous per store instruction and go out directly.
    __shared__ __attribute__((aligned(16))) unsigned char lds[OUT == 0 ? STRIP_BYTES : INV_LDS_BYTES];
    static_assert(INV_LDS_BYTES <= STRIP_BYTES, "the f32 strip covers tile + exact-tier area");
    unsigned char *xch = lds + TILE_BYTES;

    const int lane = threadIdx.x;
    int wg = blockIdx.x;
    if (prm.tune & 2) {   // XCD-private order (jpegx_device.h: xcd_private_wg)
        wg = xcd_private_wg(blockIdx.x, (nblk + 63) >> 6, (prm.tune >> 8) & 31);
        if (wg >= ((nblk + 63) >> 6)) return;
    }
    const int g0 = wg * 64;
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
    // the exact tier's tables ride along (1 KiB from L2, default cache policy)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const unsigned char *>(&c_inv_exact) + lane * 16),
                                     (__attribute__((address_space(3))) void *)(xch + XT_TAB), 16, 0, 0);
    __syncthreads();

    float v[64];
    float A = 0.f;   // weighted absolute sum of the dequantised coefficients: the error bound's only input (jpegx_math.h)
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
                A = fmaf(fabsf(d), jpegx_inv_weight(n, 0), A);
            }
        }
    }
    if (prm.mode == JPEGX_QM_DIVIDE) {   // wave-uniform: the fp32 dequantisation rounded twice (divisor, product)
#pragma unroll
        for (int n = 0; n < 64; ++n) A = fmaf(fabsf(v[n]), jpegx_inv_weight(n, 2) - jpegx_inv_weight(n, 0), A);
    }
    const float E = jpegx_inv_err_from_weighted_sum(A);
    jpegx_idct8x8_f32(v);

    // np.round of basis_change.py:43 and, per row, the distance of the worst sample to its integer
    unsigned rowmask = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        float worst = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float q = rintf(v[r * 8 + c]);
            worst = fmaxf(worst, fabsf(v[r * 8 + c] - q));
            v[r * 8 + c] = q;
            // opaque: otherwise the u8 variant keeps the unrounded sample alive next to the rounded one (it
            // re-derives the byte from it with a rounding conversion): 64 extra VGPRs
            asm volatile("" : "+v"(v[r * 8 + c]));
        }
        rowmask |= (worst + E < JPEGX_SAFE_HALF) ? 0u : (1u << r);
    }
    if (!valid) rowmask = 0;
    const unsigned long long flagged = __ballot(rowmask != 0);
    census(counters, flagged, nblk - g0, lane);
    float fix[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int fixrow = -1;
    if (flagged && !(prm.tune & 1)) inv_exact_rows(rowmask, lds, xch, xch, prm, v, fix, fixrow, lane);

    if (OUT == 0) {
        // park the lane's 8 x 32 B in the strip layout of the forward kernel (chunk c at slot strip_swz(c)),
        // lay the exact row over the fast-tier one, then 16 coalesced 1 KiB stores.  The strip covers the
        // (dead) coefficient tile and the exact tier's area.
        __syncthreads();
        const int f = ((lane >> 2) ^ (lane >> 3)) & 1;
        // clamp == 0: the bounds are infinite and the median is the sample itself (one v_med3 either way)
        const float lo = clamp ? 0.f : -__builtin_inff(), hi = clamp ? 255.f : __builtin_inff();
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float x[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) x[c] = __builtin_amdgcn_fmed3f(v[r * 8 + c], lo, hi);
            *reinterpret_cast<f32x4 *>(lds + r * 2048 + ((2 * lane + f) << 4)) = f32x4{x[0], x[1], x[2], x[3]};
            *reinterpret_cast<f32x4 *>(lds + r * 2048 + ((2 * lane + (f ^ 1)) << 4)) = f32x4{x[4], x[5], x[6], x[7]};
        }
        if (fixrow >= 0) {
#pragma unroll
            for (int c = 0; c < 8; ++c) fix[c] = __builtin_amdgcn_fmed3f(fix[c], lo, hi);
            *reinterpret_cast<f32x4 *>(lds + fixrow * 2048 + ((2 * lane + f) << 4)) = f32x4{fix[0], fix[1], fix[2], fix[3]};
            *reinterpret_cast<f32x4 *>(lds + fixrow * 2048 + ((2 * lane + (f ^ 1)) << 4)) = f32x4{fix[4], fix[5], fix[6], fix[7]};
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
    void *base = (OUT == 1)
        ? static_cast<void *>(reinterpret_cast<int16_t *>(outv) + (size_t)by * 8 * opitch + (size_t)bx * 8)
        : static_cast<void *>(reinterpret_cast<unsigned char *>(outv) + (size_t)by * 8 * (INF ? INF : rep) * opitch + (size_t)bx * 8 * (INF ? INF : rep));
    // the narrow types store straight from the registers: a lane skips the fast-tier row its exact row
    // replaces and stores that one afterwards (every byte is written once; predicated stores, no selects)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        float x[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = v[r * 8 + c];
        if (r != fixrow) store_row<OUT, NT, INF>(base, opitch, r, x, clamp, rep);
    }
    if (fixrow >= 0) store_row<OUT, NT, INF>(base, opitch, fixrow, fix, clamp, rep);
}
}  // namespace

extern "C" {

static int inverse_common(const int16_t *d_in, int H, int W, int mode, double param, unsigned flags, void *d_out,
                          ptrdiff_t out_pitch, int out_type, int inflate, jpegx_stream_t stream)
{
    int rc = check_plane(d_in, d_out, H, W, out_pitch / inflate, 1);
    if (rc) return rc;
    if (out_pitch < (ptrdiff_t)W * inflate) return fail(JPEGX_E_INVALID, "output pitch smaller than the inflated width");
    const int esz = out_type == JPEGX_OUT_F32 ? 4 : (out_type == JPEGX_OUT_I16 ? 2 : 1);
    if (out_type < 0 || out_type > 2) return fail(JPEGX_E_INVALID, "unknown output type");
    const bool pieces8 = esz == 1 && inflate != 2 && inflate != 4;      // plain u8 and the run-time factor store 8-byte pieces
    if (((size_t)out_pitch * esz) % (pieces8 ? 8 : 16) != 0 || !aligned16(d_in) || !aligned16(d_out))
        return fail(JPEGX_E_INVALID, "inverse: output rows must stay 16-byte (u8 with block_size other than 2, 4: 8-byte) aligned");
    QuantParams qp;
    rc = fill_inverse_params(mode, param, &qp);
    if (rc) return rc;
    if (flags & JPEGX_F_TUNE_SKIP_EXACT) qp.tune |= 1;
    const int wb = W / 8, nblk = (H / 8) * wb;
    dim3 grid((nblk + 63) / 64), block(64);
    qp.tune |= xcd_order_setup(flags, nblk, &grid);
    const int clamp = (flags & JPEGX_F_CLAMP_U8) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    const bool nt = (flags & JPEGX_F_TUNE_NO_NT) == 0;
#define JPEGX_LAUNCH_INV(OUT, NT, CL, INF) \
    hipLaunchKernelGGL((k_inverse_fused<OUT, NT, INF>), grid, block, 0, st, d_in, wb, nblk, qp, CL, d_out, (size_t)out_pitch, g_counters, inflate)
#define JPEGX_LAUNCH_INV_NT(OUT, CL, INF) do { if (nt) JPEGX_LAUNCH_INV(OUT, true, CL, INF); else JPEGX_LAUNCH_INV(OUT, false, CL, INF); } while (0)
    if (out_type == JPEGX_OUT_F32) JPEGX_LAUNCH_INV_NT(0, clamp, 1);
    else if (out_type == JPEGX_OUT_I16) JPEGX_LAUNCH_INV_NT(1, clamp, 1);
    else if (inflate == 1) JPEGX_LAUNCH_INV_NT(2, 1, 1);
    else if (inflate == 2) JPEGX_LAUNCH_INV_NT(2, 1, 2);
    else if (inflate == 4) JPEGX_LAUNCH_INV_NT(2, 1, 4);
    else JPEGX_LAUNCH_INV_NT(2, 1, 0);            // any other block_size: replication factor at run time
#undef JPEGX_LAUNCH_INV_NT
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
    if (bs < 1 || bs > 255) return fail(JPEGX_E_UNSUPPORTED, "fused inflate supports block_size 1..255");
    return inverse_common(d_in, H, W, mode, param, flags, d_out, out_pitch, JPEGX_OUT_U8, bs, stream);
}
int jpegx_host_inverse_fused_u8_inflated(const int16_t *h_in, int H, int W, int mode, double param, unsigned flags, int bs,
                                         uint8_t *h_out, ptrdiff_t out_pitch)
{
    if (H <= 0 || W <= 0 || bs < 1 || out_pitch < (ptrdiff_t)W * bs) return fail(JPEGX_E_INVALID, "bad plane shape");
    return host_roundtrip(h_in, (size_t)H * W * 2, h_out, (size_t)H * bs * out_pitch, [&](void *di, void *dout, jpegx_stream_t s) {
        return jpegx_inverse_fused_u8_inflated((const int16_t *)di, H, W, mode, param, flags, bs, (uint8_t *)dout, out_pitch, s);
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
}  // extern "C"
