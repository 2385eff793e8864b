// jpegx_device.h -- device-side building blocks shared by the kernel translation units of
// libjpegx.so: the reference's constant tables, the by-value quantiser parameters, cache-policy
// load/store helpers, the XOR-swizzled LDS tile, the cooperative float64 exact tier and the
// quantise/pack + tile write-out helpers of the forward kernels.  Everything lives in an anonymous
// namespace (each translation unit gets its own copy; nothing here is exported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
constexpr D64 make_cn_transposed()
{
    D64 a{{JPEGX_TABLE_DCT_NORMALIZED}}, t{};
    for (int k = 0; k < 8; ++k)
        for (int n = 0; n < 8; ++n) t.v[n * 8 + k] = a.v[k * 8 + n];
    return t;
}
__device__ const D64 c_cnT = make_cn_transposed();               // CnT[n][k] = Cn[k][n]: column n of Cn contiguous
// Everything the inverse exact tier looks up, packed into 1 KiB so that ONE LDS-DMA instruction issued
// with the coefficient tile brings it into LDS: the tier then runs without a single vector memory load
// (under a saturated HBM stream such a load queues behind the streaming traffic for microseconds, which
// at 10-16 waves per CU costs far more than the tier's arithmetic).
struct InvExactTab {
    double cnT[64];          // CnT[n][k] = Cn[k][n]
    double dinv[8];
    unsigned char pz[64];    // pz[t][k] = zigzag position of the natural coefficient (row k, column t)
    unsigned char q[64];     // q[t][k]  = luminance table entry (row k, column t)
    unsigned char pad[1024 - 512 - 64 - 64 - 64];
};
constexpr InvExactTab make_inv_exact_tab()
{
    InvExactTab t{};
    D64 cn = make_cn_transposed(), di{{JPEGX_TABLE_NORM_DIAG}};
    I64 zi = make_zzinv(), qt = make_qt();
    for (int n = 0; n < 64; ++n) t.cnT[n] = cn.v[n];
    for (int k = 0; k < 8; ++k) t.dinv[k] = di.v[k];
    for (int c = 0; c < 8; ++c)
        for (int k = 0; k < 8; ++k) {
            t.pz[c * 8 + k] = (unsigned char)zi.v[k * 8 + c];
            t.q[c * 8 + k] = (unsigned char)qt.v[k * 8 + c];
        }
    return t;
}
static_assert(sizeof(InvExactTab) == 1024, "one DMA instruction = 64 lanes x 16 B");
__device__ __attribute__((aligned(16))) const InvExactTab c_inv_exact = make_inv_exact_tab();
// The forward exact tier's tables in one DMA-sized (1 KiB) piece, same idea as InvExactTab: C[k][n], the
// luminance table (1.0 / q is recomputed in float64 from the byte: the same correctly rounded quotient the
// host table holds) and the inverse zigzag order.
struct FwdExactTab {
    double c[64];
    unsigned char q[64];
    unsigned char zzinv[64];
    unsigned char pad[1024 - 512 - 64 - 64];
};
constexpr FwdExactTab make_fwd_exact_tab()
{
    FwdExactTab t{};
    D64 c{{JPEGX_TABLE_DCT_MATRIX}};
    I64 zi = make_zzinv(), qt = make_qt();
    for (int n = 0; n < 64; ++n) {
        t.c[n] = c.v[n];
        t.q[n] = (unsigned char)qt.v[n];
        t.zzinv[n] = (unsigned char)zi.v[n];
    }
    return t;
}
static_assert(sizeof(FwdExactTab) == 1024, "one DMA instruction = 64 lanes x 16 B");
__device__ __attribute__((aligned(16))) const FwdExactTab c_fwd_exact = make_fwd_exact_tab();
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

// the same with the matrix rows read from an LDS copy of the table (no vector memory load)
__device__ __forceinline__ double coop_fwd_exact_tab(double a_own, double *sA, double *sM, int lane, const double *tabC)
{
    const int hi = lane >> 3, lo = lane & 7;
    sA[lane] = a_own;
    __syncthreads();
    const double m = jpegx_dot8_ref(&tabC[lo * 8], &sA[hi * 8], 1);  // M[i=hi][l=lo]
    sM[lo * 8 + hi] = m;
    __syncthreads();
    const double y = jpegx_dot8_ref(&tabC[hi * 8], &sM[lo * 8], 1);  // Y[k=hi][l=lo]
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

// Quantise the 64 coefficients of v (natural order, as jpegx_dct8x8_aan_f32 leaves them: scaled by g_k g_l, the
// multipliers in prm.rq32 carry 1 / (g_k g_l)) in zigzag order and pack them as int16 pairs
// (pipeline/quantization.py:8-18 + pipeline/zigzag_order.py:85-99; the zigzag is a compile-time
// renaming).  E = u S (jpegx_fwd_err_unit); a coefficient's bound is E F(k, l) / q with F = jpegx_fwd_roundings.
// Returns the worst rounding margin max(|t - rint(t)| + E F / q): the block is safe iff it stays below JPEGX_SAFE_HALF.
// Round 3: jpegx_quant_fast (magic-constant rounding of the exact product, jpegx_math.h) and one byte permute per pair.
// PIXEL: values provably fit int16, no saturation needed; the row pass's butterfly adds are exact.
// int16 pair from two magic-biased quantiser outputs (jpegx_quant_fast): the low halves of their bit patterns
__device__ __forceinline__ unsigned pack_magic_pair(float m0, float m1)
{
    return __builtin_amdgcn_perm(__float_as_uint(m1), __float_as_uint(m0), 0x05040100u);
}
// saturation of a magic-biased value to the int16 range (generic input only; one v_med3)
__device__ __forceinline__ float clamp_magic_i16(float m)
{
    return __builtin_amdgcn_fmed3f(m, JPEGX_RMAGIC - 32768.0f, JPEGX_RMAGIC + 32767.0f);
}

template <bool PIXEL, bool DC_EXACT>
__device__ __forceinline__ float quantise_zigzag_pack(const float (&v)[64], const QuantParams &prm, float E,
                                                      unsigned (&pk)[32])
{
    float worst = 0.f;
    float Ef[JPEGX_AAN_LEVELS];         // E = u S (jpegx_fwd_err_unit) times the bound's factor F(k, l), by level (jpegx_math.h)
#pragma unroll
    for (int j = 0; j < JPEGX_AAN_LEVELS; ++j) Ef[j] = E * jpegx_aan_level(j, PIXEL);
    float magic = JPEGX_RMAGIC;
    asm volatile("" : "+v"(magic));     // one live register, see jpegx_quant_fast_m
#pragma unroll
    for (int p = 0; p < 64; p += 2) {
        float m[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = kZZ.v[p + h];
            const float rq = prm.rq32[n];
            float d;
            m[h] = jpegx_quant_fast_m(v[n], rq, magic, d);
            if (!(DC_EXACT && n == 0)) worst = fmaxf(worst, fmaf(Ef[jpegx_aan_level_index(n, PIXEL)], fabsf(rq), fabsf(d)));
            if (!PIXEL) m[h] = clamp_magic_i16(m[h]);
        }
        pk[p >> 1] = pack_magic_pair(m[0], m[1]);
    }
    return worst;
}

// The same, keeping the worst margin PER COLUMN l = n % 8 of the block (eight running maxima instead of
// one: no extra instructions) and returning the set of columns that hold an unsafe coefficient -- the unit
// of work of the column-wise exact tier (fwd_exact_columns).
template <bool PIXEL, bool DC_EXACT>
__device__ __forceinline__ unsigned quantise_zigzag_pack_cols(const float (&v)[64], const QuantParams &prm, float E,
                                                              unsigned (&pk)[32])
{
    float worst[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float Ef[JPEGX_AAN_LEVELS];
#pragma unroll
    for (int j = 0; j < JPEGX_AAN_LEVELS; ++j) Ef[j] = E * jpegx_aan_level(j, PIXEL);
    float magic = JPEGX_RMAGIC;
    asm volatile("" : "+v"(magic));
#pragma unroll
    for (int p = 0; p < 64; p += 2) {
        float m[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = kZZ.v[p + h];
            const float rq = prm.rq32[n];
            float d;
            m[h] = jpegx_quant_fast_m(v[n], rq, magic, d);
            if (!(DC_EXACT && n == 0)) worst[n & 7] = fmaxf(worst[n & 7], fmaf(Ef[jpegx_aan_level_index(n, PIXEL)], fabsf(rq), fabsf(d)));
            if (!PIXEL) m[h] = clamp_magic_i16(m[h]);
        }
        pk[p >> 1] = pack_magic_pair(m[0], m[1]);
    }
    unsigned mask = 0;
#pragma unroll
    for (int l = 0; l < 8; ++l) mask |= (worst[l] < JPEGX_SAFE_HALF) ? 0u : (1u << l);
    return mask;
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

// Bytes the entropy stage will write for a block (RunLengthEncoding + RleBytestream, pipeline/run_length_encoding.py:47-64,
// pipeline/rle_byte_stream.py:48-59, util.py:134-156), from the block's 32 packed words (coefficients 2k, 2k + 1 of the
// zigzag order in word k) while they are still in registers: 8 bits of end marker + per non-zero 4 + 4 + 1 + bit_length
// bits + 8 per chain code of fifteen zeros, padded to bytes.  bad: an amplitude beyond 15 bits (util.py:140-149).
// Two coefficients per instruction where the ISA allows: |.| by v_pk_sub / v_pk_max, "non-zero" by v_pk_min_u16 with
// 1, the 64-bit non-zero mask by doubling an accumulator (word k's flags land in bits k and 16 + k: even and odd
// coefficients apart, interleaved afterwards), bit lengths through v_ffbh_u32 (which says -1 for zero: the sum is
// corrected by the number of zeros).  half: what the two-lanes-per-block emitter needs to start a lane at coefficient 32.
__device__ __forceinline__ unsigned rle_block_bytes(const unsigned (&pk)[32], bool &bad, unsigned &half)
{
    int sumf = 0, sumf_lo = 0;
    unsigned acc[2] = {0u, 0u}, any = 0u;
    // Written out instruction by instruction: left to the compiler, min(|a|, 1) on the packed halves becomes two
    // compares (one of them SDWA), two selects and a byte permute per word, and a guard against clz(0) two more
    // instructions per coefficient.  Per pair of words here: |.| (v_pk_sub, v_pk_max), the OR of all magnitudes,
    // v_ffbh_u32 of either half (32 - bit_length; -1 for zero, put right below from the number of zeros), the
    // non-zero flags of both halves in one v_pk_min_u16 and the doubling accumulator: 19 instructions for four
    // coefficients.
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h == 1) sumf_lo = sumf;
        unsigned a_h = 0u;
#pragma unroll
        for (int k = 15; k >= 1; k -= 2) {
            unsigned t0, t1, u0, u1;
            asm("v_pk_sub_i16 %[t0], 0, %[x0]\n\t"
                "v_pk_sub_i16 %[t1], 0, %[x1]\n\t"
                "v_pk_max_i16 %[t0], %[x0], %[t0]\n\t"
                "v_pk_max_i16 %[t1], %[x1], %[t1]\n\t"
                "v_or3_b32 %[any], %[any], %[t0], %[t1]\n\t"
                "v_and_b32 %[u0], 0xffff, %[t0]\n\t"
                "v_lshrrev_b32 %[u1], 16, %[t0]\n\t"
                "v_ffbh_u32 %[u0], %[u0]\n\t"
                "v_ffbh_u32 %[u1], %[u1]\n\t"
                "v_add3_u32 %[sum], %[sum], %[u0], %[u1]\n\t"
                "v_and_b32 %[u0], 0xffff, %[t1]\n\t"
                "v_lshrrev_b32 %[u1], 16, %[t1]\n\t"
                "v_ffbh_u32 %[u0], %[u0]\n\t"
                "v_ffbh_u32 %[u1], %[u1]\n\t"
                "v_add3_u32 %[sum], %[sum], %[u0], %[u1]\n\t"
                "v_pk_min_u16 %[t0], %[t0], %[ones]\n\t"
                "v_pk_min_u16 %[t1], %[t1], %[ones]\n\t"
                "v_lshl_add_u32 %[acc], %[acc], 1, %[t0]\n\t"
                "v_lshl_add_u32 %[acc], %[acc], 1, %[t1]"
                : [any] "+v"(any), [sum] "+v"(sumf), [acc] "+v"(a_h), [t0] "=&v"(t0), [t1] "=&v"(t1), [u0] "=&v"(u0), [u1] "=&v"(u1)
                : [x0] "v"(pk[16 * h + k]), [x1] "v"(pk[16 * h + k - 1]), [ones] "s"(0x00010001u));
        }
        acc[h] = a_h;
    }
    bad = (any & 0xC000C000u) != 0u;                                         // |a| > 16383 somewhere
    const unsigned nnz = (unsigned)__popc(acc[0]) + (unsigned)__popc(acc[1]);
    const unsigned sum_bl = 33u * nnz - 64u - (unsigned)sumf;                  // sum of the bit lengths: sumf = sum over non-zeros of (32 - length) - zeros
    // the non-zero mask in coefficient order
    unsigned m[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        unsigned e = acc[h] & 0xFFFFu, o = acc[h] >> 16;
        e = (e | (e << 8)) & 0x00FF00FFu; o = (o | (o << 8)) & 0x00FF00FFu;
        e = (e | (e << 4)) & 0x0F0F0F0Fu; o = (o | (o << 4)) & 0x0F0F0F0Fu;
        e = (e | (e << 2)) & 0x33333333u; o = (o | (o << 2)) & 0x33333333u;
        e = (e | (e << 1)) & 0x55555555u; o = (o | (o << 1)) & 0x55555555u;
        m[h] = e | (o << 1);
    }
    const unsigned long long M = ((unsigned long long)m[1] << 32) | m[0];
    // chain codes: one per fifteen zeros in front of a non-zero (see chain_count in jpegx_entropy.hip); rare enough to be
    // decided by the wave
    unsigned chains = 0, chains_lo = 0;
    {
        const unsigned long long z = ~M;
        const unsigned long long r2 = z & (z << 1), r4 = r2 & (r2 << 2), r8 = r4 & (r4 << 4);
        const unsigned long long r15 = r8 & (r8 << 7);
        if (__any(((r15 << 1) & M) != 0ull)) {
            const unsigned long long r30 = r15 & (r15 << 15);
            const unsigned long long r45 = r30 & (r15 << 30);
            const unsigned long long r60 = r30 & (r30 << 30);
            chains = (unsigned)(__popcll((r15 << 1) & M) + __popcll((r30 << 1) & M) + __popcll((r45 << 1) & M) + __popcll((r60 << 1) & M));
            chains_lo = (unsigned)(__popc((unsigned)(r15 << 1) & m[0]) + __popc((unsigned)(r30 << 1) & m[0]));      // runs of 45 do not end below 32 ... but 30 do
        }
    }
    const unsigned bits = 8u + sum_bl + 9u * nnz + 8u * chains;
    // for the emitter's second lane: the bits of the codes of coefficients 0..31, and 1 + the last non-zero among them
    const unsigned nnz_lo = (unsigned)__popc(acc[0]);
    const unsigned bits_lo = (33u * nnz_lo - 32u - (unsigned)sumf_lo) + 9u * nnz_lo + 8u * chains_lo;
    half = bits_lo | ((m[0] ? 32u - (unsigned)__clz((int)m[0]) : 0u) << 12);
    return (bits + 7u) >> 3;
}

// XCD-private block order.  Workgroups are dealt round-robin over the 8 XCDs, so with the natural numbering
// every XCD touches -- and translates for itself -- every 2 MiB page of both streams.  Here workgroup i of
// XCD (i % 8) takes strip ((j >> logr) * 8 + i % 8) << logr | (j & (2^logr - 1)), j = i / 8: the XCDs take
// turns in runs of 2^logr strips (logr = 7: 2 MiB of fp32 plane per run), so a page belongs to one XCD.
// logr = 31: one run per XCD (each walks a contiguous eighth).  Measured in profiles/r02_ab_xcd_order.txt.
__device__ __forceinline__ int xcd_private_wg(unsigned i, int nwg, int logr)
{
    const int j = (int)(i >> 3), x = (int)(i & 7u);
    if (logr >= 31) return x * ((nwg + 7) >> 3) + j;
    return (((j >> logr) * 8 + x) << logr) + (j & ((1 << logr) - 1));
}

// chunk permutation of the forward strip / inverse output strip: 16-B chunk c of a 2 KiB row sits at
// slot c ^ (bit3(c) ^ bit4(c)), which makes a ds_read_b128 / ds_write_b128 at a 32-byte lane stride
// bank-conflict free
__device__ __forceinline__ int strip_swz(int chunk) { return chunk ^ (((chunk >> 3) ^ (chunk >> 4)) & 1); }

constexpr int STRIP_BYTES = 8 * 2048;
constexpr int STRIP_LDS_BYTES = STRIP_BYTES + SCRATCH_DOUBLES * 8 + 128;

}  // namespace
