// jpegx_math.h -- per-block arithmetic shared by every kernel of libjpegx.so (jpegx_forward / _inverse / _stage .hip).
//
// Everything here is a plain inline function on registers so that tests/emul can compile
// the very same arithmetic for the host (g++ -mfma -ffp-contract=off) and check it against
// the oracle without a GPU.  Translation units including this file MUST be built with
// -ffp-contract=off: every fused multiply-add is an explicit fma()/fmaf() call.
//
// Reference behaviour implemented (citations relative to /root/reference):
//   transforms.py:4-11     un-normalised DCT-II matrix C[k][n] = cos(pi/8 (n+1/2) k)
//   transforms.py:36-58    transform_1d / transform_2d           (forward, rows then columns)
//   transforms.py:40-44,60-69  transform_1d_inverse / transform_2d_inverse (columns, rows)
//   quantizers.py:4-53     the four quantisers, np.round == round-half-to-even
//
// Two arithmetic tiers:
//   * fp32 fast tier: an even/odd factorisation of the 8-point DCT (35 flops per 1-D
//     transform instead of 64).  It is NOT in the reference's summation order; it only has
//     to land on the same integer after rounding, which is guaranteed by the error bound
//     jpegx_fwd_err_bound()/jpegx_inv_err_bound(): whenever the fp32 value lies within the
//     bound of a rounding boundary the coefficient is recomputed by
//   * fp64 exact tier: the reference's own operation order (the OpenBLAS dgemv orders
//     measured in the build container, see oracle/jpegx_oracle.c header), bit-identical to
//     the float64 value NumPy produces, then rounded half-to-even like np.round.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define JPEGX_HD __host__ __device__ __forceinline__
#else
#define JPEGX_HD static inline __attribute__((always_inline))
#endif

#include "../../include/jpegx_tables.inc"

// cos(m*pi/16) for the fast tier (fp32 roundings of the exact values)
#define JPEGX_A1 0.98078528040323044913f
#define JPEGX_A2 0.92387953251128675613f
#define JPEGX_A3 0.83146961230254523708f
#define JPEGX_A4 0.70710678118654752440f
#define JPEGX_A5 0.55557023301960222474f
#define JPEGX_A6 0.38268343236508977173f
#define JPEGX_A7 0.19509032201612826785f

// ---------------------------------------------------------------------------------------------
// fp32 fast tier
// ---------------------------------------------------------------------------------------------

// y = C . x for one 8-vector, in place (transforms.py:36-38, different summation order).
JPEGX_HD void jpegx_dct8_f32(float &x0, float &x1, float &x2, float &x3,
                             float &x4, float &x5, float &x6, float &x7)
{
    const float s0 = x0 + x7, s1 = x1 + x6, s2 = x2 + x5, s3 = x3 + x4;
    const float d0 = x0 - x7, d1 = x1 - x6, d2 = x2 - x5, d3 = x3 - x4;
    const float e0 = s0 + s3, e1 = s1 + s2, e2 = s0 - s3, e3 = s1 - s2;
    x0 = e0 + e1;
    x4 = (e0 - e1) * JPEGX_A4;
    x2 = fmaf(e3, JPEGX_A6, e2 * JPEGX_A2);
    x6 = fmaf(e3, -JPEGX_A2, e2 * JPEGX_A6);
    x1 = fmaf(d3, JPEGX_A7, fmaf(d2, JPEGX_A5, fmaf(d1, JPEGX_A3, d0 * JPEGX_A1)));
    x3 = fmaf(d3, -JPEGX_A5, fmaf(d2, -JPEGX_A1, fmaf(d1, -JPEGX_A7, d0 * JPEGX_A3)));
    x5 = fmaf(d3, JPEGX_A3, fmaf(d2, JPEGX_A7, fmaf(d1, -JPEGX_A1, d0 * JPEGX_A5)));
    x7 = fmaf(d3, -JPEGX_A1, fmaf(d2, JPEGX_A3, fmaf(d1, -JPEGX_A5, d0 * JPEGX_A7)));
}

// x = Cn^T . (Dinv . X) = X0/8 + 1/4 sum_{k>=1} C[k][n] X[k], in place (transforms.py:40-44).
JPEGX_HD void jpegx_idct8_f32(float &x0, float &x1, float &x2, float &x3,
                              float &x4, float &x5, float &x6, float &x7)
{
    const float q = 0.25f;
    const float p0 = x0 * 0.125f, p4 = x4 * (q * JPEGX_A4);
    const float g0 = p0 + p4, g1 = p0 - p4;
    const float h0 = fmaf(x6, q * JPEGX_A6, x2 * (q * JPEGX_A2));
    const float h1 = fmaf(x6, -q * JPEGX_A2, x2 * (q * JPEGX_A6));
    const float E0 = g0 + h0, E1 = g1 + h1, E2 = g1 - h1, E3 = g0 - h0;
    const float O0 = fmaf(x7, q * JPEGX_A7, fmaf(x5, q * JPEGX_A5, fmaf(x3, q * JPEGX_A3, x1 * (q * JPEGX_A1))));
    const float O1 = fmaf(x7, -q * JPEGX_A5, fmaf(x5, -q * JPEGX_A1, fmaf(x3, -q * JPEGX_A7, x1 * (q * JPEGX_A3))));
    const float O2 = fmaf(x7, q * JPEGX_A3, fmaf(x5, q * JPEGX_A7, fmaf(x3, -q * JPEGX_A1, x1 * (q * JPEGX_A5))));
    const float O3 = fmaf(x7, -q * JPEGX_A1, fmaf(x5, q * JPEGX_A3, fmaf(x3, -q * JPEGX_A5, x1 * (q * JPEGX_A7))));
    x0 = E0 + O0; x7 = E0 - O0;
    x1 = E1 + O1; x6 = E1 - O1;
    x2 = E2 + O2; x5 = E2 - O2;
    x3 = E3 + O3; x4 = E3 - O3;
}

// 2-D forward on a block held as v[row*8+col]: rows first, then columns (transforms.py:46-58).
JPEGX_HD void jpegx_dct8x8_f32(float (&v)[64])
{
#pragma unroll
    for (int i = 0; i < 8; ++i)
        jpegx_dct8_f32(v[i * 8 + 0], v[i * 8 + 1], v[i * 8 + 2], v[i * 8 + 3],
                       v[i * 8 + 4], v[i * 8 + 5], v[i * 8 + 6], v[i * 8 + 7]);
#pragma unroll
    for (int j = 0; j < 8; ++j)
        jpegx_dct8_f32(v[0 * 8 + j], v[1 * 8 + j], v[2 * 8 + j], v[3 * 8 + j],
                       v[4 * 8 + j], v[5 * 8 + j], v[6 * 8 + j], v[7 * 8 + j]);
}

// 2-D inverse: columns first, then rows (transforms.py:60-69).
JPEGX_HD void jpegx_idct8x8_f32(float (&v)[64])
{
#pragma unroll
    for (int j = 0; j < 8; ++j)
        jpegx_idct8_f32(v[0 * 8 + j], v[1 * 8 + j], v[2 * 8 + j], v[3 * 8 + j],
                        v[4 * 8 + j], v[5 * 8 + j], v[6 * 8 + j], v[7 * 8 + j]);
#pragma unroll
    for (int i = 0; i < 8; ++i)
        jpegx_idct8_f32(v[i * 8 + 0], v[i * 8 + 1], v[i * 8 + 2], v[i * 8 + 3],
                        v[i * 8 + 4], v[i * 8 + 5], v[i * 8 + 6], v[i * 8 + 7]);
}

// ---------------------------------------------------------------------------------------------
// fp32 fast tier, round 3: the forward transform of the fused kernels as an Arai-Agui-Nakajima flow graph with
// the multiply-adds fused -- 30 operations per 8-point transform instead of 35 (the kernels that launch it are bound
// by their instruction count).  Output k is JPEGX_AAN_G[k] times the un-normalised DCT-II coefficient of
// transforms.py:4-11; the host folds 1 / (g_k g_l) into the quantiser's reciprocals (jpegx_internal.h:
// scale_for_aan), so the quantised integers are those of the plain form wherever the bound lets the fast tier decide.
// The bound's factors F(k, l) -- |v rq' - t64| <= F u S |rq'| -- are not counted by hand any more: tests/derive_bounds.py
// runs this very flow graph symbolically (standard model of rounding, first order) and prints them; the kernels
// keep E * level for 15 levels, every F rounded UP to its level (levels are multiples of 1/64).
// tests/test_host_properties.py re-derives the factors and checks that every level here covers them; tests/test_emul.py
// checks the bound against observed errors on adversarial blocks.
// ---------------------------------------------------------------------------------------------
#define JPEGX_AAN_G \
    1, 1.9615705608064609, 1.8477590650225733, 1.6629392246050902, 1.4142135623730947, 1.1111404660392048, 0.76536686473017901, 0.3901806440322565
#define JPEGX_A2M6 0.54119610014619698440f   /* cos(2 pi/16) - cos(6 pi/16) */
#define JPEGX_A2P6 1.30656296487637652786f   /* cos(2 pi/16) + cos(6 pi/16) */

JPEGX_HD void jpegx_dct8_aan_f32(float &x0, float &x1, float &x2, float &x3, float &x4, float &x5, float &x6, float &x7)
{
    const float t0 = x0 + x7, t7 = x0 - x7, t1 = x1 + x6, t6 = x1 - x6;
    const float t2 = x2 + x5, t5 = x2 - x5, t3 = x3 + x4, t4 = x3 - x4;
    const float t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    x0 = t10 + t11;
    x4 = t10 - t11;
    const float s = t12 + t13;
    x2 = fmaf(s, JPEGX_A4, t13);
    x6 = fmaf(s, -JPEGX_A4, t13);
    const float a10 = t4 + t5, a11 = t5 + t6, a12 = t6 + t7;
    const float z5 = (a10 - a12) * JPEGX_A6;
    const float z2 = fmaf(a10, JPEGX_A2M6, z5);
    const float z4 = fmaf(a12, JPEGX_A2P6, z5);
    const float z11 = fmaf(a11, JPEGX_A4, t7), z13 = fmaf(a11, -JPEGX_A4, t7);
    x5 = z13 + z2;
    x3 = z13 - z2;
    x1 = z11 + z4;
    x7 = z11 - z4;
}

// 2-D forward on a block held as v[row*8+col]: rows first, then columns; v[k*8+l] = g_k g_l * coefficient (k, l)
JPEGX_HD void jpegx_dct8x8_aan_f32(float (&v)[64])
{
#pragma unroll
    for (int i = 0; i < 8; ++i)
        jpegx_dct8_aan_f32(v[i * 8 + 0], v[i * 8 + 1], v[i * 8 + 2], v[i * 8 + 3],
                           v[i * 8 + 4], v[i * 8 + 5], v[i * 8 + 6], v[i * 8 + 7]);
#pragma unroll
    for (int j = 0; j < 8; ++j)
        jpegx_dct8_aan_f32(v[0 * 8 + j], v[1 * 8 + j], v[2 * 8 + j], v[3 * 8 + j],
                           v[4 * 8 + j], v[5 * 8 + j], v[6 * 8 + j], v[7 * 8 + j]);
}

constexpr int JPEGX_AAN_LEVELS = 15;
struct JpegxAanLevels { float level[JPEGX_AAN_LEVELS]; unsigned char index[64]; };
constexpr JpegxAanLevels jpegx_aan_levels_pixel = {
    {1.000000f, 2.125000f, 4.312500f, 5.015625f, 5.671875f, 6.921875f, 8.015625f, 10.109375f, 11.953125f, 13.625000f, 18.203125f, 22.453125f, 26.375000f, 30.718750f, 35.640625f},
    {0, 9, 7, 8, 0, 6, 2, 4, 6, 14, 12, 13, 6, 11, 8, 8, 2, 12, 10, 11, 2, 10, 7, 7, 5, 13, 11, 12, 5, 10, 7, 7, 0, 9, 7, 8, 0, 6, 2, 4, 3, 11, 10, 10, 3, 8, 5, 5, 1, 9, 8, 8, 1, 6, 3, 2, 3, 11, 10, 10, 3, 8, 6, 2}};
constexpr JpegxAanLevels jpegx_aan_levels_generic = {
    {4.515625f, 7.000000f, 7.968750f, 10.843750f, 12.453125f, 13.375000f, 15.718750f, 18.390625f, 18.812500f, 21.921875f, 25.703125f, 28.671875f, 34.093750f, 37.734375f, 44.812500f},
    {1, 7, 5, 6, 1, 3, 1, 3, 7, 14, 12, 13, 7, 10, 6, 9, 5, 12, 10, 11, 5, 9, 4, 8, 6, 13, 11, 12, 6, 9, 5, 8, 1, 7, 5, 6, 1, 3, 1, 3, 3, 10, 9, 9, 3, 6, 3, 4, 1, 6, 4, 5, 1, 3, 1, 2, 3, 9, 8, 8, 3, 4, 2, 0}};
// level index of coefficient n = k * 8 + l
constexpr int jpegx_aan_level_index(int n, bool pixel_input)
{
    return pixel_input ? jpegx_aan_levels_pixel.index[n] : jpegx_aan_levels_generic.index[n];
}
constexpr float jpegx_aan_level(int j, bool pixel_input)
{
    return pixel_input ? jpegx_aan_levels_pixel.level[j] : jpegx_aan_levels_generic.level[j];
}

// Rigorous bound on |fp32 fast-tier coefficient - float64 reference coefficient| for a block
// whose samples have absolute sum S (any fp32 inputs).  Derivation in DESIGN.md ("error
// bound"): every intermediate of either pass is bounded by the running absolute sums, each
// 1-D output sees at most 6 roundings on a term's path (1 butterfly add, the fp32 rounding of
// the cosine, <=4 mul/fma roundings), so pass 1 errs by <= 6 u R_i per row (R_i = row abs sum);
// pass 2 propagates sum_i |C[k][i]| 6 u R_i <= 6 u S and adds <= 6 u S of its own: 12 u S with
// u = 2^-24; the quantiser's fp32 reciprocal and product add <= 2 u S.  First order 14 u S;
// 16 u S = 2^-20 S is used (margin for second-order terms).
JPEGX_HD float jpegx_fwd_err_bound(float S) { return S * 0x1p-20f; }

// The same bound per coefficient (round 2).  Counting the roundings on the worst term path of output k of
// one 1-D pass of jpegx_dct8_f32 (butterfly adds, the fp32 rounding of the cosine, mul / fma roundings):
//   k = 0: 3 (three adds)   k = 4: 5   k = 2, 6: 5   k odd: 6
// and when the pass's inputs are multiples of 2^-8 below 2^9 (the PIXEL promise: the adds and subtractions
// of the butterflies are then exact in fp32, sums stay below 2^12 with 2^-8 granularity):
//   k = 0: 0                k = 4: 2   k = 2, 6: 3   k odd: 5.
// Coefficient (k, l) goes through the row pass (output index l; exact adds only there, its outputs are
// arbitrary reals) and the column pass (output index k), each erring by at most (roundings) u S because
// |C| <= 1 and every intermediate is bounded by the abs sums; the fp32 reciprocal and the product of the
// quantiser add 2 u S (1 u S since round 3, see below).  So |t32 - t64| <= F(k, l) u S |1/q| with
// F = n_row(l) + n_col(k) + 2 between 5 and 14 (average 10.75 for pixel input) instead of the uniform 16; 2^-10
// relative margin covers the O(u^2) terms and the fp32 evaluation of S and of the bound itself.
constexpr int jpegx_dct8_roundings(int k, bool exact_adds)
{
    return k == 0 ? (exact_adds ? 0 : 3) : ((k & 1) ? (exact_adds ? 5 : 6) : (k == 4 ? (exact_adds ? 2 : 5) : (exact_adds ? 3 : 5)));
}
// Round 3: (a) the quantiser term is 1, not 2 -- jpegx_quant_fast rounds the EXACT product v * rq, so only the fp32
// rounding of the reciprocal is left (the rounding to the integer is the decision itself, and the rounding of the
// distance d is absorbed by JPEGX_SAFE_HALF); (b) with pixel input the row pass's output l = 0 is the exact sum of
// eight samples, so for the coefficients (k, 0) of the first column the COLUMN pass works on exact integers too and
// its butterfly adds are exact as well -- those are the coefficients with the largest multipliers 1/q, i.e. the
// ones that raise most flags.
constexpr int jpegx_fwd_roundings(int n, bool pixel_input)   // n = k * 8 + l
{
    return jpegx_dct8_roundings(n & 7, pixel_input) + jpegx_dct8_roundings(n >> 3, pixel_input && (n & 7) == 0) + 1;
}
JPEGX_HD float jpegx_fwd_err_unit(float S) { return S * 0x1.004p-24f; }   // u S (1 + 2^-10)

// Fast-tier quantiser (round 3): round-half-even of the EXACT product v * rq by the magic-constant addition
//   m = fma(v, rq, 1.5 * 2^23)      one rounding, at integer granularity: m = magic + rint(v * rq) for |v rq| < 2^22
//   r = m - magic                   exact
//   d = fma(v, rq, -r)              the distance to that integer, one rounding (relative 2^-24)
// instead of t = v * rq; r = rint(t); t - r: no v_rndne, no v_cvt_i32 (both issue at 2/3 of the fma rate on
// gfx950, profiles/r03_valu_rate.txt), and the int16 is the low half of m's bit pattern (two's complement), so two
// coefficients pack with one byte permute.  The product's own rounding of the old form is gone, the bound
// F(k, l) u S |rq| stays (it still charges that rounding): a coefficient is safe iff |d| + bound < JPEGX_SAFE_HALF,
// which leaves 2^-23 for d's rounding; then rint(t64) = r whatever the magnitude (for |v rq| >= 2^22, where m is no
// longer integer-grained on the negative side, every surviving r is beyond the int16 range and saturates).
#define JPEGX_RMAGIC 12582912.0f          /* 1.5 * 2^23 */
#define JPEGX_SAFE_HALF 0.49999988f       /* 0.5 - 2^-23 */
// `magic` = JPEGX_RMAGIC, handed in by the caller: the kernels keep it in ONE register the optimiser cannot see
// through -- as a literal it is re-materialised in front of every fma (v_mov + v_fmac instead of one v_fma)
JPEGX_HD float jpegx_quant_fast_m(float v, float rq, float magic, float &d)
{
    const float m = fmaf(v, rq, magic);
    const float r = m - magic;
    d = fmaf(v, rq, -r);
    return m;
}
JPEGX_HD float jpegx_quant_fast(float v, float rq, float &d) { return jpegx_quant_fast_m(v, rq, JPEGX_RMAGIC, d); }
// the quantised integer held by m (valid while |r| < 2^22)
JPEGX_HD int jpegx_quant_fast_int(float m) { return (int)(m - JPEGX_RMAGIC); }

// Same for the inverse.  A 1-D inverse pass (jpegx_idct8_f32) scales its k = 0 input by exactly 1/8 (a
// power of two; that term only meets the 3 roundings of the final additions) and every k >= 1 input by
// c/4 with |c| <= 1 (<= 6 roundings on its path: the fp32 rounding of the constant, the product, three
// fma's, the final addition).  A coefficient Z[k][l] reaches an output sample through the column pass
// (index k) and then the row pass (index l), so its share of |x32 - x64| is bounded by
//   |Z00|            * 1/8 * 1/8 * (3 + 3) u  = 3/32 u   (D  = |Z00|)
//   |Z0l|, |Zk0|     * 1/8 * 1/4 * (3 + 6) u  = 9/32 u   (A1 = abs sum of the 14 other first-row/column terms)
//   |Zkl|, k, l >= 1 * 1/4 * 1/4 * (6 + 6) u  = 3/4  u   (A2 = abs sum of the 49 inner terms)
// (pass 2 works on the computed pass-1 values; the products of two error terms are O(u^2 A), far below
// the 2^-10 relative margin added here).  `dq` = number of roundings in the fp32 dequantisation: 0 when
// z * q is exact in fp32 (modes none / discard / qtable: integers below 2^24), 2 for mode divide (the
// fp32 rounding of the divisor and of the product); each one adds 1/64, 1/32, 1/16 u to the three classes.
JPEGX_HD float jpegx_inv_err_bound(float D, float A1, float A2, float dq)
{
    const float u = 0x1.004p-24f;   // 2^-24 (1 + 2^-10)
    const float cD = fmaf(dq, 1.0f / 64, 3.0f / 32), c1 = fmaf(dq, 1.0f / 32, 9.0f / 32), c2 = fmaf(dq, 1.0f / 16, 3.0f / 4);
    return u * fmaf(D, cD, fmaf(A1, c1, A2 * c2));
}

// The same per input index (round 3).  Roundings on the path from input k of jpegx_idct8_f32 to any of its outputs
// (the fp32 rounding of the constant, the product, the fma's it passes, the additions g, E and the final one):
//   k = 0: 3 (x0 / 8 is exact)   k = 4: 5   k = 2: 5   k = 6: 4   k = 1: 6   k = 3: 5   k = 5: 4   k = 7: 3
// -- the later an input enters a fma chain, the fewer roundings see it.  With s_0 = 1/8 and s_k = 1/4 the share of
// coefficient Z[k][l] (column pass index k, then row pass index l) in |x32 - x64| is at most
//   |d_kl| s_k s_l (n_k + n_l + dq) u,      d = the dequantised coefficient, dq as above,
// so the bound is ONE weighted absolute sum, accumulated with one fma per coefficient where the three-class form
// used one add: same instruction count, 24 % smaller on average over the 49 inner terms (n_k + n_l averages 9.1
// instead of 12), hence fewer rows for the float64 tier.
constexpr int jpegx_idct8_roundings(int k)
{
    return k == 0 ? 3 : (k == 1 ? 6 : ((k == 2 || k == 3 || k == 4) ? 5 : ((k == 5 || k == 6) ? 4 : 3)));
}
// Round 3, second step: the weights themselves come from tests/derive_bounds.py, which runs jpegx_idct8_f32 symbolically
// (standard model of rounding, first order): the hand count above is what it reproduces as an upper bound, the
// derived weights are 14 percent smaller in sum (most for k or l = 4, whose constant is applied once).  Multiples of
// 2^-12, rounded up.  The dequantisation's own roundings add dq s_k s_l.
struct JpegxInvWeights { float w[64]; };
constexpr JpegxInvWeights jpegx_inv_weights = {{
    0.0937500000f, 0.2758789062f, 0.2312011719f, 0.2453613281f, 0.1770019531f, 0.2145996094f, 0.2021484375f, 0.1840820312f,
    0.2758789062f, 0.7216796875f, 0.6230468750f, 0.6613769531f, 0.4768066406f, 0.6013183594f, 0.5664062500f, 0.5412597656f,
    0.2312011719f, 0.6230468750f, 0.5336914062f, 0.5664062500f, 0.4084472656f, 0.5097656250f, 0.4802246094f, 0.4531250000f,
    0.2453613281f, 0.6613769531f, 0.5664062500f, 0.6013183594f, 0.4335937500f, 0.5412597656f, 0.5097656250f, 0.4812011719f,
    0.1770019531f, 0.4768066406f, 0.4084472656f, 0.4335937500f, 0.3125000000f, 0.3901367188f, 0.3676757812f, 0.3469238281f,
    0.2145996094f, 0.6013183594f, 0.5097656250f, 0.5412597656f, 0.3901367188f, 0.4812011719f, 0.4531250000f, 0.4208984375f,
    0.2021484375f, 0.5664062500f, 0.4802246094f, 0.5097656250f, 0.3676757812f, 0.4531250000f, 0.4270019531f, 0.3964843750f,
    0.1840820312f, 0.5412597656f, 0.4531250000f, 0.4812011719f, 0.3469238281f, 0.4208984375f, 0.3964843750f, 0.3608398438f}};
constexpr float jpegx_inv_weight(int n, int dq)   // n = k * 8 + l
{
    return jpegx_inv_weights.w[n] + ((n >> 3) == 0 ? 0.125f : 0.25f) * ((n & 7) == 0 ? 0.125f : 0.25f) * (float)dq;
}
JPEGX_HD float jpegx_inv_err_from_weighted_sum(float A) { return A * 0x1.004p-24f; }   // u (1 + 2^-10) A

// ---------------------------------------------------------------------------------------------
// fp64 exact tier (reference operation order)
// ---------------------------------------------------------------------------------------------

// C[k] . x with x strided: a_n = c_n x_n (n<4); a_n = fma(c_{n+4}, x_{n+4}, a_n); (a0+a2)+(a1+a3)
JPEGX_HD double jpegx_dot8_ref(const double *c, const double *x, int sx)
{
    double a0 = c[0] * x[0 * sx];
    double a1 = c[1] * x[1 * sx];
    double a2 = c[2] * x[2 * sx];
    double a3 = c[3] * x[3 * sx];
    a0 = fma(c[4], x[4 * sx], a0);
    a1 = fma(c[5], x[5 * sx], a1);
    a2 = fma(c[6], x[6 * sx], a2);
    a3 = fma(c[7], x[7 * sx], a3);
    return (a0 + a2) + (a1 + a3);
}

// Output sample i of Cn^T . u where u = Dinv * X already applied; cn_col[j] = Cn[j][i].
// p = w1 u1; p = fma(w0,u0,p); fma(w2,u2,p); fma(w3,u3,p); q likewise over 5,4,6,7; p + q.
JPEGX_HD double jpegx_idot8_ref(const double *w, const double *u, int su)
{
    double p = w[1] * u[1 * su];
    p = fma(w[0], u[0 * su], p);
    p = fma(w[2], u[2 * su], p);
    p = fma(w[3], u[3 * su], p);
    double q = w[5] * u[5 * su];
    q = fma(w[4], u[4 * su], q);
    q = fma(w[6], u[6 * su], q);
    q = fma(w[7], u[7 * su], q);
    return p + q;
}

// quantiser modes; numbering is part of the C ABI (include/jpegx.h)
enum { JPEGX_QM_NONE = 0, JPEGX_QM_DISCARD = 1, JPEGX_QM_DIVIDE = 2, JPEGX_QM_QTABLE = 3 };

// quantizers.py:4-49 on one float64 coefficient at block position n = i*8+j.
// rq64 = 1.0/q (correctly rounded IEEE division, what NumPy computes at quantizers.py:49).
JPEGX_HD double jpegx_quant_ref(double a, int n, int mode, double param, const double *rq64)
{
    if (mode == JPEGX_QM_QTABLE) return rint(a * rq64[n]);
    if (mode == JPEGX_QM_DIVIDE) return rint(a / param);
    if (mode == JPEGX_QM_DISCARD) {
        const int keep = (int)param;
        return ((n >> 3) >= keep || (n & 7) >= keep) ? 0.0 : rint(a);
    }
    return rint(a);
}

// the same with the table entry 1.0/q of position n already at hand (the kernels fetch it once per lane)
JPEGX_HD double jpegx_quant_lane(double a, int n, int mode, double param, double rq_n)
{
    if (mode == JPEGX_QM_QTABLE) return rint(a * rq_n);
    if (mode == JPEGX_QM_DIVIDE) return rint(a / param);
    if (mode == JPEGX_QM_DISCARD) {
        const int keep = (int)param;
        return ((n >> 3) >= keep || (n & 7) >= keep) ? 0.0 : rint(a);
    }
    return rint(a);
}

// quantizers.py:8-9,30-31,51-53 (restore)
JPEGX_HD double jpegx_restore_ref(double a, int n, int mode, double param, const int *qt)
{
    if (mode == JPEGX_QM_QTABLE) return rint(a * (double)qt[n]);
    if (mode == JPEGX_QM_DIVIDE) return a * param;
    return a;
}

JPEGX_HD int jpegx_clamp_i16(double r)
{
    return r > 32767.0 ? 32767 : (r < -32768.0 ? -32768 : (int)r);
}

// ---------------------------------------------------------------------------------------------
// synthetic planes (jpegx/synth.py is the host twin)
// ---------------------------------------------------------------------------------------------
JPEGX_HD uint32_t jpegx_hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

JPEGX_HD uint32_t jpegx_synth_pixel(int kind, uint32_t pseed, uint32_t width, uint32_t y, uint32_t x)
{
    const uint32_t h = jpegx_hash32((y * width + x) ^ pseed);
    if (kind == 0) return h >> 24;
    return (((3u * x + 5u * y) >> 4) & 127u) + 64u * (((x >> 6) ^ (y >> 6)) & 1u) + (h & 15u);
}
