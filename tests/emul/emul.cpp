// tests/emul/emul.cpp -- host build of the kernels' per-block arithmetic (csrc/jpegx_math.h).
//
// TEST-ONLY: lets the CPU test-suite check the fp32 fast tier, its error bound and the
// fp64 exact tier against the oracle before anything is run on a GPU.  It mirrors the
// per-block decision logic of k_forward_fused / k_inverse_fused (flag a coefficient whose
// fp32 value is within the bound of a rounding boundary, recompute exactly that one in the
// reference order) but has none of the kernels' data movement.  Never used by the product.
#include "../../implementing-jpeg-compression_amd/csrc/jpegx_math.h"
#include <stddef.h>

static const double T_C[64] = { JPEGX_TABLE_DCT_MATRIX };
static const double T_CN[64] = { JPEGX_TABLE_DCT_NORMALIZED };
static const double T_DINV[8] = { JPEGX_TABLE_NORM_DIAG };
static const int T_QT[64] = { JPEGX_TABLE_QTABLE };
static const int T_ZZ[64] = { JPEGX_TABLE_ZIGZAG8 };

extern "C" {

// the bound tables the kernels are compiled with (tests/test_emul.py checks them against tests/derive_bounds.py)
float emul_aan_level_of(int n, int pixel) { return jpegx_aan_level(jpegx_aan_level_index(n, pixel != 0), pixel != 0); }
float emul_inv_weight(int n, int dq) { return jpegx_inv_weight(n, dq); }
double emul_aan_g(int k) { static const double G[8] = {JPEGX_AAN_G}; return G[k]; }

// stats[0] = flagged coefficients, stats[1] = blocks with >=1 flag,
// stats[2] = max over coefficients of observed |t32 - t64| / (E F / q), the kernel's per-coefficient bound
int emul_forward(const float *in, int H, int W, int mode, double param, const float *rq32,
                 int pixel_input, int dc_exact, int16_t *out, float *out_dct32, double *stats)
{
    double rq64[64];
    for (int n = 0; n < 64; ++n) rq64[n] = 1.0 / (double)T_QT[n];
    int wb = W / 8;
    long nflag = 0, nblkflag = 0;
    double maxratio = 0.0;
    for (int by = 0; by < H / 8; ++by)
        for (int bx = 0; bx < wb; ++bx) {
            float v[64];
            double a[64];
            float S = 0.f;
            for (int i = 0; i < 8; ++i)
                for (int j = 0; j < 8; ++j) {
                    float x = in[(size_t)(by * 8 + i) * W + bx * 8 + j];
                    v[i * 8 + j] = x; a[i * 8 + j] = (double)x;
                    S += fabsf(x);
                }
            jpegx_dct8x8_aan_f32(v);                 // scaled: v[k*8+l] = g_k g_l * coefficient (k, l)
            if (pixel_input) S = v[0];
            const float E = jpegx_fwd_err_unit(S);
            // exact tier for everything (for statistics only)
            double m[64], y64[64];
            for (int i = 0; i < 8; ++i)
                for (int l = 0; l < 8; ++l) m[i * 8 + l] = jpegx_dot8_ref(&T_C[l * 8], &a[i * 8], 1);
            for (int k = 0; k < 8; ++k)
                for (int l = 0; l < 8; ++l) y64[k * 8 + l] = jpegx_dot8_ref(&T_C[k * 8], &m[l], 8);
            int16_t *o = out + ((size_t)by * wb + bx) * 64;
            int blkflag = 0;
            for (int p = 0; p < 64; ++p) {
                int n = T_ZZ[p];
                static const double G[8] = {JPEGX_AAN_G};
                if (out_dct32) out_dct32[(size_t)(by * 8 + (n >> 3)) * W + bx * 8 + (n & 7)] = (float)((double)v[n] / (G[n >> 3] * G[n & 7]));
                const float F = jpegx_aan_level(jpegx_aan_level_index(n, pixel_input != 0), pixel_input != 0);
                /* observed |v rq - t64| against the kernel's bound E F |1/q| (v rq: the exact product the fast tier
                   rounds, jpegx_quant_fast; the bound still charges a product rounding that is no longer made) */
                double t = (double)v[n] * (double)rq32[n];
                double t64 = mode == JPEGX_QM_QTABLE ? y64[n] * rq64[n] : (mode == JPEGX_QM_DIVIDE ? y64[n] / param : y64[n]);
                double bound = (double)E * F * fabs((double)rq32[n]);
                double ratio = bound > 0 ? fabs(t - t64) / bound : 0.0;
                if (rq32[n] != 0.f && ratio > maxratio) maxratio = ratio;
                float d;
                const float mg = jpegx_quant_fast(v[n], rq32[n], d);
                const float r = mg - JPEGX_RMAGIC;
                float g = fmaf(E * F, fabsf(rq32[n]), fabsf(d));
                int flag = !(g < JPEGX_SAFE_HALF);
                if (dc_exact && n == 0) flag = 0;
                int res;
                if (flag) {
                    ++nflag; blkflag = 1;
                    res = jpegx_clamp_i16(jpegx_quant_ref(y64[n], n, mode, param, rq64));
                } else {
                    res = jpegx_clamp_i16((double)r);
                }
                o[p] = (int16_t)res;
            }
            nblkflag += blkflag;
        }
    if (stats) { stats[0] = (double)nflag; stats[1] = (double)nblkflag; stats[2] = maxratio; }
    return 0;
}

// inverse: int16 zigzag -> rounded int32 plane (unclamped), same two-tier logic
int emul_inverse(const int16_t *in, int H, int W, int mode, double param, int32_t *out, double *stats)
{
    int wb = W / 8;
    long nflag = 0, nblkflag = 0;
    double maxratio = 0.0;
    for (int by = 0; by < H / 8; ++by)
        for (int bx = 0; bx < wb; ++bx) {
            const int16_t *z = in + ((size_t)by * wb + bx) * 64;
            float v[64]; double a[64]; float A = 0.f;
            for (int p = 0; p < 64; ++p) {
                int n = T_ZZ[p];
                double d = jpegx_restore_ref((double)z[p], n, mode, param, T_QT);
                a[n] = d;
                // the kernel's fp32 dequantisation: (float)z * fp32 multiplier (k_inverse_fused)
                v[n] = (mode == JPEGX_QM_QTABLE) ? (float)z[p] * (float)T_QT[n]
                       : (mode == JPEGX_QM_DIVIDE ? (float)z[p] * (float)param : (float)z[p]);
            }
            // the kernel's weighted absolute sum, in its order (zigzag order of arrival, then the divide-mode extra)
            for (int p = 0; p < 64; ++p) A = fmaf(fabsf(v[T_ZZ[p]]), jpegx_inv_weight(T_ZZ[p], 0), A);
            if (mode == JPEGX_QM_DIVIDE)
                for (int n = 0; n < 64; ++n) A = fmaf(fabsf(v[n]), jpegx_inv_weight(n, 2) - jpegx_inv_weight(n, 0), A);
            const float E = jpegx_inv_err_from_weighted_sum(A);
            jpegx_idct8x8_f32(v);
            double u[8], m[64], y64[64], w[8];
            for (int j = 0; j < 8; ++j) {       // columns first
                for (int k = 0; k < 8; ++k) u[k] = T_DINV[k] * a[k * 8 + j];
                for (int i = 0; i < 8; ++i) {
                    for (int k = 0; k < 8; ++k) w[k] = T_CN[k * 8 + i];
                    m[i * 8 + j] = jpegx_idot8_ref(w, u, 1);
                }
            }
            for (int i = 0; i < 8; ++i) {       // then rows
                for (int k = 0; k < 8; ++k) u[k] = T_DINV[k] * m[i * 8 + k];
                for (int j = 0; j < 8; ++j) {
                    for (int k = 0; k < 8; ++k) w[k] = T_CN[k * 8 + j];
                    y64[i * 8 + j] = jpegx_idot8_ref(w, u, 1);
                }
            }
            int blkflag = 0;
            for (int n = 0; n < 64; ++n) {
                double ratio = fabs((double)v[n] - y64[n]) / (double)(E > 0 ? E : 1e-30f);   /* observed error / bound */
                if (ratio > maxratio) maxratio = ratio;
                float r = rintf(v[n]);
                int flag = !((fabsf(v[n] - r) + E) < JPEGX_SAFE_HALF);
                int res = flag ? (int)rint(y64[n]) : (int)r;
                if (flag) { ++nflag; blkflag = 1; }
                out[(size_t)(by * 8 + (n >> 3)) * W + bx * 8 + (n & 7)] = res;
            }
            nblkflag += blkflag;
        }
    if (stats) { stats[0] = (double)nflag; stats[1] = (double)nblkflag; stats[2] = maxratio; }
    return 0;
}

}  // extern "C"
