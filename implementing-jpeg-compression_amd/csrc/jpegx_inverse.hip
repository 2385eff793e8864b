// jpegx_inverse.hip -- fused inverse kernel (un-zigzag + dequantise + IDCT + round [+ clamp]
// [+ up-sampling]) and its C entry points.  Part of libjpegx.so (C ABI: include/jpegx.h).
// Built with: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (explicit fma only).
#include "jpegx_internal.h"

namespace {

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
