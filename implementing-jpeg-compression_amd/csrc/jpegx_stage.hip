// jpegx_stage.hip -- the per-stage kernels behind the stand-alone step classes (fp32 DCT/IDCT,
// exact float64 DCT/IDCT/quantise/restore, zigzag permutation), the synthetic plane generator and
// their C entry points.  Part of libjpegx.so (C ABI: include/jpegx.h).
// Built with: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (explicit fma only).
#include "jpegx_internal.h"

namespace {

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

// SubSampling.execute (pipeline/subsampling.py:9-11) for ANY block_size: np.mean over bs x bs tiles of an
// integer-valued band = the exact sum (integers, far below 2^53) divided once in float64.  One thread per
// output sample; a wave reads 64 * bs contiguous samples of each input row.
template <typename T>
__global__ void k_mean_pool_f64(const T *__restrict__ in, size_t pitch, int H, int W, int bs, double *__restrict__ out,
                                size_t opitch)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    const T *p = in + (size_t)y * bs * pitch + (size_t)x * bs;
    double s = 0.0;
    for (int u = 0; u < bs; ++u)
        for (int w = 0; w < bs; ++w) s += (double)p[(size_t)u * pitch + w];
    out[(size_t)y * opitch + x] = s / (double)(bs * bs);
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

// np.dstack of Jpeg.decompress (pipeline/__init__.py:119-122) on the device: `nb` uint8 planes [rows][pitch] ->
// pixel-interleaved [rows][cols][nb].  One thread = four pixels of one row: a dword from every plane, 4 * nb
// bytes out (whole dwords when the packed rows keep them aligned).
struct InterleaveArgs { const unsigned char *plane[JPEGX_MAX_IMAGE_BANDS]; };

__global__ __launch_bounds__(256) void k_interleave_u8(InterleaveArgs a, int nb, int rows, int cols, size_t pitch,
                                                       unsigned char *__restrict__ out, size_t out_pitch)
{
    const int x4 = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    const int x = x4 * 4;
    if (x >= cols || y >= rows) return;
    unsigned w[JPEGX_MAX_IMAGE_BANDS];
#pragma unroll
    for (int k = 0; k < JPEGX_MAX_IMAGE_BANDS; ++k)
        w[k] = k < nb ? *reinterpret_cast<const unsigned *>(a.plane[k] + (size_t)y * pitch + x) : 0u;   // pitch is a multiple of 16
    unsigned char *o = out + (size_t)y * out_pitch + (size_t)x * nb;
    const int npx = min(4, cols - x);
    if (nb == 3 && npx == 4 && (out_pitch & 3) == 0) {
        const unsigned b0 = w[0], b1 = w[1], b2 = w[2];
        unsigned *o4 = reinterpret_cast<unsigned *>(o);
        o4[0] = (b0 & 0xFFu) | ((b1 & 0xFFu) << 8) | ((b2 & 0xFFu) << 16) | ((b0 & 0xFF00u) << 16);
        o4[1] = ((b1 >> 8) & 0xFFu) | (((b2 >> 8) & 0xFFu) << 8) | (b0 & 0xFF0000u) | ((b1 & 0xFF0000u) << 8);
        o4[2] = ((b2 >> 16) & 0xFFu) | ((b0 >> 24) << 8) | ((b1 >> 24) << 16) | (b2 & 0xFF000000u);
        return;
    }
    for (int i = 0; i < npx; ++i)
        for (int k = 0; k < nb; ++k) o[i * nb + k] = (unsigned char)(w[k] >> (8 * i));
}

// The way in (Jpeg.compress, pipeline/__init__.py:102-106: `image.split()`): [rows][cols][nb] packed pixels -> nb planes.
// A thread takes four pixels: 4 nb bytes in (three dwords when nb = 3 and the row keeps them aligned), one dword per plane out.
struct DeinterleaveArgs { unsigned char *plane[JPEGX_MAX_IMAGE_BANDS]; };

__global__ __launch_bounds__(256) void k_deinterleave_u8(const unsigned char *__restrict__ in, size_t in_pitch, int nb, int rows, int cols,
                                                         DeinterleaveArgs a, size_t pitch)
{
    const int x4 = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    const int x = x4 * 4;
    if (x >= cols || y >= rows) return;
    const unsigned char *p = in + (size_t)y * in_pitch + (size_t)x * nb;
    const int npx = min(4, cols - x);
    unsigned w[JPEGX_MAX_IMAGE_BANDS] = {};
    if (nb == 3 && npx == 4 && (in_pitch & 3) == 0 && (reinterpret_cast<uintptr_t>(in) & 3u) == 0) {
        const unsigned *p4 = reinterpret_cast<const unsigned *>(p);
        const unsigned d0 = p4[0], d1 = p4[1], d2 = p4[2];       // bytes: a0 b0 c0 a1 | b1 c1 a2 b2 | c2 a3 b3 c3
        w[0] = (d0 & 0xFFu) | ((d0 >> 24) << 8) | (((d1 >> 16) & 0xFFu) << 16) | (((d2 >> 8) & 0xFFu) << 24);
        w[1] = ((d0 >> 8) & 0xFFu) | ((d1 & 0xFFu) << 8) | ((d1 >> 24) << 16) | (((d2 >> 16) & 0xFFu) << 24);
        w[2] = ((d0 >> 16) & 0xFFu) | (((d1 >> 8) & 0xFFu) << 8) | ((d2 & 0xFFu) << 16) | ((d2 >> 24) << 24);
    } else {
        for (int i = 0; i < npx; ++i)
            for (int k = 0; k < nb; ++k) w[k] |= (unsigned)p[i * nb + k] << (8 * i);
    }
#pragma unroll
    for (int k = 0; k < JPEGX_MAX_IMAGE_BANDS; ++k)
        if (k < nb) *reinterpret_cast<unsigned *>(a.plane[k] + (size_t)y * pitch + x) = w[k];      // pitch is a multiple of 4 covering the row rounded up to 4
}

}  // namespace

extern "C" {

int jpegx_deinterleave_u8(const uint8_t *d_in, ptrdiff_t in_pitch, int nbands, int rows, int cols, void *const *d_planes, ptrdiff_t pitch,
                          jpegx_stream_t stream)
{
    if (!d_planes || !d_in) return fail(JPEGX_E_INVALID, "null device pointer");
    if (nbands < 1 || nbands > JPEGX_MAX_IMAGE_BANDS || rows <= 0 || cols <= 0 || rows > 65535)
        return fail(JPEGX_E_INVALID, "deinterleave: 1..JPEGX_MAX_IMAGE_BANDS planes, 1..65535 rows");
    if ((pitch % 4) != 0 || pitch < ((cols + 3) & ~3) || in_pitch < (ptrdiff_t)cols * nbands)
        return fail(JPEGX_E_INVALID, "deinterleave: plane pitch must be a multiple of 4 covering the row rounded up to 4; packed pitch >= cols * nbands");
    DeinterleaveArgs a;
    for (int k = 0; k < JPEGX_MAX_IMAGE_BANDS; ++k) {
        a.plane[k] = k < nbands ? static_cast<unsigned char *>(d_planes[k]) : nullptr;
        if (k < nbands && (!d_planes[k] || (reinterpret_cast<uintptr_t>(d_planes[k]) & 3u))) return fail(JPEGX_E_INVALID, "deinterleave: planes must be 4-byte aligned");
    }
    const dim3 block(256), grid(((cols + 3) / 4 + 255) / 256, rows);
    hipLaunchKernelGGL(k_deinterleave_u8, grid, block, 0, (hipStream_t)stream, d_in, (size_t)in_pitch, nbands, rows, cols, a, (size_t)pitch);
    HIP_TRY(hipGetLastError());
    return JPEGX_OK;
}

int jpegx_interleave_u8(const void *const *d_planes, int nbands, int rows, int cols, ptrdiff_t pitch, uint8_t *d_out,
                        ptrdiff_t out_pitch, jpegx_stream_t stream)
{
    if (!d_planes || !d_out) return fail(JPEGX_E_INVALID, "null device pointer");
    if (nbands < 1 || nbands > JPEGX_MAX_IMAGE_BANDS || rows <= 0 || cols <= 0 || rows > 65535)
        return fail(JPEGX_E_INVALID, "interleave: 1..JPEGX_MAX_IMAGE_BANDS planes, 1..65535 rows");
    if ((pitch % 4) != 0 || pitch < ((cols + 3) & ~3) || out_pitch < (ptrdiff_t)cols * nbands)
        return fail(JPEGX_E_INVALID, "interleave: plane pitch must be a multiple of 4 covering the row rounded up to 4; packed pitch >= cols * nbands");
    InterleaveArgs a;
    for (int k = 0; k < JPEGX_MAX_IMAGE_BANDS; ++k) {
        a.plane[k] = k < nbands ? static_cast<const unsigned char *>(d_planes[k]) : nullptr;
        if (k < nbands && (!d_planes[k] || (reinterpret_cast<uintptr_t>(d_planes[k]) & 3u))) return fail(JPEGX_E_INVALID, "interleave: planes must be 4-byte aligned");
    }
    const dim3 block(256), grid(((cols + 3) / 4 + 255) / 256, rows);
    hipLaunchKernelGGL(k_interleave_u8, grid, block, 0, (hipStream_t)stream, a, nbands, rows, cols, (size_t)pitch, d_out, (size_t)out_pitch);
    HIP_TRY(hipGetLastError());
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
int jpegx_mean_pool_f64(const void *d_in, int elem_size, int H, int W, ptrdiff_t pitch, int bs, double *d_out,
                        ptrdiff_t out_pitch, jpegx_stream_t stream)
{
    if (!d_in || !d_out) return fail(JPEGX_E_INVALID, "null device pointer");
    if (H <= 0 || W <= 0 || bs < 1 || bs > 4096) return fail(JPEGX_E_INVALID, "mean_pool: bad shape or block_size");
    if (pitch < (ptrdiff_t)W * bs || out_pitch < W) return fail(JPEGX_E_INVALID, "pitch smaller than the row");
    if (H > 65535) return fail(JPEGX_E_UNSUPPORTED, "mean_pool: more than 65535 output rows in one launch");
    const dim3 block(256), grid((W + 255) / 256, H);
    hipStream_t st = (hipStream_t)stream;
    if (elem_size == 1)
        hipLaunchKernelGGL((k_mean_pool_f64<uint8_t>), grid, block, 0, st, (const uint8_t *)d_in, (size_t)pitch, H, W, bs, d_out, (size_t)out_pitch);
    else if (elem_size == 4)
        hipLaunchKernelGGL((k_mean_pool_f64<float>), grid, block, 0, st, (const float *)d_in, (size_t)pitch, H, W, bs, d_out, (size_t)out_pitch);
    else
        return fail(JPEGX_E_UNSUPPORTED, "mean_pool takes uint8 (elem_size 1) or integer-valued fp32 (elem_size 4) samples");
    HIP_TRY(hipGetLastError());
    return JPEGX_OK;
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
