// jpegx_on.cpp -- explicit-device forms of the data-path entries ("every entry takes a device index
// and an optional stream handle").  jpegx_<name>_on(device, ...) = jpegx_<name>(...) executed with `device` current
// on the calling thread, whose own current device is restored afterwards: one host thread can drive all the GPUs
// of a node (pointers and streams handed in must belong to `device`).  The plain forms keep working on the
// thread's current device (jpegx_set_device), which is what a one-process-per-GPU job sets once at start.
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "../../include/jpegx.h"

extern "C" void jpegx_internal_set_error(const char *msg);

namespace {
struct DeviceGuard {
    int prev = -1, rc = JPEGX_OK;
    explicit DeviceGuard(int device)
    {
        hipError_t e = hipGetDevice(&prev);
        if (e == hipSuccess && prev != device) e = hipSetDevice(device);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            char buf[200];
            snprintf(buf, sizeof(buf), "cannot make device %d current: %s", device, hipGetErrorString(e));
            jpegx_internal_set_error(buf);
            rc = JPEGX_E_NODEVICE;
            prev = -1;
        } else if (prev == device) {
            prev = -1;                     // nothing to restore
        }
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace

#define JPEGX_ON(NAME, PARAMS, ARGS)                      \
    extern "C" int NAME##_on PARAMS                       \
    {                                                     \
        DeviceGuard g(device);                            \
        if (g.rc) return g.rc;                            \
        return NAME ARGS;                                 \
    }

JPEGX_ON(jpegx_malloc, (int device, void **dptr, size_t bytes), (dptr, bytes))
JPEGX_ON(jpegx_free, (int device, void *dptr), (dptr))
JPEGX_ON(jpegx_stream_create, (int device, jpegx_stream_t *stream), (stream))
JPEGX_ON(jpegx_generate_plane, (int device, float *d_plane, int H, int W, ptrdiff_t pitch, int kind, uint32_t seed, uint32_t plane, int row0, jpegx_stream_t stream),
         (d_plane, H, W, pitch, kind, seed, plane, row0, stream))
JPEGX_ON(jpegx_forward_fused_pooled, (int device, const float *d_in, int H, int W, ptrdiff_t pitch, int bs, int mode, double param, unsigned flags, int16_t *d_out, jpegx_stream_t stream),
         (d_in, H, W, pitch, bs, mode, param, flags, d_out, stream))
JPEGX_ON(jpegx_forward_fused_u8, (int device, const uint8_t *d_in, int H, int W, ptrdiff_t pitch, int bs, int mode, double param, unsigned flags, int16_t *d_out, jpegx_stream_t stream),
         (d_in, H, W, pitch, bs, mode, param, flags, d_out, stream))
JPEGX_ON(jpegx_forward_fused_f64, (int device, const double *d_in, int H, int W, ptrdiff_t pitch, int mode, double param, unsigned flags, int16_t *d_out, jpegx_stream_t stream),
         (d_in, H, W, pitch, mode, param, flags, d_out, stream))
JPEGX_ON(jpegx_forward_fused_planes, (int device, const jpegx_plane_desc *planes, int nplanes, int mode, double param, unsigned flags, jpegx_stream_t stream),
         (planes, nplanes, mode, param, flags, stream))
JPEGX_ON(jpegx_mean_pool_f64, (int device, const void *d_in, int elem_size, int H, int W, ptrdiff_t pitch, int bs, double *d_out, ptrdiff_t out_pitch, jpegx_stream_t stream),
         (d_in, elem_size, H, W, pitch, bs, d_out, out_pitch, stream))
JPEGX_ON(jpegx_inverse_fused_u8_inflated, (int device, const int16_t *d_in, int H, int W, int mode, double param, unsigned flags, int bs, uint8_t *d_out, ptrdiff_t out_pitch, jpegx_stream_t stream),
         (d_in, H, W, mode, param, flags, bs, d_out, out_pitch, stream))
JPEGX_ON(jpegx_entropy_sizes, (int device, const int16_t *d_zz, long long nblocks, void *d_workspace, jpegx_stream_t stream), (d_zz, nblocks, d_workspace, stream))
JPEGX_ON(jpegx_entropy_total, (int device, const void *d_workspace, unsigned long long *h_total, jpegx_stream_t stream), (d_workspace, h_total, stream))
JPEGX_ON(jpegx_entropy_block_sizes, (int device, const void *d_workspace, long long nblocks, uint32_t *h_sizes, jpegx_stream_t stream), (d_workspace, nblocks, h_sizes, stream))
JPEGX_ON(jpegx_entropy_emit, (int device, const int16_t *d_zz, long long nblocks, const void *d_workspace, uint8_t *d_out, jpegx_stream_t stream),
         (d_zz, nblocks, d_workspace, d_out, stream))
JPEGX_ON(jpegx_entropy_decode, (int device, const uint8_t *d_bytes, size_t nbytes, long long nblocks, void *d_workspace, int16_t *d_zz, int level, jpegx_stream_t stream),
         (d_bytes, nbytes, nblocks, d_workspace, d_zz, level, stream))
JPEGX_ON(jpegx_entropy_decode_status, (int device, const void *d_workspace, jpegx_stream_t stream), (d_workspace, stream))
JPEGX_ON(jpegx_host_compress_begin, (int device, const void *h_plane, int elem_size, int H, int W, ptrdiff_t pitch, int bs, int mode, double param, size_t *nbytes),
         (h_plane, elem_size, H, W, pitch, bs, mode, param, nbytes))
JPEGX_ON(jpegx_host_compress_image, (int device, const void *const *h_planes, int nbands, int elem_size, int H, int W, ptrdiff_t pitch, int bs, int mode, double param, const void *prefix, size_t prefix_len, int length_prefixes, jpegx_alloc_fn alloc, void *user, size_t *nbytes),
         (h_planes, nbands, elem_size, H, W, pitch, bs, mode, param, prefix, prefix_len, length_prefixes, alloc, user, nbytes))
JPEGX_ON(jpegx_host_compress_image_packed, (int device, const uint8_t *h_pixels, int nbands, int H, int W, ptrdiff_t pitch, int bs, int mode, double param, const void *prefix, size_t prefix_len, int length_prefixes, jpegx_alloc_fn alloc, void *user, size_t *nbytes),
         (h_pixels, nbands, H, W, pitch, bs, mode, param, prefix, prefix_len, length_prefixes, alloc, user, nbytes))
JPEGX_ON(jpegx_host_decompress_plane, (int device, const uint8_t *h_bytes, size_t nbytes, int H, int W, int bs, int mode, double param, uint8_t *h_out, ptrdiff_t out_pitch),
         (h_bytes, nbytes, H, W, bs, mode, param, h_out, out_pitch))
JPEGX_ON(jpegx_host_decompress_plane_i64, (int device, const uint8_t *h_bytes, size_t nbytes, int H, int W, int bs, int mode, double param, int64_t *h_out, int rows, int cols),
         (h_bytes, nbytes, H, W, bs, mode, param, h_out, rows, cols))
JPEGX_ON(jpegx_host_decompress_image, (int device, const uint8_t *const *h_bytes, const size_t *nbytes, int nbands, int H, int W, int bs, int mode, double param, uint8_t *h_out, ptrdiff_t out_pitch, int rows, int cols, int interleave),
         (h_bytes, nbytes, nbands, H, W, bs, mode, param, h_out, out_pitch, rows, cols, interleave))
JPEGX_ON(jpegx_host_entropy_decode_gpu, (int device, const uint8_t *h_bytes, size_t nbytes, long long nblocks, int16_t *h_zz), (h_bytes, nbytes, nblocks, h_zz))
JPEGX_ON(jpegx_host_pool_release, (int device), ())
JPEGX_ON(jpegx_comm_create_deadline, (int device, jpegx_comm_t *comm, int nranks, int rank, const void *id128, double timeout_s), (comm, nranks, rank, id128, timeout_s))
