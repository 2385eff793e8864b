/* tests/c_abi/abi_roundtrip.c -- libjpegx.so driven from plain C through include/jpegx.h only (no Python, no
 * HIP headers): what a C/C++ host of the codec would link.  Compress one synthetic 8-bit band with
 * jpegx_host_compress_begin/_finish, decode it with jpegx_host_decompress_plane, cross-check the device
 * entropy decoder against the sequential host parser and the fused kernels against the stage kernels.
 * Prints "ok ..." and returns 0, or a message and 1.  Built and run by tests/test_gpu_c_abi.py.            */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jpegx.h"

#define CHECK(call)                                                                         \
    do {                                                                                    \
        int rc_ = (call);                                                                   \
        if (rc_ != JPEGX_OK) {                                                              \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, jpegx_last_error());        \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

int main(void)
{
    enum { H = 256, W = 512, BS = 2 };                   /* band 512 x 1024, sub-sampled 2 x 2 */
    const int HH = H * BS, WW = W * BS;
    int ndev = 0;
    CHECK(jpegx_device_count(&ndev));
    if (ndev < 1) { fprintf(stderr, "no device\n"); return 1; }
    CHECK(jpegx_init(0));
    if (jpegx_version() < 100) { fprintf(stderr, "version\n"); return 1; }

    unsigned char *band = malloc((size_t)HH * WW);
    unsigned state = 12345u;
    for (int y = 0; y < HH; ++y)
        for (int x = 0; x < WW; ++x) {
            state = state * 1664525u + 1013904223u;
            band[(size_t)y * WW + x] = (unsigned char)(((3 * x + 5 * y) >> 4 & 127) + ((state >> 24) & 63));
        }

    /* forward: steps 1 + 4..8 in one native job */
    size_t nbytes = 0;
    CHECK(jpegx_host_compress_begin(band, 1, H, W, WW, BS, JPEGX_Q_QTABLE, 0.0, &nbytes));
    unsigned char *stream = malloc(nbytes ? nbytes : 1);
    CHECK(jpegx_host_compress_finish(stream));
    if (nbytes == 0 || nbytes > (size_t)H * W * 3) { fprintf(stderr, "implausible stream size %zu\n", nbytes); return 1; }

    /* the two entropy decoders agree on it */
    const long long nblocks = (long long)(H / 8) * (W / 8);
    short *zz_gpu = malloc((size_t)nblocks * 128), *zz_cpu = malloc((size_t)nblocks * 128);
    CHECK(jpegx_host_entropy_decode_gpu(stream, nbytes, nblocks, zz_gpu));
    CHECK(jpegx_host_entropy_decode(stream, nbytes, nblocks, zz_cpu));
    if (memcmp(zz_gpu, zz_cpu, (size_t)nblocks * 128) != 0) { fprintf(stderr, "device and host entropy decoders differ\n"); return 1; }

    /* the fused forward equals the stage kernels: mean-pool on the host in double, then DCT / quantise / zigzag */
    double *pooled = malloc((size_t)H * W * 8), *dct = malloc((size_t)H * W * 8), *q = malloc((size_t)H * W * 8), *zzd = malloc((size_t)H * W * 8);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const unsigned char *p = band + (size_t)(2 * y) * WW + 2 * x;
            pooled[(size_t)y * W + x] = ((double)p[0] + p[1] + p[WW] + p[WW + 1]) / 4.0;
        }
    CHECK(jpegx_host_dct8x8_f64(pooled, H, W, dct));
    CHECK(jpegx_host_quantize_f64(dct, H, W, JPEGX_Q_QTABLE, 0.0, q));
    CHECK(jpegx_host_zigzag(q, H, W, 8, zzd));
    for (long long i = 0; i < nblocks * 64; ++i)
        if ((double)zz_gpu[i] != zzd[i] + 0.0) { fprintf(stderr, "fused forward differs from the stage kernels at %lld\n", i); return 1; }

    /* back: all nine steps inverted on the device */
    const ptrdiff_t pitch = (WW + 15) / 16 * 16;
    unsigned char *rec = malloc((size_t)HH * pitch);
    CHECK(jpegx_host_decompress_plane(stream, nbytes, H, W, BS, JPEGX_Q_QTABLE, 0.0, rec, pitch));
    double se = 0.0;
    for (int y = 0; y < HH; ++y)
        for (int x = 0; x < WW; ++x) {
            const double d = (double)rec[(size_t)y * pitch + x] - band[(size_t)y * WW + x];
            se += d * d;
        }
    const double psnr = 10.0 * log10(255.0 * 255.0 / (se / ((double)HH * WW)));
    if (!(psnr > 20.0)) { fprintf(stderr, "round trip PSNR %.2f dB\n", psnr); return 1; }

    /* error contract: codes, not exceptions */
    if (jpegx_host_decompress_plane(stream, nbytes / 2, H, W, BS, JPEGX_Q_QTABLE, 0.0, rec, pitch) != JPEGX_E_INVALID) {
        fprintf(stderr, "a truncated stream was accepted\n");
        return 1;
    }
    if (jpegx_forward_fused(NULL, 8, 8, 8, JPEGX_Q_QTABLE, 0.0, 0, NULL, NULL) != JPEGX_E_INVALID || strlen(jpegx_last_error()) == 0) {
        fprintf(stderr, "argument validation\n");
        return 1;
    }
    CHECK(jpegx_host_pool_release());
    CHECK(jpegx_shutdown());
    printf("ok %zu bytes for %lld blocks, PSNR %.2f dB\n", nbytes, nblocks, psnr);
    return 0;
}
