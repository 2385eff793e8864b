// jpegx_host.cpp -- host-side (CPU) parts of libjpegx.so that the reference also runs on the host:
// the inverse of the entropy stage.  Decoding the byte stream is inherently sequential (block
// boundaries are only known by parsing), it is outside the GPU hot path and the
// reference does it in pure Python (RleBytestream.invert, pipeline/rle_byte_stream.py:61-88, then
// RunLengthEncoding.invert, pipeline/run_length_encoding.py:66-97); this is the same parse in C++
// so that decompress_band is not dominated by an interpreter loop.
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/jpegx.h"

extern "C" void jpegx_internal_set_error(const char *msg);

namespace {

struct BitReader {
    const uint8_t *p;
    size_t nbits, pos;
    bool take(int n, unsigned *out)      // n <= 16 bits, MSB first
    {
        if (pos + (size_t)n > nbits) return false;
        unsigned v = 0;
        for (int i = 0; i < n; ++i, ++pos) v = (v << 1) | ((p[pos >> 3] >> (7 - (pos & 7))) & 1u);
        *out = v;
        return true;
    }
};

int fail(const char *msg)
{
    jpegx_internal_set_error(msg);
    return JPEGX_E_INVALID;
}

}  // namespace

extern "C" int jpegx_host_entropy_decode(const uint8_t *h_bytes, size_t nbytes, long long nblocks, int16_t *h_zz)
{
    if (!h_bytes || !h_zz) return fail("null host pointer");
    if (nblocks <= 0) return fail("block count must be positive");
    BitReader br{h_bytes, nbytes * 8, 0};
    for (long long b = 0; b < nblocks; ++b) {
        int16_t *blk = h_zz + b * 64;
        int n = 0;                                   // coefficients written so far
        for (;;) {
            unsigned run, size;
            if (!br.take(4, &run) || !br.take(4, &size)) return fail("entropy stream ends inside a block");
            if (run == 0 && size == 0) {             // EOB: zero fill, skip the byte padding
                for (; n < 64; ++n) blk[n] = 0;
                br.pos = (br.pos + 7) & ~(size_t)7;
                break;
            }
            if (run == 15 && size == 0) {            // zero chain: FIFTEEN zeros (util.py:134-154)
                if (n + 15 > 64) return fail("zero chain overruns the block");
                for (int i = 0; i < 15; ++i) blk[n++] = 0;
                continue;
            }
            if (size == 0) return fail("BadRleCodeError: zero size with a non-terminal run");
            unsigned bits;
            if (!br.take((int)size, &bits)) return fail("entropy stream ends inside an amplitude");
            const unsigned mag = bits & ((1u << (size - 1)) - 1u);
            const int amp = (bits >> (size - 1)) ? (int)mag : -(int)mag;     // sign bit '1' = positive
            if (n + (int)run + 1 > 64) return fail("run overruns the block");
            for (unsigned i = 0; i < run; ++i) blk[n++] = 0;
            blk[n++] = (int16_t)amp;
        }
    }
    return JPEGX_OK;
}
