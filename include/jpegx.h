/*
 * jpegx.h -- C ABI of libjpegx.so: the MI355X (gfx950) implementation of the per-8x8-block
 * transform path of X-rayLaser/Implementing-JPEG-compression (steps 4-6 of its pipeline and
 * their inverses).
 *
 * Boundary contract:
 *   - extern "C", plain pointers and sizes only; every function returns 0 on success or a
 *     negative JPEGX_E_* code, and jpegx_last_error() returns a thread-local message.
 *   - The caller owns host buffers; device buffers are plain device pointers (from
 *     jpegx_malloc, hipMalloc or torch .data_ptr() -- they are interchangeable).
 *   - Every compute entry takes a stream handle (hipStream_t cast to void*; NULL = the
 *     default stream) and only ENQUEUES work: no allocation, no host synchronisation.
 *     jpegx_host_* variants are the synchronous
 *     host-pointer conveniences used by the Python step classes.
 *   - One host thread per device is safe; the library keeps no mutable global state besides
 *     the thread-local error string and the current HIP device of the calling thread.
 *
 * The reference has no native boundary (it is pure Python); each entry point below names the
 * reference function(s) whose work it replaces, relative to /root/reference.
 *
 * Plane layout: row-major, `pitch` in ELEMENTS between rows, H and W multiples of 8.
 * Coefficient stream layout ("zigzag stream"): int16 [H/8][W/8][64], block-row-major, the 64
 * coefficients of a block contiguous in zigzag order -- pipeline/zigzag_order.py:85-99.
 */
#ifndef JPEGX_H
#define JPEGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JPEGX_VERSION 100 /* major*10000 + minor*100 + patch */

/* error codes */
#define JPEGX_OK 0
#define JPEGX_E_INVALID (-1)  /* bad argument (shape not a multiple of 8, null pointer, bad mode ...) */
#define JPEGX_E_HIP (-2)      /* a HIP runtime call failed; message has hipGetErrorString */
#define JPEGX_E_NODEVICE (-3) /* no usable GPU */
#define JPEGX_E_UNSUPPORTED (-4)
#define JPEGX_E_TIMEOUT (-5)  /* a wait with a deadline ran out (RCCL communicator creation / first exchange) */

/* Quantiser selection: pipeline/__init__.py:13-19 (QuantizationMethod.name_to_quantizer). */
typedef enum jpegx_quant_mode {
    JPEGX_Q_NONE = 0,    /* RoundingQuantizer        quantizers.py:4-9    round(a)                 */
    JPEGX_Q_DISCARD = 1, /* DiscardingQuantizer      quantizers.py:12-20  param = keep             */
    JPEGX_Q_DIVIDE = 2,  /* DivisionQuantizer        quantizers.py:23-31  param = divisor          */
    JPEGX_Q_QTABLE = 3   /* JpegQuantizationTable    quantizers.py:34-53  fixed luminance table    */
} jpegx_quant_mode;

/* flags for the fused kernels */
#define JPEGX_F_PIXEL_INPUT 1u /* forward: every sample is a non-negative multiple of 2^-8 below 2^9  \
                                  (8-bit pixels, or their 2x2/4x4 means), so partial sums are exact   \
                                  in fp32; lets the kernel use DC as the error-bound scale            */
#define JPEGX_F_CLAMP_U8 2u    /* inverse: fuse the clamp to [0,255] of pipeline/normalization.py:10-14 */
/* tuning switches (A/B measurements in one process; results are identical either way) */
#define JPEGX_F_TUNE_NO_NT 0x100u /* use the default cache policy instead of nontemporal accesses */
#define JPEGX_F_TUNE_SKIP_EXACT 0x400u /* TIMING ONLY: skip the float64 exact tier (output no longer bit-exact) */
#define JPEGX_F_TUNE_WAVE_PER_BLOCK 0x800u /* forward: one-wavefront-per-block kernel (lane = coefficient,   \
                                              ds_bpermute 1-D passes) instead of lane-per-block; same output */
#define JPEGX_F_TUNE_F64_KERNEL 0x4000u    /* forward: force the all-float64 kernel                          */
#define JPEGX_F_TUNE_NO_F64_KERNEL 0x8000u /* forward: never pick it (fp32 tier + exact tier always)          */
#define JPEGX_F_TUNE_F64_LANE_PER_BLOCK 0x4u /* forward: the all-float64 kernel in its lane-per-block form (default: 8 lanes per block) */
#define JPEGX_F_TUNE_POOL_ROWS_LO 0x1000u /* pooled forward: fewer input rows per LDS phase (experiment) */
#define JPEGX_F_TUNE_POOL_ROWS_HI 0x2000u /* pooled forward: more input rows per LDS phase (experiment)  */
#define JPEGX_F_TUNE_XCD_CONTIG 0x10000u    /* XCD-private block order: the XCDs take turns in runs of 128 strips, \
                                              so a 2 MiB page is touched and translated by ONE XCD (default for   \
                                              launches of 2^22 blocks and more; forward strip and inverse)        */
#define JPEGX_F_TUNE_XCD_RUN(logr) (((unsigned)(logr) & 31u) << 20) /* other run length 2^logr; 31 = one run per XCD */
#define JPEGX_F_TUNE_NO_XCD_CONTIG 0x20000u /* never: plain round-robin order */
#define JPEGX_F_TUNE_COLUMN_UNITS 0x40000u    /* forward: force the strip kernel with the column-wise exact tier     */
#define JPEGX_F_TUNE_NO_COLUMN_UNITS 0x80000u /* forward: never pick it                                              */
#define JPEGX_F_TUNE_DIRECT_STORE 0x8u /* uint8 forward: per-lane 16-byte stores instead of the LDS output tile (A/B) */
#define JPEGX_F_TUNE_NO_STRIP 0x200u /* forward: per-lane global loads instead of LDS-DMA staging    */

/* output element type of jpegx_inverse_fused */
typedef enum jpegx_out_type {
    JPEGX_OUT_F32 = 0, /* float samples (integers after np.round, basis_change.py:43) */
    JPEGX_OUT_I16 = 1, /* int16 */
    JPEGX_OUT_U8 = 2   /* uint8, implies clamping */
} jpegx_out_type;

typedef void *jpegx_stream_t; /* hipStream_t */
typedef void *jpegx_event_t;  /* hipEvent_t  */

/* ---- library / device management ------------------------------------------------------- */
/* jpegx_init selects `device` for the calling thread and creates its HIP context up front (optional:
 * every entry initialises lazily); jpegx_shutdown waits for outstanding work.  Neither resets the
 * device, so the library can share a process with other HIP users (e.g. PyTorch).              */
int jpegx_init(int device);
int jpegx_shutdown(void);
const char *jpegx_last_error(void);
int jpegx_version(void);
int jpegx_device_count(int *count);
int jpegx_set_device(int device);
int jpegx_get_device(int *device);
int jpegx_device_name(int device, char *buf, size_t buflen);
int jpegx_device_synchronize(void);

int jpegx_malloc(void **dptr, size_t bytes);
int jpegx_free(void *dptr);
int jpegx_memset(void *dptr, int value, size_t bytes, jpegx_stream_t stream);
int jpegx_memcpy_h2d(void *dst, const void *src, size_t bytes, jpegx_stream_t stream);
int jpegx_memcpy_d2h(void *dst, const void *src, size_t bytes, jpegx_stream_t stream);
int jpegx_memcpy_d2d(void *dst, const void *src, size_t bytes, jpegx_stream_t stream);

int jpegx_stream_create(jpegx_stream_t *stream);
int jpegx_stream_destroy(jpegx_stream_t stream);
int jpegx_stream_synchronize(jpegx_stream_t stream);
int jpegx_event_create(jpegx_event_t *event);
int jpegx_event_destroy(jpegx_event_t event);
int jpegx_event_record(jpegx_event_t event, jpegx_stream_t stream);
int jpegx_event_synchronize(jpegx_event_t event);
/* make later work on `stream` wait for `event` (hipStreamWaitEvent): orders the gather stream behind
 * the transform stream chunk by chunk (jpegx.multigpu.transform_and_gather) */
int jpegx_stream_wait_event(jpegx_stream_t stream, jpegx_event_t event);
int jpegx_event_elapsed_ms(jpegx_event_t start, jpegx_event_t stop, float *ms);

/* ---- synthetic planes (no reference counterpart; stands in for util.band_to_array,
 *      util.py:110-112).  Writes rows row0..row0+H-1 of plane `plane` of the integer-only
 *      generator documented in jpegx/synth.py: kind 0 = uniform noise 0..255, 1 = smooth. --- */
int jpegx_generate_plane(float *d_plane, int H, int W, ptrdiff_t pitch, int kind, uint32_t seed,
                         uint32_t plane, int row0, jpegx_stream_t stream);

/* ---- the hot path, fused ------------------------------------------------------------------
 * Forward: replaces BasisChange.execute (pipeline/basis_change.py:11-18) + Quantization.execute
 * (pipeline/quantization.py:8-18) + ZigzagOrder.execute (pipeline/zigzag_order.py:85-99) for
 * transform='DCT', dct_size=8.  d_in: fp32 plane [H][pitch]; d_out: int16 zigzag stream.
 * Output integers are bit-exact with the float64 reference (values saturate to int16).      */
int jpegx_forward_fused(const float *d_in, int H, int W, ptrdiff_t pitch, int mode, double param,
                        unsigned flags, int16_t *d_out, jpegx_stream_t stream);

/* Same with the SubSampling mean-pool prologue fused (pipeline/subsampling.py:9-11): the
 * input plane is [H*bs][W*bs] and is averaged over bs x bs tiles (bs in {1,2,4}) on load.    */
int jpegx_forward_fused_pooled(const float *d_in, int H, int W, ptrdiff_t pitch, int bs, int mode,
                               double param, unsigned flags, int16_t *d_out, jpegx_stream_t stream);

/* The same on a FLOAT64 plane, everything in float64 in the reference's operation order (bit-exact by
 * construction): the input of step 4 when samples are not exact in fp32, e.g. after SubSampling with
 * block_size 3, 5, 6 ... (pipeline/subsampling.py:9-11).  pitch in elements, even.               */
int jpegx_forward_fused_f64(const double *d_in, int H, int W, ptrdiff_t pitch, int mode, double param,
                            unsigned flags, int16_t *d_out, jpegx_stream_t stream);

/* SubSampling.execute for ANY block_size (pipeline/subsampling.py:9-11, np.mean of bs x bs tiles): d_in is
 * [H*bs][pitch] of uint8 (elem_size 1) or integer-valued fp32 (elem_size 4); d_out [H][out_pitch] float64:
 * the exact tile sum divided once, which is what np.mean returns for integer bands.             */
int jpegx_mean_pool_f64(const void *d_in, int elem_size, int H, int W, ptrdiff_t pitch, int bs,
                        double *d_out, ptrdiff_t out_pitch, jpegx_stream_t stream);

/* Several planes in ONE launch (BASELINE.json configs[2]: the Y, Cb and Cr bands that
 * pipeline/__init__.py:104-106 compresses one after another): every descriptor is one
 * jpegx_forward_fused_pooled job; the pooled planes' workgroups are dispatched first and the
 * bs = 1 planes fill in behind them.  Same quantiser and flags for all planes.               */
#define JPEGX_MAX_PLANES 8
typedef struct jpegx_plane_desc {
    const float *d_in; /* fp32 plane [H*bs][pitch]                          */
    int16_t *d_out;    /* this plane's zigzag stream [H/8][W/8][64]         */
    int H, W;          /* size AFTER pooling, multiples of 8                */
    ptrdiff_t pitch;   /* elements between input rows (>= W*bs)             */
    int bs;            /* SubSampling factor fused on load: 1, 2 or 4       */
} jpegx_plane_desc;
int jpegx_forward_fused_planes(const jpegx_plane_desc *planes, int nplanes, int mode, double param,
                               unsigned flags, jpegx_stream_t stream);

/* The same on uint8 planes -- the form in which image bands arrive (util.band_to_array,
 * util.py:110-112): d_in is [H*bs][pitch] bytes, bs in {1,2,4} (bs > 1 fuses the bs x bs mean of
 * pipeline/subsampling.py:9-11).  Reads 64 bs^2 bytes per block instead of 256 bs^2.
 * bs = 1 needs W % 16 == 0; pitch in BYTES, a multiple of 16.                                  */
int jpegx_forward_fused_u8(const uint8_t *d_in, int H, int W, ptrdiff_t pitch, int bs, int mode,
                           double param, unsigned flags, int16_t *d_out, jpegx_stream_t stream);

/* Inverse: replaces ZigzagOrder.invert (zigzag_order.py:101-119) + Quantization.invert
 * (quantization.py:20-30) + BasisChange.invert (basis_change.py:28-43, including its final
 * np.round).  d_in: int16 zigzag stream; d_out: [H][out_pitch] of out_type.                  */
int jpegx_inverse_fused(const int16_t *d_in, int H, int W, int mode, double param, unsigned flags,
                        void *d_out, ptrdiff_t out_pitch, int out_type, jpegx_stream_t stream);

/* The two hot entries with an explicit device index ("every entry takes a device index and
 * an optional stream handle"): the pointers and the stream must belong to `device`; the calling thread's
 * current device is left as it was.  The other data-path entries have the same form further down
 * (jpegx_<name>_on); the plain forms work on the thread's current device (jpegx_set_device), which is what a
 * one-process-per-GPU job sets once at start.                                                               */
int jpegx_forward_fused_on(int device, const float *d_in, int H, int W, ptrdiff_t pitch, int mode, double param,
                           unsigned flags, int16_t *d_out, jpegx_stream_t stream);
int jpegx_inverse_fused_on(int device, const int16_t *d_in, int H, int W, int mode, double param, unsigned flags,
                           void *d_out, ptrdiff_t out_pitch, int out_type, jpegx_stream_t stream);

/* Inverse straight to displayable samples: uint8 with the clamp of pipeline/normalization.py:10-14
 * AND SubSampling.invert (pipeline/subsampling.py:13-14 -> util.inflate, util.py:6-14) fused:
 * every sample is replicated bs x bs times, bs in 1..255 (1, 2, 4: compile-time forms; any other factor is a
 * run-time argument of the same kernel); d_out is [H*bs][out_pitch >= W*bs], rows 8-byte aligned (bs 2, 4: 16). */
int jpegx_inverse_fused_u8_inflated(const int16_t *d_in, int H, int W, int mode, double param,
                                    unsigned flags, int bs, uint8_t *d_out, ptrdiff_t out_pitch,
                                    jpegx_stream_t stream);

/* ---- unfused fp32 stage kernels (per-stage parity; coefficients within 1e-4 of the float64
 *      reference, normalised by the block maximum) ---------------------------------------- */
/* DCT.transform_2d blockwise: transforms.py:46-58 via basis_change.py:15-18 */
int jpegx_dct8x8_f32(const float *d_in, int H, int W, ptrdiff_t pitch, float *d_out,
                     ptrdiff_t out_pitch, jpegx_stream_t stream);
/* DCT.transform_2d_inverse blockwise (no rounding): transforms.py:60-69 */
int jpegx_idct8x8_f32(const float *d_in, int H, int W, ptrdiff_t pitch, float *d_out,
                      ptrdiff_t out_pitch, jpegx_stream_t stream);

/* ---- exact float64 stage kernels: bit-identical to the reference's float64 arrays; these
 *      back the stand-alone step classes (BasisChange / Quantization / ZigzagOrder) -------- */
/* BasisChange.execute, DCT branch: basis_change.py:15-18 */
int jpegx_dct8x8_f64(const double *d_in, int H, int W, ptrdiff_t pitch, double *d_out,
                     ptrdiff_t out_pitch, jpegx_stream_t stream);
/* BasisChange.invert, DCT branch: basis_change.py:33-35 (+ np.round of :43 when do_round) */
int jpegx_idct8x8_f64(const double *d_in, int H, int W, ptrdiff_t pitch, double *d_out,
                      ptrdiff_t out_pitch, int do_round, jpegx_stream_t stream);
/* Quantization.execute / invert: quantization.py:8-30 with quantizers.py:4-53 */
int jpegx_quantize_f64(const double *d_in, int H, int W, ptrdiff_t pitch, int mode, double param,
                       double *d_out, ptrdiff_t out_pitch, jpegx_stream_t stream);
int jpegx_restore_f64(const double *d_in, int H, int W, ptrdiff_t pitch, int mode, double param,
                      double *d_out, ptrdiff_t out_pitch, jpegx_stream_t stream);
/* ZigzagOrder.execute / invert: zigzag_order.py:85-119; elem_size in {2,4,8,16} bytes */
int jpegx_zigzag(const void *d_in, int H, int W, ptrdiff_t pitch, int elem_size, void *d_out,
                 jpegx_stream_t stream);
int jpegx_unzigzag(const void *d_in, int H, int W, int elem_size, void *d_out, ptrdiff_t out_pitch,
                   jpegx_stream_t stream);

/* ---- entropy stage on the device ---------------------------------------
 * Replaces RunLengthEncoding.execute (pipeline/run_length_encoding.py:47-64) + RleBytestream.execute
 * (pipeline/rle_byte_stream.py:48-59) for dct_size 8: int16 zigzag stream of `nblocks` blocks ->
 * the reference's byte stream (4-bit run, 4-bit size, sign + magnitude bits, 15-zero chain codes,
 * EOB byte, per-block byte alignment).  Usage: jpegx_entropy_sizes (enqueue: per-block sizes and
 * byte offsets into the workspace) -> jpegx_entropy_total (synchronises, returns the total and
 * JPEGX_E_INVALID if an amplitude exceeds 15 bits, the reference's BadRleCodeError) ->
 * jpegx_entropy_emit into a buffer of at least `total` bytes.                                  */
size_t jpegx_entropy_workspace_bytes(long long nblocks);
int jpegx_entropy_sizes(const int16_t *d_zz, long long nblocks, void *d_workspace, jpegx_stream_t stream);
int jpegx_entropy_total(const void *d_workspace, unsigned long long *h_total, jpegx_stream_t stream);
/* bytes of every block's code string (after jpegx_entropy_sizes; synchronises): block b starts at the sum
 * of the sizes before it -- the index a decoder needs to find block boundaries without parsing          */
int jpegx_entropy_block_sizes(const void *d_workspace, long long nblocks, uint32_t *h_sizes,
                              jpegx_stream_t stream);
/* enqueue; writes nothing at all when the sizes pass flagged an amplitude beyond 15 bits */
int jpegx_entropy_emit(const int16_t *d_zz, long long nblocks, const void *d_workspace, uint8_t *d_out,
                       jpegx_stream_t stream);
/* host convenience; h_out may be NULL to query the size only */
int jpegx_host_entropy_encode(const int16_t *h_zz, long long nblocks, uint8_t *h_out, size_t cap,
                              size_t *nbytes);

/* Inverse of the entropy stage ON THE DEVICE, device pointers in and out (what jpegx_host_entropy_decode_gpu and the
 * jpegx_host_decompress_* jobs run on their pooled buffers): RleBytestream.invert (pipeline/rle_byte_stream.py:61-88) +
 * RunLengthEncoding.invert (pipeline/run_length_encoding.py:66-97) for dct_size 8, bytes -> int16 [nblocks][64], three
 * kernel launches, no host round trip.  d_bytes must be readable (and zero) for 16 bytes behind the stream.
 * `level` 0: segments sized from the stream's average block length, candidates that cannot start a block dropped;
 * `level` 1: 256-byte segments, every candidate (the second try).  jpegx_entropy_decode only enqueues (it clears the
 * workspace's state first); jpegx_entropy_decode_status synchronises the stream and answers JPEGX_OK, 1 = "this level
 * could not take the stream, enqueue the next one" (a stretch far denser in blocks than the average, or a block the
 * first try's filter dropped), or JPEGX_E_INVALID for a stream that is not a sequence of well-formed blocks of this
 * plane (the sequential host parser names the fault).  After level 1 answers 1 (blocks of one byte each: more
 * candidates than any segment's tables hold) the whole-stream scheme of the host conveniences is what is left. */
size_t jpegx_entropy_decode_workspace_bytes(size_t nbytes, long long nblocks);
int jpegx_entropy_decode(const uint8_t *d_bytes, size_t nbytes, long long nblocks, void *d_workspace, int16_t *d_zz,
                         int level, jpegx_stream_t stream);
int jpegx_entropy_decode_status(const void *d_workspace, jpegx_stream_t stream);

/* Inverse of the entropy stage, ON THE HOST (sequential parse, as in the reference):
 * RleBytestream.invert (pipeline/rle_byte_stream.py:61-88) + RunLengthEncoding.invert
 * (pipeline/run_length_encoding.py:66-97) for dct_size 8: bytes -> int16 [nblocks][64].        */
int jpegx_host_entropy_decode(const uint8_t *h_bytes, size_t nbytes, long long nblocks, int16_t *h_zz);

/* ---- multi-GPU: the one exchange step of the path ---------------------------
 * ncclGather-style gather of raw bytes (the int16 zigzag stream or the entropy-coded stream) over
 * RCCL/xGMI.  No reference counterpart (the reference is single-process).  librccl is bound at run
 * time; JPEGX_E_UNSUPPORTED if it cannot be loaded.  One communicator per process/GPU:
 *   rank 0: jpegx_comm_unique_id -> ship the 128 bytes to every rank (any side channel) ->
 *   all ranks: jpegx_comm_create(nranks, rank, id) on the thread's current device ->
 *   jpegx_comm_gather_bytes(...): every rank sends send_bytes; the root receives recv_bytes[r] bytes
 *   from rank r at d_recv + recv_offsets[r].  Enqueued on `stream`.
 * No call blocks without a deadline: the communicator is created non-blocking (ncclCommInitRankConfig,
 * blocking = 0) and polled; when `timeout_s` (jpegx_comm_create: JPEGX_COMM_TIMEOUT_S from the
 * environment, else 120 s) runs out in creation or in a gather round's enqueue (the first round sets up the
 * peer connections) the call returns JPEGX_E_TIMEOUT with the phase in jpegx_last_error().  JPEGX_COMM_LOG=1
 * in the environment logs init-enter / init-exit / first-round lines with time stamps to stderr.          */
typedef void *jpegx_comm_t;
int jpegx_comm_available(void); /* 0 iff librccl could be bound (no communicator is created) */
int jpegx_comm_unique_id(void *id128);
int jpegx_comm_create(jpegx_comm_t *comm, int nranks, int rank, const void *id128);
int jpegx_comm_create_deadline(jpegx_comm_t *comm, int nranks, int rank, const void *id128, double timeout_s);
int jpegx_comm_destroy(jpegx_comm_t comm);
int jpegx_comm_abort(jpegx_comm_t comm);  /* ncclCommAbort: give up outstanding operations, free the handle */
int jpegx_comm_count(jpegx_comm_t comm, int *nranks); /* ncclCommCount: the size RCCL reports */
int jpegx_comm_gather_bytes(jpegx_comm_t comm, const void *d_send, size_t send_bytes, void *d_recv,
                            const size_t *recv_bytes, const size_t *recv_offsets, int root,
                            jpegx_stream_t stream);

/* ---- synchronous host-pointer conveniences (H2D, kernel, D2H on an internal stream) ------ */
int jpegx_host_forward_fused(const float *h_in, int H, int W, ptrdiff_t pitch, int mode,
                             double param, unsigned flags, int16_t *h_out);
int jpegx_host_forward_fused_f64(const double *h_in, int H, int W, int mode, double param, unsigned flags,
                                 int16_t *h_out);
int jpegx_host_inverse_fused(const int16_t *h_in, int H, int W, int mode, double param,
                             unsigned flags, void *h_out, ptrdiff_t out_pitch, int out_type);
int jpegx_host_inverse_fused_u8_inflated(const int16_t *h_in, int H, int W, int mode, double param,
                                         unsigned flags, int bs, uint8_t *h_out, ptrdiff_t out_pitch);
int jpegx_host_dct8x8_f64(const double *h_in, int H, int W, double *h_out);
int jpegx_host_idct8x8_f64(const double *h_in, int H, int W, double *h_out, int do_round);
int jpegx_host_quantize_f64(const double *h_in, int H, int W, int mode, double param, double *h_out);
int jpegx_host_restore_f64(const double *h_in, int H, int W, int mode, double param, double *h_out);
int jpegx_host_zigzag(const void *h_in, int H, int W, int elem_size, void *h_out);
int jpegx_host_unzigzag(const void *h_in, int H, int W, int elem_size, void *h_out);
int jpegx_host_dct8x8_f32(const float *h_in, int H, int W, float *h_out);
int jpegx_host_idct8x8_f32(const float *h_in, int H, int W, float *h_out);

/* ---- one band of compress_band, natively (pipeline/__init__.py:71-76 for transform 'DCT', dct_size 8) -----
 * h_plane: [H*bs][pitch] samples of elem_size 1 (uint8), 4 (int32) or 8 (int64) bytes -- the dtype
 * util.band_to_array hands over (util.py:110-112); wide integers are range-checked (0..255, else
 * JPEGX_E_UNSUPPORTED) and narrowed by a few host threads.  _begin uploads, runs steps 1+4+5+6 (fused) and
 * 7+8 (device entropy stage) on the device's pooled stream and buffers, and reports the size of the byte
 * stream; _finish copies it into h_out (>= that many bytes).  Between the two calls one of the device's job contexts
 * (four per device: streams + grow-only buffers each, so that the jobs of several host threads overlap on the device)
 * is held by the calling thread: any other pooled entry -- a second _begin, the host conveniences, _decompress_*,
 * _pool_release -- returns JPEGX_E_INVALID on that thread until then; other threads take another context, or wait
 * when all are busy; _finish / _abort act on
 * the job the thread opened, whatever its current device has become; _abort gives the pool back without copying.  bs 1, 2, 4: the uint8 kernels with
 * the mean folded in (W*bs a multiple of 16); any other bs: jpegx_mean_pool_f64 + jpegx_forward_fused_f64. */
int jpegx_host_compress_begin(const void *h_plane, int elem_size, int H, int W, ptrdiff_t pitch, int bs,
                              int mode, double param, size_t *nbytes);
int jpegx_host_compress_finish(uint8_t *h_out);
int jpegx_host_compress_abort(void);
/* The way back (decompress_band, pipeline/__init__.py:79-88, transform 'DCT', dct_size 8): the band's byte
 * stream up, RleBytestream.invert + RunLengthEncoding.invert (pipeline/rle_byte_stream.py:61-88,
 * pipeline/run_length_encoding.py:66-97) ON THE DEVICE -- the stream has no index, block starts are recovered
 * in parallel from the zero byte every block ends with (csrc/jpegx_entropy_decode.hip) -- then the fused
 * inverse with clamp and SubSampling.invert; uint8 samples [H*bs][out_pitch] down.  JPEGX_E_INVALID when the
 * bytes are not H/8 * W/8 well-formed blocks.                                                              */
int jpegx_host_decompress_plane(const uint8_t *h_bytes, size_t nbytes, int H, int W, int bs, int mode,
                                double param, uint8_t *h_out, ptrdiff_t out_pitch);
/* the same, as the int64 [rows][cols] array decompress_band returns (cropped to the band's configured size;
 * rows <= H*bs, cols <= W*bs): samples staged in pinned memory and widened by a few host threads          */
int jpegx_host_decompress_plane_i64(const uint8_t *h_bytes, size_t nbytes, int H, int W, int bs, int mode,
                                    double param, int64_t *h_out, int rows, int cols);
/* ---- whole image in one native job (Jpeg.compress / Jpeg.decompress, pipeline/__init__.py:102-124: Y, Cb, Cr
 * through compress_band / decompress_band one after another) -------------------------------------------------
 * The bands of one picture go through ONE lock of the device's pool on two alternating streams: the upload and
 * transform of band k + 1 overlap the entropy stage and the download of band k.  All bands share shape, block_size
 * and quantiser (the reference applies one Configuration to the three bands).
 * compress_image: once every band's byte count is known, `alloc(user, total)` is called ONCE (on the calling
 * thread) for the destination of the whole result -- e.g. a fresh bytes object of exactly that size -- laid out as
 * [prefix][u32 LE count][band 0][u32 LE count][band 1] ... (the counts only with length_prefixes != 0): with the
 * container header as prefix that is the reference's file (file_format.py:86-93), nothing is concatenated on the
 * host.  Fresh destination pages are touched by a few host threads, then every band is copied from the device
 * straight to its place.  nbytes[k] = the bands' byte counts.
 * decompress_image: uint8 samples cropped to rows x cols, either as planes stacked behind each other
 * ([band][rows][out_pitch], interleave = 0) or pixel-interleaved ([rows][cols][nbands], rows out_pitch bytes
 * apart, interleave = 1: the np.dstack of pipeline/__init__.py:121 done on the device).                         */
#define JPEGX_MAX_IMAGE_BANDS 4
typedef void *(*jpegx_alloc_fn)(void *user, size_t nbytes);
int jpegx_host_compress_image(const void *const *h_planes, int nbands, int elem_size, int H, int W, ptrdiff_t pitch,
                              int bs, int mode, double param, const void *prefix, size_t prefix_len,
                              int length_prefixes, jpegx_alloc_fn alloc, void *user, size_t *nbytes);
int jpegx_host_decompress_image(const uint8_t *const *h_bytes, const size_t *nbytes, int nbands, int H, int W, int bs,
                                int mode, double param, uint8_t *h_out, ptrdiff_t out_pitch, int rows, int cols,
                                int interleave);
/* compress_image from pixel-interleaved samples, [H*bs][W*bs][nbands] uint8, rows `pitch` bytes apart -- what
 * np.asarray(image) hands over in half the host time of `image.split()` (pipeline/__init__.py:104) plus one array per
 * band: one upload, the planes are made on the device (jpegx_deinterleave_u8), then as above.  Same bytes out. */
int jpegx_host_compress_image_packed(const uint8_t *h_pixels, int nbands, int H, int W, ptrdiff_t pitch, int bs,
                                     int mode, double param, const void *prefix, size_t prefix_len,
                                     int length_prefixes, jpegx_alloc_fn alloc, void *user, size_t *nbytes);
/* pixel-interleaved [rows][cols][nbands] (rows in_pitch bytes apart) -> device planes [rows][pitch] (uint8, pitch a multiple of 4) */
int jpegx_deinterleave_u8(const uint8_t *d_in, ptrdiff_t in_pitch, int nbands, int rows, int cols, void *const *d_planes,
                          ptrdiff_t pitch, jpegx_stream_t stream);
/* device planes [rows][pitch] (uint8, pitch a multiple of 4) -> pixel-interleaved [rows][cols][nbands] */
int jpegx_interleave_u8(const void *const *d_planes, int nbands, int rows, int cols, ptrdiff_t pitch, uint8_t *d_out,
                        ptrdiff_t out_pitch, jpegx_stream_t stream);
/* the entropy decoding alone on the device: bytes -> int16 [nblocks][64] (= jpegx_host_entropy_decode) */
int jpegx_host_entropy_decode_gpu(const uint8_t *h_bytes, size_t nbytes, long long nblocks, int16_t *h_zz);
/* frees the pooled device / pinned buffers and streams of every job context of the current device (waits for jobs
 * of other threads to finish; JPEGX_E_INVALID while the calling thread itself holds a context) */
int jpegx_host_pool_release(void);

/* ---- explicit-device forms: jpegx_<name>_on(device, ...) == jpegx_<name>(...) with `device` made current for the
 * call and the thread's own current device restored afterwards (csrc/jpegx_on.cpp).  With them one host thread
 * can drive every GPU of a node; device pointers and streams handed in must belong to `device`.
 * jpegx_host_compress_finish / _abort need no such form: they act on the job the calling thread opened. --------- */
int jpegx_malloc_on(int device, void **dptr, size_t bytes);
int jpegx_free_on(int device, void *dptr);
int jpegx_stream_create_on(int device, jpegx_stream_t *stream);
int jpegx_generate_plane_on(int device, float *d_plane, int H, int W, ptrdiff_t pitch, int kind, uint32_t seed,
                            uint32_t plane, int row0, jpegx_stream_t stream);
int jpegx_forward_fused_pooled_on(int device, const float *d_in, int H, int W, ptrdiff_t pitch, int bs, int mode,
                                  double param, unsigned flags, int16_t *d_out, jpegx_stream_t stream);
int jpegx_forward_fused_u8_on(int device, const uint8_t *d_in, int H, int W, ptrdiff_t pitch, int bs, int mode,
                              double param, unsigned flags, int16_t *d_out, jpegx_stream_t stream);
int jpegx_forward_fused_f64_on(int device, const double *d_in, int H, int W, ptrdiff_t pitch, int mode, double param,
                               unsigned flags, int16_t *d_out, jpegx_stream_t stream);
int jpegx_forward_fused_planes_on(int device, const jpegx_plane_desc *planes, int nplanes, int mode, double param,
                                  unsigned flags, jpegx_stream_t stream);
int jpegx_mean_pool_f64_on(int device, const void *d_in, int elem_size, int H, int W, ptrdiff_t pitch, int bs,
                           double *d_out, ptrdiff_t out_pitch, jpegx_stream_t stream);
int jpegx_inverse_fused_u8_inflated_on(int device, const int16_t *d_in, int H, int W, int mode, double param,
                                       unsigned flags, int bs, uint8_t *d_out, ptrdiff_t out_pitch,
                                       jpegx_stream_t stream);
int jpegx_entropy_sizes_on(int device, const int16_t *d_zz, long long nblocks, void *d_workspace, jpegx_stream_t stream);
int jpegx_entropy_total_on(int device, const void *d_workspace, unsigned long long *h_total, jpegx_stream_t stream);
int jpegx_entropy_block_sizes_on(int device, const void *d_workspace, long long nblocks, uint32_t *h_sizes,
                                 jpegx_stream_t stream);
int jpegx_entropy_emit_on(int device, const int16_t *d_zz, long long nblocks, const void *d_workspace, uint8_t *d_out,
                          jpegx_stream_t stream);
int jpegx_entropy_decode_on(int device, const uint8_t *d_bytes, size_t nbytes, long long nblocks, void *d_workspace,
                            int16_t *d_zz, int level, jpegx_stream_t stream);
int jpegx_entropy_decode_status_on(int device, const void *d_workspace, jpegx_stream_t stream);
int jpegx_host_compress_begin_on(int device, const void *h_plane, int elem_size, int H, int W, ptrdiff_t pitch, int bs,
                                 int mode, double param, size_t *nbytes);
int jpegx_host_compress_image_on(int device, const void *const *h_planes, int nbands, int elem_size, int H, int W,
                                 ptrdiff_t pitch, int bs, int mode, double param, const void *prefix, size_t prefix_len,
                                 int length_prefixes, jpegx_alloc_fn alloc, void *user, size_t *nbytes);
int jpegx_host_compress_image_packed_on(int device, const uint8_t *h_pixels, int nbands, int H, int W, ptrdiff_t pitch, int bs,
                                        int mode, double param, const void *prefix, size_t prefix_len, int length_prefixes,
                                        jpegx_alloc_fn alloc, void *user, size_t *nbytes);
int jpegx_host_decompress_plane_on(int device, const uint8_t *h_bytes, size_t nbytes, int H, int W, int bs, int mode,
                                   double param, uint8_t *h_out, ptrdiff_t out_pitch);
int jpegx_host_decompress_plane_i64_on(int device, const uint8_t *h_bytes, size_t nbytes, int H, int W, int bs, int mode,
                                       double param, int64_t *h_out, int rows, int cols);
int jpegx_host_decompress_image_on(int device, const uint8_t *const *h_bytes, const size_t *nbytes, int nbands, int H,
                                   int W, int bs, int mode, double param, uint8_t *h_out, ptrdiff_t out_pitch, int rows,
                                   int cols, int interleave);
int jpegx_host_entropy_decode_gpu_on(int device, const uint8_t *h_bytes, size_t nbytes, long long nblocks, int16_t *h_zz);
int jpegx_host_pool_release_on(int device);
int jpegx_comm_create_deadline_on(int device, jpegx_comm_t *comm, int nranks, int rank, const void *id128,
                                  double timeout_s);

/* ---- instrumentation ---------------------------------------------------------------------
 * When d_counters is non-NULL the fused kernels atomically add, per launch:
 * [0] blocks that took the float64 exact tier, [1] blocks processed.  Pass NULL (default) in
 * production.  Set with jpegx_set_debug_counters for the calling thread.                     */
int jpegx_set_debug_counters(unsigned long long *d_counters);

#ifdef __cplusplus
}
#endif
#endif /* JPEGX_H */
