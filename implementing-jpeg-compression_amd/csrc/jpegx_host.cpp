// jpegx_host.cpp -- host-side (CPU) parts of libjpegx.so that the reference also runs on the host:
// the inverse of the entropy stage.  Decoding the byte stream is inherently sequential (block
// boundaries are only known by parsing), it is outside the GPU hot path and the
// reference does it in pure Python (RleBytestream.invert, pipeline/rle_byte_stream.py:61-88, then
// RunLengthEncoding.invert, pipeline/run_length_encoding.py:66-97); this is the same parse in C++
// so that decompress_band is not dominated by an interpreter loop.
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/jpegx.h"

extern "C" void jpegx_internal_set_error(const char *msg);

namespace {

// MSB-first view of the byte stream at an arbitrary bit position: peek() returns the next 64 bits
// left-aligned (at least 57 of them valid), enough for a whole code (8 header + at most 15 amplitude bits)
// without a refill loop.  Reads beyond the end of the buffer are served from a zero-padded copy of the tail.
struct BitCursor {
    const uint8_t *p;
    size_t nbytes;
    size_t bitpos = 0;
    uint8_t tail[16] = {0};
    size_t tail_from;                         // byte offset from which `tail` takes over

    BitCursor(const uint8_t *bytes, size_t n) : p(bytes), nbytes(n)
    {
        tail_from = n >= 8 ? n - 8 : 0;
        for (size_t i = tail_from; i < n; ++i) tail[i - tail_from] = bytes[i];
    }
    uint64_t peek() const
    {
        const size_t byte = bitpos >> 3;
        const uint8_t *src = byte < tail_from ? p + byte : tail + (byte - tail_from < 8 ? byte - tail_from : 8);
        uint64_t w;
        __builtin_memcpy(&w, src, 8);
        return __builtin_bswap64(w) << (bitpos & 7);
    }
    size_t bits_left() const { return nbytes * 8 > bitpos ? nbytes * 8 - bitpos : 0; }
    void align_to_byte() { bitpos = (bitpos + 7) & ~(size_t)7; }   // drop the zero padding after an EOB
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
    BitCursor cur(h_bytes, nbytes);
    for (long long b = 0; b < nblocks; ++b) {
        int16_t *blk = h_zz + b * 64;
        memset(blk, 0, 128);                         // zeros are the common case: only non-zeros are written below
        int n = 0;                                   // coefficients placed so far
        for (;;) {
            if (cur.bits_left() < 8) return fail("entropy stream ends inside a block");
            const uint64_t w = cur.peek();
            const unsigned run = (unsigned)(w >> 60), size = (unsigned)(w >> 56) & 15u;
            if (size == 0) {
                cur.bitpos += 8;
                if (run == 0) {                      // EOB: the rest of the block stays zero, skip the byte padding
                    cur.align_to_byte();
                    break;
                }
                if (run != 15) return fail("BadRleCodeError: zero size with a non-terminal run");
                if (n + 15 > 64) return fail("zero chain overruns the block");      // FIFTEEN zeros (util.py:134-154)
                n += 15;
                continue;
            }
            if (size == 1) return fail("ValueError: run-length code with a sign bit but no amplitude bits (size 1)");   // rle_byte_stream.py:35-42
            if (cur.bits_left() < 8 + size) return fail("entropy stream ends inside an amplitude");
            const unsigned bits = (unsigned)((w << 8) >> (64 - size));
            cur.bitpos += 8 + size;
            const unsigned mag = bits & ((1u << (size - 1)) - 1u);
            n += (int)run;
            if (n >= 64) return fail("run overruns the block");
            blk[n++] = (int16_t)((bits >> (size - 1)) ? (int)mag : -(int)mag);       // sign bit '1' = positive
        }
    }
    // bytes (whole blocks) behind the last block: the reference parses them too and then fails in its reshape
    // (run_length_encoding.py:77-79)
    if (cur.bits_left() > 0) return fail("ValueError: the entropy stream holds more than the plane's blocks");
    return JPEGX_OK;
}

// ------------------------------------------------------------------------------------------------
// Multi-GPU gather of the coefficient / entropy-coded stream over RCCL: the one
// exchange step of the path.  librccl is bound at run time (dlopen) so that libjpegx.so itself
// carries no RCCL dependency; an RCCL that is already loaded in the process (e.g. PyTorch's) is
// reused.  ncclGather-style: grouped ncclSend (every rank) / ncclRecv (root, one per rank) of raw
// bytes, enqueued on the caller's stream.  The unique id is exchanged by the caller (any side
// channel: torch.distributed, MPI, a file).
//
// Nothing in here may block without a deadline (round 3): the communicator is created NON-BLOCKING
// (ncclCommInitRankConfig with blocking = 0) and every RCCL call that may answer ncclInProgress is
// followed by a poll of ncclCommGetAsyncError against a deadline, so a peer that never arrives, or an
// RCCL initialisation that sits in one of its own dependencies (rocm_smi's process-shared mutex, see
// DESIGN.md section 6), ends in JPEGX_E_TIMEOUT naming the phase instead of a silent hang.  With
// JPEGX_COMM_LOG=1 every phase is logged to stderr with a time stamp.
// ------------------------------------------------------------------------------------------------
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <rccl/rccl.h>   // types and constants only: every function is bound with dlsym

namespace {

typedef ncclResult_t (*fn_get_unique_id)(ncclUniqueId *);
typedef ncclResult_t (*fn_comm_init_rank)(ncclComm_t *, int, ncclUniqueId, int);
typedef ncclResult_t (*fn_comm_init_rank_config)(ncclComm_t *, int, ncclUniqueId, int, ncclConfig_t *);
typedef ncclResult_t (*fn_comm_async_error)(ncclComm_t, ncclResult_t *);
typedef ncclResult_t (*fn_comm_op)(ncclComm_t);
typedef ncclResult_t (*fn_comm_count)(const ncclComm_t, int *);
typedef ncclResult_t (*fn_send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
typedef ncclResult_t (*fn_recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
typedef ncclResult_t (*fn_group)(void);
typedef const char *(*fn_error_string)(ncclResult_t);

struct Rccl {
    void *handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_init_rank_config comm_init_rank_config = nullptr;
    fn_comm_async_error comm_async_error = nullptr;
    fn_comm_op comm_destroy = nullptr, comm_abort = nullptr, comm_finalize = nullptr;
    fn_comm_count comm_count = nullptr;
    fn_send send = nullptr;
    fn_recv recv = nullptr;
    fn_group group_start = nullptr, group_end = nullptr;
    fn_error_string error_string = nullptr;
};

Rccl g_rccl;

int load_rccl()
{
    if (g_rccl.handle) return JPEGX_OK;
    static const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)                      // prefer a copy that is already in the process
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
    if (!h)
        for (const char *n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) {
        jpegx_internal_set_error("librccl.so could not be loaded (multi-GPU gather unavailable)");
        return JPEGX_E_UNSUPPORTED;
    }
    Rccl r;
    r.handle = h;
    r.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
    r.comm_init_rank_config = (fn_comm_init_rank_config)dlsym(h, "ncclCommInitRankConfig");
    r.comm_async_error = (fn_comm_async_error)dlsym(h, "ncclCommGetAsyncError");
    r.comm_destroy = (fn_comm_op)dlsym(h, "ncclCommDestroy");
    r.comm_abort = (fn_comm_op)dlsym(h, "ncclCommAbort");
    r.comm_finalize = (fn_comm_op)dlsym(h, "ncclCommFinalize");
    r.comm_count = (fn_comm_count)dlsym(h, "ncclCommCount");
    r.send = (fn_send)dlsym(h, "ncclSend");
    r.recv = (fn_recv)dlsym(h, "ncclRecv");
    r.group_start = (fn_group)dlsym(h, "ncclGroupStart");
    r.group_end = (fn_group)dlsym(h, "ncclGroupEnd");
    r.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.send || !r.recv || !r.group_start || !r.group_end) {
        jpegx_internal_set_error("librccl.so lacks a required symbol");
        return JPEGX_E_UNSUPPORTED;
    }
    g_rccl = r;
    return JPEGX_OK;
}

int rccl_fail(const char *what, int code)
{
    char buf[300];
    snprintf(buf, sizeof(buf), "%s failed: %s", what, g_rccl.error_string ? g_rccl.error_string((ncclResult_t)code) : "RCCL error");
    jpegx_internal_set_error(buf);
    return JPEGX_E_HIP;
}

double now_s()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

struct Comm {
    ncclComm_t nccl;
    int nranks, rank;
    bool nonblocking;
    double timeout_s;       // deadline of every later wait on this communicator
    double t0;              // creation time: log stamps are relative to it
    long rounds;            // gather rounds issued so far (the first one sets up the connections)
};

bool comm_log_on()
{
    const char *e = getenv("JPEGX_COMM_LOG");
    return e && *e && strcmp(e, "0") != 0;
}

void comm_log(int rank, int nranks, double t0, const char *phase, const char *detail = "")
{
    if (!comm_log_on()) return;
    fprintf(stderr, "[jpegx comm rank %d/%d pid %d +%.3fs] %s%s\n", rank, nranks, (int)getpid(), now_s() - t0, phase, detail);
    fflush(stderr);
}

double default_timeout_s()
{
    const char *e = getenv("JPEGX_COMM_TIMEOUT_S");
    const double v = e ? atof(e) : 0.0;
    return v > 0.0 ? v : 120.0;
}

// Wait until the communicator has left ncclInProgress, at most until `deadline`.  JPEGX_OK, JPEGX_E_TIMEOUT
// (message names `phase`) or JPEGX_E_HIP (RCCL's own error).
int wait_ready(ncclComm_t c, double deadline, const char *phase, int rank, int nranks)
{
    if (!g_rccl.comm_async_error) return JPEGX_OK;
    useconds_t nap = 50;
    for (;;) {
        ncclResult_t state = ncclSuccess;
        const ncclResult_t e = g_rccl.comm_async_error(c, &state);
        if (e != ncclSuccess) return rccl_fail("ncclCommGetAsyncError", e);
        if (state == ncclSuccess) return JPEGX_OK;
        if (state != ncclInProgress) {
            char what[120];
            snprintf(what, sizeof(what), "RCCL (%s, rank %d of %d)", phase, rank, nranks);
            return rccl_fail(what, state);
        }
        if (now_s() > deadline) {
            char buf[300];
            snprintf(buf, sizeof(buf), "RCCL %s did not complete before the deadline on rank %d of %d (still ncclInProgress)", phase, rank, nranks);
            jpegx_internal_set_error(buf);
            return JPEGX_E_TIMEOUT;
        }
        usleep(nap);
        if (nap < 2000) nap *= 2;
    }
}

const ncclDataType_t kBytes = ncclUint8;

}  // namespace

extern "C" {

int jpegx_comm_available(void) { return load_rccl(); }

int jpegx_comm_unique_id(void *id128)
{
    if (!id128) return fail("null id buffer");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    const ncclResult_t e = g_rccl.get_unique_id(&id);
    if (e) return rccl_fail("ncclGetUniqueId", e);
    memcpy(id128, id.internal, 128);
    return JPEGX_OK;
}

int jpegx_comm_create_deadline(jpegx_comm_t *comm, int nranks, int rank, const void *id128, double timeout_s)
{
    if (!comm || !id128) return fail("null pointer");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail("bad rank / nranks");
    int rc = load_rccl();
    if (rc) return rc;
    if (!(timeout_s > 0.0)) timeout_s = default_timeout_s();
    ncclUniqueId id;
    memcpy(id.internal, id128, 128);
    const double t0 = now_s();
    ncclComm_t c = nullptr;
    const bool nb = g_rccl.comm_init_rank_config && g_rccl.comm_async_error;
    comm_log(rank, nranks, t0, "init-enter", nb ? " (ncclCommInitRankConfig, blocking = 0)" : " (ncclCommInitRank, blocking: this librccl has no config entry)");
    if (nb) {
        ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
        cfg.blocking = 0;
        const ncclResult_t e = g_rccl.comm_init_rank_config(&c, nranks, id, rank, &cfg);   // binds to the calling thread's current HIP device
        if (e != ncclSuccess && e != ncclInProgress) return rccl_fail("ncclCommInitRankConfig", e);
        rc = wait_ready(c, t0 + timeout_s, "communicator creation (ncclCommInitRankConfig)", rank, nranks);
        if (rc) {
            comm_log(rank, nranks, t0, "init-FAILED: ", rc == JPEGX_E_TIMEOUT ? "deadline passed" : "RCCL error");
            // no ncclCommAbort here: it joins RCCL's init thread, which is the one that is stuck; the handle is
            // leaked and the caller is expected to leave the process
            return rc;
        }
    } else {
        const ncclResult_t e = g_rccl.comm_init_rank(&c, nranks, id, rank);
        if (e) return rccl_fail("ncclCommInitRank", e);
    }
    comm_log(rank, nranks, t0, "init-exit");
    *comm = new Comm{c, nranks, rank, nb, timeout_s, t0, 0};
    return JPEGX_OK;
}

int jpegx_comm_create(jpegx_comm_t *comm, int nranks, int rank, const void *id128)
{
    return jpegx_comm_create_deadline(comm, nranks, rank, id128, 0.0);
}

int jpegx_comm_destroy(jpegx_comm_t comm)
{
    if (!comm) return JPEGX_OK;
    Comm *c = static_cast<Comm *>(comm);
    ncclResult_t e = ncclSuccess;
    if (c->nonblocking && g_rccl.comm_finalize) {
        // a non-blocking communicator is flushed first (ncclCommFinalize -> poll), then freed locally
        e = g_rccl.comm_finalize(c->nccl);
        if (e == ncclSuccess || e == ncclInProgress) {
            if (wait_ready(c->nccl, now_s() + c->timeout_s, "ncclCommFinalize", c->rank, c->nranks) == JPEGX_OK) e = g_rccl.comm_destroy(c->nccl);
            else e = g_rccl.comm_abort ? g_rccl.comm_abort(c->nccl) : ncclInternalError;
        }
    } else {
        e = g_rccl.comm_destroy(c->nccl);
    }
    comm_log(c->rank, c->nranks, c->t0, "destroyed");
    delete c;
    return e ? rccl_fail("ncclCommDestroy", e) : JPEGX_OK;
}

int jpegx_comm_abort(jpegx_comm_t comm)
{
    if (!comm) return JPEGX_OK;
    Comm *c = static_cast<Comm *>(comm);
    const ncclResult_t e = g_rccl.comm_abort ? g_rccl.comm_abort(c->nccl) : g_rccl.comm_destroy(c->nccl);
    comm_log(c->rank, c->nranks, c->t0, "aborted");
    delete c;
    return e ? rccl_fail("ncclCommAbort", e) : JPEGX_OK;
}

int jpegx_comm_count(jpegx_comm_t comm, int *nranks)
{
    if (!comm || !nranks) return fail("null pointer");
    Comm *c = static_cast<Comm *>(comm);
    if (!g_rccl.comm_count) {
        jpegx_internal_set_error("librccl.so lacks ncclCommCount");
        return JPEGX_E_UNSUPPORTED;
    }
    const ncclResult_t e = g_rccl.comm_count(c->nccl, nranks);      // what RCCL itself believes, not what we passed in
    return e ? rccl_fail("ncclCommCount", e) : JPEGX_OK;
}

int jpegx_comm_gather_bytes(jpegx_comm_t comm, const void *d_send, size_t send_bytes, void *d_recv,
                            const size_t *recv_bytes, const size_t *recv_offsets, int root, jpegx_stream_t stream)
{
    if (!comm) return fail("null communicator");
    Comm *c = static_cast<Comm *>(comm);
    if (root < 0 || root >= c->nranks) return fail("bad root rank");
    if (c->rank == root && (!d_recv || !recv_bytes || !recv_offsets)) return fail("root needs receive buffer, sizes and offsets");
    if (send_bytes && !d_send) return fail("null send buffer");
    const bool first = c->rounds++ == 0;
    if (first) comm_log(c->rank, c->nranks, c->t0, "first-round enter (connection setup happens here)");
    hipStream_t st = (hipStream_t)stream;
    ncclResult_t e = g_rccl.group_start();
    if (e) return rccl_fail("ncclGroupStart", e);
    if (c->rank == root)
        for (int r = 0; r < c->nranks && (e == ncclSuccess || e == ncclInProgress); ++r)
            if (recv_bytes[r])
                e = g_rccl.recv(static_cast<char *>(d_recv) + recv_offsets[r], recv_bytes[r], kBytes, r, c->nccl, st);
    if ((e == ncclSuccess || e == ncclInProgress) && send_bytes) e = g_rccl.send(d_send, send_bytes, kBytes, root, c->nccl, st);
    const ncclResult_t e2 = g_rccl.group_end();
    if (e != ncclSuccess && e != ncclInProgress) return rccl_fail("ncclSend/ncclRecv", e);
    if (e2 != ncclSuccess && e2 != ncclInProgress) return rccl_fail("ncclGroupEnd", e2);
    if (c->nonblocking) {
        // the group is ENQUEUED once the communicator reports ncclSuccess again (the transfer itself runs on `stream`)
        const int rc = wait_ready(c->nccl, now_s() + c->timeout_s, first ? "first gather round (connection setup)" : "gather round (enqueue)", c->rank, c->nranks);
        if (rc) {
            comm_log(c->rank, c->nranks, c->t0, first ? "first-round FAILED" : "round FAILED");
            return rc;
        }
    }
    if (first) comm_log(c->rank, c->nranks, c->t0, "first-round enqueued");
    return JPEGX_OK;
}

}  // extern "C"
