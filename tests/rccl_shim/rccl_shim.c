/* TEST INFRASTRUCTURE ONLY -- a stand-in for the eight RCCL entry points csrc/comm.cpp resolves (ncclGetUniqueId, ncclCommInitRank,
 * ncclCommDestroy, ncclGroupStart, ncclGroupEnd, ncclSend, ncclRecv, ncclGetErrorString), so that the library's own exchanges
 * (vhr_comm_*: plans, pieces, packing into / unpacking from the staging buffers, the grouped batch, the `broken` state) run at world
 * size 2 and 4 on ONE GPU, where RCCL itself refuses two ranks per device.  Selected with VHR_RCCL_LIBRARY=<this .so> (csrc/comm.cpp).
 *
 * Transport: one process per rank; a message is a file under /dev/shm/<unique id>/ (written under a temporary name, then renamed:
 * the receiver never sees half a message); the k-th send from a to b matches the k-th receive of b from a, as in NCCL.  Semantics kept:
 * sends and receives issued between ncclGroupStart and ncclGroupEnd progress together (all sends are posted before any receive is
 * waited for, so no order of calls deadlocks), operations are ordered with the stream they are given (the stream is synchronised
 * before data is read from or written to device memory -- host-blocking where RCCL is asynchronous, which only makes the schedule
 * stricter), byte counts of a matched pair must agree.  What it does not do: collectives, other data types than bytes, more than one
 * communicator per process, any performance.
 *
 * Failure injection (tests/test_comm_shim.py): VHR_RCCL_SHIM_FAIL = "<what>:<n>" makes the n-th call (1-based, per process) of
 * <what> in { init, send, recv, groupstart, groupend } return ncclInternalError on the rank VHR_RCCL_SHIM_FAIL_RANK (default 0).
 * A receive that waits longer than VHR_RCCL_SHIM_TIMEOUT_S (default 60) seconds, or sees the communicator's `abort` mark, fails with
 * ncclRemoteError: a rank whose peer died does not hang.  A rank that fails any call leaves the `abort` mark for its peers. */
#include <errno.h>
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5, ncclRemoteError = 6 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclDataType_t;

struct ncclComm {
    char dir[160];
    int rank, nranks;
    uint64_t send_seq[64], recv_seq[64];
};
typedef struct ncclComm *ncclComm_t;

typedef struct { int send; void *buf; size_t bytes; int peer; ncclComm_t comm; hipStream_t stream; } Op;
static __thread int g_depth = 0;
static __thread Op g_ops[1024];
static __thread int g_nops = 0;
static int g_calls[5] = { 0, 0, 0, 0, 0 };     /* init, send, recv, groupstart, groupend */
static int g_rank = -1;
static char g_dir[160] = "";

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static double timeout_s(void) { const char *e = getenv("VHR_RCCL_SHIM_TIMEOUT_S"); return e && *e ? atof(e) : 60.0; }

static void mark_abort(void) {
    if (!g_dir[0]) return;
    char path[256];
    snprintf(path, sizeof path, "%s/abort", g_dir);
    FILE *f = fopen(path, "w");
    if (f) fclose(f);
}
static int aborted(const char *dir) {
    char path[256];
    snprintf(path, sizeof path, "%s/abort", dir);
    return access(path, F_OK) == 0;
}

/* the injected failure, if this call is the one */
static int injected(int what, int rank) {
    static const char *names[5] = { "init", "send", "recv", "groupstart", "groupend" };
    const int n = ++g_calls[what];
    const char *spec = getenv("VHR_RCCL_SHIM_FAIL");
    if (!spec || !*spec) return 0;
    const char *fr = getenv("VHR_RCCL_SHIM_FAIL_RANK");
    if (rank != (fr && *fr ? atoi(fr) : 0)) return 0;
    const size_t len = strlen(names[what]);
    if (strncmp(spec, names[what], len) != 0 || spec[len] != ':') return 0;
    if (atoi(spec + len + 1) != n) return 0;
    fprintf(stderr, "[rccl_shim] rank %d: injected failure of %s call %d\n", rank, names[what], n);
    mark_abort();
    return 1;
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error (shim)";
        case ncclUnhandledCudaError: return "unhandled HIP error (shim)";
        case ncclSystemError: return "system error (shim)";
        case ncclInternalError: return "internal error (shim: injected failure)";
        case ncclInvalidArgument: return "invalid argument (shim)";
        case ncclInvalidUsage: return "invalid usage (shim)";
        case ncclRemoteError: return "remote error (shim: a peer did not deliver in time, or aborted)";
        default: return "unknown error (shim)";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    memset(id->internal, 0, sizeof id->internal);
    struct timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    /* VHR_RCCL_SHIM_TAG: a run's own mark in the directory name, so that whoever cleans up after an injected failure removes ITS directories only */
    const char *tag = getenv("VHR_RCCL_SHIM_TAG");
    snprintf(id->internal, sizeof id->internal, "/dev/shm/vhr_rccl_shim_%.24s_%d_%lld_%ld", tag && *tag ? tag : "run", (int)getpid(), (long long)t.tv_sec, t.tv_nsec);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks || strncmp(id.internal, "/dev/shm/vhr_rccl_shim_", 23) != 0) return ncclInvalidArgument;
    struct ncclComm *c = (struct ncclComm *)calloc(1, sizeof *c);
    if (!c) return ncclSystemError;
    memcpy(c->dir, id.internal, sizeof id.internal);
    c->dir[sizeof id.internal - 1] = 0;
    c->rank = rank; c->nranks = nranks;
    g_rank = rank;
    snprintf(g_dir, sizeof g_dir, "%s", c->dir);
    if (mkdir(c->dir, 0700) != 0 && errno != EEXIST) { free(c); return ncclSystemError; }
    if (injected(0, rank)) { free(c); return ncclInternalError; }
    char path[256];
    snprintf(path, sizeof path, "%s/rank%d", c->dir, rank);
    FILE *f = fopen(path, "w");
    if (!f) { free(c); return ncclSystemError; }
    fclose(f);
    const double t0 = now_s();                    /* the rendezvous: every rank's mark is there */
    for (int r = 0; r < nranks; ++r) {
        snprintf(path, sizeof path, "%s/rank%d", c->dir, r);
        while (access(path, F_OK) != 0) {
            if (aborted(c->dir) || now_s() - t0 > timeout_s()) { free(c); return ncclRemoteError; }
            usleep(200);
        }
    }
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclInvalidArgument;
    char path[256];
    snprintf(path, sizeof path, "%s/done%d", c->dir, c->rank);
    FILE *f = fopen(path, "w");
    if (f) fclose(f);
    int all = 1;                                   /* the last rank out removes the directory (best effort) */
    for (int r = 0; r < c->nranks; ++r) { snprintf(path, sizeof path, "%s/done%d", c->dir, r); if (access(path, F_OK) != 0) all = 0; }
    if (all) {
        char cmd[256];
        snprintf(cmd, sizeof cmd, "rm -rf '%s'", c->dir);
        if (system(cmd) != 0) { /* best effort */ }
    }
    free(c);
    return ncclSuccess;
}

static ncclResult_t do_send(const Op *o) {
    struct ncclComm *c = o->comm;
    void *host = malloc(o->bytes ? o->bytes : 1);
    if (!host) return ncclSystemError;
    if (o->bytes && hipMemcpy(host, o->buf, o->bytes, hipMemcpyDeviceToHost) != hipSuccess) { free(host); return ncclUnhandledCudaError; }
    char tmp[300], fin[300];
    const unsigned long long seq = (unsigned long long)c->send_seq[o->peer]++;
    snprintf(tmp, sizeof tmp, "%s/t_%d_%d_%llu", c->dir, c->rank, o->peer, seq);
    snprintf(fin, sizeof fin, "%s/m_%d_%d_%llu", c->dir, c->rank, o->peer, seq);
    FILE *f = fopen(tmp, "wb");
    if (!f) { free(host); return ncclSystemError; }
    const size_t w = o->bytes ? fwrite(host, 1, o->bytes, f) : 0;
    fclose(f);
    free(host);
    if (w != o->bytes || rename(tmp, fin) != 0) return ncclSystemError;
    return ncclSuccess;
}

static ncclResult_t do_recv(const Op *o) {
    struct ncclComm *c = o->comm;
    char fin[300];
    const unsigned long long seq = (unsigned long long)c->recv_seq[o->peer]++;
    snprintf(fin, sizeof fin, "%s/m_%d_%d_%llu", c->dir, o->peer, c->rank, seq);
    const double t0 = now_s();
    while (access(fin, F_OK) != 0) {
        if (aborted(c->dir) || now_s() - t0 > timeout_s()) { fprintf(stderr, "[rccl_shim] rank %d: receive %llu from %d never arrived\n", c->rank, seq, o->peer); return ncclRemoteError; }
        usleep(100);
    }
    struct stat st;
    if (stat(fin, &st) != 0 || (size_t)st.st_size != o->bytes) { fprintf(stderr, "[rccl_shim] rank %d: receive of %zu bytes from %d matched a send of %lld\n", c->rank, o->bytes, o->peer, (long long)st.st_size); return ncclInvalidUsage; }
    void *host = malloc(o->bytes ? o->bytes : 1);
    if (!host) return ncclSystemError;
    FILE *f = fopen(fin, "rb");
    if (!f) { free(host); return ncclSystemError; }
    const size_t r = o->bytes ? fread(host, 1, o->bytes, f) : 0;
    fclose(f);
    unlink(fin);
    ncclResult_t rc = ncclSuccess;
    if (r != o->bytes) rc = ncclSystemError;
    else if (o->bytes && hipMemcpy(o->buf, host, o->bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
    free(host);
    return rc;
}

static ncclResult_t progress(Op *ops, int n) {
    for (int i = 0; i < n; ++i) {                  /* everything the streams hold in front of the batch has to be through */
        int seen = 0;
        for (int j = 0; j < i; ++j) seen |= ops[j].stream == ops[i].stream;
        if (!seen && hipStreamSynchronize(ops[i].stream) != hipSuccess) return ncclUnhandledCudaError;
    }
    for (int i = 0; i < n; ++i)
        if (ops[i].send) { const ncclResult_t rc = do_send(&ops[i]); if (rc != ncclSuccess) { mark_abort(); return rc; } }
    for (int i = 0; i < n; ++i)
        if (!ops[i].send) { const ncclResult_t rc = do_recv(&ops[i]); if (rc != ncclSuccess) { mark_abort(); return rc; } }
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) {
    if (injected(3, g_rank)) return ncclInternalError;
    ++g_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd(void) {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    const int n = g_nops;
    g_nops = 0;
    if (injected(4, g_rank)) return ncclInternalError;         /* (the queued operations are dropped: the peers see the abort mark) */
    return progress(g_ops, n);
}

static ncclResult_t post(int send, void *buf, size_t count, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!comm || peer < 0 || peer >= comm->nranks || peer == comm->rank || (!buf && count)) return ncclInvalidArgument;
    if (injected(send ? 1 : 2, comm->rank)) return ncclInternalError;
    Op o = { send, buf, count, peer, comm, stream };
    if (g_depth > 0) {
        if (g_nops >= 1024) return ncclInternalError;
        g_ops[g_nops++] = o;
        return ncclSuccess;
    }
    return progress(&o, 1);
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    if (type != 1 && type != 0) return ncclInvalidArgument;         /* bytes (ncclUint8 = 1, ncclInt8 = 0) */
    return post(1, (void *)buf, count, peer, comm, stream);
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    if (type != 1 && type != 0) return ncclInvalidArgument;
    return post(0, buf, count, peer, comm, stream);
}
