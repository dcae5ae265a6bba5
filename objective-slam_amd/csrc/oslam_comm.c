/*
 * oslam_comm.c -- the two transports behind oslam_comm (see oslam_comm.h): RCCL over xGMI for one
 * process per GPU, and an in-process loopback for N emulated ranks on one device.
 * The reference has no multi-GPU code (src/cuda/ppf.cu:45 picks one device); the exchange these
 * serve is the global threshold of model.cu:164-170 applied across the shards.
 */
#include <errno.h>
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <rccl/rccl.h>

#include "oslam_comm.h"

/* ---------------------------------------------------------------------------------------------
 * RCCL
 * ------------------------------------------------------------------------------------------- */
typedef struct rccl_ctx {
    ncclComm_t nccl;
    int world;
} rccl_ctx;

static int nccl_fail(ncclResult_t r, const char *what)
{
    char msg[256];
    snprintf(msg, sizeof msg, "%s: %s", what, ncclGetErrorString(r));
    return oslam_fail(OSLAM_E_DEVICE, msg);
}

static int rccl_all_reduce_max(void *ctx, int rank, uint32_t *d_buf, size_t n, void *stream)
{
    rccl_ctx *x = (rccl_ctx *)ctx;
    ncclResult_t r;
    (void)rank;
    if (!x->nccl) return oslam_fail(OSLAM_E_DEVICE, "the RCCL communicator was aborted");
    r = ncclAllReduce(d_buf, d_buf, n, ncclUint32, ncclMax, x->nccl, (hipStream_t)stream);
    return r == ncclSuccess ? OSLAM_OK : nccl_fail(r, "ncclAllReduce");
}

static int rccl_all_gather(void *ctx, int rank, const uint32_t *d_send, uint32_t *d_recv, size_t n, void *stream)
{
    rccl_ctx *x = (rccl_ctx *)ctx;
    ncclResult_t r;
    (void)rank;
    if (!x->nccl) return oslam_fail(OSLAM_E_DEVICE, "the RCCL communicator was aborted");
    r = ncclAllGather(d_send, d_recv, n, ncclUint32, x->nccl, (hipStream_t)stream);
    return r == ncclSuccess ? OSLAM_OK : nccl_fail(r, "ncclAllGather");
}

/* an all-gather with exact sizes: one broadcast per rank that has something, grouped */
static int rccl_all_gather_v(void *ctx, int rank, const void *d_send, void *d_recv, const size_t *bytes, void *stream)
{
    rccl_ctx *x = (rccl_ctx *)ctx;
    ncclResult_t r;
    size_t off = 0;
    int q;
    (void)rank;
    if (!x->nccl) return oslam_fail(OSLAM_E_DEVICE, "the RCCL communicator was aborted");
    r = ncclGroupStart();
    if (r != ncclSuccess) return nccl_fail(r, "ncclGroupStart");
    for (q = 0; q < x->world; q++) {
        if (bytes[q]) {
            r = ncclBroadcast(d_send, (char *)d_recv + off, bytes[q], ncclUint8, q, x->nccl, (hipStream_t)stream);
            if (r != ncclSuccess) {
                (void)ncclGroupEnd();
                return nccl_fail(r, "ncclBroadcast");
            }
        }
        off += bytes[q];
    }
    r = ncclGroupEnd();
    return r == ncclSuccess ? OSLAM_OK : nccl_fail(r, "ncclGroupEnd");
}

static void rccl_abort(void *ctx, int rank)
{
    rccl_ctx *x = (rccl_ctx *)ctx;
    (void)rank;
    if (x->nccl) (void)ncclCommAbort(x->nccl);       /* does not wait for peers that hang in a collective */
    x->nccl = NULL;
}

static void rccl_destroy(void *ctx, int rank)
{
    rccl_ctx *x = (rccl_ctx *)ctx;
    (void)rank;
    if (x->nccl) (void)ncclCommDestroy(x->nccl);
    free(x);
}

static const oslam_coll_ops k_rccl_ops = {rccl_all_reduce_max, rccl_all_gather, rccl_all_gather_v, rccl_abort,
                                          rccl_destroy, "rccl"};

/* ---------------------------------------------------------------------------------------------
 * Loopback: N ranks = N threads of this process on one device.  A collective is: publish the
 * pointers, meet, copy device-to-device, meet again.  The barrier can be aborted (every waiter
 * returns an error) and times out, so a rank that never arrives does not hang the others.
 * ------------------------------------------------------------------------------------------- */
#define LOOP_MAX_RANKS 64
#define LOOP_TIMEOUT_S 120
typedef struct loop_hub {
    int world, refs, aborted, arrived;
    unsigned gen;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    const void *send[LOOP_MAX_RANKS];
    size_t count[LOOP_MAX_RANKS];
} loop_hub;

static int loop_barrier(loop_hub *h)
{
    int rc = OSLAM_OK;
    struct timespec until;
    clock_gettime(CLOCK_REALTIME, &until);
    until.tv_sec += LOOP_TIMEOUT_S;
    pthread_mutex_lock(&h->mu);
    if (h->aborted) {
        rc = OSLAM_E_DEVICE;
    } else if (++h->arrived == h->world) {
        h->arrived = 0;
        h->gen++;
        pthread_cond_broadcast(&h->cv);
    } else {
        const unsigned gen = h->gen;
        while (gen == h->gen && !h->aborted)
            if (pthread_cond_timedwait(&h->cv, &h->mu, &until) == ETIMEDOUT) {
                h->aborted = 1;              /* a rank never came: everybody leaves */
                pthread_cond_broadcast(&h->cv);
                break;
            }
        if (gen == h->gen) rc = OSLAM_E_DEVICE;
    }
    pthread_mutex_unlock(&h->mu);
    return rc == OSLAM_OK ? rc : oslam_fail(rc, "loopback communicator: a peer rank left the collective (aborted or timed out)");
}

#define LHIP(call)                                                                   \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) return oslam_fail(OSLAM_E_DEVICE, hipGetErrorString(e_)); \
    } while (0)

static int loop_all_reduce_max(void *ctx, int rank, uint32_t *d_buf, size_t n, void *stream)
{
    loop_hub *h = (loop_hub *)ctx;
    uint32_t mx[16], tmp[16];
    size_t i;
    int q, rc;
    if (n > 16) return oslam_fail(OSLAM_E_INVALID, "loopback all-reduce of more than 16 words");
    LHIP(hipStreamSynchronize((hipStream_t)stream));       /* what this rank contributes is in memory */
    h->send[rank] = d_buf;
    rc = loop_barrier(h);
    if (rc != OSLAM_OK) return rc;
    for (i = 0; i < n; i++) mx[i] = 0;
    for (q = 0; q < h->world; q++) {
        LHIP(hipMemcpy(tmp, h->send[q], sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
        for (i = 0; i < n; i++) if (tmp[i] > mx[i]) mx[i] = tmp[i];
    }
    rc = loop_barrier(h);                                   /* everybody has read before anybody writes */
    if (rc != OSLAM_OK) return rc;
    LHIP(hipMemcpy(d_buf, mx, sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    return OSLAM_OK;
}

static int loop_all_gather_v(void *ctx, int rank, const void *d_send, void *d_recv, const size_t *bytes, void *stream)
{
    loop_hub *h = (loop_hub *)ctx;
    size_t off = 0;
    int q, rc;
    LHIP(hipStreamSynchronize((hipStream_t)stream));
    h->send[rank] = d_send;
    h->count[rank] = bytes[rank];
    rc = loop_barrier(h);
    if (rc != OSLAM_OK) return rc;
    for (q = 0; q < h->world; q++) {
        if (h->count[q] != bytes[q]) return oslam_fail(OSLAM_E_DEVICE, "loopback all-gather: the ranks disagree on the sizes");
        if (bytes[q]) LHIP(hipMemcpyAsync((char *)d_recv + off, h->send[q], bytes[q], hipMemcpyDeviceToDevice, (hipStream_t)stream));
        off += bytes[q];
    }
    LHIP(hipStreamSynchronize((hipStream_t)stream));
    return loop_barrier(h);                                 /* the send buffers are free again */
}

static int loop_all_gather(void *ctx, int rank, const uint32_t *d_send, uint32_t *d_recv, size_t n, void *stream)
{
    loop_hub *h = (loop_hub *)ctx;
    size_t bytes[LOOP_MAX_RANKS];
    int q;
    for (q = 0; q < h->world; q++) bytes[q] = sizeof(uint32_t) * n;
    return loop_all_gather_v(ctx, rank, d_send, d_recv, bytes, stream);
}

static void loop_abort(void *ctx, int rank)
{
    loop_hub *h = (loop_hub *)ctx;
    (void)rank;
    pthread_mutex_lock(&h->mu);
    h->aborted = 1;
    pthread_cond_broadcast(&h->cv);
    pthread_mutex_unlock(&h->mu);
}

static void loop_destroy(void *ctx, int rank)
{
    loop_hub *h = (loop_hub *)ctx;
    int last;
    (void)rank;
    pthread_mutex_lock(&h->mu);
    last = --h->refs == 0;
    pthread_mutex_unlock(&h->mu);
    if (last) {
        pthread_mutex_destroy(&h->mu);
        pthread_cond_destroy(&h->cv);
        free(h);
    }
}

static const oslam_coll_ops k_loop_ops = {loop_all_reduce_max, loop_all_gather, loop_all_gather_v, loop_abort,
                                          loop_destroy, "loopback"};

/* ---------------------------------------------------------------------------------------------
 * handles
 * ------------------------------------------------------------------------------------------- */
static int comm_alloc_small(oslam_comm *c)
{
    const size_t words = 4 + 4 * (size_t)c->world;
    if (hipSetDevice(c->dev) != hipSuccess) return oslam_fail(OSLAM_E_DEVICE, "hipSetDevice failed");
    if (hipMalloc((void **)&c->d_small, sizeof(uint32_t) * words) != hipSuccess) {
        c->d_small = NULL;
        return oslam_fail(OSLAM_E_NOMEM, "no device memory for the communicator");
    }
    c->h_small = (uint32_t *)malloc(sizeof(uint32_t) * words);
    if (!c->h_small) return oslam_fail(OSLAM_E_NOMEM, "host allocation failed");
    return OSLAM_OK;
}

void oslam_comm_destroy(oslam_comm *c)
{
    if (!c) return;
    (void)hipSetDevice(c->dev);
    if (c->ctx && c->ops) c->ops->destroy(c->ctx, c->rank);
    if (c->d_small) (void)hipFree(c->d_small);
    free(c->h_small);
    free(c);
}

int oslam_comm_unique_id(void *id_out)
{
    ncclUniqueId id;
    ncclResult_t r;
    if (!id_out) return oslam_fail(OSLAM_E_INVALID, "id_out is NULL");
    r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return nccl_fail(r, "ncclGetUniqueId");
    memcpy(id_out, &id, OSLAM_COMM_ID_BYTES);
    return OSLAM_OK;
}

int oslam_comm_create(const void *id, int rank, int world, int dev, oslam_comm **out)
{
    int rc = OSLAM_OK, n_dev = 0;
    oslam_comm *c;
    rccl_ctx *x;
    ncclUniqueId uid;
    ncclResult_t r;
    if (!out) return oslam_fail(OSLAM_E_INVALID, "out is NULL");
    *out = NULL;
    if (!id || world < 1 || rank < 0 || rank >= world) return oslam_fail(OSLAM_E_INVALID, "bad communicator arguments");
    if (sizeof(ncclUniqueId) != OSLAM_COMM_ID_BYTES) return oslam_fail(OSLAM_E_DEVICE, "ncclUniqueId is not 128 bytes in this RCCL");
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return oslam_fail(OSLAM_E_DEVICE, "no HIP device available");
    c = (oslam_comm *)calloc(1, sizeof *c);
    x = (rccl_ctx *)calloc(1, sizeof *x);
    if (!c || !x) { free(c); free(x); return oslam_fail(OSLAM_E_NOMEM, "host allocation failed"); }
    c->ops = &k_rccl_ops;
    c->ctx = x;
    c->rank = rank;
    c->world = world;
    c->dev = dev < 0 ? 0 : dev < n_dev - 1 ? dev : n_dev - 1;          /* ppf.cu:45 */
    x->world = world;
    rc = comm_alloc_small(c);
    if (rc == OSLAM_OK) {
        memcpy(&uid, id, sizeof uid);
        r = ncclCommInitRank(&x->nccl, world, uid, rank);
        if (r != ncclSuccess) { x->nccl = NULL; rc = nccl_fail(r, "ncclCommInitRank"); }
    }
    if (rc != OSLAM_OK) { oslam_comm_destroy(c); return rc; }
    *out = c;
    return OSLAM_OK;
}

int oslam_comm_create_loopback(int world, int dev, oslam_comm **out)
{
    int rc = OSLAM_OK, r, n_dev = 0;
    loop_hub *h;
    if (!out) return oslam_fail(OSLAM_E_INVALID, "out is NULL");
    for (r = 0; r < world; r++) out[r] = NULL;
    if (world < 1 || world > LOOP_MAX_RANKS) return oslam_fail(OSLAM_E_INVALID, "a loopback communicator has 1 to 64 ranks");
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return oslam_fail(OSLAM_E_DEVICE, "no HIP device available");
    h = (loop_hub *)calloc(1, sizeof *h);
    if (!h) return oslam_fail(OSLAM_E_NOMEM, "host allocation failed");
    h->world = world;
    pthread_mutex_init(&h->mu, NULL);
    pthread_cond_init(&h->cv, NULL);
    for (r = 0; r < world && rc == OSLAM_OK; r++) {
        oslam_comm *c = (oslam_comm *)calloc(1, sizeof *c);
        if (!c) { rc = oslam_fail(OSLAM_E_NOMEM, "host allocation failed"); break; }
        c->ops = &k_loop_ops;
        c->ctx = h;
        c->rank = r;
        c->world = world;
        c->dev = dev < 0 ? 0 : dev < n_dev - 1 ? dev : n_dev - 1;
        h->refs++;
        out[r] = c;
        rc = comm_alloc_small(c);
    }
    if (rc != OSLAM_OK) {
        for (r = 0; r < world; r++) { oslam_comm_destroy(out[r]); out[r] = NULL; }
        if (h->refs == 0) { pthread_mutex_destroy(&h->mu); pthread_cond_destroy(&h->cv); free(h); }
    }
    return rc;
}

int oslam_comm_inject_failure(oslam_comm *c, int stage)
{
    if (!c || stage < OSLAM_STAGE_NONE || stage > OSLAM_STAGE_GROW) return oslam_fail(OSLAM_E_INVALID, "bad stage");
    c->inject_stage = stage;
    return OSLAM_OK;
}

int oslam_comm_abort(oslam_comm *c)
{
    if (!c) return oslam_fail(OSLAM_E_INVALID, "NULL handle");
    if (!c->broken) {
        c->ops->abort(c->ctx, c->rank);
        c->broken = 1;
    }
    return OSLAM_OK;
}

int oslam_comm_info(const oslam_comm *c, int *rank, int *world, int *broken)
{
    if (!c) return oslam_fail(OSLAM_E_INVALID, "NULL handle");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (broken) *broken = c->broken;
    return OSLAM_OK;
}

/* ---- the guarded operations ---- */
static int guard(oslam_comm *c)
{
    if (!c) return oslam_fail(OSLAM_E_INVALID, "NULL communicator");
    if (c->broken) return oslam_fail(OSLAM_E_DEVICE, "the communicator was aborted after a failed collective: make a new one");
    return OSLAM_OK;
}

static int after(oslam_comm *c, int rc)
{
    if (rc != OSLAM_OK) {                /* never leave the peers with a half-alive communicator */
        c->ops->abort(c->ctx, c->rank);
        c->broken = 1;
    }
    return rc;
}

int oslam_comm_all_reduce_max(oslam_comm *c, uint32_t *d_buf, size_t n, void *stream)
{
    int rc = guard(c);
    return rc != OSLAM_OK ? rc : after(c, c->ops->all_reduce_max_u32(c->ctx, c->rank, d_buf, n, stream));
}

int oslam_comm_all_gather(oslam_comm *c, const uint32_t *d_send, uint32_t *d_recv, size_t n, void *stream)
{
    int rc = guard(c);
    return rc != OSLAM_OK ? rc : after(c, c->ops->all_gather_u32(c->ctx, c->rank, d_send, d_recv, n, stream));
}

int oslam_comm_all_gather_v(oslam_comm *c, const void *d_send, void *d_recv, const size_t *bytes, void *stream)
{
    int rc = guard(c);
    return rc != OSLAM_OK ? rc : after(c, c->ops->all_gather_v(c->ctx, c->rank, d_send, d_recv, bytes, stream));
}
