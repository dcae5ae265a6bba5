/*
 * oslam_comm.h -- the collectives of the multi-GPU exchange behind a table of operations
 * (internal to liboslam_hip.so; the public handle is `oslam_comm` of include/oslam.h).
 *
 * The exchange of one registration (oslam_align_multi) needs three operations on device buffers:
 * an all-reduce(MAX) of a few words, an all-gather of a few words per rank, and an all-gather
 * with a different size per rank.  Two transports implement them:
 *   - RCCL (one process per GPU, xGMI): what a multi-GPU node runs;
 *   - loopback: N emulated ranks that share ONE device inside one process, one thread per rank,
 *     meeting at pthread barriers and copying device-to-device.  It exists so that the SAME C
 *     function that runs over RCCL -- the per-rank state machine with its error paths -- can be
 *     executed with N = 2, 3, 8 on a one-GPU box (tests/test_gpu_multi.py).
 * The reference has no multi-GPU code (src/cuda/ppf.cu:45 picks one device).
 */
#ifndef OSLAM_COMM_H
#define OSLAM_COMM_H

#include <stddef.h>
#include <stdint.h>

#include "oslam.h"

typedef struct oslam_coll_ops {
    /* d_buf[0..n): in place, every rank ends with the element-wise maximum over ranks */
    int (*all_reduce_max_u32)(void *ctx, int rank, uint32_t *d_buf, size_t n, void *stream);
    /* d_recv[r * n .. r * n + n) = rank r's d_send[0..n) */
    int (*all_gather_u32)(void *ctx, int rank, const uint32_t *d_send, uint32_t *d_recv, size_t n, void *stream);
    /* rank r contributes bytes[r] bytes from its d_send; every rank receives them back to back, rank after
     * rank, at d_recv (which must not overlap d_send) */
    int (*all_gather_v)(void *ctx, int rank, const void *d_send, void *d_recv, const size_t *bytes, void *stream);
    /* gives up the communicator after a failed operation: peers that are still inside a collective are
     * not waited for */
    void (*abort)(void *ctx, int rank);
    void (*destroy)(void *ctx, int rank);
    const char *name;
} oslam_coll_ops;

struct oslam_comm {
    const oslam_coll_ops *ops;
    void *ctx;
    int rank, world, dev;
    int broken;                        /* a collective failed: the communicator was aborted, every later call is refused */
    int inject_stage;                  /* test tap (oslam_comm_inject_failure): fail locally at this stage of the next exchange */
    uint32_t *d_small;                 /* [4 + 4 * world] device words of the exchange */
    uint32_t *h_small;
};

/* stages of the exchange at which a rank can fail on its own (oslam_comm_inject_failure) */
#define OSLAM_STAGE_NONE 0
#define OSLAM_STAGE_VOTE 1             /* before the maxima are exchanged */
#define OSLAM_STAGE_SELECT 2           /* before the survivor counts are exchanged */
#define OSLAM_STAGE_GROW 3             /* while making room for the union */

/* a collective over c; on failure the communicator is aborted and marked broken */
int oslam_comm_all_reduce_max(oslam_comm *c, uint32_t *d_buf, size_t n, void *stream);
int oslam_comm_all_gather(oslam_comm *c, const uint32_t *d_send, uint32_t *d_recv, size_t n, void *stream);
int oslam_comm_all_gather_v(oslam_comm *c, const void *d_send, void *d_recv, const size_t *bytes, void *stream);

/* error text of the calling thread (oslam_host.c) */
int oslam_fail(int code, const char *what);

#endif
