/*
 * oslam_vote_wide.hip -- the re-vote of the workgroups whose 16-bit counters overflowed (oslam_vote_body.inc: a counter
 * word of the accumulator holds two 16-bit counters; counts beyond 65 535 need large planar surfaces in both clouds).
 * The same vote_body with 32-bit counters for one half of the slice's model points (PASS 1), then for the other (PASS 2).
 * A small fixed grid walks the redo list in a loop.  A translation unit of its own because of that loop: with machine
 * LICM on, the compiler hoists the loop-invariant constants of the whole vote body into vector registers that stay
 * live across it -- 128 registers and 36 bytes of scratch per lane; built with -mllvm -disable-machine-licm the body
 * fits as it does in k_vote (103 registers, no scratch; profiles/r03_kernel_resources.txt).
 */
#include <hip/hip_runtime.h>

#include "oslam_kernels.h"
#include "ppf_core.h"
#include "oslam_vote_body.inc"

#define VOTE_WIDE_GRID 256
template <int MODE, int PASS>
__global__ __launch_bounds__(VOTE_THREADS) void k_vote_wide(oslamk_vote_args a)
{
    const uint32_t n = a.counters->redo_count;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        vote_body<MODE, PASS>(a, a.redo[i]);
        __syncthreads();                            /* the LDS of this workgroup is reused by the next entry */
    }
}

/* both passes over the redo list of the launch that has just been queued on `stream` (almost always empty) */
extern "C" int oslamk_vote_wide(const oslamk_vote_args *a, void *stream)
{
    if (a->mode == 0) {
        hipLaunchKernelGGL((k_vote_wide<0, 1>), dim3(VOTE_WIDE_GRID), dim3(VOTE_THREADS), 0, (hipStream_t)stream, *a);
        hipLaunchKernelGGL((k_vote_wide<0, 2>), dim3(VOTE_WIDE_GRID), dim3(VOTE_THREADS), 0, (hipStream_t)stream, *a);
    } else {
        hipLaunchKernelGGL((k_vote_wide<1, 1>), dim3(VOTE_WIDE_GRID), dim3(VOTE_THREADS), 0, (hipStream_t)stream, *a);
        hipLaunchKernelGGL((k_vote_wide<1, 2>), dim3(VOTE_WIDE_GRID), dim3(VOTE_THREADS), 0, (hipStream_t)stream, *a);
    }
    return (int)hipGetLastError();
}
