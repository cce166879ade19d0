// blsgpu_probe.hip -- measurement helpers of the timing entries (blsgpu_timing_mad_probe, blsgpu_timing_mark; include/blsgpu.h): no part of the
// verification path.  Included by blsgpu_api.hip like the other kernel files; emitted by the translation unit BLSGPU_TU_FXW.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "blsgpu_tu.h"

namespace blsgpu {
// ---- the chip's multiply-add rate, measured in the process that is about to be timed (blsgpu_timing_mad_probe) ---------------
// Every lane runs `iters` rounds of eight independent v_mad_i64_i32 chains; a grid of 2048 workgroups of 256 threads (eight
// wavefronts per SIMD) keeps every SIMD's issue slot busy.  The boxes of the pool differ by a few per cent in the clock the
// power limit leaves them, so bench.py prices its roofline against THIS run's figure as well as the
// constant of profiles/r01_intrate_microbench.txt.
namespace probe {
__global__ void __launch_bounds__(256) k_mad_probe(uint32_t* __restrict__ out, uint32_t iters, uint32_t seed)
#if BLSGPU_EMIT(BLSGPU_TU_FXW)
{
    const int32_t a = (int32_t)(threadIdx.x * 2654435761u + seed) | 1, b = (int32_t)(blockIdx.x * 40503u + 12345u + seed) | 1;
    int64_t acc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = a + k;
#pragma unroll 1
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = (int64_t)(int32_t)acc[k] * (int64_t)b + acc[k];
    }
    int64_t r = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) r ^= acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)r ^ (uint32_t)(r >> 32);
}
#else
;
#endif
// an empty kernel whose dispatches bracket a timed region in a rocprofv3 trace (blsgpu_timing_mark; tools/collect_profiles.py sums
// the counters of the dispatches between two of them)
__global__ void k_mark(uint32_t tag)
#if BLSGPU_EMIT(BLSGPU_TU_FXW)
{
    (void)tag;
}
#else
;
#endif
}  // namespace probe
}  // namespace blsgpu
