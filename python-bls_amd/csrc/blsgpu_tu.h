// blsgpu_tu.h -- how libblsgpu.so is cut into translation units.
//
// Every source is one text (blsgpu_api.hip includes the kernel files), compiled several times with -DBLSGPU_TU=<n>:
// a kernel's BODY is emitted by exactly one of them, the others see its declaration, and the host side of the C ABI is
// emitted by BLSGPU_TU_HOST alone.  The objects are linked into one library (csrc/Makefile); kernels launched from the
// host translation unit resolve to the stubs of the one that defines them.  Without -DBLSGPU_TU (0) everything is
// emitted at once -- the single-file build of rounds 1-3, 5.5 minutes against ~2 with `make -j`.
#pragma once
#define BLSGPU_TU_HOST 1
#define BLSGPU_TU_VM 2      // blsgpu_kernels.hip: the wavefront VM's Miller / reduce / final exponentiation kernels
#define BLSGPU_TU_ML 3      // blsgpu_ml.hip: line-stream Miller stage
#define BLSGPU_TU_FX 4      // blsgpu_fexp.hip: batched final exponentiations
#define BLSGPU_TU_MSM 5     // blsgpu_msm.hip: multi-scalar sums
#define BLSGPU_TU_H2C 6     // blsgpu_h2c.hip: hash to G2, decompression
#define BLSGPU_TU_FXW 7     // blsgpu_fexpw.hip: one final exponentiation per wavefront
#ifndef BLSGPU_TU
#define BLSGPU_TU 0
#endif
#define BLSGPU_EMIT(g) (BLSGPU_TU == 0 || BLSGPU_TU == (g))


// Workgroup shape of the register kernels (one unit of work per lane or lane team, nothing shared between wavefronts).
// Measured (tools/microbench/wg_placement.hip, profiles/r04_workgroup_shape.txt): a launch of fewer wavefronts than the
// chip holds, issued as 64-thread workgroups of a kernel with a large register allocation, gets wavefronts stacked two
// to a SIMD while other SIMDs stay empty (k_msm_horner_quads, 625 wavefronts: 4.2 ms; as 157 workgroups of 256 threads
// -- four wavefronts, one per SIMD of a CU -- 2.4 ms).  Such kernels index by wave_index() / the global thread index
// and are launched through blsgpu_api.hip's wave_shape().
#if defined(__HIPCC__)
namespace blsgpu {
__device__ __forceinline__ unsigned wave_index() { return blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); }
}
#endif
