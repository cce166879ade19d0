#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "fq_cios_ref.h"
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

template<int ILP> __global__ void k_mad64(uint32_t* out, int iters, uint32_t s) {
    uint32_t a = threadIdx.x*2654435761u + s, b = blockIdx.x*40503u + 12345u + s;
    uint64_t acc[ILP];
#pragma unroll
    for (int k=0;k<ILP;k++) acc[k] = a + k;
    for (int i=0;i<iters;i++) {
#pragma unroll
        for (int k=0;k<ILP;k++) acc[k] = (uint64_t)(uint32_t)acc[k] * b + acc[k];
    }
    uint64_t r=0;
#pragma unroll
    for (int k=0;k<ILP;k++) r ^= acc[k];
    out[blockIdx.x*blockDim.x+threadIdx.x] = (uint32_t)r ^ (uint32_t)(r>>32);
}
template<int ILP> __global__ void k_mullohi(uint32_t* out, int iters, uint32_t s) {
    uint32_t b = blockIdx.x*40503u + 12345u + s;
    uint32_t x[ILP], y[ILP];
#pragma unroll
    for (int k=0;k<ILP;k++) { x[k] = threadIdx.x + k + s; y[k]=k; }
    for (int i=0;i<iters;i++) {
#pragma unroll
        for (int k=0;k<ILP;k++) { uint32_t lo = x[k]*b; uint32_t hi = __umulhi(x[k], b); x[k] = lo ^ y[k]; y[k] = hi + 1; }
    }
    uint32_t r=0;
#pragma unroll
    for (int k=0;k<ILP;k++) r ^= x[k]^y[k];
    out[blockIdx.x*blockDim.x+threadIdx.x] = r;
}
template<int ILP> __global__ void k_add(uint32_t* out, int iters, uint32_t s) {
    uint32_t b = blockIdx.x*40503u + 12345u + s;
    uint64_t x[ILP];
#pragma unroll
    for (int k=0;k<ILP;k++) { x[k] = threadIdx.x + k + s; }
    for (int i=0;i<iters;i++) {
#pragma unroll
        for (int k=0;k<ILP;k++) { x[k] = x[k] + (x[k]>>7) + b; }   // 64-bit adds: add_co + addc
    }
    uint64_t r=0;
#pragma unroll
    for (int k=0;k<ILP;k++) r ^= x[k];
    out[blockIdx.x*blockDim.x+threadIdx.x] = (uint32_t)r^(uint32_t)(r>>32);
}
template<int ILP> __global__ void k_add32(uint32_t* out, int iters, uint32_t s) {
    uint32_t b = blockIdx.x*40503u + 12345u + s;
    uint32_t x[ILP];
#pragma unroll
    for (int k=0;k<ILP;k++) { x[k] = threadIdx.x + k + s; }
    for (int i=0;i<iters;i++) {
#pragma unroll
        for (int k=0;k<ILP;k++) { x[k] = (x[k] + b) ^ i; }  // 2 full-rate ops
    }
    uint32_t r=0;
#pragma unroll
    for (int k=0;k<ILP;k++) r ^= x[k];
    out[blockIdx.x*blockDim.x+threadIdx.x] = r;
}
template<int ILP> __global__ void k_fma64(uint32_t* out, int iters, uint32_t s) {
    double b = 1.0 + 1e-9*(blockIdx.x + s);
    double x[ILP];
#pragma unroll
    for (int k=0;k<ILP;k++) { x[k] = threadIdx.x + k + s; }
    for (int i=0;i<iters;i++) {
#pragma unroll
        for (int k=0;k<ILP;k++) { x[k] = __builtin_fma(x[k], b, 0.5); }
    }
    double r=0;
#pragma unroll
    for (int k=0;k<ILP;k++) r += x[k];
    out[blockIdx.x*blockDim.x+threadIdx.x] = (uint32_t)r;
}
__global__ void k_fqmul(uint32_t* out, int iters, uint32_t s) {
    uint32_t a[12], b[12];
#pragma unroll
    for (int j=0;j<12;j++) { a[j] = threadIdx.x*977u + j*s + 1; b[j] = blockIdx.x*31u + j + s; }
    a[11] &= 0x0fffffff; b[11] &= 0x0fffffff;
    for (int i=0;i<iters;i++) {
        uint32_t r[12];
        fq_mul_cios(r, a, b);
#pragma unroll
        for (int j=0;j<12;j++) { b[j] = a[j]; a[j] = r[j]; }
    }
    uint32_t x=0;
#pragma unroll
    for (int j=0;j<12;j++) x ^= a[j];
    out[blockIdx.x*blockDim.x+threadIdx.x] = x;
}
template<typename F> double timeit(F f, int reps=3) {
    hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    double best=1e30;
    for (int r=0;r<reps;r++){ CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); if(ms<best)best=ms; }
    return best;
}
int main() {
    uint32_t* d; CK(hipMalloc(&d, 1<<26));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
    printf("dev %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    int cus = p.multiProcessorCount;
    // waves per SIMD sweep: blocks of 256 threads = 4 waves = 1/SIMD per CU
    for (int wps : {1,2,4,8}) {
        int blocks = cus*wps; int iters=20000;
        constexpr int ILP=8;
        double ms;
        ms = timeit([&]{ hipLaunchKernelGGL(k_mad64<ILP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u); });
        double ops = (double)blocks*256*iters*ILP;
        printf("mad_u64_u32  wps=%d  %.3f ms  %.2f Tops/s  cyc/wave-instr/SIMD=%.2f\n", wps, ms, ops/ms*1e-9, ms*1e-3*2.4e9/((double)iters*ILP*wps));
        ms = timeit([&]{ hipLaunchKernelGGL(k_mullohi<ILP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u); });
        printf("mul_lo+hi    wps=%d  %.3f ms  %.2f Tpairs/s cyc/pair=%.2f\n", wps, ms, ops/ms*1e-9, ms*1e-3*2.4e9/((double)iters*ILP*wps));
        ms = timeit([&]{ hipLaunchKernelGGL(k_add<ILP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u); });
        printf("add64 x2+shift wps=%d  %.3f ms  cyc/iter=%.2f\n", wps, ms, ms*1e-3*2.4e9/((double)iters*ILP*wps));
        ms = timeit([&]{ hipLaunchKernelGGL(k_add32<ILP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u); });
        printf("add32+xor    wps=%d  %.3f ms  cyc/iter(2 ops)=%.2f\n", wps, ms, ms*1e-3*2.4e9/((double)iters*ILP*wps));
        ms = timeit([&]{ hipLaunchKernelGGL(k_fma64<ILP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u); });
        printf("fma_f64      wps=%d  %.3f ms  %.2f Tfma/s  cyc/wave-instr/SIMD=%.2f\n", wps, ms, ops/ms*1e-9, ms*1e-3*2.4e9/((double)iters*ILP*wps));
        int it2=2000;
        ms = timeit([&]{ hipLaunchKernelGGL(k_fqmul, dim3(blocks), dim3(256), 0, 0, d, it2, 3u); });
        printf("fq_mul_cios  wps=%d  %.3f ms  %.2f Gmul/s  cyc/mul/wave=%.0f\n", wps, ms, (double)blocks*256*it2/ms*1e-6, ms*1e-3*2.4e9/((double)it2*wps));
    }
    return 0;
}
