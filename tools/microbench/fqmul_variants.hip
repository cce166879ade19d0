#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../python-bls_amd/csrc/fq32.h"
#include "fips_proto.h"
#include "../../python-bls_amd/csrc/fq_mul_gfx950.h"
#include "fq_mul_n0.h"
#include "fq_mul_n1.h"
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
template<int V> __global__ void k_fqmul(uint32_t* out, int iters, uint32_t s) {
    uint32_t a[12], b[12];
#pragma unroll
    for (int j=0;j<12;j++) { a[j] = threadIdx.x*977u + j*s + 1; b[j] = blockIdx.x*31u + j + s; }
    a[11] &= 0x0fffffff; b[11] &= 0x0fffffff;
    for (int i=0;i<iters;i++) {
        uint32_t r[12];
        if (V==0) bls::fq_mul(r, a, b); else if (V==1) fq_mul_fips(r,a,b); else if (V==2) bls::fq_mul_dev<true>(r,a,b); else if (V==3) bls::fq_mul_dev_n0<true>(r,a,b); else bls::fq_mul_dev_n1<true>(r,a,b);
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
    uint32_t* h=(uint32_t*)malloc(1<<20);
    int cus=256;
    // correctness cross-check of the two variants
    hipLaunchKernelGGL(k_fqmul<0>, dim3(4), dim3(256), 0, 0, d, 50, 3u); CK(hipMemcpy(h,d,4096,hipMemcpyDeviceToHost));
    uint32_t ref[1024]; memcpy(ref,h,4096);
    hipLaunchKernelGGL(k_fqmul<1>, dim3(4), dim3(256), 0, 0, d, 50, 3u); CK(hipMemcpy(h,d,4096,hipMemcpyDeviceToHost));
    printf("variants agree: %d\n", memcmp(ref,h,4096)==0);
    hipLaunchKernelGGL(k_fqmul<2>, dim3(4), dim3(256), 0, 0, d, 50, 3u); CK(hipMemcpy(h,d,4096,hipMemcpyDeviceToHost));
    printf("generated variant agrees: %d\n", memcmp(ref,h,4096)==0);
    for (int wps : {1,2,4}) {
        int blocks = cus*wps; int it2=2000; double ms;
        ms = timeit([&]{ hipLaunchKernelGGL(k_fqmul<0>, dim3(blocks), dim3(256), 0, 0, d, it2, 3u); });
        printf("cios  wps=%d  %.3f ms  %.2f Gmul/s  cyc/mul/wave=%.0f\n", wps, ms, (double)blocks*256*it2/ms*1e-6, ms*1e-3*2.4e9/((double)it2*wps));
        ms = timeit([&]{ hipLaunchKernelGGL(k_fqmul<2>, dim3(blocks), dim3(256), 0, 0, d, it2, 3u); });
        printf("gen   wps=%d  %.3f ms  %.2f Gmul/s  cyc/mul/wave=%.0f\n", wps, ms, (double)blocks*256*it2/ms*1e-6, ms*1e-3*2.4e9/((double)it2*wps));
        ms = timeit([&]{ hipLaunchKernelGGL(k_fqmul<3>, dim3(blocks), dim3(256), 0, 0, d, it2, 3u); });
        printf("gen+nop0 wps=%d  %.3f ms  %.2f Gmul/s  cyc/mul/wave=%.0f\n", wps, ms, (double)blocks*256*it2/ms*1e-6, ms*1e-3*2.4e9/((double)it2*wps));
        ms = timeit([&]{ hipLaunchKernelGGL(k_fqmul<4>, dim3(blocks), dim3(256), 0, 0, d, it2, 3u); });
        printf("gen+nop1 wps=%d  %.3f ms  %.2f Gmul/s  cyc/mul/wave=%.0f\n", wps, ms, (double)blocks*256*it2/ms*1e-6, ms*1e-3*2.4e9/((double)it2*wps));
        ms = timeit([&]{ hipLaunchKernelGGL(k_fqmul<1>, dim3(blocks), dim3(256), 0, 0, d, it2, 3u); });
        printf("fips1 wps=%d  %.3f ms  %.2f Gmul/s  cyc/mul/wave=%.0f\n", wps, ms, (double)blocks*256*it2/ms*1e-6, ms*1e-3*2.4e9/((double)it2*wps));
    }
    return 0;
}
