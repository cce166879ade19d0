// How the dispatcher places the workgroups of a launch that does not fill the chip (fewer wavefronts than the 1024
// SIMDs), and what that costs a latency-bound kernel: W wavefronts of a dependent multiply-add chain launched as
//   (a) W workgroups of 64 threads, (b) W/4 workgroups of 256 threads, (c) as (b) with 96 KB of dynamic LDS per
//   workgroup (one workgroup per CU), (d) as (a) with 48 KB of LDS (three workgroups per CU at most).
// Every wavefront records the CU / SIMD it ran on (HW_ID) so the stacking can be counted.
// build: hipcc --offload-arch=gfx950 -O3 -o wg_placement wg_placement.hip ; run: ./wg_placement [wavefronts ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
extern __shared__ unsigned char dyn_lds[];
__global__ void __launch_bounds__(256) chain(unsigned long long* out, unsigned* where, int iters, int use_lds) {
    unsigned long long a = threadIdx.x + 1, b = 0x9E3779B97F4A7C15ull;
    int x = (int)threadIdx.x * 7 + 3;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) a = (unsigned long long)((long long)x * (long long)(int)(a >> 7) + (long long)a);   // v_mad_i64_i32, dependent
    }
    if (use_lds && a == 12345) dyn_lds[threadIdx.x] = 1;
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    out[gid] = a + b;
    if ((threadIdx.x & 63) == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        where[gid >> 6] = (hw & 0xFFFFFFu) | ((xcc & 0xFu) << 24);
    }
}
int main(int argc, char** argv) {
    std::vector<int> sizes;
    for (int i = 1; i < argc; i++) sizes.push_back(atoi(argv[i]));
    if (sizes.empty()) sizes = {256, 512, 625, 1000, 1024};
    unsigned long long* out; unsigned* where;
    hipMalloc(&out, 8ull * 64 * 4096); hipMalloc(&where, 4 * 4096);
    hipFuncSetAttribute((const void*)chain, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int W : sizes) {
        struct { const char* name; int threads; size_t lds; } forms[] = {{"64-thread workgroups", 64, 0}, {"256-thread workgroups", 256, 0},
            {"256-thread workgroups, 96 KB LDS", 256, 96 * 1024}, {"64-thread workgroups, 48 KB LDS", 64, 48 * 1024}};
        for (auto& f : forms) {
            const int blocks = (W * 64 + f.threads - 1) / f.threads;
            float best = 1e9, worst = 0;
            int maxstack = 0, stacked = 0;
            for (int rep = 0; rep < 6; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(chain, dim3(blocks), dim3(f.threads), f.lds, 0, out, where, iters, f.lds ? 1 : 0);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
                if (ms > worst) worst = ms;
                std::vector<unsigned> h(blocks * f.threads / 64);
                hipMemcpy(h.data(), where, 4 * h.size(), hipMemcpyDeviceToHost);
                std::map<unsigned, int> cnt;                       // (xcc, se, sh, cu, simd) -> wavefronts
                for (unsigned v : h) cnt[((v >> 24) << 16) | (((v >> 13) & 7) << 12) | (((v >> 12) & 1) << 11) | (((v >> 8) & 15) << 4) | ((v >> 4) & 3)]++;
                maxstack = 0; stacked = 0;
                for (auto& kv : cnt) { if (kv.second > maxstack) maxstack = kv.second; if (kv.second > 1) stacked += kv.second; }
                if (rep == 5) printf("W=%5d %-36s blocks %5d: %.3f .. %.3f ms; last run: %zu SIMDs used, most on one SIMD %d, wavefronts sharing a SIMD %d\n",
                                     W, f.name, blocks, best, worst, cnt.size(), maxstack, stacked);
            }
        }
    }
    return 0;
}
