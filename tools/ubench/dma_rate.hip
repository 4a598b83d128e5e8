// Micro-benchmark: per-CU throughput of global_load_lds (LDS-DMA) vs global_load_dwordx4 (+ds_write_b128)
// for 1 KiB-per-wave-instruction streaming of an L2-resident 64 KiB window per workgroup.
// (measurement aid; not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

template <int MODE>   // 0: global_load_lds, 1: global_load + ds_write_b128, 2: global_load only
__global__ __launch_bounds__(512) void stream(const char* __restrict__ src, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (size_t)blockIdx.x * 65536;
    uint4 accv = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int off = ((it * 8 + j) * 8192 + wave * 1024 + lane * 16) & 65535;
            if constexpr (MODE == 0) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                                 (__attribute__((address_space(3))) void*)(smem + j * 8192 + wave * 1024), 16, 0, 0);
            } else {
                const uint4 v = *(const uint4*)(base + off);
                if constexpr (MODE == 1) *(uint4*)(smem + j * 8192 + wave * 1024 + lane * 16) = v;
                else { accv.x ^= v.x; accv.y ^= v.y; accv.z ^= v.z; accv.w ^= v.w; }
            }
        }
        if constexpr (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (MODE != 2) accv = *(const uint4*)(smem + tid * 16);
    out[blockIdx.x * 512 + tid] = (float)(accv.x ^ accv.y ^ accv.z ^ accv.w);
}

int main() {
    const int blocks = 256, iters = 2000;
    char* d_src; float* d_out;
    hipMalloc(&d_src, (size_t)blocks * 65536); hipMalloc(&d_out, blocks * 512 * 4);
    hipMemset(d_src, 1, (size_t)blocks * 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)stream<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)stream<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)stream<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const char* names[3] = {"global_load_lds (LDS-DMA)", "global_load_dwordx4 + ds_write_b128", "global_load_dwordx4 only"};
    for (int mode = 0; mode < 3; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(stream<0>, dim3(blocks), dim3(512), 65536, 0, d_src, d_out, iters);
            if (mode == 1) hipLaunchKernelGGL(stream<1>, dim3(blocks), dim3(512), 65536, 0, d_src, d_out, iters);
            if (mode == 2) hipLaunchKernelGGL(stream<2>, dim3(blocks), dim3(512), 65536, 0, d_src, d_out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)blocks * iters * 8 * 8192;
            if (rep == 1) printf("%-40s %.3f ms  %.2f TB/s chip  %.1f GB/s per CU  (%.1f ns per 1 KiB wave-instr per CU)\n", names[mode], ms,
                                 bytes / ms / 1e9, bytes / ms / 1e6 / blocks, ms * 1e6 / (iters * 64.0));
        }
    return 0;
}
