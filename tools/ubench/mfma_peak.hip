// Micro-benchmark: what does the MFMA pipe of this MI355X sustain on random vs zero operands?
// (measurement aid for roofline interpretation; not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_loop(const uint4* __restrict__ in, float* __restrict__ out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint4 va = in[(tid * 8 + i) & 0xfffff], vb = in[(tid * 8 + 4 + i) & 0xfffff];
        a[i] = __builtin_bit_cast(bf16x8, va);
        b[i] = __builtin_bit_cast(bf16x8, vb);
    }
    float s = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[i], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[i][e];
    } else {
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k], b[i & 3], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    out[tid] = s;
}

int main() {
    const int blocks = 256 * 4, threads = 512, iters = 4000;
    std::vector<uint16_t> h(8 << 20);
    uint4* d_in; float* d_out;
    hipMalloc(&d_in, h.size() * 2); hipMalloc(&d_out, blocks * threads * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int fill = 0; fill < 2; ++fill) {
        for (auto& v : h) { float f = fill ? (float)rand() / RAND_MAX * 2.f - 1.f : 0.f; uint32_t u; memcpy(&u, &f, 4); v = u >> 16; }
        hipMemcpy(d_in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        for (int shape : {32, 16}) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (shape == 32) hipLaunchKernelGGL(mfma_loop<32>, dim3(blocks), dim3(threads), 0, 0, d_in, d_out, iters);
                else hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(threads), 0, 0, d_in, d_out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double flops_per_wave_iter = shape == 32 ? 16.0 * 2 * 32 * 32 * 16 : 32.0 * 2 * 16 * 16 * 32;
                const double fl = flops_per_wave_iter * iters * (double)blocks * threads / 64;
                if (rep == 2) printf("fill=%s mfma=%dx%d  %.2f ms  %.0f TFLOP/s\n", fill ? "random" : "zero", shape, shape, ms, fl / ms / 1e9);
            }
        }
    }
    return 0;
}
