// svr_mfma_calibrate: what the matrix pipes of THIS device sustain right now on a bare v_mfma_f32_32x32x16_bf16 stream -- the
// power-limited ceiling that bench.py reports next to the nominal peak, so that a record separates "the box clocks lower" from
// "the kernel got worse".  Not on the data path.  One 512-thread workgroup per CU x 4 (two waves per SIMD, 4 x 16 accumulators
// each): no LDS, no memory traffic inside the loop; operands are pseudo-random bf16 in (-1, 1) derived from the lane id (the chip
// is power-managed: all-zero operands clock ~20 % higher), accumulators are written out once so nothing is dead code.
#pragma once
#include "svr_common.h"

namespace svr {

__global__ __launch_bounds__(512) void mfma_calibrate_kernel(float* __restrict__ out, const int iters) {
    typedef __attribute__((ext_vector_type(16))) float f32x16_t;
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[4], b[4];
    unsigned h = tid * 2654435761u + 0x9e3779b9u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            h ^= h << 13; h ^= h >> 17; h ^= h << 5;           // xorshift32
            // a bf16 with a random sign and mantissa and an exponent in [2^-4, 1): 0x3d80..0x3f7f
            w[e] = ((h >> 16) & 0x8000u) | (0x3d80u + ((h >> 3) & 0x1ffu) % 0x200u);
        }
        uint4 va = {w[0] | (w[1] << 16), w[2] | (w[3] << 16), w[4] | (w[5] << 16), w[6] | (w[7] << 16)};
        uint4 vb = {w[1] | (w[0] << 16), w[3] | (w[2] << 16), w[5] | (w[4] << 16), w[7] | (w[6] << 16)};
        a[i] = __builtin_bit_cast(bf16x8, va);
        b[i] = __builtin_bit_cast(bf16x8, vb);
    }
    f32x16_t acc[4];
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
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[tid] = s;
}

}  // namespace svr
