// Gate 2 of the Winograd question, priced with the operand-cost model that reproduces the shipped conv kernel
// (tools/ubench/operand_cost.hip: "4 W + 4 LDS + 2/3 DMA" measured 1 398 TFLOP/s where conv_halo2_kernel runs 1 345-1 418): a 32x32x16
// bf16 MFMA loop on random data, 16 MFMAs per wave and iteration, ONE wave per SIMD (512 registers, as the 8-row kernel), plus per
// iteration and wave
//   NW  x 1 KiB global_load_dwordx4 from an L1/L2-hot buffer   (weight fragments; hand-issued two iterations ahead, counted vmcnt)
//   NL  x 1 KiB ds_read_b128                                   (halo-row / transformed-input fragments)
//   ND3 x 1 KiB global_load_lds per THREE iterations from an HBM-sized buffer   (halo staging)
//   NV  x v_pk_add_f16 on live registers per iteration         (the input transform's adds, interleaved with the MFMAs)
// Operand traffic per MFMA of the candidate register blockings (256 accumulators per wave in every case):
//   direct, 8 rows x 64 couts (shipped)        : a weight fragment serves 8 MFMAs, a halo-row fragment ~5   -> NW 2, NL 3, ND 2/3
//   F(2x2,3x3), 16 positions x (32 tiles x 32 couts): every MFMA needs a fresh A (built from 16 halo reads per 16 positions) AND a
//                                                fresh B fragment                                             -> NW 16, NL 16, NV 128
//   F(2x2,3x3), 4 positions x (64 tiles x 64 couts) per wave, the 16 positions split over the 4 waves          -> NW 8, NL 8, NV 64
//   temporal F(2,3), wave = position (8 rows x 64 couts per wave as shipped, its own 10x34 halo per wave: 2.2x the staging per MFMA)
//                                                                                                              -> NW 2, NL 3, ND 4/3..5/3
// A Winograd form wins only if  rate(form) / rate(shipped) x (MFMA reduction: 2.25 spatial, 1.5 temporal)  > 1.25  BEFORE its output
// transform, epilogue growth and (temporal) the doubled GroupNorm-apply writes are charged.   (measurement aid, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int NW, int NL, int ND3, int NV>
__global__ __launch_bounds__(256, 1) void loop(const uint4* __restrict__ wbuf, const uint4* __restrict__ hbuf, size_t hmask,
                                               float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 128 KiB: 32 KiB per wave (16 fragment slots + 16 staging slots)
    constexpr int WN = NW > 0 ? NW : 1, LN = NL > 0 ? NL : 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* mine = smem + wave * 32768;
    for (int i = 0; i < 16; ++i) *(uint4*)(mine + i * 1024 + lane * 16) = wbuf[(blockIdx.x * 1024 + wave * 256 + i * 64 + lane) & 0xffff];
    __syncthreads();
    // (hand-issued asynchronous loads are only safe while hipcc never copies a pending destination register: with NW >= 8 three
    // rotating sets overflow into AGPR copies -- the first version of this file faulted there -- so those forms run TWO sets, i.e. a
    // prefetch distance of one iteration instead of two)
    constexpr bool TWO = NW >= 8;
    bf16x8 w0[WN], w1[WN], w2[TWO ? 1 : WN], a[LN];
#pragma unroll
    for (int i = 0; i < WN; ++i) { w0[i] = __builtin_bit_cast(bf16x8, wbuf[(i * 64 + lane + wave * 512) & 0xffff]); w1[i] = w0[i]; }
#pragma unroll
    for (int i = 0; i < (TWO ? 1 : WN); ++i) w2[i] = w0[0];
#pragma unroll
    for (int i = 0; i < LN; ++i) a[i] = *(const bf16x8*)(mine + (i & 15) * 1024 + lane * 16);
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    h2 tv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) tv[e] = h2{(_Float16)(0.001f * lane + e), (_Float16)(0.5f - e)};
    size_t hpos = ((size_t)blockIdx.x * 4 + wave) * 977 * 64 + lane;
    const unsigned lds_mine = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)mine + lane * 16;
    const int voff = lane * 16;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    auto body = [&](auto ph, int it, bf16x8 (&wc)[WN], bf16x8 (&wn)[WN]) {
        constexpr int PH = decltype(ph)::value;
        constexpr int D0 = ND3 / 3 + (PH < ND3 % 3 ? 1 : 0);                       // LDS-DMA pieces issued in this iteration
        constexpr int D1 = ND3 / 3 + (((PH + 2) % 3) < ND3 % 3 ? 1 : 0);           // ... in the previous one
        const char* wp = (const char*)wbuf + ((it * WN) & 63) * 1024;
#pragma unroll
        for (int i = 0; i < NW; ++i)
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(wn[i]) : "v"(voff + i * 1024), "s"(wp) : "memory");
#pragma unroll
        for (int d = 0; d < D0; ++d) {
            glds16(hbuf + (hpos & hmask), mine + 16384 + ((it * 2 + d) & 15) * 1024);
            hpos += 64;
        }
#pragma unroll
        for (int i = 0; i < NL; ++i)
            asm volatile("ds_read_b128 %0, %1" : "=v"(a[i]) : "v"(lds_mine + (unsigned)(((it * NL + i) & 15) * 1024)) : "memory");
        // this iteration's weights were issued two (TWO: one) iterations ago: younger VMEM ops = the weight loads + DMA pieces since
        constexpr int YOUNGER = TWO ? NW + D0 : 2 * NW + D0 + D1;
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(YOUNGER > 63 ? 63 : YOUNGER) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            acc[j & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[j % WN], a[j % LN], acc[j & 7], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV / 16; ++v) tv[(j + v) & 7] = tv[(j + v) & 7] + tv[(j + v + 3) & 7];   // v_pk_add_f16
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (TWO) {
        for (int it = 0; it < iters; it += 6) {
            body(std::integral_constant<int, 0>{}, it, w0, w1);
            body(std::integral_constant<int, 1>{}, it + 1, w1, w0);
            body(std::integral_constant<int, 2>{}, it + 2, w0, w1);
            body(std::integral_constant<int, 0>{}, it + 3, w1, w0);
            body(std::integral_constant<int, 1>{}, it + 4, w0, w1);
            body(std::integral_constant<int, 2>{}, it + 5, w1, w0);
        }
    } else {
        for (int it = 0; it < iters; it += 3) {
            body(std::integral_constant<int, 0>{}, it, w0, w2);
            body(std::integral_constant<int, 1>{}, it + 1, w1, w0);
            body(std::integral_constant<int, 2>{}, it + 2, w2, w1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
#pragma unroll
    for (int e = 0; e < 8; ++e) s += (float)tv[e][0] + (float)tv[e][1];
#pragma unroll
    for (int i = 0; i < WN; ++i) s += (float)w0[i][0] + (float)w1[i][0];
    s += (float)w2[0][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static uint4* d_w; static uint4* d_h; static float* d_out;
static hipEvent_t e0, e1;
static double g_base = 0.0;

template <int NW, int NL, int ND3, int NV>
static double run(const char* label, size_t hmask, double mfma_reduction) {
    const int blocks = 256 * 8, iters = 3000;
    auto kern = loop<NW, NL, ND3, NV>;
    const int lds = 131072;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d_w, d_h, hmask, d_out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double fl = 16.0 * 2 * 32 * 32 * 16 * iters * (double)blocks * 4;
    const double rate = fl / best / 1e9;
    if (g_base == 0.0 && mfma_reduction == 1.0 && NW == 2) g_base = rate;
    printf("%-78s NW=%2d NL=%2d ND=%d/3 NV=%3d  %8.2f ms  %6.0f TFLOP/s (MFMA rate)", label, NW, NL, ND3, NV, best, rate);
    if (g_base > 0.0) printf("  -> effective x%.2f vs shipped mix", rate / g_base * mfma_reduction);
    printf("\n");
    fflush(stdout);
    return rate;
}

int main() {
    const size_t hbytes = (size_t)2 << 30;
    std::vector<uint16_t> h(32 << 20);
    for (auto& v : h) { float f = (float)rand() / RAND_MAX * 2.f - 1.f; uint32_t u; memcpy(&u, &f, 4); v = u >> 16; }
    hipMalloc(&d_w, 1 << 20); hipMalloc(&d_h, hbytes); hipMalloc(&d_out, 256 * 8 * 256 * 4);
    hipMemcpy(d_w, h.data(), 1 << 20, hipMemcpyHostToDevice);
    for (size_t off = 0; off < hbytes; off += h.size() * 2) hipMemcpy((char*)d_h + off, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t HBM = (hbytes / 16) - 1;
    run<0, 0, 0, 0>("bare MFMA loop, one wave per SIMD", HBM, 1.0);
    run<2, 3, 2, 0>("shipped blocking: direct, 8 rows x 64 couts per wave (2 W + 3 LDS + 2/3 DMA)", HBM, 1.0);
    // where the staged halo comes from: the same mix with the LDS-DMA source confined to windows that fit the L2s (4 MiB per XCD),
    // the memory-side cache (256 MiB MALL) or nothing
    run<2, 3, 2, 0>("   shipped mix, halo source window  4 MiB (L2-resident)", ((size_t)4 << 20) / 16 - 1, 1.0);
    run<2, 3, 2, 0>("   shipped mix, halo source window 16 MiB (half of all L2s together)", ((size_t)16 << 20) / 16 - 1, 1.0);
    run<2, 3, 2, 0>("   shipped mix, halo source window 128 MiB (MALL-resident)", ((size_t)128 << 20) / 16 - 1, 1.0);
    run<2, 3, 2, 0>("   shipped mix, halo source window 512 MiB (beyond the MALL)", ((size_t)512 << 20) / 16 - 1, 1.0);
    run<16, 16, 2, 0>("F(2x2,3x3) 16 positions x (32 tiles x 32 couts): operands only, no transform VALU", HBM, 2.25);
    run<16, 16, 2, 128>("F(2x2,3x3) 16 positions x (32 x 32) + the input transform's 128 packed adds", HBM, 2.25);
    run<8, 8, 2, 0>("F(2x2,3x3) 4 positions x (64 tiles x 64 couts) per wave: operands only", HBM, 2.25);
    run<8, 8, 2, 64>("F(2x2,3x3) 4 positions x (64 x 64) per wave + 64 packed adds", HBM, 2.25);
    // ... and with the halo staging such a workgroup really needs: 256 accumulators per wave hold 16 positions x 64 tiles x 64 couts per
    // WORKGROUP = 256 output voxels x 64 couts (the shipped tile: 512 voxels x 128 couts), so a 6 x 66 halo (25 KiB per 32-channel slice)
    // feeds 4 x 32 MFMAs: 3.2 KiB of LDS-DMA per 16 MFMAs and wave, five times the shipped kernel's 0.54 KiB
    run<8, 8, 5, 64>("F(2x2,3x3) 4 positions x (64 x 64) + adds, 5/3 KiB DMA per iteration", HBM, 2.25);
    run<8, 8, 10, 64>("F(2x2,3x3) 4 positions x (64 x 64) + adds, 10/3 KiB DMA per iteration (its real halo)", HBM, 2.25);
    run<8, 8, 10, 64>("   the same, staging from a cache-resident source", ((size_t)4 << 20) / 16 - 1, 2.25);
    run<2, 3, 4, 0>("temporal F(2,3), wave = position, own 10x34 halo per wave (4/3 DMA)", HBM, 1.5);
    run<2, 3, 5, 0>("temporal F(2,3), wave = position, own 10x34 halo per wave (5/3 DMA)", HBM, 1.5);
    run<2, 6, 2, 24>("temporal F(2,3), four raw frame halos shared, A = fp16 sum of two reads (6 LDS + 24 adds)", HBM, 1.5);
    return 0;
}
