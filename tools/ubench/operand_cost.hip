// Micro-benchmark: what does each operand path cost a power-capped MFMA loop on MI355X?
// A 32x32x16 bf16 MFMA loop on random data (16 MFMAs per wave and iteration, 8 accumulators -- the conv kernel's interval)
// plus, per iteration and wave, a selectable amount of operand traffic of the three kinds the LDS-halo conv kernel uses:
//   NW  x 1 KiB global_load_dwordx4 from an L1/L2-hot buffer, consumed as MFMA A operands   (its weight fragments: 4)
//   NL  x 1 KiB ds_read_b128, consumed as MFMA B operands                                    (its halo-row fragments: 4 on average)
//   ND3 x 1 KiB global_load_lds per THREE iterations from a cache-resident or an HBM-sized buffer   (its halo staging: 2 per 3)
// Loads are issued one iteration ahead (register double buffer), two workgroups of four waves per CU like the conv kernel.
// (measurement aid for DESIGN.md section 3.1; not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_w(bf16x8 (&w)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : "n"(N));
    __builtin_amdgcn_sched_barrier(0);
}

// Loads are hand-issued (inline asm) with counted waits, like the conv kernel's: weights two iterations ahead in three
// rotating register sets, LDS reads right before the burst (the partner wave on the SIMD covers their latency).
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + fast_exp2(-1.4426950408889634f * x)); }

// NG3 > 0: instead of LDS-DMA the halo chunk is loaded into registers (16 bytes per lane), normalised (x * a + b per channel),
// SiLU'd, rounded to bf16 and written to LDS with ds_write_b128 one G-phase later -- what fusing GroupNorm-apply + SiLU into the
// conv kernel's halo staging would add to a wave; the VALU work is interleaved with the MFMAs by sched_group_barrier.
template <int NW, int NL, int ND3, int WPS, int NG3 = 0>
__global__ __launch_bounds__(256, WPS) void loop(const uint4* __restrict__ wbuf, const uint4* __restrict__ hbuf, size_t hmask,
                                                 float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 64 KiB: 16 KiB per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* mine = smem + wave * 16384;
    for (int i = 0; i < 16; ++i) *(uint4*)(mine + i * 1024 + lane * 16) = wbuf[(blockIdx.x * 1024 + wave * 256 + i * 64 + lane) & 0xffff];
    __syncthreads();
    bf16x8 w0[4], w1[4], w2[4], a[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        w0[i] = __builtin_bit_cast(bf16x8, wbuf[(i * 64 + lane + wave * 512) & 0xffff]);
        w1[i] = w0[i]; w2[i] = w0[i];
        a[i] = *(const bf16x8*)(mine + i * 1024 + lane * 16);
    }
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    size_t hpos = ((size_t)blockIdx.x * 4 + wave) * 977 * 64 + lane;    // 16-byte units; every wave streams its own region
    const unsigned lds_mine = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)mine + lane * 16;
    const int voff = lane * 16;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    u32x4 gx = __builtin_bit_cast(u32x4, wbuf[lane]);                  // pending register-staged chunk
    float ga[8], gb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ga[e] = 1.0f + 0.01f * (lane & 7) + 0.1f * e; gb[e] = 0.05f * e - 0.2f; }
    auto body = [&](auto ph, int it, bf16x8 (&wc)[4], bf16x8 (&wn)[4]) {
        constexpr int PH = decltype(ph)::value;                       // it % 3
        constexpr int NX3 = ND3 + NG3;                                // one extra VMEM op in the first NX3 phases of every three
        constexpr int D0 = PH < NX3 ? 1 : 0, D1 = ((PH + 2) % 3) < NX3 ? 1 : 0;   // issued this / the previous iteration
        // weight fragments of iteration it + 2 (a 64 KiB window: L1/L2-hot)
        const char* wp = (const char*)wbuf + ((it * NW) & 63) * 1024;
#pragma unroll
        for (int i = 0; i < NW; ++i)
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(wn[i]) : "v"(voff + i * 1024), "s"(wp) : "memory");
        u32x4 gy = gx;
        if constexpr (D0 != 0 && NG3 == 0) {
            glds16(hbuf + (hpos & hmask), mine + 8192 + (it & 7) * 1024);
            hpos += 64;
        }
        if constexpr (D0 != 0 && NG3 != 0) {
            // the chunk loaded in the previous G phase: younger VMEM ops = the weight loads since then
            constexpr int GAP = PH == 0 ? (NX3 == 1 ? 3 : 2) : 1;      // iterations since that load (phases 0..NX3-1 are G phases)
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(gx) : "n"(GAP * NW));
            gy = gx;
            const uint4* src = hbuf + (hpos & hmask);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gx) : "v"(src) : "memory");
            hpos += 64;
        }
#pragma unroll
        for (int i = 0; i < NL; ++i)
            asm volatile("ds_read_b128 %0, %1" : "=v"(a[i]) : "v"(lds_mine + (unsigned)(((it * NL + i) & 7) * 1024)) : "memory");
        wait_w<2 * NW + D0 + D1>(wc);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (D0 != 0 && NG3 != 0) {
            // 16 MFMAs with the chunk's VALU work spread between them
            const unsigned u[4] = {gy[0], gy[1], gy[2], gy[3]};
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = __builtin_bit_cast(float, (e & 1) ? (u[e >> 1] & 0xffff0000u) : (u[e >> 1] << 16));
                f[e] = silu(x * ga[e] + gb[e]);
            }
            uint4 o;
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o.x) : "v"(f[0]), "v"(f[1]));
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o.y) : "v"(f[2]), "v"(f[3]));
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o.z) : "v"(f[4]), "v"(f[5]));
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o.w) : "v"(f[6]), "v"(f[7]));
            *(uint4*)(mine + 8192 + (it & 7) * 1024 + lane * 16) = o;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[j & 3], a[(j >> 2) & 3], acc[j & 7], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x006, 5, 0);     // five VALU / SALU
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[j & 3], a[(j >> 2) & 3], acc[j & 7], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int it = 0; it < iters; it += 3) {
        body(std::integral_constant<int, 0>{}, it, w0, w2);
        body(std::integral_constant<int, 1>{}, it + 1, w1, w0);
        body(std::integral_constant<int, 2>{}, it + 2, w2, w1);
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(w0[0]), "+v"(w0[1]), "+v"(w0[2]), "+v"(w0[3]));
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(w1[0]), "+v"(w1[1]), "+v"(w1[2]), "+v"(w1[3]));
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(w2[0]), "+v"(w2[1]), "+v"(w2[2]), "+v"(w2[3]));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static uint4* d_w; static uint4* d_h; static float* d_out;
static hipEvent_t e0, e1;

template <int NW, int NL, int ND3, int WPS, int NG3 = 0>
static void run(const char* label, size_t hmask) {
    const int blocks = 256 * 2 * 8, iters = 3000;
    auto kern = loop<NW, NL, ND3, WPS, NG3>;
    const int lds = WPS == 1 ? 100000 : 65536;
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
    printf("%-64s NW=%d NL=%d ND=%d/3 NG=%d/3 wg/CU=%d  %8.2f ms  %6.0f TFLOP/s\n", label, NW, NL, ND3, NG3, WPS, best, fl / best / 1e9);
    fflush(stdout);
}

int main() {
    const size_t hbytes = (size_t)2 << 30;
    std::vector<uint16_t> h(32 << 20);
    for (auto& v : h) { float f = (float)rand() / RAND_MAX * 2.f - 1.f; uint32_t u; memcpy(&u, &f, 4); v = u >> 16; }
    hipMalloc(&d_w, 1 << 20); hipMalloc(&d_h, hbytes); hipMalloc(&d_out, 256 * 2 * 8 * 256 * 4);
    hipMemcpy(d_w, h.data(), 1 << 20, hipMemcpyHostToDevice);
    for (size_t off = 0; off < hbytes; off += h.size() * 2) hipMemcpy((char*)d_h + off, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t HBM = (hbytes / 16) - 1, L2 = ((size_t)4 << 20) / 16 - 1;
    run<0, 0, 0, 2>("pure MFMA loop (random operands)", L2);
    run<0, 0, 0, 1>("pure MFMA loop, one workgroup per CU", L2);
    run<4, 0, 0, 2>("+ 4 weight fragments from L1 (the conv kernel's)", L2);
    run<2, 0, 0, 2>("+ 2 weight fragments from L1", L2);
    run<0, 4, 0, 2>("+ 4 LDS fragment reads (the conv kernel's average)", L2);
    run<0, 2, 0, 2>("+ 2 LDS fragment reads", L2);
    run<0, 0, 2, 2>("+ 2/3 KiB LDS-DMA per iteration, cache-resident source", L2);
    run<0, 0, 2, 2>("+ 2/3 KiB LDS-DMA per iteration, HBM stream", HBM);
    run<0, 0, 1, 2>("+ 1/3 KiB LDS-DMA per iteration, HBM stream", HBM);
    run<4, 4, 2, 2>("the conv kernel's mix: 4 W + 4 LDS + 2/3 DMA (HBM)", HBM);
    run<4, 4, 2, 1>("the conv kernel's mix, one workgroup per CU", HBM);
    run<2, 4, 2, 2>("half the weight loads (8 rows x 64 couts per wave)", HBM);
    run<4, 2, 1, 2>("half the LDS reads and staging (two output frames)", HBM);
    run<2, 2, 1, 2>("both halved", HBM);
    // the 8-row kernel's mix (one wave per SIMD) and what fusing GroupNorm-apply + SiLU into its halo staging would add
    run<2, 3, 2, 1>("8-row kernel's mix: 2 W + 3 LDS + 2/3 DMA (HBM), 1 wg/CU", HBM);
    run<2, 3, 0, 1>("8-row mix without halo staging", HBM);
    run<2, 3, 0, 1, 2>("8-row mix, halo through registers + GroupNorm + SiLU (2/3 chunk/iter)", HBM);
    run<2, 3, 0, 1, 1>("8-row mix, halo through registers + GroupNorm + SiLU (1/3 chunk/iter)", HBM);
    return 0;
}
