// Standalone probe: what does this MI355X give a 2-byte-in / 2-byte-out streaming pass, and what does the GroupNorm-apply + SiLU math of
// svr_groupnorm_apply (6.8 % of a BASELINE config 3 step at 5.0 TB/s; MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy) cost on top?
// The kernel below is that pass re-stated with its knobs as template arguments -- same per-element arithmetic (svr_common.h: h16 unpack,
// x * a_c + b_c, SiLU by v_exp_f32 + v_rcp_f32, hardware bf16 pack), so every variant of a MATH level writes the SAME BITS (checksum printed):
//   MATH      0 plain copy of the 16-byte chunks | 1 unpack + affine + pack | 2 + SiLU compiled in | 4 + SiLU behind a run-time flag (the product's
//             hot form: 115 full-rate + 32 quarter-rate VALU instructions per two chunks = 972 issue cycles per wave, 16 of them v_cndmask)
//             | 3 = 2 on packed fp32 (v_pk_fma / v_pk_mul / v_pk_add: same bits, half the issue slots for those operations)
//   NT        0 default cache policy | 1 non-temporal loads and stores (the product) | 2 non-temporal stores only
//   PATTERN   0 grid-stride over a frame's chunks, grid (8192, T) (the product) | 1 every workgroup owns ONE contiguous span of its frame
//             | 2 persistent: 8 workgroups per CU walk contiguous spans of the whole tensor
//   INFLIGHT  chunks loaded before the first is used: 1 | 2 (the product) | 4 | 8
// usage: stream_ab [reps] [case ...]      cases: gn128 (5 x 1024^2 x 128, default), gn256 (5 x 512^2 x 256), gn512 (5 x 256^2 x 512), big (25 x 1024^2 x 128)
// Prints one JSON line per (case, variant): microseconds, TB/s over the algorithmic bytes (2 B read + 2 B written per element), checksum.
// What to read off: copy vs MATH 2 at the product's knobs = what the arithmetic costs; the best (NT, PATTERN, INFLIGHT) row = what the pass could
// be rewritten to.  build: tools/ubench/build_ubench.sh (links nothing of the product; measurement aid only)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>
#include "../../comfyui-seedvr2_videoupscaler_amd/csrc/svr_common.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
using namespace svr;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int NT> __device__ __forceinline__ uint4 ld16(const uint4* p) {
    if constexpr (NT == 1) { const u32x4 v = __builtin_nontemporal_load((const u32x4*)p); return make_uint4(v.x, v.y, v.z, v.w); }
    else return *p;
}
template <int NT> __device__ __forceinline__ void st16(uint4* p, const uint4& v) {
    if constexpr (NT != 0) __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, (u32x4*)p);
    else *p = v;
}

// one 16-byte chunk = 8 consecutive channels of one voxel; channel block = chunk index % (C / 8).
// MATH 2: SiLU compiled in;  4: behind a run-time flag, as in the product (a v_cndmask per element on top);  3: the same arithmetic on
// float2 vectors -- v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 issue two lanes' worth per slot, IEEE-identical to the scalar forms.
typedef __attribute__((ext_vector_type(2))) float f32x2_s;
template <int MATH> __device__ __forceinline__ uint4 transform(const uint4& v, const float* a_s, const float* b_s, int c0, int apply_silu) {
    if constexpr (MATH == 0) return v;
    float f[8];
    unpack8h_raw(v, f);
    if constexpr (MATH == 3) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const f32x2_s x = {f[e], f[e + 1]}, a = {a_s[c0 + e], a_s[c0 + e + 1]}, b = {b_s[c0 + e], b_s[c0 + e + 1]};
            const f32x2_s u = __builtin_elementwise_fma(x, a, b);
            const f32x2_s t = u * f32x2_s{-1.4426950408889634f, -1.4426950408889634f};
            const f32x2_s d = f32x2_s{fast_exp2(t[0]), fast_exp2(t[1])} + f32x2_s{1.0f, 1.0f};
            const f32x2_s r = u * f32x2_s{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
            f[e] = r[0]; f[e + 1] = r[1];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float u = f[e] * a_s[c0 + e] + b_s[c0 + e];
            f[e] = MATH == 2 ? silu(u) : MATH == 4 ? (apply_silu ? silu(u) : u) : u;
        }
    }
    return pack8(f);
}

template <int MATH, int NT, int PATTERN, int INFLIGHT>
__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, const float* __restrict__ scale,
                                                     const float* __restrict__ offset, int64_t chunks_per_frame, int cchunks, int frames,
                                                     int apply_silu) {
    __shared__ float a_s[512], b_s[512];
    for (int c = threadIdx.x; c < cchunks * 8; c += 256) { a_s[c] = scale[c]; b_s[c] = offset[c]; }
    __syncthreads();
    int64_t first, last, step;          // this thread's chunks: first, first + step, ... < last (indices into the whole tensor)
    if constexpr (PATTERN == 0) {       // grid (gx, T): grid-stride inside frame blockIdx.y
        const int64_t base = (int64_t)blockIdx.y * chunks_per_frame;
        first = base + (int64_t)blockIdx.x * 256 + threadIdx.x; last = base + chunks_per_frame; step = (int64_t)gridDim.x * 256;
    } else if constexpr (PATTERN == 1) {   // grid (gx, T): workgroup blockIdx.x owns one contiguous span of frame blockIdx.y
        const int64_t base = (int64_t)blockIdx.y * chunks_per_frame;
        const int64_t span = (chunks_per_frame + gridDim.x - 1) / gridDim.x;
        first = base + (int64_t)blockIdx.x * span + threadIdx.x;
        last = base + min((int64_t)(blockIdx.x + 1) * span, chunks_per_frame); step = 256;
    } else {                            // grid (gx): persistent workgroups, contiguous spans of the whole tensor
        const int64_t total = chunks_per_frame * frames;
        const int64_t span = ((total + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
        first = (int64_t)blockIdx.x * span + threadIdx.x; last = min((int64_t)(blockIdx.x + 1) * span, total); step = 256;
    }
    auto run = [&](auto fixedc) {
        constexpr bool FIXED = decltype(fixedc)::value;       // every chunk of this thread starts at the same channel: scale / offset in registers
        float sa[8], sb[8];
        if constexpr (FIXED) {
            const int c0 = (int)(first % cchunks) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) { sa[e] = a_s[c0 + e]; sb[e] = b_s[c0 + e]; }
        }
        auto tf = [&](const uint4& v, int64_t idx) {
            if constexpr (FIXED) return transform<MATH>(v, sa, sb, 0, apply_silu);
            else return transform<MATH>(v, a_s, b_s, (int)(idx % cchunks) * 8, apply_silu);
        };
        int64_t i = first;
        for (; i + (INFLIGHT - 1) * step < last; i += INFLIGHT * step) {
            uint4 v[INFLIGHT];
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) v[k] = ld16<NT>(x + i + k * step);
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) st16<NT>(y + i + k * step, tf(v[k], i + k * step));
        }
        for (; i < last; i += step) st16<NT>(y + i, tf(ld16<NT>(x + i), i));
    };
    if (MATH != 0 && (step % cchunks) == 0) run(std::true_type{}); else run(std::false_type{});
}

__global__ void fill_h16(_Float16* p, int64_t n, uint32_t seed, float amp) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (_Float16)(((float)(h & 0xffff) / 32768.0f - 1.0f) * amp * 0.015625f);
    }
}
__global__ void checksum(const uint16_t* p, int64_t n, unsigned long long* out) {
    unsigned long long s = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        s += (unsigned long long)p[i] * (unsigned long long)((i % 1000003) + 1);
    atomicAdd(out, s);
}

struct Variant { int math, nt, pattern, inflight; void (*fn)(const uint4*, uint4*, const float*, const float*, int64_t, int, int, int); };
#define V(M, N, P, I) {M, N, P, I, stream_kernel<M, N, P, I>}
static const Variant VARIANTS[] = {
    // the product's knobs: copy, affine, + SiLU compiled in, + SiLU behind the run-time flag (the product), + SiLU on packed fp32
    V(0, 1, 0, 2), V(1, 1, 0, 2), V(2, 1, 0, 2), V(4, 1, 0, 2), V(3, 1, 0, 2), V(3, 1, 0, 4), V(3, 1, 1, 4), V(3, 1, 2, 4),
    // cache policy
    V(0, 0, 0, 2), V(0, 2, 0, 2), V(2, 0, 0, 2), V(2, 2, 0, 2),
    // chunks in flight
    V(0, 1, 0, 1), V(0, 1, 0, 4), V(0, 1, 0, 8), V(2, 1, 0, 1), V(2, 1, 0, 4), V(2, 1, 0, 8),
    // contiguous spans per workgroup; persistent workgroups
    V(0, 1, 1, 2), V(0, 1, 1, 4), V(0, 1, 1, 8), V(2, 1, 1, 2), V(2, 1, 1, 4), V(2, 1, 1, 8),
    V(0, 1, 2, 2), V(0, 1, 2, 4), V(0, 1, 2, 8), V(0, 0, 2, 4), V(2, 1, 2, 2), V(2, 1, 2, 4), V(2, 1, 2, 8), V(2, 0, 2, 4),
};
#undef V

struct Case { const char* name; int T, H, W, C; };
static const Case CASES[] = {{"gn128", 5, 1024, 1024, 128}, {"gn256", 5, 512, 512, 256}, {"gn512", 5, 256, 256, 512}, {"big", 25, 1024, 1024, 128}};

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 5;
    std::vector<std::string> names;
    for (int i = 2; i < argc; ++i) names.push_back(argv[i]);
    if (names.empty()) names.push_back("gn128");
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("# %s CUs=%d\n", prop.gcnArchName, prop.multiProcessorCount);
    unsigned long long* d_sum; CK(hipMalloc(&d_sum, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const std::string& nm : names) {
        const Case* c = nullptr;
        for (const Case& k : CASES) if (nm == k.name) c = &k;
        if (!c) { fprintf(stderr, "unknown case %s\n", nm.c_str()); return 2; }
        const int64_t n = (int64_t)c->T * c->H * c->W * c->C, chunks_per_frame = (int64_t)c->H * c->W * (c->C / 8);
        _Float16* x; uint16_t* y; float *scale, *offset;
        CK(hipMalloc(&x, n * 2)); CK(hipMalloc(&y, n * 2)); CK(hipMalloc(&scale, 512 * 4)); CK(hipMalloc(&offset, 512 * 4));
        hipLaunchKernelGGL(fill_h16, dim3(4096), dim3(256), 0, 0, x, n, 1u, 3.0f);
        std::vector<float> hs(512), ho(512);
        for (int i = 0; i < 512; ++i) { hs[i] = 64.0f * (0.6f + 0.001f * i); ho[i] = -0.3f + 0.002f * i; }      // (a_c carries the h16 factor 2^6)
        CK(hipMemcpy(scale, hs.data(), 512 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(offset, ho.data(), 512 * 4, hipMemcpyHostToDevice));
        CK(hipDeviceSynchronize());
        for (const Variant& v : VARIANTS) {
            unsigned gx = (unsigned)((chunks_per_frame + 1023) / 1024);
            if (gx > 8192) gx = 8192;                      // (the product's launch: svr_api.hip svr_groupnorm_apply)
            const dim3 grid = v.pattern == 2 ? dim3(prop.multiProcessorCount * 8) : dim3(gx, c->T);
            auto launch = [&]() { hipLaunchKernelGGL(v.fn, grid, dim3(256), 0, 0, (const uint4*)x, (uint4*)y, scale, offset, chunks_per_frame, c->C / 8, c->T, 1); };
            CK(hipMemset(y, 0, n * 2));
            launch(); launch();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipGetLastError());
            CK(hipMemset(d_sum, 0, 8));
            hipLaunchKernelGGL(checksum, dim3(2048), dim3(256), 0, 0, y, n, d_sum);
            unsigned long long sum = 0;
            CK(hipMemcpy(&sum, d_sum, 8, hipMemcpyDeviceToHost));
            const double us = ms * 1e3 / reps;
            printf("{\"case\": \"%s\", \"math\": %d, \"nt\": %d, \"pattern\": %d, \"inflight\": %d, \"us\": %.1f, \"tb_s\": %.2f, \"checksum\": \"%016llx\"}\n",
                   c->name, v.math, v.nt, v.pattern, v.inflight, us, (double)n * 4 / (us * 1e-6) / 1e12, sum);
            fflush(stdout);
        }
        CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(scale)); CK(hipFree(offset));
    }
    return 0;
}
