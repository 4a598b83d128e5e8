// Common device helpers for the SeedVR2 gfx950 (CDNA4) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svr {

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define SVR_DEVICE __device__ __forceinline__

SVR_DEVICE float bf2f(bf16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }

// fp32 -> bf16, round-to-nearest-even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, two values per
// instruction); the integer sequence it replaces cost ~6 VALU instructions per value in every epilogue
SVR_DEVICE uint32_t pack2bf(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

SVR_DEVICE bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.0f) & 0xffffu); }

// v_exp_f32 (2^x) without the denormal-range fix-up code of exp2f()
SVR_DEVICE float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// x * sigmoid(x) with v_exp_f32 + v_rcp_f32 (1 ulp; an IEEE fp32 division costs ~10 VALU instructions per element and
// turned the bandwidth-bound GroupNorm+SiLU pass ALU-bound)
SVR_DEVICE float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + fast_exp2(-1.4426950408889634f * x)); }
// nn.GELU("tanh"): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))),  tanh(u) = 1 - 2 / (exp(2u) + 1)
SVR_DEVICE float gelu_tanh(float x) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return 0.5f * x * (2.0f - 2.0f * __builtin_amdgcn_rcpf(fast_exp2(2.8853900817779268f * u) + 1.0f));
}

// 8 bf16 (one 16-byte chunk) -> 8 floats
SVR_DEVICE void unpack8(const uint4& v, float* o) {
    o[0] = __builtin_bit_cast(float, v.x << 16); o[1] = __builtin_bit_cast(float, v.x & 0xffff0000u);
    o[2] = __builtin_bit_cast(float, v.y << 16); o[3] = __builtin_bit_cast(float, v.y & 0xffff0000u);
    o[4] = __builtin_bit_cast(float, v.z << 16); o[5] = __builtin_bit_cast(float, v.z & 0xffff0000u);
    o[6] = __builtin_bit_cast(float, v.w << 16); o[7] = __builtin_bit_cast(float, v.w & 0xffff0000u);
}

// ---- "h16": the 2-byte WIDE storage format of the residual trunk (SVR_STORE_H16, include/seedvr2_hip.h) ----------------------
// An IEEE half holding x * 2^-6: 11 significant bits instead of bf16's 8 for the same bytes -- measured on the production-width
// chain it is as good as fp32 storage (tools/error_budget.py: 49.99 vs 50.01 dB) -- and the exponent shift moves half's range to
// +-4.2e6 with an absolute floor of 3.8e-6 (un-normalised VAE activations of real checkpoints can exceed half's 65504).
// Only GroupNorm and residual adds read such a tensor; every MFMA operand stays bf16.
constexpr float H16_SCALE = 0.015625f, H16_INV = 64.0f;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
SVR_DEVICE uint32_t pack2h_raw(float lo, float hi) {            // RNE (v_cvt_pk_f16_f32), values already scaled
    const f16x2_t v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
SVR_DEVICE uint4 pack8h(const float* o) {                       // fp32 values -> 8 h16
    uint4 v;
    v.x = pack2h_raw(o[0] * H16_SCALE, o[1] * H16_SCALE); v.y = pack2h_raw(o[2] * H16_SCALE, o[3] * H16_SCALE);
    v.z = pack2h_raw(o[4] * H16_SCALE, o[5] * H16_SCALE); v.w = pack2h_raw(o[6] * H16_SCALE, o[7] * H16_SCALE);
    return v;
}
// half -> float as a plain fp32 register value.  The empty asm keeps hipcc from folding the conversion into v_fma_mix_f32 / packed
// mixed-precision forms of the consumer: with those, the fused GroupNorm statistics of conv_halo2_kernel<8, thin>'s h16 instance
// came out DIFFERENT FROM RUN TO RUN on MI355X (outputs identical; found by bench.py's determinism guard, isolated with
// tools/debug_thin_stats.py, gpurun r4e -> r4f) -- the conversions are a handful of VALU instructions in HBM-bound epilogues.
SVR_DEVICE float h2f(_Float16 h) {
    float t = (float)h;
    asm volatile("" : "+v"(t));
    return t;
}
SVR_DEVICE void unpack8h_raw(const uint4& v, float* o) {        // 8 h16 -> 8 floats STILL scaled by 2^-6
    const f16x2_t a = __builtin_bit_cast(f16x2_t, v.x), b = __builtin_bit_cast(f16x2_t, v.y),
                  c = __builtin_bit_cast(f16x2_t, v.z), d = __builtin_bit_cast(f16x2_t, v.w);
    o[0] = h2f(a[0]); o[1] = h2f(a[1]); o[2] = h2f(b[0]); o[3] = h2f(b[1]);
    o[4] = h2f(c[0]); o[5] = h2f(c[1]); o[6] = h2f(d[0]); o[7] = h2f(d[1]);
}
SVR_DEVICE void unpack8h(const uint4& v, float* o) {            // 8 h16 -> the fp32 values they stand for
    unpack8h_raw(v, o);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] *= H16_INV;
}

// 8 consecutive activations starting at element index e8 (a multiple of 8) of a tensor stored as KIND: 0 bf16, 1 fp32, 2 h16
// (the SVR_STORE_* codes; bool arguments keep their old meaning: false bf16, true fp32)
template <int KIND> SVR_DEVICE void load8(const void* base, int64_t e8, float* o) {
    if constexpr (KIND == 1) {
        const float4 a = *(const float4*)((const float*)base + e8), b = *(const float4*)((const float*)base + e8 + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    } else if constexpr (KIND == 2) {
        unpack8h(*(const uint4*)((const bf16_t*)base + e8), o);
    } else {
        unpack8(*(const uint4*)((const bf16_t*)base + e8), o);
    }
}

SVR_DEVICE uint4 pack8(const float* o) {
    uint4 v;
    v.x = pack2bf(o[0], o[1]); v.y = pack2bf(o[2], o[3]);
    v.z = pack2bf(o[4], o[5]); v.w = pack2bf(o[6], o[7]);
    return v;
}

// wave64 all-reduce helpers (ds_bpermute based shuffles; fine off the MFMA critical path)
SVR_DEVICE float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
SVR_DEVICE float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// async 16-byte global -> LDS copy (LDS destination = wave-uniform base + lane*16)
SVR_DEVICE void glds16(const void* gptr, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE function attribute: a process that drives several GPUs (one
// HipOps per device, e.g. ComfyUI nodes pinned to different cuda:N) must set it on each of them.  One bit per device and
// launch site (`done` is the site's static mask); hipGetDevice is a thread-local read.
static inline int set_max_dynamic_lds(const void* kern, int bytes, uint64_t& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipErrorInvalidDevice;
    const uint64_t bit = 1ull << (dev & 63);
    if (done & bit) return 0;
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    done |= bit;
    return 0;
}

// Every entry point launches on the caller's stream; the per-device state above (function attributes, CU count) and the launch
// itself belong to THAT stream's device, which need not be the thread's current one (a ComfyUI process with nodes pinned to
// several cuda:N only switches devices when torch itself launches).  Entry points open one of these: a no-op read pair when
// the devices agree, hipSetDevice there and back when they do not.
struct StreamDeviceGuard {
    int prev = -1;
    explicit StreamDeviceGuard(void* stream) {
        int cur = 0, dev = 0;
        if (hipGetDevice(&cur) != hipSuccess) return;
        if (stream == nullptr || hipStreamGetDevice((hipStream_t)stream, &dev) != hipSuccess || dev == cur) return;
        if (hipSetDevice(dev) == hipSuccess) prev = cur;
    }
    ~StreamDeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    StreamDeviceGuard(const StreamDeviceGuard&) = delete;
    StreamDeviceGuard& operator=(const StreamDeviceGuard&) = delete;
};

// Workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0) + s_barrier).  __syncthreads() also carries a memory fence, for
// which hipcc waits vmcnt(0): between the passes of an LDS-staged epilogue that is a wait for the previous pass's global stores
// (and for any residual loads already on their way) that nothing needs -- the stores only have to leave before the kernel ends.
SVR_DEVICE void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// compute units of the current device (persistent kernels launch one workgroup per CU); cached per device
static inline int device_cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int& c = cached[dev & 63];
    if (c == 0) {
        int n = 0;
        c = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return c;
}

}  // namespace svr
