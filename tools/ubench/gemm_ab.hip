// Standalone A/B of svr_gemm_bf16 on the four big GEMM forms a NaDiT-3B block issues (dit.py: qkv, attention out, MLP in, MLP out) through
// the C ABI -- no Python: seconds per GPU call instead of the minutes a fresh box spends on `import torch`.
//   usage: gemm_ab <reps> <case>[,<case>...] ["<key>=<v>[,<key>=<v>...]" ...]
// Every option set on the command line ("" = the library's defaults) is applied with svr_set_option, then each case runs two warm-up
// calls + `reps` timed calls (HIP events) and prints microseconds, TFLOP/s (2 M N K), the kernel class the library routes the launch
// to, and a 64-bit checksum of the output of ONE call on freshly filled operands -- builds / option sets that only change scheduling
// must print the same checksum.  Cases at BASELINE config 3's token count (M = 291 658 = 243 windows x 1 200 video rows + 58 text rows):
//   qkv     [M, 2560] x [7680, 2560]^T             no bias, bf16 out                                  mmattn.py:173
//   out     [M, 2560] x [2560, 2560]^T  + bias     hid = hid + gate * (.)  fp32 stream in place       mmattn.py:269, mmsr_block.py:108-109
//   mlpin   [M, 2560] x [13824, 2560]^T            SwiGLU -> [M, 6912] bf16                           mlp.py:60-61
//   mlpout  [M, 6912] x [2560, 6912]^T  + bias     hid = hid + gate * (.)  fp32 stream in place       mlp.py:61, mmsr_block.py:125-126
//   outb / mlpoutb: the same two with a bf16 stream (round 2's storage regime; the epilogue-bytes A/B)
// Every case passes the fragment-ordered weight copy (svr_gemm_pack_frag) as W_frag, like dit.NaDiTEngine (-> gemm_w4r_kernel); the case
// suffix ":q" (e.g. qkv:q) leaves it out (-> gemm_w4q_kernel, both operands through LDS).  GEMM_AB_M=<rows> overrides M.
// build: tools/ubench/build_ubench.sh   (measurement aid, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/seedvr2_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_bf16(uint16_t* p, int64_t n, uint32_t seed, float amp) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        const float v = ((float)(h & 0xffff) / 32768.0f - 1.0f) * amp;
        p[i] = (uint16_t)(__builtin_bit_cast(uint32_t, v) >> 16);
    }
}
__global__ void fill_f32_hash(float* p, int64_t n, uint32_t seed, float amp) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((float)(h & 0xffff) / 32768.0f - 1.0f) * amp;
    }
}
__global__ void checksum(const uint16_t* p, int64_t n, unsigned long long* out) {
    unsigned long long s = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        s += (unsigned long long)p[i] * (unsigned long long)((i % 1000003) + 1);
    atomicAdd(out, s);                                     // integer: order-independent
}

struct Case { const char* name; int N, K, epilogue; bool bias, stream, stream_f32; };
static const Case CASES[] = {
    {"qkv", 7680, 2560, SVR_EPI_BIAS, false, false, false},
    {"out", 2560, 2560, SVR_EPI_RESID_GATE, true, true, true},
    {"mlpin", 13824, 2560, SVR_EPI_SWIGLU, false, false, false},
    {"mlpout", 2560, 6912, SVR_EPI_RESID_GATE, true, true, true},
    {"outb", 2560, 2560, SVR_EPI_RESID_GATE, true, true, false},
    {"mlpoutb", 2560, 6912, SVR_EPI_RESID_GATE, true, true, false},
};

static int apply_options(const std::string& set) {
    size_t pos = 0;
    while (pos < set.size()) {
        size_t end = set.find(',', pos);
        if (end == std::string::npos) end = set.size();
        const std::string item = set.substr(pos, end - pos);
        const size_t eq = item.find('=');
        if (eq == std::string::npos) { fprintf(stderr, "bad option %s\n", item.c_str()); return 1; }
        if (svr_set_option(item.substr(0, eq).c_str(), atoi(item.c_str() + eq + 1)) != 0) { fprintf(stderr, "%s\n", svr_last_error()); return 1; }
        pos = end + 1;
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: gemm_ab <reps> <case>[:q][,<case>...] [\"key=v,...\" ...]   (GEMM_AB_M=<rows> overrides M)\n"); return 2; }
    const int reps = atoi(argv[1]);
    std::vector<std::string> names;
    { std::string s = argv[2]; size_t p = 0; while (p <= s.size()) { size_t e = s.find(',', p); if (e == std::string::npos) e = s.size(); names.push_back(s.substr(p, e - p)); p = e + 1; } }
    std::vector<std::string> sets;
    for (int i = 3; i < argc; ++i) sets.push_back(argv[i]);
    if (sets.empty()) sets.push_back("");
    const int M = getenv("GEMM_AB_M") ? atoi(getenv("GEMM_AB_M")) : 243 * 1200 + 58;
    char info[256];
    svr_device_info(info, 256);
    printf("# %s | build %s | M %d\n", info, svr_build_id(), M);
    unsigned long long* d_sum; CK(hipMalloc(&d_sum, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const std::string& full : names) {
        const size_t colon = full.find(':');
        const std::string nm = full.substr(0, colon);
        const bool no_frag = colon != std::string::npos && full.substr(colon + 1) == "q";
        const Case* c = nullptr;
        for (const Case& k : CASES) if (nm == k.name) c = &k;
        if (!c) { fprintf(stderr, "unknown case %s\n", nm.c_str()); return 2; }
        const int N = c->N, K = c->K, Nout = c->epilogue == SVR_EPI_SWIGLU ? N / 2 : N;
        const int64_t n_a = (int64_t)M * K, n_w = (int64_t)N * K, n_out = (int64_t)M * Nout;
        const int out_bytes = c->stream && c->stream_f32 ? 4 : 2;
        uint16_t *A, *W, *Wf;
        void* out;
        float *bias, *gate;
        CK(hipMalloc(&A, n_a * 2)); CK(hipMalloc(&W, n_w * 2)); CK(hipMalloc(&Wf, n_w * 2)); CK(hipMalloc(&out, n_out * out_bytes));
        CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&gate, N * 4));
        hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, A, n_a, 1u, 1.5f);
        hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, 0, W, n_w, 2u, 0.03f);
        hipLaunchKernelGGL(fill_f32_hash, dim3(8), dim3(256), 0, 0, bias, (int64_t)N, 3u, 0.05f);
        hipLaunchKernelGGL(fill_f32_hash, dim3(8), dim3(256), 0, 0, gate, (int64_t)N, 4u, 0.5f);
        if (svr_gemm_pack_frag(W, Wf, N, K, nullptr) != 0) { fprintf(stderr, "svr_gemm_pack_frag: %s\n", svr_last_error()); return 1; }
        auto fill_stream = [&]() {                       // the residual stream the epilogue updates in place
            if (!c->stream) return;
            if (c->stream_f32) hipLaunchKernelGGL(fill_f32_hash, dim3(4096), dim3(256), 0, 0, (float*)out, n_out, 5u, 2.0f);
            else hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, (uint16_t*)out, n_out, 5u, 2.0f);
        };
        fill_stream();
        svr_gemm_args a;
        memset(&a, 0, sizeof(a));
        a.A = A; a.lda = K; a.W = W; a.C = out; a.ldc = Nout; a.M = M; a.N = N; a.K = K;
        a.bias = c->bias ? bias : nullptr;
        a.epilogue = c->epilogue;
        a.W_frag = no_frag ? nullptr : Wf;
        if (c->stream) {
            a.gate = gate; a.resid = out; a.ldr = Nout;
            a.out_f32 = c->stream_f32 ? SVR_STORE_FP32 : SVR_STORE_BF16;
            a.resid_f32 = c->stream_f32 ? SVR_STORE_FP32 : SVR_STORE_BF16;
        }
        CK(hipDeviceSynchronize());
        const double flops = 2.0 * (double)M * N * K;
        for (const std::string& set : sets) {
            if (apply_options(set)) return 1;
            const int cls = svr_gemm_kernel_class(&a);
            if (cls < 0) { fprintf(stderr, "svr_gemm_kernel_class: %s\n", svr_last_error()); return 1; }
            for (int i = 0; i < 2; ++i)
                if (svr_gemm_bf16(&a, nullptr) != 0) { fprintf(stderr, "svr_gemm_bf16: %s\n", svr_last_error()); return 1; }
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) svr_gemm_bf16(&a, nullptr);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            fill_stream();                                // one call on fresh operands: the checksum does not depend on `reps`
            if (svr_gemm_bf16(&a, nullptr) != 0) { fprintf(stderr, "svr_gemm_bf16: %s\n", svr_last_error()); return 1; }
            CK(hipMemset(d_sum, 0, 8));
            hipLaunchKernelGGL(checksum, dim3(2048), dim3(256), 0, 0, (const uint16_t*)out, n_out * (out_bytes / 2), d_sum);
            unsigned long long sum = 0;
            CK(hipMemcpy(&sum, d_sum, 8, hipMemcpyDeviceToHost));
            printf("{\"case\": \"%s\", \"options\": \"%s\", \"kernel\": \"%s\", \"W_frag\": %s, \"us\": %.1f, \"tflops\": %.1f, \"checksum\": \"%016llx\"}\n",
                   full.c_str(), set.c_str(), svr_gemm_kernel_name(cls), no_frag ? "false" : "true", ms * 1e3 / reps,
                   flops / (ms * 1e-3 / reps) / 1e12, sum);
            fflush(stdout);
            fill_stream();
        }
        hipFree(A); hipFree(W); hipFree(Wf); hipFree(out); hipFree(bias); hipFree(gate);
    }
    return 0;
}
