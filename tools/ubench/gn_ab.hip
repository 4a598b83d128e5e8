// Standalone A/B of svr_groupnorm_apply (GroupNorm apply + SiLU over an NDHWC tensor: 6.8 % of a BASELINE config 3 step) through the C ABI.
//   usage: gn_ab [reps] [case ...] [key=value ...]     (key=value: svr_set_option before the runs)     cases (h16 trunk input -> bf16 output, 32 groups, as a VAE tile issues them):
//     gn128  5 x 1024^2 x 128 (default)    gn256  5 x 512^2 x 256    gn512  5 x 256^2 x 512    gn128b  the same tensor as bf16 input
// Each case runs with SiLU and without; prints microseconds, TB/s over the algorithmic bytes (2 B read + 2 B written per element) and a
// 64-bit checksum of the output -- an experiment library (tools/ubench/build_variant.sh -D... -> gn_ab_x) must print the same.
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

__global__ void fill_h16(_Float16* p, int64_t n, uint32_t seed, float amp) {      // h16 = half of x * 2^-6
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (_Float16)(((float)(h & 0xffff) / 32768.0f - 1.0f) * amp * 0.015625f);
    }
}
__global__ void fill_bf16(uint16_t* p, int64_t n, uint32_t seed, float amp) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float v = ((float)(h & 0xffff) / 32768.0f - 1.0f) * amp;
        p[i] = (uint16_t)(__builtin_bit_cast(uint32_t, v) >> 16);
    }
}
__global__ void checksum(const uint16_t* p, int64_t n, unsigned long long* out) {
    unsigned long long s = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        s += (unsigned long long)p[i] * (unsigned long long)((i % 1000003) + 1);
    atomicAdd(out, s);
}

struct Case { const char* name; int T, H, W, C, kind; };
static const Case CASES[] = {{"gn128", 5, 1024, 1024, 128, SVR_STORE_H16}, {"gn256", 5, 512, 512, 256, SVR_STORE_H16},
                             {"gn512", 5, 256, 256, 512, SVR_STORE_H16}, {"gn128b", 5, 1024, 1024, 128, SVR_STORE_BF16}};

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 10;
    std::vector<std::string> names;
    std::string opts;
    for (int i = 2; i < argc; ++i) {
        const std::string arg = argv[i];
        const size_t eq = arg.find('=');
        if (eq == std::string::npos) { names.push_back(arg); continue; }
        if (svr_set_option(arg.substr(0, eq).c_str(), atoi(arg.c_str() + eq + 1)) != 0) { fprintf(stderr, "%s\n", svr_last_error()); return 1; }
        opts += (opts.empty() ? "" : ",") + arg;
    }
    if (names.empty()) names.push_back("gn128");
    char info[256];
    svr_device_info(info, 256);
    printf("# %s | build %s | options %s\n", info, svr_build_id(), opts.empty() ? "-" : opts.c_str());
    unsigned long long* d_sum; CK(hipMalloc(&d_sum, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int groups = 32;
    for (const std::string& nm : names) {
        const Case* c = nullptr;
        for (const Case& k : CASES) if (nm == k.name) c = &k;
        if (!c) { fprintf(stderr, "unknown case %s\n", nm.c_str()); return 2; }
        const int64_t HW = (int64_t)c->H * c->W, n = (int64_t)c->T * HW * c->C;
        void* x; uint16_t* y; double* stats; float *gamma, *beta;
        CK(hipMalloc(&x, n * 2)); CK(hipMalloc(&y, n * 2)); CK(hipMalloc(&stats, (size_t)c->T * groups * 2 * 8));
        CK(hipMalloc(&gamma, c->C * 4)); CK(hipMalloc(&beta, c->C * 4));
        if (c->kind == SVR_STORE_H16) hipLaunchKernelGGL(fill_h16, dim3(4096), dim3(256), 0, 0, (_Float16*)x, n, 1u, 3.0f);
        else hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, (uint16_t*)x, n, 1u, 3.0f);
        // statistics of a tensor with mean 0.1 and variance 2.9 per (frame, group) -- plausible values, all that matters here is that they are fixed
        const double cnt = (double)HW * (c->C / groups);
        std::vector<double> hs((size_t)c->T * groups * 2);
        for (int t = 0; t < c->T; ++t)
            for (int g = 0; g < groups; ++g) {
                const double mean = 0.1 + 0.01 * g - 0.02 * t, var = 2.9 + 0.05 * g;
                hs[((size_t)t * groups + g) * 2] = mean * cnt;
                hs[((size_t)t * groups + g) * 2 + 1] = (var + mean * mean) * cnt;
            }
        CK(hipMemcpy(stats, hs.data(), hs.size() * 8, hipMemcpyHostToDevice));
        std::vector<float> hg(c->C), hb(c->C);
        for (int i = 0; i < c->C; ++i) { hg[i] = 0.8f + 0.003f * i; hb[i] = -0.2f + 0.002f * i; }
        CK(hipMemcpy(gamma, hg.data(), c->C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(beta, hb.data(), c->C * 4, hipMemcpyHostToDevice));
        CK(hipDeviceSynchronize());
        for (int silu = 1; silu >= 0; --silu) {
            auto call = [&]() { return svr_groupnorm_apply(x, y, stats, gamma, beta, c->T, HW, c->C, groups, 1e-6f, silu, c->kind, nullptr); };
            for (int i = 0; i < 2; ++i) if (call() != 0) { fprintf(stderr, "svr_groupnorm_apply: %s\n", svr_last_error()); return 1; }
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) call();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemset(d_sum, 0, 8));
            hipLaunchKernelGGL(checksum, dim3(2048), dim3(256), 0, 0, y, n, d_sum);
            unsigned long long sum = 0;
            CK(hipMemcpy(&sum, d_sum, 8, hipMemcpyDeviceToHost));
            const double us = ms * 1e3 / reps;
            printf("{\"case\": \"%s\", \"silu\": %d, \"us\": %.1f, \"tb_s\": %.2f, \"checksum\": \"%016llx\"}\n", c->name, silu, us,
                   (double)n * 4 / (us * 1e-6) / 1e12, sum);
            fflush(stdout);
        }
        CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(stats)); CK(hipFree(gamma)); CK(hipFree(beta));
    }
    return 0;
}
