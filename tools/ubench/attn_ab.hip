// Standalone A/B of the window-attention builds through the C ABI (no Python: a fresh GPU box spends one to two minutes on its first
// `import torch`; this binary runs in seconds).  BASELINE config 3's regular window family: 243 windows x (1200 video + 58 text) rows,
// 20 heads x 128, qkv [291 658, 7 680] bf16 filled with hashed values.  For every attn_variant on the command line: two warm-up calls,
// `reps` timed calls (HIP events), TFLOP/s, and a 64-bit checksum of the output -- all builds must print the SAME checksum (same MFMAs
// in the same order per query row).   usage: attn_ab [reps] [variant ...]      (default: 10 reps, variants 0 1 3 4)
// build: tools/ubench/build_ubench.sh   (links libseedvr2_hip.so by rpath; measurement aid, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/seedvr2_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill_bf16(uint16_t* p, int64_t n, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        const float v = ((float)(h & 0xffff) / 32768.0f - 1.0f) * 1.5f;           // uniform in [-1.5, 1.5)
        p[i] = (uint16_t)(__builtin_bit_cast(uint32_t, v) >> 16);                  // truncation to bf16 is fine for a filler
    }
}
__global__ void checksum(const uint16_t* p, int64_t n, unsigned long long* out) {
    unsigned long long s = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        s += (unsigned long long)p[i] * (unsigned long long)((i % 1000003) + 1);
    atomicAdd(out, s);                                                             // integer: order-independent
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 10;
    std::vector<int> variants;
    for (int i = 2; i < argc; ++i) variants.push_back(atoi(argv[i]));
    if (variants.empty()) variants = {0, 1, 3, 4};
    const int n_win = 243, per_win = 1200, Lt = 58, heads = 20, D = 128;
    const int64_t N = (int64_t)n_win * per_win, rows_in = N + Lt, rows_out = N + Lt + (int64_t)n_win * Lt;
    const int L = per_win + Lt;
    std::vector<int32_t> seq((size_t)n_win * L), dst((size_t)n_win * L), cu(n_win + 1);
    for (int w = 0; w < n_win; ++w) {
        cu[w] = w * L;
        for (int i = 0; i < per_win; ++i) seq[(size_t)w * L + i] = dst[(size_t)w * L + i] = w * per_win + i;
        for (int i = 0; i < Lt; ++i) { seq[(size_t)w * L + per_win + i] = (int32_t)(N + i); dst[(size_t)w * L + per_win + i] = (int32_t)(N + Lt + (int64_t)w * Lt + i); }
    }
    cu[n_win] = n_win * L;
    uint16_t *qkv, *out;
    int32_t *d_seq, *d_dst, *d_cu;
    unsigned long long* d_sum;
    const int64_t n_qkv = rows_in * 3 * heads * D, n_out = rows_out * heads * D;
    CK(hipMalloc(&qkv, n_qkv * 2)); CK(hipMalloc(&out, n_out * 2));
    CK(hipMalloc(&d_seq, seq.size() * 4)); CK(hipMalloc(&d_dst, dst.size() * 4)); CK(hipMalloc(&d_cu, cu.size() * 4)); CK(hipMalloc(&d_sum, 8));
    CK(hipMemcpy(d_seq, seq.data(), seq.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_dst, dst.data(), dst.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cu, cu.data(), cu.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, qkv, n_qkv, 12345u);
    CK(hipDeviceSynchronize());
    char info[256];
    svr_device_info(info, 256);
    printf("# %s | build %s\n", info, svr_build_id());
    const double flops = 4.0 * heads * D * (double)L * L * n_win;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int v : variants) {
        if (svr_set_option("attn_variant", v) != 0) { fprintf(stderr, "set_option: %s\n", svr_last_error()); return 1; }
        CK(hipMemset(out, 0xff, n_out * 2));
        for (int i = 0; i < 2; ++i)
            if (svr_attn_varlen(qkv, 3 * heads * D, out, heads * D, d_seq, d_dst, d_cu, n_win, L, heads, D, 0.08838834764831845f, nullptr) != 0) {
                fprintf(stderr, "svr_attn_varlen: %s\n", svr_last_error()); return 1;
            }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i)
            svr_attn_varlen(qkv, 3 * heads * D, out, heads * D, d_seq, d_dst, d_cu, n_win, L, heads, D, 0.08838834764831845f, nullptr);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemset(d_sum, 0, 8));
        hipLaunchKernelGGL(checksum, dim3(2048), dim3(256), 0, 0, out, n_out, d_sum);
        unsigned long long sum = 0;
        CK(hipMemcpy(&sum, d_sum, 8, hipMemcpyDeviceToHost));
        printf("{\"kernel\": \"attn window 720p regular (243 x 1258 rows, 20 heads) attn_variant %d\", \"us\": %.1f, \"tflops\": %.1f, \"checksum\": \"%016llx\"}\n",
               v, ms * 1e3 / reps, flops / (ms * 1e-3 / reps) / 1e12, sum);
        fflush(stdout);
    }
    svr_set_option("attn_variant", 0);
    return 0;
}
