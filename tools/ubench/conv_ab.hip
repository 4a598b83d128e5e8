// Standalone A/B of svr_gemm_bf16 in conv mode through the C ABI (no Python: seconds per GPU call instead of minutes).
//   usage: conv_ab <reps> <case>[,<case>...] ["<key>=<v>[,<key>=<v>...]" ...]
// Every option set on the command line ("" = the library's defaults) is applied with svr_set_option, then each case runs two warm-up
// calls + `reps` timed calls (HIP events) and prints microseconds, TFLOP/s (algorithmic: 2 Cin Cout taps per output voxel), the kernel
// class the library routes the launch to, and a 64-bit checksum of the output -- option sets that only change scheduling must print the
// same checksum.  Cases (BASELINE config 3 shapes, one 1024-px tile x 5 frames at the layer's resolution):
//   c128   3x3x3 128->128 @1024^2   conv2 form: fragment-ordered weights, h16 residual in, h16 out, fused GroupNorm statistics (LDS-halo kernel)
//   c256   3x3x3 256->256 @512^2    same          c512   3x3x3 512->512 @256^2   same     c128r / c256s   ragged image borders, 1x3x3 taps
//   sc256  1x1x1 256->128 @1024^2   decoder shortcut (generic kernel, h16 out)        sc128  1x1x1 128->256 @512^2   encoder shortcut
//   ds128  3x3x3 128->128 @1024^2 stride (1,2,2) pad (0,1)  encoder downsampler       ds256  3x3x3 256->256 @512^2 stride (2,2,2)
//   sp256  (3,2,2)-tap phases 256->256 @512^2 -> 1024^2: the four spatial phases of the spatial sub-pixel upsampler as ONE quad launch
//   cin    3x3x3 4->128 @1024^2 (encoder.conv_in: the thin-input kernel, h16 out, fused GroupNorm statistics; its run time is its epilogue)
//   sp512  (2,2,2)-tap phases 512->512 @256^2 -> 512^2, output frames interleaved (t_stride 2): one temporal phase of the temporal upsampler
//          (conv_sub_kernel; bf16 out, fused GroupNorm statistics, a halo tensor for the causal head -- as vae.py::_upsample_subpixel)
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
__global__ void fill_h16(_Float16* p, int64_t n, uint32_t seed, float amp) {      // h16 = half of x * 2^-6
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (_Float16)(((float)(h & 0xffff) / 32768.0f - 1.0f) * amp * 0.015625f);
    }
}
__global__ void fill_f32(float* p, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v * (float)((i % 7) - 3);
}
__global__ void checksum(const uint16_t* p, int64_t n, unsigned long long* out) {
    unsigned long long s = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        s += (unsigned long long)p[i] * (unsigned long long)((i % 1000003) + 1);
    atomicAdd(out, s);
}

struct Case { const char* name; int T, H, W, Cin, Cout, kt, kh, kw, st, sh, sw, pt, ph, pw; bool conv2; int sub_tstride = 0; bool thin = false; };
static const Case CASES[] = {
    {"c128", 5, 1024, 1024, 128, 128, 3, 3, 3, 1, 1, 1, 2, 1, 1, true},
    {"c256", 5, 512, 512, 256, 256, 3, 3, 3, 1, 1, 1, 2, 1, 1, true},
    {"c512", 5, 256, 256, 512, 512, 3, 3, 3, 1, 1, 1, 2, 1, 1, true},
    {"c128r", 3, 1000, 1000, 128, 128, 3, 3, 3, 1, 1, 1, 2, 1, 1, true},       // ragged patches on both image borders
    {"c256s", 4, 200, 328, 256, 128, 1, 3, 3, 1, 1, 1, 0, 1, 1, true},          // 1x3x3 taps, Cin != Cout, ragged rows
    {"sc256", 5, 1024, 1024, 256, 128, 1, 1, 1, 1, 1, 1, 0, 0, 0, false},
    {"sc128", 5, 512, 512, 128, 256, 1, 1, 1, 1, 1, 1, 0, 0, 0, false},
    {"ds128", 5, 1024, 1024, 128, 128, 3, 3, 3, 1, 2, 2, 2, 0, 0, false},
    {"ds256", 5, 512, 512, 256, 256, 3, 3, 3, 2, 2, 2, 2, 0, 0, false},
    {"sp256", 5, 512, 512, 256, 256, 3, 2, 2, 1, 1, 1, 2, 1, 1, false, 1},
    {"sp512", 5, 256, 256, 512, 512, 2, 2, 2, 1, 1, 1, 1, 1, 1, false, 2},
    {"cin", 5, 1024, 1024, 4, 128, 3, 3, 3, 1, 1, 1, 2, 1, 1, false, 0, true},      // encoder.conv_in: RGB padded to 4 channels, K padded to 128
    {"cout", 9, 1024, 1024, 128, 3, 3, 3, 3, 1, 1, 1, 2, 1, 1, false},              // decoder.conv_out: 128 -> 3 at full resolution (conv_thinout4_kernel), bf16 out
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
    if (argc < 3) { fprintf(stderr, "usage: conv_ab <reps> <case>[,<case>...] [\"key=v,...\" ...]\n"); return 2; }
    const int reps = atoi(argv[1]);
    std::vector<std::string> names;
    { std::string s = argv[2]; size_t p = 0; while (p <= s.size()) { size_t e = s.find(',', p); if (e == std::string::npos) e = s.size(); names.push_back(s.substr(p, e - p)); p = e + 1; } }
    std::vector<std::string> sets;
    for (int i = 3; i < argc; ++i) sets.push_back(argv[i]);
    if (sets.empty()) sets.push_back("");
    if (apply_options(sets[0])) return 1;            // (buffers that depend on the route -- the fused-statistics partials -- are sized under the FIRST set:
                                                     //  option sets that change the route of a case belong in separate invocations)
    char info[256];
    svr_device_info(info, 256);
    printf("# %s | build %s\n", info, svr_build_id());
    void* zeros; CK(hipMalloc(&zeros, 256)); CK(hipMemset(zeros, 0, 256));
    unsigned long long* d_sum; CK(hipMalloc(&d_sum, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const std::string& nm : names) {
        const Case* c = nullptr;
        for (const Case& k : CASES) if (nm == k.name) c = &k;
        if (!c) { fprintf(stderr, "unknown case %s\n", nm.c_str()); return 2; }
        const int To = (c->T + c->pt - c->kt) / c->st + 1;
        const int Ho = c->sh == 1 ? c->H : c->H / 2, Wo = c->sw == 1 ? c->W : c->W / 2;
        const int K = c->thin ? 128 : c->kt * c->kh * c->kw * c->Cin, N = c->Cout, Npad = (N + 127) / 128 * 128;
        const bool sub = c->sub_tstride > 0;             // quad phase launch: dense [To * t_stride, 2 H, 2 W, N] output
        const int64_t n_in = (int64_t)c->T * c->H * c->W * c->Cin, n_w = (int64_t)Npad * K;
        const int64_t n_out = sub ? (int64_t)To * c->sub_tstride * 4 * Ho * Wo * N : (int64_t)To * Ho * Wo * N;
        uint16_t *x, *w, *wf = nullptr, *out, *halo = nullptr;
        float* bias_border = nullptr;
        _Float16* resid = nullptr;
        float* bias;
        void* partial = nullptr;
        int64_t partial_bytes = 0;
        CK(hipMalloc(&x, n_in * 2)); CK(hipMalloc(&w, n_w * 2)); CK(hipMalloc(&out, n_out * 2)); CK(hipMalloc(&bias, Npad * 4));
        hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, x, n_in, 1u, 1.5f);
        hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, 0, w, n_w, 2u, 0.03f);
        hipLaunchKernelGGL(fill_f32, dim3(4), dim3(256), 0, 0, bias, (int64_t)Npad, 0.01f);
        svr_gemm_args a;
        memset(&a, 0, sizeof(a));
        a.A = x; a.W = w; a.C = out; a.ldc = N; a.M = To * Ho * Wo; a.N = N; a.K = K; a.bias = bias;
        a.epilogue = SVR_EPI_BIAS; a.out_f32 = N < 32 ? SVR_STORE_BF16 : SVR_STORE_H16;
        a.conv.enabled = 1;
        a.conv.T = c->T; a.conv.H = c->H; a.conv.W = c->W; a.conv.Cin = c->Cin; a.conv.To = To; a.conv.Ho = Ho; a.conv.Wo = Wo;
        a.conv.kt = c->kt; a.conv.kh = c->kh; a.conv.kw = c->kw; a.conv.st = c->st; a.conv.sh = c->sh; a.conv.sw = c->sw;
        a.conv.pt = c->pt; a.conv.ph = c->ph; a.conv.pw = c->pw; a.conv.zeros = zeros;
        if (c->conv2) {
            CK(hipMalloc(&wf, n_w * 2)); CK(hipMalloc(&resid, n_out * 2));
            if (svr_conv_pack_frag(w, wf, Npad, K, c->kt, c->Cin, nullptr) != 0) { fprintf(stderr, "%s\n", svr_last_error()); return 1; }
            hipLaunchKernelGGL(fill_h16, dim3(4096), dim3(256), 0, 0, resid, n_out, 3u, 2.0f);
            a.W_frag = wf; a.epilogue = SVR_EPI_RESID_GATE; a.resid = resid; a.ldr = N; a.resid_f32 = SVR_STORE_H16; a.gn_groups = 32;
            const int nblk = svr_gemm_gn_blocks(&a);
            if (nblk > 0) { partial_bytes = (int64_t)To * nblk * 32 * 16; CK(hipMalloc(&partial, partial_bytes)); a.gn_partial = partial; } else a.gn_groups = 0;
        }
        if (c->thin) {
            a.gn_groups = 32;
            const int nblk = svr_gemm_gn_blocks(&a);
            if (nblk > 0) { partial_bytes = (int64_t)To * nblk * 32 * 16; CK(hipMalloc(&partial, partial_bytes)); a.gn_partial = partial; } else a.gn_groups = 0;
        }
        if (sub) {
            // one weight set for the four phases (timing / checksum do not care), a halo tensor holding the causal head frames
            const int64_t n_halo = (int64_t)c->pt * c->H * c->W * c->Cin;
            CK(hipMalloc(&wf, n_w * 2)); CK(hipMalloc(&halo, n_halo * 2)); CK(hipMalloc(&bias_border, 3 * Npad * 4));
            if (svr_conv_pack_frag_taps(w, wf, Npad, K, c->kt, 2, 2, c->Cin, nullptr) != 0) { fprintf(stderr, "%s\n", svr_last_error()); return 1; }
            hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, halo, n_halo, 7u, 1.5f);
            hipLaunchKernelGGL(fill_f32, dim3(4), dim3(256), 0, 0, bias_border, (int64_t)3 * Npad, 0.02f);
            a.out_f32 = SVR_STORE_BF16; a.ldc = 0;
            a.conv.halo = halo; a.conv.halo_frames = c->pt;
            a.phase.enabled = 1; a.phase.t_stride = c->sub_tstride; a.phase.quad = 1;
            for (int p = 0; p < 4; ++p) { a.phase.W_frag4[p] = wf; a.phase.bias4[p] = bias; a.phase.bias_border4[p] = bias_border; }
            a.W_frag = wf; a.phase.bias_border = bias_border;
            a.gn_groups = 32;
            const int nblk = svr_gemm_gn_blocks(&a);
            if (nblk > 0) { partial_bytes = (int64_t)To * c->sub_tstride * nblk * 32 * 16; CK(hipMalloc(&partial, partial_bytes)); CK(hipMemset(partial, 0, partial_bytes)); a.gn_partial = partial; } else a.gn_groups = 0;
        }
        CK(hipDeviceSynchronize());
        const double flops = (sub ? 4.0 : 1.0) * 2.0 * c->Cin * N * c->kt * c->kh * c->kw * (double)To * Ho * Wo;
        for (const std::string& set : sets) {
            if (apply_options(set)) return 1;
            const int cls = svr_gemm_kernel_class(&a);
            for (int i = 0; i < 2; ++i)
                if (svr_gemm_bf16(&a, nullptr) != 0) { fprintf(stderr, "svr_gemm_bf16: %s\n", svr_last_error()); return 1; }
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) svr_gemm_bf16(&a, nullptr);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemset(d_sum, 0, 8));
            hipLaunchKernelGGL(checksum, dim3(2048), dim3(256), 0, 0, out, n_out, d_sum);
            unsigned long long sum = 0;
            CK(hipMemcpy(&sum, d_sum, 8, hipMemcpyDeviceToHost));
            unsigned long long gsum = 0;
            if (partial) {                               // the fused GroupNorm partials (fp64 pairs), as 16-bit words
                CK(hipMemset(d_sum, 0, 8));
                hipLaunchKernelGGL(checksum, dim3(64), dim3(256), 0, 0, (const uint16_t*)partial, partial_bytes / 2, d_sum);
                CK(hipMemcpy(&gsum, d_sum, 8, hipMemcpyDeviceToHost));
            }
            printf("{\"case\": \"%s\", \"options\": \"%s\", \"kernel\": \"%s\", \"us\": %.1f, \"tflops\": %.1f, \"checksum\": \"%016llx\", \"gn_checksum\": \"%016llx\"}\n",
                   c->name, set.c_str(), svr_gemm_kernel_name(cls), ms * 1e3 / reps, flops / (ms * 1e-3 / reps) / 1e12, sum, gsum);
            fflush(stdout);
            // back to the defaults for the next set: every key of this set to 0 would be wrong for keys whose default is 1 -- sets
            // are expected to name the keys they change AND the harness is restarted per experiment; nothing to undo here.
        }
        hipFree(x); hipFree(w); hipFree(out); hipFree(bias);
        if (wf) hipFree(wf);
        if (resid) hipFree(resid);
        if (halo) hipFree(halo);
        if (bias_border) hipFree(bias_border);
        if (partial) hipFree(partial);
    }
    return 0;
}
