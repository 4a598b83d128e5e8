// extern "C" entry points of libseedvr2_hip.so (see include/seedvr2_hip.h for the contract).
#include "svr_common.h"
#include "../../include/seedvr2_hip.h"
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cstring>

// single translation unit: the kernel sources are included here so one hipcc call builds the library
#include "svr_gemm.hip"
#include "svr_conv_halo2.hip"
#include "svr_conv_sub.hip"
#include "svr_conv_thinout.hip"
#include "svr_attn_win.hip"
#include "svr_attn.hip"
#include "svr_elementwise.hip"
#include "svr_calibrate.hip"

using namespace svr;

static thread_local char g_err[512] = "";

static int fail(const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return -1;
}
static int check(int hip_status, const char* what) {
    if (hip_status == 0) return 0;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString((hipError_t)hip_status));
    return hip_status;
}
static inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

extern "C" {

const char* svr_last_error(void) { return g_err; }
int svr_abi_version(void) { return SVR_ABI_VERSION; }
#ifndef SVR_BUILD_ID
#define SVR_BUILD_ID "unknown"
#endif
// (stored behind a marker so that the loader can read it from the file without dlopen()ing a possibly stale binary)
static const char g_build_id[] = "SVR_BUILD_ID=" SVR_BUILD_ID;
const char* svr_build_id(void) { return g_build_id + 13; }


int64_t svr_mfma_calibrate_workspace_bytes(void) { return (int64_t)device_cu_count() * 4 * 512 * (int64_t)sizeof(float); }

int svr_mfma_calibrate(void* workspace, int32_t iters, double* flops, void* stream) {
    StreamDeviceGuard on_stream_device(stream);
    if (!workspace || iters <= 0) return fail("svr_mfma_calibrate: workspace of svr_mfma_calibrate_workspace_bytes() and iters > 0");
    const unsigned grid = (unsigned)device_cu_count() * 4;
    hipLaunchKernelGGL(mfma_calibrate_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, (float*)workspace, iters);
    // 8 waves per workgroup, 16 MFMAs of 2 * 32 * 32 * 16 FLOP per wave and iteration
    if (flops) *flops = (double)grid * 8.0 * 16.0 * (2.0 * 32 * 32 * 16) * (double)iters;
    return check(hipGetLastError(), "svr_mfma_calibrate");
}

int svr_set_option(const char* key, int32_t value) {
    if (!key) return fail("svr_set_option: null key");
    if (!strcmp(key, "pipe_abl")) { g_pipe_abl = value; return 0; }
    if (!strcmp(key, "conv_impl")) { g_conv_impl = value; return 0; }
    if (!strcmp(key, "gemm_epi")) { g_gemm_epi = value; return 0; }
    if (!strcmp(key, "gemm_w4")) { g_gemm_w4 = value; return 0; }
    if (!strcmp(key, "gemm_w4r")) { if ((unsigned)value > 1u) return fail("svr_set_option: gemm_w4r is 0 or 1"); g_gemm_w4r = value; return 0; }
    if (!strcmp(key, "conv_rows")) { g_conv_rows = value; return 0; }
    if (!strcmp(key, "conv_band")) { g_conv_band = value; return 0; }
    if (!strcmp(key, "conv_sub")) { g_conv_sub = value; return 0; }
    if (!strcmp(key, "conv_lds")) { g_conv_lds_dbg = value; return 0; }
    if (!strcmp(key, "conv_thinout4")) { g_conv_thinout4 = value; return 0; }
    if (!strcmp(key, "attn_impl")) { g_attn_impl = value; return 0; }
    if (!strcmp(key, "attn_variant")) { g_attn_variant = value; return 0; }
    return fail("svr_set_option: unknown key");
}

int svr_device_info(char* buf, int32_t buflen) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail("svr_device_info: no HIP device");
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return fail("svr_device_info: hipGetDeviceProperties failed");
    if (buf && buflen > 0)
        snprintf(buf, buflen, "%s arch=%s CUs=%d LDS/block=%zu mem=%.1fGB", p.name, p.gcnArchName,
                 p.multiProcessorCount, (size_t)p.maxSharedMemoryPerMultiProcessor, p.totalGlobalMem / 1e9);
    return strstr(p.gcnArchName, "gfx950") ? 0 : fail("svr_device_info: current device is not gfx950");
}

int svr_gemm_bf16(const svr_gemm_args* args, void* stream) {
    StreamDeviceGuard on_stream_device(stream);
    if (!args) return fail("svr_gemm_bf16: null args");
    const char* why = nullptr;
    const int rc = gemm_dispatch(*args, (hipStream_t)stream, &why);
    if (why) return fail(why);
    return check(rc, "svr_gemm_bf16");
}

int32_t svr_gemm_kernel_class(const svr_gemm_args* args) {
    if (!args) { fail("svr_gemm_kernel_class: null args"); return -1; }
    const char* why = nullptr;
    const int cls = gemm_route(*args, &why);
    if (why) fail(why);
    return cls;
}

const char* svr_gemm_kernel_name(int32_t cls) {
    switch (cls) {
        case SVR_KERNEL_NONE: return "none";
        case SVR_KERNEL_GEMM: return "svr::gemm_kernel";
        case SVR_KERNEL_GEMM_PERSISTENT: return "svr::gemm_w4r_kernel";      // (gemm_w4q_kernel when the launch carries no W_frag)
        case SVR_KERNEL_CONV_HALO: return "svr::conv_halo2_kernel";
        case SVR_KERNEL_CONV_SUBPIXEL: return "svr::conv_sub_kernel";
        case SVR_KERNEL_CONV_THIN_IN: return "svr::conv_halo2_kernel<8, 2> (thin input)";
        case SVR_KERNEL_CONV_THIN_OUT: return "svr::conv_thinout_kernel";
        case SVR_KERNEL_CONV_GENERIC: return "svr::gemm_kernel<..., CONV>";
        default: return "invalid";
    }
}

int32_t svr_gemm_gn_blocks(const svr_gemm_args* args) {
    if (!args || !args->conv.enabled || args->gn_groups <= 0) return 0;
    return conv_gn_blocks(*args);
}

int svr_groupnorm_reduce(const void* partial, double* stats, int32_t T, int32_t nblk, int32_t groups, void* stream) {
    StreamDeviceGuard on_stream_device(stream);
    if (T <= 0 || nblk <= 0) return 0;
    if (groups <= 0 || groups > 65535) return fail("svr_groupnorm_reduce: 1..65535 groups");
    if (!partial || !stats) return fail("svr_groupnorm_reduce: null pointer");
    hipLaunchKernelGGL(groupnorm_reduce_kernel, dim3(T, groups), dim3(256), 0, (hipStream_t)stream,
                       (const double2*)partial, stats, (int)nblk, groups);
    return check(hipGetLastError(), "svr_groupnorm_reduce");
}

int svr_rmsnorm_mod(const void* x, void* y, int64_t rows, int32_t dim, float eps, const float* w, const float* scale,
                    const float* shift, int32_t x_f32, void* stream) {
    StreamDeviceGuard on_stream_device(stream);
    if (rows <= 0) return 0;
    if (dim <= 0 || dim % 8 || dim > 64 * 8 * 8) return fail("svr_rmsnorm_mod: dim must be a multiple of 8 and <= 4096");
    if (!x || !y) return fail("svr_rmsnorm_mod: null pointer");
    if ((unsigned)x_f32 > (unsigned)SVR_STORE_H16) return fail("svr_rmsnorm_mod: x_f32 must be SVR_STORE_BF16 / _FP32 / _H16");
    const unsigned grid = (unsigned)(rows < 4 * 2048 ? blocks_for(rows, 4) : 2048);      // 8 blocks per CU, rows strided
    const int nc = (dim + 511) / 512;
#define SVR_RMS_LAUNCH(NC) do { if (x_f32 == SVR_STORE_FP32) hipLaunchKernelGGL((rmsnorm_mod_kernel<NC, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, \
                                              x, (bf16_t*)y, rows, dim, eps, w, scale, shift); \
                           else if (x_f32 == SVR_STORE_H16) hipLaunchKernelGGL((rmsnorm_mod_kernel<NC, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, \
                                              x, (bf16_t*)y, rows, dim, eps, w, scale, shift); \
                           else hipLaunchKernelGGL((rmsnorm_mod_kernel<NC, 0>), dim3(grid), dim3(256), 0, (hipStream_t)stream, \
                                              x, (bf16_t*)y, rows, dim, eps, w, scale, shift); } while (0)
    switch (nc) {
        case 1: SVR_RMS_LAUNCH(1); break; case 2: SVR_RMS_LAUNCH(2); break; case 3: SVR_RMS_LAUNCH(3); break;
        case 4: SVR_RMS_LAUNCH(4); break; case 5: SVR_RMS_LAUNCH(5); break; case 6: SVR_RMS_LAUNCH(6); break;
        case 7: SVR_RMS_LAUNCH(7); break; default: SVR_RMS_LAUNCH(8); break;
    }
#undef SVR_RMS_LAUNCH
    return check(hipGetLastError(), "svr_rmsnorm_mod");
}

int svr_ada_combine(const void* emb, const void* params, const int32_t* slot, float* out, int32_t n_vec, int32_t dim,
                    void* stream) {
    StreamDeviceGuard on_stream_device(stream);
    if (n_vec <= 0 || dim <= 0) return 0;
    if (n_vec > 65535) return fail("svr_ada_combine: at most 65535 vectors per call");
    if (!emb || !params || !slot || !out) return fail("svr_ada_combine: null pointer");
    hipLaunchKernelGGL(ada_combine_kernel, dim3(blocks_for(dim, 256), n_vec), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)emb, (const bf16_t*)params, slot, out, n_vec, dim);
    return check(hipGetLastError(), "svr_ada_combine");
}

int svr_qknorm_rope(void* qkv, int64_t rows, int32_t heads, const int16_t* pos, int32_t t_offset, const float* cos_tab,
                    const float* sin_tab, int32_t n_pos, int32_t n_freq, const float* wq, const float* wk, float eps,
                    void* stream) {
    StreamDeviceGuard on_stream_device(stream);
    if (rows <= 0) return 0;
    if (n_freq <= 0 || n_freq * 3 > 64) return fail("svr_qknorm_rope: 1..21 frequencies per axis (head_dim 128)");
    if (heads <= 0 || n_pos <= 0) return fail("svr_qknorm_rope: heads and n_pos must be positive");
    if (!qkv || !pos || !cos_tab || !sin_tab || !wq || !wk) return fail("svr_qknorm_rope: null pointer (wq / wk are required)");
    // 8 rows (q and k of each: 16 groups of 16 lanes) per block and step; 16 blocks per CU, the rest in the grid-stride loop
    const int64_t nblk = (rows + 7) / 8;
    const unsigned grid = (unsigned)std::min<int64_t>(nblk, (int64_t)device_cu_count() * 16);
    hipLaunchKernelGGL(qknorm_rope_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (bf16_t*)qkv, rows, heads, pos, t_offset, cos_tab, sin_tab, n_pos, n_freq, wq, wk, eps);
    return check(hipGetLastError(), "svr_qknorm_rope");
}

int svr_attn_varlen(const void* qkv, int64_t ld_qkv, void* out, int64_t ld_out, const int32_t* seq_rows,
                    const int32_t* out_rows, const int32_t* cu, int32_t n_seq, int32_t max_len, int32_t heads,
                    int32_t head_dim, float scale, void* stream) {
    StreamDeviceGuard on_stream_device(stream);
    const char* why = nullptr;
    const int rc = attn_dispatch(qkv, ld_qkv, out, ld_out, seq_rows, out_rows, cu, n_seq, max_len, heads, head_dim,
                                 scale, (hipStream_t)stream, &why);
    if (why) return fail(why);
    return check(rc, "svr_attn_varlen");
}

#ifdef SVR_ABLATIONS
#include "measure/svr_api_measure_1.inc"
#endif

int svr_conv_pack_frag_taps(const void* W, void* out, int32_t N, int32_t K, int32_t kt, int32_t kh, int32_t kw, int32_t Cin, void* stream) {
    StreamDeviceGuard on_stream_device(stream);
    if (N <= 0 || N % 32 || Cin <= 0 || Cin % 32 || kt < 1 || kt > 3 || kh < 1 || kh > 3 || kw < 1 || kw > 3 || K != kt * kh * kw * Cin)
        return fail("svr_conv_pack_frag_taps: need N % 32 == 0, Cin % 32 == 0, kt, kh, kw in 1..3, K == kt * kh * kw * Cin");
    const int64_t chunks = (int64_t)N * K / 8;
    hipLaunchKernelGGL(conv_pack_frag_taps_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)W, (uint4*)out, N, K, kt, kh, kw, Cin);
    return check(hipGetLastError(), "svr_conv_pack_frag_taps");
}

int svr_gemm_pack_frag(const void* W, void* out, int32_t N, int32_t K, void* stream) {
    StreamDeviceGuard on_stream_device(stream);
    if (N <= 0 || N % 128 || K <= 0 || K % 64) return fail("svr_gemm_pack_frag: need N % 128 == 0, K % 64 == 0");
    const int64_t chunks = (int64_t)N * K / 8;
    hipLaunchKernelGGL(gemm_pack_frag_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)W, (uint4*)out, N, K);
    return check(hipGetLastError(), "svr_gemm_pack_frag");
}

int svr_conv_pack_frag(const void* W, void* out, int32_t N, int32_t K, int32_t kt, int32_t Cin, void* stream) {
    StreamDeviceGuard on_stream_device(stream);
    if (N <= 0 || N % 32 || Cin <= 0 || Cin % 32 || kt < 1 || kt > 3 || K != kt * 9 * Cin)
        return fail("svr_conv_pack_frag: need N % 32 == 0, Cin % 32 == 0, kt in 1..3, K == kt * 9 * Cin");
    const int64_t chunks = (int64_t)N * K / 8;
    hipLaunchKernelGGL(conv_pack_frag_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)W, (uint4*)out, N, K, kt, Cin);
    return check(hipGetLastError(), "svr_conv_pack_frag");
}

int svr_softmax_rows(const float* S, void* P, int64_t rows, int32_t cols, int64_t ld_s, int64_t ld_p, float scale,
                     void* stream) {
    StreamDeviceGuard on_stream_device(stream);
    if (rows <= 0 || cols <= 0) return 0;
    if (!S || !P) return fail("svr_softmax_rows: null pointer");
    if (rows > 0x7fffffff) return fail("svr_softmax_rows: at most 2^31 - 1 rows per call");
    if (cols % 4 || cols > 1024 * SM_MAXV * 4 || ld_s % 4 || ld_p % 4)
        return fail("svr_softmax_rows: cols must be a multiple of 4 and <= 65536, leading dimensions multiples of 4");
    if (cols <= 256 * SM_MAXV * 4)
        hipLaunchKernelGGL(softmax_rows_kernel<256>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, S, (bf16_t*)P, cols,
                           ld_s, ld_p, scale * 1.4426950408889634f);
    else
        hipLaunchKernelGGL(softmax_rows_kernel<1024>, dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream, S, (bf16_t*)P, cols,
                           ld_s, ld_p, scale * 1.4426950408889634f);
    return check(hipGetLastError(), "svr_softmax_rows");
}

int svr_rows_mean(const void* src, void* dst, int32_t n_groups, int32_t rows_per_group, int32_t dim, void* stream) {
    StreamDeviceGuard on_stream_device(stream);
    if (n_groups <= 0 || rows_per_group <= 0 || dim <= 0) return 0;
    if (dim % 8) return fail("svr_rows_mean: dim must be a multiple of 8");
    if (rows_per_group > 65535) return fail("svr_rows_mean: at most 65535 rows per group");
    if (!src || !dst) return fail("svr_rows_mean: null pointer");
    hipLaunchKernelGGL(rows_mean_kernel, dim3(blocks_for(dim / 8, 64), rows_per_group), dim3(64), 0,
                       (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, n_groups, rows_per_group, dim);
    return check(hipGetLastError(), "svr_rows_mean");
}

int svr_patchify(const void* in, void* out, int32_t T, int32_t H, int32_t W, int32_t C, int32_t kpad, void* stream) {
    StreamDeviceGuard on_stream_device(stream);
    if ((H & 1) || (W & 1) || C <= 0 || kpad < 4 * C) return fail("svr_patchify: H, W must be even, C positive and kpad >= 4*C");
    if (!in || !out) return fail("svr_patchify: null pointer");
    const int64_t total = (int64_t)T * (H / 2) * (W / 2) * kpad;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(patchify_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)in, (bf16_t*)out, T, H, W, C, kpad);
    return check(hipGetLastError(), "svr_patchify");
}

int svr_unpatchify_euler(const void* pred, int64_t ldp, const void* x_t, void* out, int32_t T, int32_t H, int32_t W,
                         int32_t C, void* stream) {
    StreamDeviceGuard on_stream_device(stream);
    if ((H & 1) || (W & 1)) return fail("svr_unpatchify_euler: H, W must be even");
    if (!pred || !out) return fail("svr_unpatchify_euler: null pointer");
    if (ldp < 4 * (int64_t)C) return fail("svr_unpatchify_euler: ldp must cover the 4*C prediction columns");
    const int64_t total = (int64_t)T * H * W * C;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(unpatchify_euler_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)pred, ldp, (const bf16_t*)x_t, (bf16_t*)out, T, H, W, C);
    return check(hipGetLastError(), "svr_unpatchify_euler");
}

int64_t svr_groupnorm_workspace_bytes(int32_t T, int64_t HW, int32_t groups) {
    return (int64_t)T * blocks_for(HW, GN_ROWS_PER_BLOCK) * groups * 16;
}

int svr_groupnorm_stats(const void* x, double* stats, void* workspace, int32_t T, int64_t HW, int32_t C, int32_t groups,
                        int32_t x_f32, void* stream) {
    StreamDeviceGuard on_stream_device(stream);
    if (T <= 0 || HW <= 0) return 0;
    if (C <= 0 || groups <= 0 || C % groups || T > 65535) return fail("svr_groupnorm_stats: need C > 0, 1 <= groups dividing C, T <= 65535");
    if (!x || !stats) return fail("svr_groupnorm_stats: null pointer");
    if (C % 8 || C > 512 || groups > 32 || (C / groups) % 4 || (256 % (C / 8)))
        return fail("svr_groupnorm_stats: need C in {128,256,512}-like (C%8==0, C<=512, groups<=32, (C/groups)%4==0)");
    if (!workspace) return fail("svr_groupnorm_stats: workspace missing (svr_groupnorm_workspace_bytes)");
    const unsigned nblk = blocks_for(HW, GN_ROWS_PER_BLOCK);
    if ((unsigned)x_f32 > (unsigned)SVR_STORE_H16) return fail("svr_groupnorm_stats: x_f32 must be SVR_STORE_BF16 / _FP32 / _H16");
    if (x_f32 == SVR_STORE_FP32) hipLaunchKernelGGL(groupnorm_stats_kernel<1>, dim3(nblk, T), dim3(256), 0, (hipStream_t)stream, x, (double2*)workspace, HW, C, groups);
    else if (x_f32 == SVR_STORE_H16) hipLaunchKernelGGL(groupnorm_stats_kernel<2>, dim3(nblk, T), dim3(256), 0, (hipStream_t)stream, x, (double2*)workspace, HW, C, groups);
    else hipLaunchKernelGGL(groupnorm_stats_kernel<0>, dim3(nblk, T), dim3(256), 0, (hipStream_t)stream, x, (double2*)workspace, HW, C, groups);
    hipLaunchKernelGGL(groupnorm_reduce_kernel, dim3(T, groups), dim3(256), 0, (hipStream_t)stream,
                       (const double2*)workspace, stats, (int)nblk, groups);
    return check(hipGetLastError(), "svr_groupnorm_stats");
}

int svr_groupnorm_apply(const void* x, void* y, const double* stats, const float* gamma, const float* beta, int32_t T,
                        int64_t HW, int32_t C, int32_t groups, float eps, int32_t apply_silu, int32_t x_f32, void* stream) {
    StreamDeviceGuard on_stream_device(stream);
    if (T <= 0 || HW <= 0) return 0;
    if (C <= 0 || C % 8 || C > 512) return fail("svr_groupnorm_apply: C must be a multiple of 8 and <= 512");
    if (groups <= 0 || C % groups || T > 65535) return fail("svr_groupnorm_apply: need 1 <= groups dividing C, T <= 65535");
    if (!x || !y || !stats || !gamma || !beta) return fail("svr_groupnorm_apply: null pointer");
    const int64_t nchunks = HW * (C / 8);
    // workgroups per frame, each streaming one contiguous span of >= 4 x 4 KiB (measured: spans of 16 KiB 6.4 TB/s, 32 KiB 6.25,
    // 64 KiB 5.65, 128 KiB 5.2 on 5 x 1024^2 x 128: profiles/r5_gn_apply_ab.txt)
    unsigned gx = blocks_for(nchunks, 256 * 4);
    if (gx > 65535u) gx = 65535u;
    if ((unsigned)x_f32 > (unsigned)SVR_STORE_H16) return fail("svr_groupnorm_apply: x_f32 must be SVR_STORE_BF16 / _FP32 / _H16");
#define SVR_GN(K, S) hipLaunchKernelGGL((groupnorm_apply_kernel<K, S>), dim3(gx, T), dim3(256), 0, (hipStream_t)stream, x, \
                                        (bf16_t*)y, stats, gamma, beta, HW, C, groups, eps)
    if (x_f32 == SVR_STORE_FP32) { if (apply_silu) SVR_GN(1, true); else SVR_GN(1, false); }
    else if (x_f32 == SVR_STORE_H16) { if (apply_silu) SVR_GN(2, true); else SVR_GN(2, false); }
    else { if (apply_silu) SVR_GN(0, true); else SVR_GN(0, false); }
#undef SVR_GN
    return check(hipGetLastError(), "svr_groupnorm_apply");
}

int svr_im2col_causal(const void* in, void* out, const svr_conv_geom* g, int32_t kpad, void* stream) {
    StreamDeviceGuard on_stream_device(stream);
    if (!g || !in || !out) return fail("svr_im2col_causal: null geometry / pointer");
    if (g->Cin <= 0 || g->Cin % 4 || kpad % g->Cin || kpad < g->kt * g->kh * g->kw * g->Cin)
        return fail("svr_im2col_causal: Cin % 4 == 0 and kpad a multiple of Cin covering all taps required");
    const int64_t total = (int64_t)g->To * g->Ho * g->Wo * (kpad / g->Cin);
    if (total <= 0) return 0;
    hipLaunchKernelGGL(im2col_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)in, (bf16_t*)out, *g, kpad);
    return check(hipGetLastError(), "svr_im2col_causal");
}

int svr_blend_accumulate(const void* tile, float* acc, float* cnt, const float* wy, const float* wx, int32_t T,
                         int32_t h, int32_t w, int32_t C, int32_t H, int32_t W, int32_t y0, int32_t x0, void* stream) {
    StreamDeviceGuard on_stream_device(stream);
    const int64_t total = (int64_t)T * h * w * C;
    if (total <= 0) return 0;
    if (y0 < 0 || x0 < 0 || y0 + h > H || x0 + w > W) return fail("svr_blend_accumulate: tile outside the canvas");
    if (!tile || !acc || !cnt || !wy || !wx) return fail("svr_blend_accumulate: null pointer");
    hipLaunchKernelGGL(blend_accumulate_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)tile, acc, cnt, wy, wx, T, h, w, C, H, W, y0, x0);
    return check(hipGetLastError(), "svr_blend_accumulate");
}

int svr_blend_finalize(const float* acc, const float* cnt, void* out, int32_t T, int64_t HW, int32_t C, int32_t c_take,
                       float scale, float shift, void* stream) {
    StreamDeviceGuard on_stream_device(stream);
    const int64_t total = (int64_t)T * HW * c_take;
    if (total <= 0) return 0;
    if (c_take > C) return fail("svr_blend_finalize: c_take > C");
    if (!acc || !cnt || !out) return fail("svr_blend_finalize: null pointer");
    hipLaunchKernelGGL(blend_finalize_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream, acc, cnt,
                       (bf16_t*)out, T, HW, C, c_take, scale, shift);
    return check(hipGetLastError(), "svr_blend_finalize");
}

int svr_affine_slice(const void* in, void* out, int64_t rows, int32_t c_in, int32_t c_out, float scale, float shift,
                     void* stream) {
    StreamDeviceGuard on_stream_device(stream);
    if (rows <= 0) return 0;
    if (c_out <= 0 || c_out > c_in) return fail("svr_affine_slice: need 0 < c_out <= c_in");
    if (!in || !out) return fail("svr_affine_slice: null pointer");
    hipLaunchKernelGGL(affine_slice_kernel, dim3(blocks_for(rows * c_out, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)in, (bf16_t*)out, rows, c_in, c_out, scale, shift);
    return check(hipGetLastError(), "svr_affine_slice");
}

}  // extern "C"
