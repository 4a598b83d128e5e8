/*
 * libseedvr2_hip.so -- C ABI of the MI355X (gfx950) SeedVR2 hot path.
 *
 * The reference (numz/ComfyUI-SeedVR2_VideoUpscaler) is pure Python/PyTorch and has no FFI
 * layer; its hot path runs as torch ops below three call sites of the runner
 * (src/core/infer.py:164-184 vae.encode, :249-266 vae.decode, :361-367 dit forward).  Each entry
 * point below replaces the torch op sequence cited next to it.  All pointers are raw DEVICE
 * pointers (tensor.data_ptr()), all tensors are dense row-major, activations bf16 unless noted,
 * `stream` is a hipStream_t.  Functions return 0 on success, non-zero on failure
 * (svr_last_error() gives the message).  The library never allocates caller-visible memory.
 *
 * Layouts: DiT activations  [tokens, channels]; VAE activations NDHWC = [T, H, W, C].
 */
#ifndef SEEDVR2_HIP_H
#define SEEDVR2_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* v8 (round 5): svr_rmsnorm_mod takes SVR_STORE_H16 inputs; svr_gemm_bf16's persistent kernel serves the two h16 forms of the NaDiT's
 * residual stream (bias -> h16; gate * (acc + bias) + h16 residual -> h16); svr_qknorm_rope / svr_groupnorm_apply refuse NULL weights;
 * svr_set_option keys "gemm_asym", "gn_grid_cap", attn_variant 5..10 and gemm_w4r = 2 are gone (measured, deleted); "conv_thinout16" became "conv_thinout4" (1 default: N <= 4 thin-output
 * convs on conv_thinout4_kernel, 0: on the 32-cout kernel).  No signature changed.
 * v9 (round 6): + svr_mfma_calibrate() / svr_mfma_calibrate_workspace_bytes() (measurement aid, not on the data path); a thin-output
 * conv with N <= 4 couts accepts an exact [N, K] weight (the 4-row units take rows >= N from the zero page).  No signature changed. */
#define SVR_ABI_VERSION 9

/* ---- GEMM / implicit-GEMM convolution epilogues ------------------------------------------ */
/* Storage kinds of activation tensors that are NOT MFMA operands (svr_gemm_args.out_f32 / .resid_f32, the x_f32 arguments).
 * SVR_STORE_H16 (ABI v6): an IEEE half holding x * 2^-6 -- 11 significant bits for bf16's bytes, range +-4.2e6, absolute floor
 * 3.8e-6: the residual trunk of the VAE (ResnetBlock3D / attention outputs, attn_video_vae.py:311-362, 615-665), a block's conv1
 * output and (v8) the NaDiT's residual stream (mmsr_block.py:108-126): tensors read only by GroupNorm / RMSNorm and residual adds.
 * Every MFMA operand is bf16. */
#define SVR_STORE_BF16 0
#define SVR_STORE_FP32 1
#define SVR_STORE_H16  2

#define SVR_EPI_BIAS        0   /* C = acc + bias                                              */
#define SVR_EPI_BIAS_SILU   1   /* C = silu(acc + bias)                 embedding.py:56-61     */
#define SVR_EPI_RESID_GATE  2   /* C = resid + gate * (acc + bias)      mmsr_block.py:108-109,
                                   125-126 (AdaSingle "out" + residual); attn_video_vae.py:360 */
#define SVR_EPI_BIAS_GELU   4   /* C = gelu_tanh(acc + bias)            dit_7b/mlp.py:36-41    */
#define SVR_EPI_SWIGLU      3   /* C[:, h] = silu(acc[:, gate h]) * acc[:, in h]; W rows are
                                   interleaved in blocks of 16 (gate16 | in16)  mlp.py:60-61   */

typedef struct svr_conv_geom {
    int32_t enabled;            /* 0: plain GEMM, A is [M, K]                                  */
    int32_t T, H, W, Cin;       /* input  [T, H, W, Cin]                                       */
    int32_t To, Ho, Wo;         /* output [To, Ho, Wo, N];  M = To*Ho*Wo                       */
    int32_t kt, kh, kw;         /* kernel taps; W is [Npad, kt*kh*kw*Cin], tap-major, Cin minor */
    int32_t st, sh, sw;         /* strides                                                     */
    int32_t pt, ph, pw;         /* front pads: pt causal head frames, ph/pw zero pad (low side;
                                   reads past H/W on the high side are zero too)               */
    int32_t halo_frames;        /* frames in `halo` (previous temporal slice tail), 0 = none   */
    const void* halo;           /* bf16 [halo_frames, H, W, Cin] or NULL: replicate frame 0
                                   (causal_inflation_lib.py:422-437 extend_head / :260-278)    */
    const void* zeros;          /* >= 16 zero bytes on the device (spatial padding source)     */
} svr_conv_geom;

typedef struct svr_pixel_shuffle {
    int32_t enabled;            /* scatter-store epilogue of Upsample3D.upscale_conv:
                                   "b (x y z c) f h w -> b c (f z) (h x) (w y)"
                                   attn_video_vae.py:137-143, + remove_head :152-153           */
    int32_t F, H, W;            /* input grid (M = F*H*W)                                      */
    int32_t rz;                 /* temporal ratio 1 | 2 (spatial ratio is always 2)            */
    int32_t C;                  /* output channels (N = 4*rz*C)                                */
    int32_t drop_first;         /* 1: drop the duplicated 2nd output frame (first slice only)  */
} svr_pixel_shuffle;

typedef struct svr_phase_scatter {
    int32_t enabled;            /* conv mode, stride 1: this launch computes ONE output phase of a 2x spatially upsampled
                                   grid ("sub-pixel convolution"): output voxel (to, yo, xo) of the conv is stored at
                                   C[to * t_stride][2*yo + py][2*xo + px][n], C dense [*, 2*Ho, 2*Wo, N] (the caller offsets C to
                                   the first frame of the launch).  Used to run Upsample3D's 1x1x1 upscale_conv + pixel
                                   shuffle + 3x3x3 conv (attn_video_vae.py:110-174) as (kt', 2, 2)-tap convs over the
                                   LOW-resolution input with merged weights: four spatial phases, and for a temporal
                                   upsampler two temporal phases (t_stride 2) interleaving their frames.              */
    int32_t py, px;             /* 0 | 1                                                                              */
    int32_t t_stride;           /* 1 | 2: output frames between consecutive conv output frames                        */
    const float* bias_border;   /* fp32 [3][N] or NULL: bias used INSTEAD of `bias` on the voxels whose window loses its
                                   border tap to the zero padding of the upsampled grid -- row yo == (py ? Ho-1 : 0):
                                   [0]; column xo == (px ? Wo-1 : 0): [1]; both: [2]                                  */
    int32_t quad;               /* 1 (ABI v6): ONE launch computes all four spatial phases -- phase p = py * 2 + px takes its
                                   fragment-ordered weights, bias and border bias from the arrays below and its spatial pads
                                   (1 - py, 1 - px); py / px / bias_border above, conv.ph / conv.pw, W_frag and bias of the
                                   launch are ignored.  The phase is the FASTEST index of the tile order, so the four
                                   workgroups that stage the same low-resolution halo run next to each other on one XCD
                                   (three of the four stagings come from L2 instead of HBM).  Served by the sub-pixel conv
                                   kernel only (svr_gemm_kernel_class() == SVR_KERNEL_CONV_SUBPIXEL), refused otherwise.   */
    int32_t reserved_;
    const void* W_frag4[4];
    const float* bias4[4];
    const float* bias_border4[4];
} svr_phase_scatter;

typedef struct svr_gemm_args {
    const void* A;  int64_t lda;        /* bf16 [M, K] (ignored rows/K layout when conv.enabled: A = input tensor) */
    const void* W;                      /* bf16 [Npad, K] row-major (nn.Linear layout), K % 64 == 0,
                                           Npad = N rounded up to the tile width (128)                       */
    void* C;        int64_t ldc;        /* bf16 (or fp32 if out_f32) [M, N(/2 for SWIGLU)]                     */
    int32_t M, N, K;
    const float* bias;                  /* fp32 [N] or NULL                                                    */
    const float* gate;                  /* fp32 [N] or NULL (=1)            SVR_EPI_RESID_GATE                 */
    const void* resid; int64_t ldr;     /* bf16 [M, N] or NULL (may alias C) SVR_EPI_RESID_GATE                */
    int32_t epilogue;
    int32_t out_f32;
    svr_conv_geom conv;
    svr_pixel_shuffle ps;
    /* Optional fused GroupNorm statistics of the STORED output (conv mode, N = channel count): when the
     * launch is served by the LDS-halo conv kernel (svr_gemm_gn_blocks(args) > 0) every workgroup writes
     * the (sum, sum of squares) of its patch per group to gn_partial[frame][block][group] (fp64 pairs,
     * svr_gemm_gn_blocks() blocks per frame); svr_groupnorm_reduce() turns them into `stats`.  Fixed
     * reduction order, like svr_groupnorm_stats.  NULL / 0: off.                                          */
    void* gn_partial;
    int32_t gn_groups;
    /* Optional (conv mode, 3x3 spatial taps, stride 1, Cin % 32 == 0, N % 128 == 0): the same weights in
     * MFMA-fragment order as written by svr_conv_pack_frag().  When set, the LDS-halo conv kernel streams the
     * weights from this copy straight into registers instead of staging W through LDS.  NULL: off.
     * Plain GEMMs (ABI v7): the copy written by svr_gemm_pack_frag().  When set and the problem is served by the persistent
     * GEMM kernel (svr_gemm_kernel_class() == SVR_KERNEL_GEMM_PERSISTENT), that kernel streams the weights from it into
     * registers and only the activations pass through LDS (by LDS-DMA); same results, bit for bit.  Ignored by every other
     * plain-GEMM kernel. */
    const void* W_frag;
    svr_phase_scatter phase;            /* conv mode only; not together with ps / SWIGLU / resid                        */
    int32_t resid_f32;                  /* 1: `resid` is fp32 [M, N] (ldr in elements).  Wide residual trunk (ABI v5): with out_f32
                                           the skip path of ResnetBlock3D (attn_video_vae.py:311-362) / the NaDiT residual stream
                                           (mmsr_block.py:108-126) is carried in fp32 and only MFMA operands are rounded to bf16 */
} svr_gemm_args;

/* W [N, K = kt * 9 * Cin] (conv weight rows, K order (dt, dy, dx, c)) -> out (same byte size, N * K bf16) in the
 * fragment order svr_gemm_args.W_frag expects.  N % 32 == 0, Cin % 32 == 0.  Done once per checkpoint.     */
int svr_conv_pack_frag(const void* W, void* out, int32_t N, int32_t K, int32_t kt, int32_t Cin, void* stream);

/* W [N, K] (plain GEMM weight rows, N % 128 == 0 -- rows past the problem's N zero-filled --, K % 64 == 0) -> out (same byte
 * size) in the fragment order a plain GEMM's svr_gemm_args.W_frag expects: 16-byte unit
 * (((n / 128) * (K / 32) + k / 32) * 8 + (n % 128) / 16) * 64 + lane holds W[(n & ~15) + (lane & 15)][(k & ~31) + (lane >> 4) * 8 .. + 7],
 * the first operand of v_mfma_f32_16x16x32_bf16.  Done once per checkpoint (replaces nothing in the reference: its nn.Linear weights,
 * src/models/dit_3b/nablocks/attention/mmattn.py:173,207-208,269 and mlp.py:60-61, have no kernel-side layout).  ABI v7.          */
int svr_gemm_pack_frag(const void* W, void* out, int32_t N, int32_t K, void* stream);

/* Same for any spatial tap grid (K = kt * kh * kw * Cin, K order (dt, dy, dx, c)): kh = kw = 3 is svr_conv_pack_frag; kh = kw = 2
 * feeds the sub-pixel upsampler conv kernel (stride 1, pads 0 | 1, same-size output, N % 128 == 0, plain bias epilogue,
 * with or without svr_gemm_args.phase).                                                                        */
int svr_conv_pack_frag_taps(const void* W, void* out, int32_t N, int32_t K, int32_t kt, int32_t kh, int32_t kw, int32_t Cin,
                            void* stream);

/* Number of per-frame partial blocks the launch described by `args` will write to args->gn_partial
 * (0: the kernel that serves this problem does not produce fused statistics -- use svr_groupnorm_stats). */
int32_t svr_gemm_gn_blocks(const svr_gemm_args* args);

/* Which kernel svr_gemm_bf16 would launch for `args` (no launch): one of SVR_KERNEL_*, or -1 with svr_last_error() set when
 * svr_gemm_bf16 would refuse the arguments.  ABI v6.  bench.py attributes launch times to kernels with it (the `roofline`
 * object is the SVR_KERNEL_CONV_HALO launches only), tests pin it on the shapes a VAE tile issues.
 * One class depends on the DEVICE: SVR_KERNEL_GEMM_PERSISTENT needs at least 8 compute units (one workgroup per CU in multiples of
 * the 8 XCDs; a smaller partition gets SVR_KERNEL_GEMM), and this function reads the CU count of the calling thread's CURRENT
 * device (256 when there is no GPU), whereas svr_gemm_bf16 launches on the device of its stream.  Callers that attribute by it
 * (bench.py) make the stream's device current first, as torch.cuda.set_device does; every other class is a function of `args` alone. */
#define SVR_KERNEL_NONE            0   /* empty problem: nothing is launched                                           */
#define SVR_KERNEL_GEMM            1   /* gemm_kernel (eight waves, 256x256 / 256x128 tiles)                           */
#define SVR_KERNEL_GEMM_PERSISTENT 2   /* gemm_w4r_kernel / gemm_w4q_kernel (persistent four-wave workgroups, the NaDiT's big GEMMs; with / without W_frag) */
#define SVR_KERNEL_CONV_HALO       3   /* conv_halo2_kernel: stride-1 3x3 spatial taps, LDS halo (the dominant kernel) */
#define SVR_KERNEL_CONV_SUBPIXEL   4   /* conv_sub_kernel: (kt, 2, 2)-tap phases of the sub-pixel upsamplers           */
#define SVR_KERNEL_CONV_THIN_IN    5   /* conv_halo2_kernel<8, thin>: Cin = 4 (encoder.conv_in)                        */
#define SVR_KERNEL_CONV_THIN_OUT   6   /* conv_thinout4_kernel (Cout <= 4) / conv_thinout_kernel (Cout <= 32): conv_out */
#define SVR_KERNEL_CONV_GENERIC    7   /* gemm_kernel in conv mode: strided / 1x1x1 / everything else                  */
int32_t svr_gemm_kernel_class(const svr_gemm_args* args);
const char* svr_gemm_kernel_name(int32_t kernel_class);

/* nn.Linear / F.conv3d replacement (MFMA bf16, fp32 accumulate).
 * Replaces: every nn.Linear in src/models/dit_3b (mmattn.py:173,269; mlp.py:60-61; patch_v1.py:96,113;
 * embedding.py:56-61; nadit.py:211) and InflatedCausalConv3d.forward (causal_inflation_lib.py:213-305). */
int svr_gemm_bf16(const svr_gemm_args* args, void* stream);

/* ---- DiT elementwise / normalisation ------------------------------------------------------- */
/* y = rms_norm(x) [* w] * scale + shift, per row.  normalization.py:88-109 + modulation.py:110.
 * x [rows, dim] of storage kind x_f32 (SVR_STORE_BF16 | _FP32 | _H16: the NaDiT's residual stream is wide), y bf16; w (affine
 * weight) / scale / shift fp32 [dim] or NULL. */
int svr_rmsnorm_mod(const void* x, void* y, int64_t rows, int32_t dim, float eps,
                    const float* w, const float* scale, const float* shift, int32_t x_f32, void* stream);

/* mod[l][i] = emb[i*stride + off(l)] + param[l][i]: builds the fused AdaSingle vectors
 * (scaleA+scaleB, shiftA+shiftB, gateA+gateB).  modulation.py:76,88-113.
 * emb bf16 [dim*6] viewed "(d l g)"; params bf16 [n_vec, dim]; slot[n_vec] = l*3+g; out fp32.    */
int svr_ada_combine(const void* emb, const void* params, const int32_t* slot, float* out,
                    int32_t n_vec, int32_t dim, void* stream);

/* In-place q/k RMSNorm(head_dim, affine) + 3-axis interleaved-pair RoPE on a packed qkv buffer.
 * mmattn.py:207-208 + rope.py:118-126,172-173.  qkv bf16 [rows, 3*heads*128]; pos int16 [rows,3];
 * cs fp32 [n_pos, 2, 63] = cos|sin(pos * freq) table per axis is folded on the host into one
 * table indexed by position; wq,wk fp32 [128]; t_offset is added to pos[:,0] (text length).     */
int svr_qknorm_rope(void* qkv, int64_t rows, int32_t heads, const int16_t* pos, int32_t t_offset,
                    const float* cos_tab, const float* sin_tab, int32_t n_pos, int32_t n_freq,
                    const float* wq, const float* wk, float eps, void* stream);

/* Variable-length window attention, softmax(q k^T / sqrt(d)) v per window.  attention.py:27-64,
 * mmattn.py:245-264 (gather by window, text rows appended to every window, scatter back).
 * qkv bf16 [*, 3*heads*D] (q|k|v); seq_rows int32 [total] = source row of each window position;
 * out_rows int32 [total] = destination row in `out` (bf16 [*, heads*D]); cu int32 [n_seq+1].       */
int svr_attn_varlen(const void* qkv, int64_t ld_qkv, void* out, int64_t ld_out,
                    const int32_t* seq_rows, const int32_t* out_rows, const int32_t* cu,
                    int32_t n_seq, int32_t max_len, int32_t heads, int32_t head_dim, float scale,
                    void* stream);

/* dst[j] = mean_w src[w*n_txt + j] (text outputs coalesced over windows, na.py:396-417).          */
int svr_rows_mean(const void* src, void* dst, int32_t n_groups, int32_t rows_per_group, int32_t dim,
                  void* stream);

/* patchify "(T t)(H h)(W w) c -> T H W (t h w c)" (t,h,w = 1,2,2), zero-padded to kpad columns.
 * patch_v1.py:91.  in bf16 [T,H,W,C] -> out bf16 [T*(H/2)*(W/2), kpad]                             */
int svr_patchify(const void* in, void* out, int32_t T, int32_t H, int32_t W, int32_t C, int32_t kpad,
                 void* stream);

/* un-patchify + one-step Euler endpoint x0 = x_t - pred  (patch_v1.py:119-123, euler.py:60-63,
 * schedules/base.py:108-110 with A=0,B=1).  pred bf16 [T*(H/2)*(W/2), ldp] (first 4*C cols),
 * x_t bf16 [T,H,W,C] -> out bf16 [T,H,W,C]; x_t NULL -> out = pred.                                */
int svr_unpatchify_euler(const void* pred, int64_t ldp, const void* x_t, void* out,
                         int32_t T, int32_t H, int32_t W, int32_t C, void* stream);

/* ---- VAE elementwise ------------------------------------------------------------------------- */
/* Per-frame GroupNorm statistics over NDHWC.  causal_inflation_lib.py:366-408.
 * x bf16 (fp32 if x_f32) [T, HW, C]; stats fp64 [T, groups, 2] (sum, sumsq), fully overwritten.  Reductions run in a
 * fixed order (no atomics): bit-reproducible, independent of temporal slicing.  `workspace` is a
 * caller-provided scratch of svr_groupnorm_workspace_bytes(T, HW, groups) bytes.                    */
int64_t svr_groupnorm_workspace_bytes(int32_t T, int64_t HW, int32_t groups);
int svr_groupnorm_stats(const void* x, double* stats, void* workspace, int32_t T, int64_t HW, int32_t C,
                        int32_t groups, int32_t x_f32, void* stream);
/* stats[t][g] = fixed-order sum over the `nblk` block partials [T][nblk][groups] (fp64 pairs) written by
 * svr_groupnorm_stats' first stage or by a conv launch with gn_partial set.                           */
int svr_groupnorm_reduce(const void* partial, double* stats, int32_t T, int32_t nblk, int32_t groups, void* stream);
/* y = [silu](gamma * (x - mean) * rstd + beta); x bf16 (fp32 if x_f32), y bf16.  attn_video_vae.py:316-323,343-350. */
int svr_groupnorm_apply(const void* x, void* y, const double* stats, const float* gamma, const float* beta,
                        int32_t T, int64_t HW, int32_t C, int32_t groups, float eps, int32_t apply_silu,
                        int32_t x_f32, void* stream);
/* P[r, :] = softmax(scale * S[r, :]): fp32 scores [rows, cols] (ld_s) -> bf16 probabilities (ld_p); cols <= 65536.
 * The softmax of the VAE mid-block attention (diffusers Attention, 1 head x 512; attn_video_vae.py:659-665) when it is
 * run as two MFMA GEMMs around a materialised score matrix.                                            */
int svr_softmax_rows(const float* S, void* P, int64_t rows, int32_t cols, int64_t ld_s, int64_t ld_p, float scale,
                     void* stream);
/* im2col for thin-input causal convs (Cin = 3 / 16): out bf16 [To*Ho*Wo, kpad].                    */
int svr_im2col_causal(const void* in, void* out, const svr_conv_geom* g, int32_t kpad, void* stream);
/* acc[T, y0:y0+h, x0:x0+w, C] += tile * wy[h] * wx[w]  (fp32 accumulator; tiled_encode/decode
 * blending, attn_video_vae.py:1445-1457, 1606-1619) and cnt[y, x] += wy*wx.                       */
int svr_blend_accumulate(const void* tile, float* acc, float* cnt, const float* wy, const float* wx,
                         int32_t T, int32_t h, int32_t w, int32_t C, int32_t H, int32_t W,
                         int32_t y0, int32_t x0, void* stream);
/* out[..., :c_take] = (acc / max(cnt, 1e-6) - shift) * scale -> bf16.  attn_video_vae.py:1463 + infer.py:188. */
int svr_blend_finalize(const float* acc, const float* cnt, void* out, int32_t T, int64_t HW, int32_t C,
                       int32_t c_take, float scale, float shift, void* stream);
/* out[r, :c_out] = (in[r, :c_out] - shift) * scale   (latent (de)scaling + mean slice, infer.py:188, 236). */
int svr_affine_slice(const void* in, void* out, int64_t rows, int32_t c_in, int32_t c_out,
                     float scale, float shift, void* stream);

/* ---- misc ------------------------------------------------------------------------------------ */
/* Tuning / measurement knobs (no effect on results):
 * "conv_impl" 0 auto (LDS-halo kernels for stride-1 3x3 convs; register-streamed weights when W_frag is given)
 * | 1 generic implicit GEMM everywhere | 3 LDS-halo kernel ignoring W_frag (weights through LDS),
 * "conv_rows" patch rows per wave of the register-streamed conv kernel: 8 (default: 256 accumulators, one wave per SIMD) | 4
 * (two workgroups per CU),
 * "conv_band" tile rows per band of the conv kernel's frame-inner tile order (default 1; 0: frame outermost),
 * "conv_sub" 1 (default) (kt, 2, 2)-tap convs with W_frag run on the sub-pixel conv kernel | 0 on the generic kernel,
 * "conv_lds" dynamic LDS bytes to request for the halo kernel (> 80 KiB forces one workgroup per CU),
 * "conv_thinout4" 1 (default) stride-1 3x3 convs with Cout <= 4 (decoder conv_out) on the resident-weight frame-streaming kernel | 0 on
 * the 32-cout thin-output kernel,
 * "gemm_w4" 1 (default) plain GEMMs with N % 256 == 0 and >= 256 tiles on the persistent four-wave kernel (256 accumulators per
 * wave, the vendor library's tile shape and MFMA) | 0 on the eight-wave kernel like everything else,
 * "gemm_w4r" 1 (default) the persistent kernel streams the weights of a launch that carries W_frag from that copy into registers and
 * brings the activations into LDS by LDS-DMA | 0 it stages both operands through LDS whether W_frag is given or not,
 * "gemm_epi" epilogue of the GEMM kernel: 0 auto | 1 stores straight from the accumulators | 2 through LDS wherever possible,
 * "attn_impl" 0 auto (second-generation window kernel for head_dim 128 / windows <= 2048 rows) | 1 first kernel everywhere,
 * "attn_variant" build variant of the second-generation window kernel (0 default = 8 waves x 32 queries; 1 / 3 / 4: 4-wave and
 * s_setprio builds -- see svr_attn_win.hip),
 * "pipe_abl" measurement-only ablations in -DSVR_ABLATIONS builds (non-zero values give garbage). */
int svr_set_option(const char* key, int32_t value);
/* Calibration aid, NOT on the data path (ABI v9; the reference has no counterpart -- its timing report is src/utils/debug.py): launches
 * a bare v_mfma_f32_32x32x16_bf16 loop on pseudo-random operands, four 512-thread workgroups per CU, `iters` x 16 MFMAs per wave, no
 * memory traffic, on `stream`; `*flops` (if not NULL) receives the FLOPs the launch executes.  The caller times it with events on the
 * same stream: FLOPs / time is the matrix-pipe rate this device sustains under its power limit at that moment (bench.py reports it
 * as roofline.power_limited_peak next to the nominal 2.5 PFLOP/s).  `workspace`: svr_mfma_calibrate_workspace_bytes() of device memory. */
int64_t svr_mfma_calibrate_workspace_bytes(void);
int svr_mfma_calibrate(void* workspace, int32_t iters, double* flops, void* stream);
const char* svr_last_error(void);
int svr_abi_version(void);
/* hex SHA-256 of the sources (every .hip and .h file under csrc/ and this header; sorted by name) AND the compile configuration (hipcc flags
 * and -D defines: a measurement build is not the product build) this binary was compiled from, as passed by the
 * build (-DSVR_BUILD_ID=...); "unknown" if the build did not pass one.  The Python loader refuses a library whose id differs
 * from the sources next to it (a stale binary shipped with newer sources). */
const char* svr_build_id(void);
/* prints device name / CU count into buf; returns 0 if a gfx950 device is current. */
int svr_device_info(char* buf, int32_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* SEEDVR2_HIP_H */
