// HBM-bound side kernels of the SeedVR2 hot path (gfx950): normalisation, modulation, RoPE,
// patch (un)folding, GroupNorm, im2col for thin convs, tile blending.  All loads/stores are
// 16-byte (8 x bf16) per lane where the layout allows; reductions are wave shuffles + LDS.
#include "svr_common.h"
#include "../../include/seedvr2_hip.h"

namespace svr {

// ------------------------------------------------------------------------------------------------
// RMSNorm (+ optional affine weight) + AdaLN "in" modulation.  One wave per row.
// normalization.py:88-109, modulation.py:110.
// ------------------------------------------------------------------------------------------------
constexpr int RMS_MAXC = 8;   // 16-byte chunks per lane -> dim <= 4096

// One wave per row, rows strided over the grid.  Round 5 form: the pass is LATENCY-bound, not bandwidth-bound -- with the per-channel
// affine (w * scale, shift) and two unpacked rows in registers (~200 VGPRs) only two waves per SIMD were resident, each with one row
// in flight, and the 2-byte inputs (half the bytes in flight per wave) ran SLOWER than fp32 (rocprof, round 5: 291 600 x 2560 from
// h16 1.06 ms against 0.86 ms from fp32).  Now the affine lives in LDS (written once per workgroup), the rows that are in flight are
// held PACKED as loaded (16 B per 8 elements of a 2-byte input), and a wave keeps D rows in flight -- 2 for the 2-byte kinds, 1 for
// fp32, i.e. ~10 KiB per wave either way -- at ~110 registers: four waves per SIMD.
template <int NC, int KIND>   // NC: 16-byte (2-byte kinds) / 32-byte (fp32) chunks per lane = ceil(dim / 512); KIND: SVR_STORE_* of x
__global__ __launch_bounds__(256) void rmsnorm_mod_kernel(const void* __restrict__ x, bf16_t* __restrict__ y,
                                                          int64_t rows, int dim, float eps,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift) {
    __shared__ float mul_s[RMS_MAXC * 512], add_s[RMS_MAXC * 512];
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int nchunk = dim >> 3;
    const bool affine = w || scale || shift;
    if (affine) {
        for (int ch = threadIdx.x; ch < dim; ch += 256) {
            float m = 1.f;
            if (w) m *= w[ch];
            if (scale) m *= scale[ch];
            mul_s[ch] = m;
            add_s[ch] = shift ? shift[ch] : 0.f;
        }
        __syncthreads();
    }
    constexpr int D = KIND == 1 ? 1 : 2;                 // rows in flight per wave
    constexpr int RW = KIND == 1 ? 2 : 1;                // 16-byte registers per chunk as loaded
    uint4 raw[D][NC * RW];
    auto fetch = [&](int64_t row, uint4 (&dst)[NC * RW]) {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                if constexpr (KIND == 1) {
                    const uint4* p = (const uint4*)((const float*)x + row * dim + c * 8);
                    dst[2 * i] = p[0]; dst[2 * i + 1] = p[1];
                } else {
                    dst[i] = *(const uint4*)((const bf16_t*)x + row * dim + c * 8);
                }
            }
        }
    };
    auto unpack = [&](const uint4 (&src)[NC * RW], int i, float* f) {
        if constexpr (KIND == 1) {
            const uint4 a = src[2 * i], b = src[2 * i + 1];
            f[0] = __uint_as_float(a.x); f[1] = __uint_as_float(a.y); f[2] = __uint_as_float(a.z); f[3] = __uint_as_float(a.w);
            f[4] = __uint_as_float(b.x); f[5] = __uint_as_float(b.y); f[6] = __uint_as_float(b.z); f[7] = __uint_as_float(b.w);
        } else if constexpr (KIND == 2) {
            unpack8h(src[i], f);
        } else {
            unpack8(src[i], f);
        }
    };
    auto one_row = [&](int64_t row, uint4 (&slot)[NC * RW]) {      // consume the row held in `slot`, then refill the slot D rows ahead
        float v[NC][8];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            if (lane + 64 * i < nchunk) {
                unpack(slot, i, v[i]);
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += v[i][e] * v[i][e];
            }
        }
        if (row + D * nwaves < rows) fetch(row + D * nwaves, slot);
        ss = wave_sum(ss);
        const float inv = rsqrtf(ss / (float)dim + eps);
        uint4* yp = (uint4*)(y + row * dim);
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                float f[8];
                if (affine) {
                    const float4 m0 = *(const float4*)(mul_s + c * 8), m1 = *(const float4*)(mul_s + c * 8 + 4);
                    const float4 a0 = *(const float4*)(add_s + c * 8), a1 = *(const float4*)(add_s + c * 8 + 4);
                    const float m[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
                    const float ad[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = (v[i][e] * inv) * m[e] + ad[e];   // ((x * inv) * w * scale) + shift  (modulation.py:110)
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = v[i][e] * inv;
                }
                yp[c] = pack8(f);
            }
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (wave0 + d * nwaves < rows) fetch(wave0 + d * nwaves, raw[d]);
    for (int64_t row = wave0; row < rows; row += D * nwaves) {
        one_row(row, raw[0]);
        if constexpr (D > 1) {
            if (row + nwaves < rows) one_row(row + nwaves, raw[1]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// AdaSingle vectors: out[v][i] = emb[i*6 + slot[v]] + params[v][i].  modulation.py:76,88-113
// ------------------------------------------------------------------------------------------------
__global__ void ada_combine_kernel(const bf16_t* __restrict__ emb, const bf16_t* __restrict__ params,
                                   const int32_t* __restrict__ slot, float* __restrict__ out, int n_vec, int dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = blockIdx.y;
    if (i >= dim || v >= n_vec) return;
    out[(int64_t)v * dim + i] = bf2f(emb[(int64_t)i * 6 + slot[v]]) + bf2f(params[(int64_t)v * dim + i]);
}

// ------------------------------------------------------------------------------------------------
// q/k RMSNorm(128, affine) + interleaved-pair RoPE on 3 * n_freq pairs of the 64, in place.  mmattn.py:207-208, rope.py:118-126.
// Round 4: one THREAD per 16-byte chunk (8 dims = 4 pairs) of a q or k head vector, 16 lanes per vector, the sum of squares
// reduced with four DPP adds inside the 16-lane row; a block walks the vectors in a grid-stride loop, so a thread keeps its
// chunk of the norm weight and the (axis, frequency) of its four pairs in registers and only the position-dependent cos / sin
// are looked up per row.  (The first version spent one wave on 256 B -- 373 M threads at BASELINE config 3 -- and moved
// 2.8 TB/s; this is a plain streaming pass.)
// ------------------------------------------------------------------------------------------------
SVR_DEVICE float row16_sum(float v) {                   // sum over the 16 lanes of a DPP row, result in every lane
    auto dpp_add = [](float x, auto ctrl) {
        constexpr int CTRL = decltype(ctrl)::value;
        const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true);
        return x + __builtin_bit_cast(float, t);
    };
    v = dpp_add(v, std::integral_constant<int, 0xB1>{});     // quad_perm [1, 0, 3, 2]
    v = dpp_add(v, std::integral_constant<int, 0x4E>{});     // quad_perm [2, 3, 0, 1]
    v = dpp_add(v, std::integral_constant<int, 0x141>{});    // row_half_mirror: the other quad of the half row
    v = dpp_add(v, std::integral_constant<int, 0x140>{});    // row_mirror: the other half row
    return v;
}

__global__ __launch_bounds__(256) void qknorm_rope_kernel(bf16_t* __restrict__ qkv, int64_t rows, int heads,
                                                          const int16_t* __restrict__ pos, int t_offset,
                                                          const float* __restrict__ cos_tab,
                                                          const float* __restrict__ sin_tab, int n_pos, int n_freq,
                                                          const float* __restrict__ wq, const float* __restrict__ wk,
                                                          float eps) {
    // item = (row, q | k): a 16-lane group walks the `heads` head vectors of its item, four at a time (their loads in flight
    // together), so the position-dependent cos / sin of its four pairs are gathered ONCE per row and reused for every head
    // (the first streaming version gathered them per head vector and stayed latency-bound at 2.6 TB/s: gpurun r4h)
    const int sub16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int is_k = grp & 1;
    const int d0 = sub16 * 8;
    const float* wsrc = (is_k ? wk : wq) + d0;
    const float4 w0 = *(const float4*)wsrc, w1 = *(const float4*)(wsrc + 4);
    const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    int axis[4], fi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pl = sub16 * 4 + i;                    // pair index 0..63
        axis[i] = pl / n_freq;
        fi[i] = pl - axis[i] * n_freq;
    }
    const int64_t ld = (int64_t)3 * heads * 128;
    for (int64_t row = (int64_t)blockIdx.x * 8 + (grp >> 1); row < rows; row += (int64_t)gridDim.x * 8) {
        float c[4], sn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            c[i] = 1.f; sn[i] = 0.f;                      // (pairs beyond 3 * n_freq pass through)
            if (axis[i] < 3) {
                int pp = pos[row * 3 + axis[i]] + (axis[i] == 0 ? t_offset : 0);
                pp = min(max(pp, 0), n_pos - 1);
                c[i] = cos_tab[pp * n_freq + fi[i]];
                sn[i] = sin_tab[pp * n_freq + fi[i]];
            }
        }
        bf16_t* base = qkv + row * ld + (int64_t)is_k * heads * 128 + d0;
        for (int h0 = 0; h0 < heads; h0 += 4) {
            uint4 raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) raw[u] = h0 + u < heads ? *(const uint4*)(base + (int64_t)(h0 + u) * 128) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float x[8];
                unpack8(raw[u], x);
                float ss = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += x[e] * x[e];
                ss = row16_sum(ss);
                const float inv = rsqrtf(ss * (1.0f / 128.0f) + eps);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = x[e] * inv * wv[e];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float a0 = x[2 * i], a1 = x[2 * i + 1];
                    x[2 * i] = axis[i] < 3 ? a0 * c[i] - a1 * sn[i] : a0;
                    x[2 * i + 1] = axis[i] < 3 ? a1 * c[i] + a0 * sn[i] : a1;
                }
                if (h0 + u < heads) *(uint4*)(base + (int64_t)(h0 + u) * 128) = pack8(x);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dst[j] = mean_g src[g * rows_per_group + j]    (na.py:411-417 text coalescing)
// ------------------------------------------------------------------------------------------------
__global__ void rows_mean_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int n_groups,
                                 int rows_per_group, int dim) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;      // 16-byte chunk
    const int j = blockIdx.y;
    if (c * 8 >= dim) return;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int gidx = 0; gidx < n_groups; ++gidx) {
        const uint4 v = *(const uint4*)(src + ((int64_t)gidx * rows_per_group + j) * dim + c * 8);
        float f[8];
        unpack8(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += f[e];
    }
    const float inv = 1.0f / (float)n_groups;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= inv;
    *(uint4*)(dst + (int64_t)j * dim + c * 8) = pack8(acc);
}

// ------------------------------------------------------------------------------------------------
// patchify (1,2,2): out[(t, hp, wp)][(dh*2 + dw)*C + c] = in[t][2hp+dh][2wp+dw][c], zero pad to kpad
// ------------------------------------------------------------------------------------------------
__global__ void patchify_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int T, int H, int W, int C,
                                int kpad) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)T * (H / 2) * (W / 2) * kpad;
    if (idx >= total) return;
    const int k = (int)(idx % kpad);
    const int64_t tokn = idx / kpad;
    bf16_t v = 0;
    if (k < 4 * C) {
        const int c = k % C, hw = k / C, dh = hw >> 1, dw = hw & 1;
        const int wp = (int)(tokn % (W / 2));
        const int64_t r = tokn / (W / 2);
        const int hp = (int)(r % (H / 2));
        const int t = (int)(r / (H / 2));
        v = in[(((int64_t)t * H + 2 * hp + dh) * W + 2 * wp + dw) * C + c];
    }
    out[idx] = v;
}

// un-patchify + Euler endpoint: out[t][y][x][c] = x_t[...] - pred[token][(dh*2+dw)*C + c]
__global__ void unpatchify_euler_kernel(const bf16_t* __restrict__ pred, int64_t ldp, const bf16_t* __restrict__ x_t,
                                        bf16_t* __restrict__ out, int T, int H, int W, int C) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)T * H * W * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    int64_t r = idx / C;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int t = (int)(r / H);
    const int64_t tokn = ((int64_t)t * (H / 2) + (y >> 1)) * (W / 2) + (x >> 1);
    const float p = bf2f(pred[tokn * ldp + ((y & 1) * 2 + (x & 1)) * C + c]);
    out[idx] = x_t ? f2bf(bf2f(x_t[idx]) - p) : f2bf(p);
}

// ------------------------------------------------------------------------------------------------
// Per-frame GroupNorm over NDHWC.  Statistics: fp32 per thread over a fixed row set, then fixed-order
// fp64 reductions (inside the block through LDS, across blocks by a second kernel) -- no atomics, so
// the result is bit-reproducible and independent of how the clip is cut into temporal slices.
// ------------------------------------------------------------------------------------------------
constexpr int GN_ROWS_PER_BLOCK = 2048;

// partial[(t * nblk + blockIdx.x) * groups + g] = {sum, sum of squares} of this block's rows
template <int XF32>   // storage kind of x (SVR_STORE_*: 0 bf16, 1 fp32, 2 h16)
__global__ __launch_bounds__(256) void groupnorm_stats_kernel(const void* __restrict__ x, double2* __restrict__ partial,
                                                              int64_t HW, int C, int groups) {
    __shared__ float red[256][4];
    __shared__ double qsum[128][2];                 // per 4-channel quad (C/4 <= 128)
    const int t = blockIdx.y;
    const int cchunks = C >> 3;                     // 16-byte chunks per row
    const int tid = threadIdx.x;
    const int cc = tid % cchunks;
    const int rstep = 256 / cchunks;
    const int64_t r0 = (int64_t)blockIdx.x * GN_ROWS_PER_BLOCK;
    const int64_t r1 = min(r0 + GN_ROWS_PER_BLOCK, HW);
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
    const int64_t base = ((int64_t)t * HW) * C + cc * 8;
    for (int64_t r = r0 + tid / cchunks; r < r1; r += rstep) {
        float f[8];
        load8<XF32>(x, base + r * C, f);
        s0 += f[0] + f[1] + f[2] + f[3];
        q0 += f[0] * f[0] + f[1] * f[1] + f[2] * f[2] + f[3] * f[3];
        s1 += f[4] + f[5] + f[6] + f[7];
        q1 += f[4] * f[4] + f[5] * f[5] + f[6] * f[6] + f[7] * f[7];
    }
    red[tid][0] = s0; red[tid][1] = q0; red[tid][2] = s1; red[tid][3] = q1;
    __syncthreads();
    if (tid < cchunks * 2) {                        // quad = chunk * 2 + half: fixed-order sum over the row lanes
        const int c = tid >> 1, h = tid & 1;
        double s = 0.0, q = 0.0;
        for (int j = 0; j < rstep; ++j) {
            s += (double)red[j * cchunks + c][2 * h];
            q += (double)red[j * cchunks + c][2 * h + 1];
        }
        qsum[tid][0] = s; qsum[tid][1] = q;
    }
    __syncthreads();
    const int quads_per_group = (C / groups) >> 2;  // channels per group / 4 (>= 1)
    if (tid < groups) {
        double s = 0.0, q = 0.0;
        for (int i = 0; i < quads_per_group; ++i) {
            s += qsum[tid * quads_per_group + i][0];
            q += qsum[tid * quads_per_group + i][1];
        }
        partial[((int64_t)t * gridDim.x + blockIdx.x) * groups + tid] = make_double2(s, q);
    }
}

// stats[t][g] = fixed-order sum of the nblk block partials.  One 256-thread block per (frame, group): thread i adds the
// partials i, i + 256, ... in order, then a fixed binary tree over the 256 threads in LDS -- the order depends only on
// nblk, never on timing.  (This kernel sits between every fused conv and its GroupNorm apply with the rest of the GPU idle,
// so it is built for latency: the 8-lanes-per-group version took 50 us at nblk = 4096.)
__global__ __launch_bounds__(256) void groupnorm_reduce_kernel(const double2* __restrict__ partial, double* __restrict__ stats,
                                                               int nblk, int groups) {
    __shared__ double2 red[256];
    const int t = blockIdx.x, g = blockIdx.y, i = threadIdx.x;
    double s = 0.0, q = 0.0;
    for (int b = i; b < nblk; b += 256) {
        const double2 v = partial[((int64_t)t * nblk + b) * groups + g];
        s += v.x; q += v.y;
    }
    red[i] = make_double2(s, q);
    __syncthreads();
#pragma unroll
    for (int o = 128; o > 0; o >>= 1) {
        if (i < o) { red[i].x += red[i + o].x; red[i].y += red[i + o].y; }
        __syncthreads();
    }
    if (i == 0) {
        stats[((int64_t)t * groups + g) * 2] = red[0].x;
        stats[((int64_t)t * groups + g) * 2 + 1] = red[0].y;
    }
}

// GroupNorm apply (+ SiLU): y = act(x * a_c + b_c) with a_c = gamma_c * rstd_g (times 2^6 for an h16 input), b_c = beta_c - mean_g * a_c.
// A streaming pass over a tensor far larger than L2, 2 B (or 4 B) in and 2 B out per element.  Round 5 form, chosen by measurement
// (tools/ubench/stream_ab, profiles/r5_gn_stream_ab.txt: 27 variants of this pass on 5 x 1024^2 x 128):
//   * every workgroup owns ONE CONTIGUOUS SPAN of its frame (256 threads x 16 B = 4 KiB per sweep) instead of a grid-stride walk over
//     the frame: 6.1-6.2 TB/s against 4.9-5.0 (the open DRAM pages of a span are used up by the workgroup that opened them);
//   * four chunks in flight per thread (with the grid-stride walk four were SLOWER than two; with contiguous spans they are faster);
//   * SiLU is a template argument (the run-time flag cost a v_cndmask per element) and the affine + SiLU arithmetic runs on float2
//     vectors (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: the same IEEE operations in the same order, two elements per issue slot);
//   * mean / rstd are derived once per GROUP (one fp64 division chain on `groups` lanes), not once per channel.
// All of it is bit-identical to the round 1-4 kernel (tools/ubench/gn_ab prints the same output checksums).
template <bool SILU> SVR_DEVICE void gn_affine_act8(float* f, const float* sa, const float* sb) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_p;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const f32x2_p x = {f[e], f[e + 1]}, a = {sa[e], sa[e + 1]}, b = {sb[e], sb[e + 1]};
        f32x2_p u = __builtin_elementwise_fma(x, a, b);
        if constexpr (SILU) {
            const f32x2_p t = u * f32x2_p{-1.4426950408889634f, -1.4426950408889634f};
            const f32x2_p d = f32x2_p{1.0f, 1.0f} + f32x2_p{fast_exp2(t[0]), fast_exp2(t[1])};
            u = u * f32x2_p{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        }
        f[e] = u[0]; f[e + 1] = u[1];
    }
}

template <int XF32, bool SILU>    // XF32: storage kind of x (SVR_STORE_*: 0 bf16, 1 fp32, 2 h16)
__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const void* __restrict__ x, bf16_t* __restrict__ y,
                                                              const double* __restrict__ stats,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int64_t HW, int C,
                                                              int groups, float eps) {
    __shared__ float a_s[512], b_s[512];            // per-channel scale / offset (C <= 512); first: per-group mean / rstd
    const int t = blockIdx.y;
    const int cpg = C / groups;
    for (int gidx = threadIdx.x; gidx < groups; gidx += 256) {
        const double n = (double)HW * (double)cpg;
        const double mean = stats[((int64_t)t * groups + gidx) * 2] / n;
        double var = stats[((int64_t)t * groups + gidx) * 2 + 1] / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        a_s[gidx] = (float)(1.0 / sqrt(var + (double)eps));
        b_s[gidx] = (float)mean;
    }
    __syncthreads();
    float ga[2], gb[2];                              // (C <= 512: at most two channels per thread)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = threadIdx.x + j * 256;
        if (c < C) {
            const float g = gamma[c] * a_s[c / cpg];
            ga[j] = XF32 == 2 ? g * H16_INV : g;     // (h16 input: the stored value is x * 2^-6 -- the factor absorbs the 2^6)
            gb[j] = beta[c] - b_s[c / cpg] * g;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = threadIdx.x + j * 256;
        if (c < C) { a_s[c] = ga[j]; b_s[c] = gb[j]; }
    }
    __syncthreads();
    const int cchunks = C >> 3;
    const int64_t nchunks = HW * cchunks;            // 8-channel chunks of this frame
    const int64_t xo = (int64_t)t * HW * C;          // element offset of this frame
    bf16_t* yb = y + xo;
    // this workgroup's contiguous span of the frame's chunks
    const int64_t span = (nchunks + gridDim.x - 1) / gridDim.x;
    const int64_t first = (int64_t)blockIdx.x * span + threadIdx.x;
    const int64_t last = min((int64_t)(blockIdx.x + 1) * span, nchunks);
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    auto load = [&](int64_t i, float* o) {           // chunk i of the frame -> 8 floats (h16: still scaled by 2^-6)
        if constexpr (XF32 == 1) {
            const float* xf = (const float*)x + xo + i * 8;
            const f32x4 v0 = __builtin_nontemporal_load((const f32x4*)xf), v1 = __builtin_nontemporal_load((const f32x4*)(xf + 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = v0[e]; o[4 + e] = v1[e]; }
        } else {
            const u32x4 v = __builtin_nontemporal_load((const u32x4*)((const bf16_t*)x + xo + i * 8));
            if constexpr (XF32 == 2) unpack8h_raw(make_uint4(v.x, v.y, v.z, v.w), o); else unpack8(make_uint4(v.x, v.y, v.z, v.w), o);
        }
    };
    auto store = [&](int64_t i, const float* f) {
        const uint4 o = pack8(f);
        __builtin_nontemporal_store(u32x4{o.x, o.y, o.z, o.w}, (u32x4*)(yb + i * 8));
    };
    if ((256 % cchunks) == 0) {
        // every chunk of this thread starts at the same channel (the sweep of 256 chunks is a multiple of C / 8): its 8 scale /
        // offset pairs live in registers, no per-chunk index arithmetic or LDS reads
        const int c0 = (int)(first % cchunks) * 8;
        float sa[8], sb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sa[e] = a_s[c0 + e]; sb[e] = b_s[c0 + e]; }
        constexpr int INF = XF32 == 1 ? 2 : 4;       // chunks in flight per thread
        int64_t i = first;
        for (; i + (INF - 1) * 256 < last; i += INF * 256) {
            float f[INF][8];
#pragma unroll
            for (int k = 0; k < INF; ++k) load(i + k * 256, f[k]);
#pragma unroll
            for (int k = 0; k < INF; ++k) { gn_affine_act8<SILU>(f[k], sa, sb); store(i + k * 256, f[k]); }
        }
        for (; i < last; i += 256) {
            float f[8];
            load(i, f);
            gn_affine_act8<SILU>(f, sa, sb);
            store(i, f);
        }
    } else {
        for (int64_t i = first; i < last; i += 256) {
            const int c0 = (int)(i % cchunks) * 8;
            float f[8];
            load(i, f);
            gn_affine_act8<SILU>(f, a_s + c0, b_s + c0);
            store(i, f);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Row softmax of fp32 scores -> bf16 probabilities: P[r, :] = softmax(scale * S[r, :]).  One 256-thread block
// per row, the row lives in registers (cols <= 16384), one read and one write of the matrix.  Used by the VAE
// mid-block attention (single 512-wide head over H*W tokens, attn_video_vae.py:659-665), which runs as
// Q K^T (MFMA GEMM, fp32 out) -> this kernel -> P V (MFMA GEMM).
// ------------------------------------------------------------------------------------------------
constexpr int SM_MAXV = 16;   // float4 per thread -> 256 * 16 * 4 = 16384 columns (1024 threads: 65536)

// NT = 256 threads for rows up to 16384 columns (every tiled call); NT = 1024 for the untiled 2K / 4K frames of BASELINE config 2
// (65536 tokens per frame).  The row lives in registers either way.
template <int NT>
__global__ __launch_bounds__(NT) void softmax_rows_kernel(const float* __restrict__ S, bf16_t* __restrict__ P, int cols,
                                                          int64_t ld_s, int64_t ld_p, float scale_log2) {
    constexpr int NW = NT / 64;
    __shared__ float red[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4* sp = (const float4*)(S + (int64_t)blockIdx.x * ld_s);
    const int nv = cols >> 2;
    float4 v[SM_MAXV];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < SM_MAXV; ++i) {
        const int c = tid + NT * i;
        if (c < nv) {
            v[i] = sp[c];
            v[i].x *= scale_log2; v[i].y *= scale_log2; v[i].z *= scale_log2; v[i].w *= scale_log2;
            mx = fmaxf(fmaxf(mx, fmaxf(v[i].x, v[i].y)), fmaxf(v[i].z, v[i].w));
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    if constexpr (NW == 4) {
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    } else {
        mx = red[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w]);
    }
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < SM_MAXV; ++i) {
        const int c = tid + NT * i;
        if (c < nv) {
            v[i].x = fast_exp2(v[i].x - mx); v[i].y = fast_exp2(v[i].y - mx);
            v[i].z = fast_exp2(v[i].z - mx); v[i].w = fast_exp2(v[i].w - mx);
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    float tot;
    if constexpr (NW == 4) {
        tot = (red[0] + red[1]) + (red[2] + red[3]);
    } else {
        tot = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += red[w];       // fixed order
    }
    const float inv = 1.0f / tot;
    uint2* pp = (uint2*)(P + (int64_t)blockIdx.x * ld_p);
#pragma unroll
    for (int i = 0; i < SM_MAXV; ++i) {
        const int c = tid + NT * i;
        if (c < nv) pp[c] = make_uint2(pack2bf(v[i].x * inv, v[i].y * inv), pack2bf(v[i].z * inv, v[i].w * inv));
    }
}

// ------------------------------------------------------------------------------------------------
// im2col for thin causal convs (Cin = 4 (RGB padded) / 16): one thread per (voxel, tap), 8-byte units
// ------------------------------------------------------------------------------------------------
__global__ void im2col_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, svr_conv_geom g, int kpad) {
    const int taps = g.kt * g.kh * g.kw;
    const int slots = kpad / g.Cin;                 // tap slots per output row (>= taps), rest zero
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t M = (int64_t)g.To * g.Ho * g.Wo;
    if (idx >= M * slots) return;
    const int tap = (int)(idx % slots);
    const int64_t m = idx / slots;
    uint2* dst = (uint2*)(out + m * kpad + (int64_t)tap * g.Cin);
    const int units = g.Cin >> 2;
    const uint2* src = nullptr;
    if (tap < taps) {
        const int xo = (int)(m % g.Wo);
        const int64_t r = m / g.Wo;
        const int yo = (int)(r % g.Ho), to = (int)(r / g.Ho);
        const int dx = tap % g.kw, r2 = tap / g.kw, dy = r2 % g.kh, dt = r2 / g.kh;
        int ts = to * g.st - g.pt + dt;
        const int ys = yo * g.sh - g.ph + dy, xs = xo * g.sw - g.pw + dx;
        if ((unsigned)ys < (unsigned)g.H && (unsigned)xs < (unsigned)g.W) {
            const bf16_t* basep = in;
            if (ts < 0) {
                if (g.halo) { basep = (const bf16_t*)g.halo; ts += g.halo_frames; }
                else ts = 0;
            }
            src = (const uint2*)(basep + (((int64_t)ts * g.H + ys) * g.W + xs) * g.Cin);
        }
    }
    for (int u = 0; u < units; ++u) dst[u] = src ? src[u] : make_uint2(0u, 0u);
}

// ------------------------------------------------------------------------------------------------
// Tile blending (tiled_encode / tiled_decode): fp32 accumulate with separable cosine ramps.
// ------------------------------------------------------------------------------------------------
__global__ void blend_accumulate_kernel(const bf16_t* __restrict__ tile, float* __restrict__ acc,
                                        float* __restrict__ cnt, const float* __restrict__ wy,
                                        const float* __restrict__ wx, int T, int h, int w, int C, int H, int W,
                                        int y0, int x0) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // over T*h*w*C
    const int64_t total = (int64_t)T * h * w * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    int64_t r = idx / C;
    const int x = (int)(r % w); r /= w;
    const int y = (int)(r % h);
    const int t = (int)(r / h);
    const float wgt = wy[y] * wx[x];
    acc[(((int64_t)t * H + y0 + y) * W + x0 + x) * C + c] += bf2f(tile[idx]) * wgt;
    if (t == 0 && c == 0) cnt[(int64_t)(y0 + y) * W + x0 + x] += wgt;
}

__global__ void blend_finalize_kernel(const float* __restrict__ acc, const float* __restrict__ cnt,
                                      bf16_t* __restrict__ out, int T, int64_t HW, int C, int c_take, float scale,
                                      float shift) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // over T*HW*c_take
    const int64_t total = (int64_t)T * HW * c_take;
    if (idx >= total) return;
    const int c = (int)(idx % c_take);
    const int64_t vox = idx / c_take;
    const int64_t p = vox % HW;
    const float v = acc[vox * C + c] / fmaxf(cnt[p], 1e-6f);
    out[idx] = f2bf((v - shift) * scale);
}

// out[r][:c_out] = (in[r][:c_out] - shift) * scale      (latent scaling, infer.py:188 / :236)
__global__ void affine_slice_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int64_t rows, int c_in,
                                    int c_out, float scale, float shift) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * c_out) return;
    const int c = (int)(idx % c_out);
    const int64_t r = idx / c_out;
    out[idx] = f2bf((bf2f(in[r * c_in + c]) - shift) * scale);
}

}  // namespace svr
