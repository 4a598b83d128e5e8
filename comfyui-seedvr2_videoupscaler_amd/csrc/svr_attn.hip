// Variable-length (window) attention for gfx950: out = softmax(q k^T * scale) v per sequence.
//
// Replaces pytorch_varlen_attention (src/models/dit_3b/attention.py:27-64: a python loop of SDPA
// calls per window) together with the window gather / text concat / scatter around it
// (mmattn.py:199,245-264): sequences are described by index vectors, so K/V/Q rows are gathered
// straight from the token-ordered qkv GEMM output and results are scattered back in place.
// Also serves the VAE mid-block attention (1 head, D = 512; attn_video_vae.py:659-665).
//
// Formulation: everything is computed transposed so that softmax statistics are lane-local:
//   S^T[key, q] = K Q^T   (mfma 16x16x32: A = K rows from LDS, B = Q rows held in registers)
//   O^T[d,   q] = V^T P^T (A = V^T from a transposed LDS image, B = P^T straight from the S^T
//                          accumulators -- the key order inside each 32-key group is permuted
//                          identically on the P and V^T sides, so no cross-lane traffic)
// Lane (q = lane & 15, g = lane >> 4) owns query q; the 4 lanes sharing a query only exchange
// the running max (2 shuffles per tile) and, once at the end, the row sum.
// Block = 4 waves, each wave 16*QSUB queries; K tile [KT][D] and V^T tile [D][KT] live in LDS
// (K via 16-byte global_load_lds with a source-side XOR swizzle, V via registers + 8-byte
// transposing stores).
#include "svr_common.h"
#include "../../include/seedvr2_hip.h"
#include <cstdlib>

namespace svr {

template <int D, int QSUB, int KT>
__global__ __launch_bounds__(256) void attn_kernel(
    const bf16_t* __restrict__ qkv, int64_t ld_qkv, bf16_t* __restrict__ out, int64_t ld_out,
    const int32_t* __restrict__ seq_rows, const int32_t* __restrict__ out_rows,
    const int32_t* __restrict__ cu, int heads, float scale_log2) {
    constexpr int QB = 64 * QSUB;          // queries per block
    constexpr int DS = D / 32;             // k-steps of the QK^T contraction
    constexpr int DB = D / 16;             // 16-wide output column blocks
    constexpr int KB = KT / 16;            // 16-key blocks per tile
    constexpr int KG = KT / 32;            // 32-key groups per tile (PV contraction steps)
    constexpr int KCH = D / 8;             // 16-byte chunks per K row
    constexpr int VCH = KT / 8;            // 16-byte chunks per V^T row
    constexpr int VSH = (VCH == 8) ? 1 : 2;  // rows of V^T per 256-byte bank row -> swizzle shift
    constexpr int K_BYTES = KT * D * 2;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;
    char* sV = smem + K_BYTES;

    const int seq = blockIdx.z, head = blockIdx.y;
    const int beg = cu[seq];
    const int L = cu[seq + 1] - beg;
    const int q0 = blockIdx.x * QB;
    if (q0 >= L) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 15, g = lane >> 4;

    const bf16_t* qbase = qkv + (int64_t)head * D;
    const bf16_t* kbase = qbase + (int64_t)heads * D;
    const bf16_t* vbase = kbase + (int64_t)heads * D;

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q][32 ds + 8 g .. +8]
    bf16x8 qf[QSUB][DS];
    int qpos[QSUB];
#pragma unroll
    for (int s = 0; s < QSUB; ++s) {
        qpos[s] = q0 + (wave * QSUB + s) * 16 + ql;
        const int row = seq_rows[beg + min(qpos[s], L - 1)];
        const bf16_t* p = qbase + (int64_t)row * ld_qkv + 8 * g;
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) qf[s][ds] = *(const bf16x8*)(p + 32 * ds);
    }

    f32x4 o[QSUB][DB];
    float m_run[QSUB], l_run[QSUB];
#pragma unroll
    for (int s = 0; s < QSUB; ++s) {
        m_run[s] = -INFINITY; l_run[s] = 0.f;
#pragma unroll
        for (int db = 0; db < DB; ++db) o[s][db] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int kt0 = 0; kt0 < L; kt0 += KT) {
        __syncthreads();                                   // previous tile fully consumed
        // ---- stage K: LDS position p (16-byte units) = key * KCH + slot; source chunk = slot ^ (key & 15)
        {
            constexpr int ITERS = KT * KCH / 256;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int p = it * 256 + tid;
                const int key = p / KCH, slot = p % KCH;
                const int chunk = (slot & ~15) | ((slot ^ key) & 15);
                const int row = seq_rows[beg + min(kt0 + key, L - 1)];
                glds16(kbase + (int64_t)row * ld_qkv + chunk * 8, sK + (it * 256 + wave * 64) * 16);
            }
        }
        // ---- stage V^T: micro-tile = 4 keys (one slot quad) x 8 d
        {
            constexpr int MT = (KT / 4) * (D / 8);
#pragma unroll
            for (int it = 0; it < MT / 256; ++it) {
                const int mt = it * 256 + tid;
                const int dc = mt % (D / 8);               // d chunk
                const int kq = mt / (D / 8);               // key quad: keys 4 kq .. 4 kq + 3
                uint4 v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = seq_rows[beg + min(kt0 + 4 * kq + i, L - 1)];
                    v[i] = *(const uint4*)(vbase + (int64_t)row * ld_qkv + dc * 8);
                }
                // key = 32 G + 16 kb + 4 gq + i  ->  slot = 32 G + 8 gq + 4 kb + i
                const int G = kq >> 3, kb = (kq >> 2) & 1, gq = kq & 3;
                const int cc = G * 4 + gq;                 // 16-byte chunk inside the V^T row
                const uint32_t* w0 = (const uint32_t*)&v[0];
                const uint32_t* w1 = (const uint32_t*)&v[1];
                const uint32_t* w2 = (const uint32_t*)&v[2];
                const uint32_t* w3 = (const uint32_t*)&v[3];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int d = dc * 8 + e;
                    const int sh = (e & 1) * 16;
                    uint2 t;
                    t.x = ((w0[e >> 1] >> sh) & 0xffffu) | (((w1[e >> 1] >> sh) & 0xffffu) << 16);
                    t.y = ((w2[e >> 1] >> sh) & 0xffffu) | (((w3[e >> 1] >> sh) & 0xffffu) << 16);
                    const int ccs = cc ^ ((d >> VSH) & (VCH - 1));
                    *(uint2*)(sV + d * (KT * 2) + ccs * 16 + kb * 8) = t;
                }
            }
        }
        __syncthreads();                                   // LDS-DMA drained (vmcnt(0)) + V stores visible

        // ---- S^T = K Q^T
        f32x4 sacc[QSUB][KB];
#pragma unroll
        for (int s = 0; s < QSUB; ++s)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) sacc[s][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int key = kb * 16 + ql;
                const int c = 4 * ds + g;
                const int slot = (c & ~15) | ((c ^ key) & 15);
                const bf16x8 kf = *(const bf16x8*)(sK + key * (D * 2) + slot * 16);
#pragma unroll
                for (int s = 0; s < QSUB; ++s)
                    sacc[s][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s][ds], sacc[s][kb], 0, 0, 0);
            }
        }

        // ---- online softmax (lane-local; lanes g = 0..3 of a query exchange only the max)
        bf16x8 pf[QSUB][KG];
#pragma unroll
        for (int s = 0; s < QSUB; ++s) {
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt0 + kb * 16 + 4 * g + r;
                    float v = sacc[s][kb][r] * scale_log2;
                    v = key < L ? v : -INFINITY;
                    sacc[s][kb][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[s], mx);      // finite: every tile holds >= 1 valid key
            const float alpha = fast_exp2(m_run[s] - m_new);
            m_run[s] = m_new;
            float psum = 0.f;
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) {
                float p[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[r] = fast_exp2(sacc[s][2 * kg][r] - m_new);
                    p[4 + r] = fast_exp2(sacc[s][2 * kg + 1][r] - m_new);
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) psum += p[r];
                const uint4 pk = pack8(p);
                pf[s][kg] = __builtin_bit_cast(bf16x8, pk);
            }
            l_run[s] = l_run[s] * alpha + psum;
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                o[s][db][0] *= alpha; o[s][db][1] *= alpha; o[s][db][2] *= alpha; o[s][db][3] *= alpha;
            }
        }

        // ---- O^T += V^T P^T
#pragma unroll
        for (int db = 0; db < DB; ++db) {
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) {
                const int d = db * 16 + ql;
                const int cc = (kg * 4 + g) ^ ((d >> VSH) & (VCH - 1));
                const bf16x8 vf = *(const bf16x8*)(sV + d * (KT * 2) + cc * 16);
#pragma unroll
                for (int s = 0; s < QSUB; ++s)
                    o[s][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[s][kg], o[s][db], 0, 0, 0);
            }
        }
    }

    // ---- normalise and scatter: lane holds O[q][16 db + 4 g + 0..3]
#pragma unroll
    for (int s = 0; s < QSUB; ++s) {
        float l = l_run[s];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (qpos[s] >= L) continue;
        const float inv = 1.0f / l;
        const int row = out_rows[beg + qpos[s]];
        bf16_t* op = out + (int64_t)row * ld_out + (int64_t)head * D + 4 * g;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            uint2 t;
            t.x = pack2bf(o[s][db][0] * inv, o[s][db][1] * inv);
            t.y = pack2bf(o[s][db][2] * inv, o[s][db][3] * inv);
            *(uint2*)(op + 16 * db) = t;
        }
    }
}

template <int D, int QSUB, int KT>
static int launch_attn(const void* qkv, int64_t ld_qkv, void* out, int64_t ld_out, const int32_t* seq_rows,
                       const int32_t* out_rows, const int32_t* cu, int n_seq, int max_len, int heads,
                       float scale, hipStream_t s) {
    constexpr int QB = 64 * QSUB;
    const size_t lds = (size_t)KT * D * 2 * 2;
    auto kern = attn_kernel<D, QSUB, KT>;
    static uint64_t lds_attr_done = 0;               // per device (svr_common.h)
    {
        const int e = (lds > 48 * 1024) ? set_max_dynamic_lds((const void*)kern, (int)lds, lds_attr_done) : 0;
        if (e != 0) return e;
    }
    dim3 grid((max_len + QB - 1) / QB, heads, n_seq);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const bf16_t*)qkv, ld_qkv, (bf16_t*)out, ld_out,
                       seq_rows, out_rows, cu, heads, scale * 1.4426950408889634f);
    return (int)hipGetLastError();
}

// svr_set_option("attn_impl", v): 0 auto (second-generation window kernel, svr_attn_win.hip, wherever it applies),
// 1 = this file's kernel everywhere.  Used by A/B measurements and the kernel tests.
int g_attn_impl = 0;

int attn_dispatch(const void* qkv, int64_t ld_qkv, void* out, int64_t ld_out, const int32_t* seq_rows,
                  const int32_t* out_rows, const int32_t* cu, int n_seq, int max_len, int heads, int head_dim,
                  float scale, hipStream_t s, const char** why) {
    *why = nullptr;
    if (n_seq <= 0 || max_len <= 0) return 0;
    if (n_seq > 65535 || heads > 65535) { *why = "svr_attn_varlen: grid too large"; return -1; }
    if (head_dim == 128 && max_len <= AW_MAXL && (ld_qkv % 8) == 0 && g_attn_impl != 1)
        return launch_attn_win(qkv, ld_qkv, out, ld_out, seq_rows, out_rows, cu, n_seq, max_len, heads, scale, s);
    if (head_dim == 128)
        return launch_attn<128, 2, 64>(qkv, ld_qkv, out, ld_out, seq_rows, out_rows, cu, n_seq, max_len, heads, scale, s);
    if (head_dim == 512)
        return launch_attn<512, 1, 32>(qkv, ld_qkv, out, ld_out, seq_rows, out_rows, cu, n_seq, max_len, heads, scale, s);
    *why = "svr_attn_varlen: head_dim must be 128 or 512";
    return -1;
}

}  // namespace svr
