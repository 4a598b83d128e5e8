// bf16 MFMA GEMM / implicit-GEMM causal Conv3d for gfx950 (MI355X).
// (A second, deeper-pipelined 256x256x64 kernel was built in round 1, measured equal (-4 .. +5 %, profiles/r1_kbench_ab.txt)
// and removed in round 2; its ablations are what located this kernel's limit: staging, not barriers or fragment reads.)
//
//   C[M, N] = A[M, K] * W[N, K]^T  (+ fused epilogue)
//
// One kernel serves every nn.Linear of the NaDiT (mmattn.py:173,269; mlp.py:60-61; ...) and,
// in "conv" mode, InflatedCausalConv3d (causal_inflation_lib.py:213-305) as an implicit GEMM over
// NDHWC activations: row m is an output voxel, the K axis runs tap-major / channel-minor, and the A
// tile is gathered straight from the input tensor (spatial zero padding reads a zero page, the
// causal temporal head reads the previous slice's tail frames or replicates frame 0).
//
// Structure (CDNA4): 512 threads = 8 waves, block tile BM x BN x 64, v_mfma_f32_16x16x32_bf16,
// both operands streamed HBM/L2 -> LDS with 16-byte global_load_lds (no VGPR round trip), two LDS
// stages, one barrier per K tile.  LDS rows are 128 B (64 bf16); the 16-byte chunk index is XORed
// with (row & 7) on the *source* address and on the ds_read_b128 side (the LDS-DMA destination must
// stay lane-linear), which makes every ds_read_b128 lane group hit 16 distinct bank slots.
// Operands are issued swapped (mfma(Wfrag, Afrag)) so each lane ends up with 4 consecutive output
// columns of one row -> 8-byte bf16 stores and float4 bias/gate loads in the epilogue.
// Block ids are remapped so every XCD works on a contiguous band of tiles (own L2), swept in
// groups of 4 row-panels x all column panels.
#include "svr_common.h"
#include "../../include/seedvr2_hip.h"
#include <cstdlib>
#include <cstring>

namespace svr {

constexpr int BK = 64;
constexpr int THREADS = 512;

struct RowSrc {           // per-thread description of one A-tile row it stages
    const char* base;     // plain: row pointer;  conv: unused
    int t, y, x;          // conv: top-left-front input coordinate of the receptive field
};

// Sub-pixel convolution (a.phase): element offset of conv output voxel m, column n in the 2x upsampled tensor, and the border
// class of the voxel (0 interior, 1 row border, 2 column border, 3 both) that selects its bias vector.
SVR_DEVICE int64_t phase_offset(const svr_gemm_args& a, int m, int n, int& border) {
    const int xo = m % a.conv.Wo;
    const int r2 = m / a.conv.Wo;
    const int yo = r2 % a.conv.Ho;
    const int to = r2 / a.conv.Ho;
    border = (yo == (a.phase.py ? a.conv.Ho - 1 : 0) ? 1 : 0) | (xo == (a.phase.px ? a.conv.Wo - 1 : 0) ? 2 : 0);
    return (((int64_t)to * a.phase.t_stride * (2 * a.conv.Ho) + 2 * yo + a.phase.py) * (2 * a.conv.Wo) + 2 * xo + a.phase.px) * a.N + n;
}

// One lane's 4 consecutive output columns of row m (n .. n+3): fused epilogue + store.
SVR_DEVICE void epilogue_store(const svr_gemm_args& a, const f32x4 accv, const f32x4 u, int m, int n) {
    float v[4] = {accv[0], accv[1], accv[2], accv[3]};
    const int epi = a.epilogue;
    if (epi == SVR_EPI_SWIGLU) {
        // n = 32*hb + 4g (gate block of hidden block hb) -> hidden index 16*hb + 4g
        const int hid = ((n >> 5) << 4) + (n & 15);
        if (a.out_f32) {        // fp32-store test epilogue (tests/test_gpu_kernels.py: the 1e-3 contract)
            *(float4*)((float*)a.C + (int64_t)m * a.ldc + hid) =
                make_float4(silu(v[0]) * u[0], silu(v[1]) * u[1], silu(v[2]) * u[2], silu(v[3]) * u[3]);
            return;
        }
        uint2 o;
        o.x = pack2bf(silu(v[0]) * u[0], silu(v[1]) * u[1]);
        o.y = pack2bf(silu(v[2]) * u[2], silu(v[3]) * u[3]);
        *(uint2*)((char*)a.C + ((int64_t)m * a.ldc + hid) * 2) = o;
        return;
    }
    const bool full = (n + 3 < a.N);
    int border = 0;
    const int64_t poff = a.phase.enabled ? phase_offset(a, m, n, border) : 0;
    {
        const float* bias = (border && a.phase.bias_border) ? a.phase.bias_border + (int64_t)(border - 1) * a.N : a.bias;
        if (bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < a.N) v[r] += bias[n + r];
        }
    }
    if (epi == SVR_EPI_BIAS_SILU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu(v[r]);
    } else if (epi == SVR_EPI_BIAS_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_tanh(v[r]);
    } else if (epi == SVR_EPI_RESID_GATE) {
        if (a.gate) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < a.N) v[r] *= a.gate[n + r];
        }
        if (a.resid && a.resid_f32) {
            const float* rp = (const float*)a.resid + (int64_t)m * a.ldr + n;
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < a.N) v[r] += rp[r];
        } else if (a.resid) {
            const bf16_t* rp = (const bf16_t*)a.resid + (int64_t)m * a.ldr + n;
            if (full) {
                const uint2 rr = *(const uint2*)rp;
                v[0] += bf2f((bf16_t)(rr.x & 0xffff)); v[1] += bf2f((bf16_t)(rr.x >> 16));
                v[2] += bf2f((bf16_t)(rr.y & 0xffff)); v[3] += bf2f((bf16_t)(rr.y >> 16));
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < a.N) v[r] += bf2f(rp[r]);
            }
        }
    }
    int64_t off;   // element offset of v[0] in C
    if (a.phase.enabled) {
        off = poff;
    } else if (a.ps.enabled) {
        const int pw = m % a.ps.W;
        const int r2 = m / a.ps.W;
        const int ph = r2 % a.ps.H;
        const int pf = r2 / a.ps.H;
        const int C = a.ps.C;
        const int blk = n / C, c = n - blk * C;           // blk = (x*2 + y)*rz + z
        const int z = blk % a.ps.rz;
        const int xy = blk / a.ps.rz;
        int fo = pf * a.ps.rz + z;
        if (a.ps.drop_first) {
            if (fo == 1) return;                          // duplicated head frame (remove_head)
            if (fo > 1) fo -= 1;
        }
        const int yo = ph * 2 + (xy >> 1), xo = pw * 2 + (xy & 1);
        off = (((int64_t)fo * (2 * a.ps.H) + yo) * (2 * a.ps.W) + xo) * C + c;
    } else {
        off = (int64_t)m * a.ldc + n;
    }
    if (a.out_f32) {
        float* cp = (float*)a.C + off;
        if (full) *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < a.N) cp[r] = v[r];
        }
    } else {
        bf16_t* cp = (bf16_t*)a.C + off;
        if (full) {
            uint2 o; o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
            *(uint2*)cp = o;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < a.N) cp[r] = f2bf(v[r]);
        }
    }
}

// Store side of the LDS-staged epilogue: one thread finishes 8 consecutive output columns (n .. n + 7) of row m -- same
// arithmetic, in the same order, as epilogue_store() -- and writes them with one 16-byte store (two for fp32 output);
// the residual is read the same way.  `u`: the SwiGLU "in" values (ignored otherwise).  bias8 / gate8: this thread's
// columns (loaded once, every store iteration of a thread has the same column chunk).
SVR_DEVICE void epilogue_store8(const svr_gemm_args& a, const float (&acc8)[8], const float (&u)[8], int m, int n,
                                const float (&bias8)[8], const float (&gate8)[8]) {
    float v[8];
    const int epi = a.epilogue;
    if (epi == SVR_EPI_SWIGLU) {
        const int hid = ((n >> 5) << 4) + (n & 15);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = silu(acc8[e]) * u[e];
        if (a.out_f32) {
            float* cp = (float*)a.C + (int64_t)m * a.ldc + hid;
            *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            *(uint4*)((bf16_t*)a.C + (int64_t)m * a.ldc + hid) = pack8(v);
        }
        return;
    }
    int border = 0;
    const int64_t poff = a.phase.enabled ? phase_offset(a, m, n, border) : 0;
    {
        // (border voxels -- a one-voxel frame of the image -- take their bias from the table; element-wise selects keep v[] in registers)
        const bool tab = border && a.phase.bias_border;
        const float* bb = tab ? a.phase.bias_border + (int64_t)(border - 1) * a.N + n : nullptr;
        float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0;
        if (tab) { t0 = *(const float4*)bb; t1 = *(const float4*)(bb + 4); }
        const float bsel[8] = {tab ? t0.x : bias8[0], tab ? t0.y : bias8[1], tab ? t0.z : bias8[2], tab ? t0.w : bias8[3],
                               tab ? t1.x : bias8[4], tab ? t1.y : bias8[5], tab ? t1.z : bias8[6], tab ? t1.w : bias8[7]};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc8[e] + bsel[e];
    }
    if (epi == SVR_EPI_BIAS_SILU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = silu(v[e]);
    } else if (epi == SVR_EPI_BIAS_GELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_tanh(v[e]);
    } else if (epi == SVR_EPI_RESID_GATE) {
        if (a.gate) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= gate8[e];
        }
        if (a.resid) {
            float r8[8];
            if (a.resid_f32) load8<true>(a.resid, (int64_t)m * a.ldr + n, r8);
            else load8<false>(a.resid, (int64_t)m * a.ldr + n, r8);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r8[e];
        }
    }
    int64_t off;
    if (a.phase.enabled) {
        off = poff;
    } else if (a.ps.enabled) {
        const int pw = m % a.ps.W;
        const int r2 = m / a.ps.W;
        const int ph = r2 % a.ps.H;
        const int pf = r2 / a.ps.H;
        const int C = a.ps.C;
        const int blk = n / C, c = n - blk * C;
        const int z = blk % a.ps.rz;
        const int xy = blk / a.ps.rz;
        int fo = pf * a.ps.rz + z;
        if (a.ps.drop_first) {
            if (fo == 1) return;
            if (fo > 1) fo -= 1;
        }
        const int yo = ph * 2 + (xy >> 1), xo = pw * 2 + (xy & 1);
        off = (((int64_t)fo * (2 * a.ps.H) + yo) * (2 * a.ps.W) + xo) * C + c;
    } else {
        off = (int64_t)m * a.ldc + n;
    }
    if (a.out_f32) {
        float* cp = (float*)a.C + off;
        *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        *(uint4*)((bf16_t*)a.C + off) = pack8(v);
    }
}

template <int BM, int BN, int WM, int WN, bool CONV, bool EPI_LDS = false>
__global__ __launch_bounds__(THREADS) void gemm_kernel(const svr_gemm_args a) {
    constexpr int WAVES_N = BN / WN;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int A_ITERS = BM / 64, B_ITERS = BN / 64;
    constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
    static_assert((BM / WM) * (BN / WN) == 8, "8 waves");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- tile id: XCD-contiguous bands, then grouped (4 row panels x all column panels) order
    const int tiles_m = (a.M + BM - 1) / BM;
    const int tiles_n = (a.N + BN - 1) / BN;
    int t;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, j = bid >> 3, q = nwg >> 3, r = nwg & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    constexpr int GM = 4;
    const int group_size = GM * tiles_n;
    const int group = t / group_size;
    const int first_m = group * GM;
    const int gm = min(tiles_m - first_m, GM);
    const int tm = first_m + (t % group_size) % gm;
    const int tn = (t % group_size) / gm;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging roles: thread stages 16-byte chunk (lane & 7) of rows (tid >> 3) + 64 * i
    const int chunk_src = (lane & 7) ^ (lane >> 3);       // source-side XOR swizzle
    RowSrc rows[A_ITERS];
    const svr_conv_geom& g = a.conv;
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
        int m = m0 + (tid >> 3) + 64 * i;
        m = min(m, a.M - 1);
        if constexpr (CONV) {
            const int xo = m % g.Wo;
            const int r2 = m / g.Wo;
            const int yo = r2 % g.Ho;
            const int to = r2 / g.Ho;
            rows[i].t = to * g.st - g.pt;
            rows[i].y = yo * g.sh - g.ph;
            rows[i].x = xo * g.sw - g.pw;
            rows[i].base = nullptr;
        } else {
            rows[i].base = (const char*)a.A + (int64_t)m * a.lda * 2 + chunk_src * 16;
            rows[i].t = rows[i].y = rows[i].x = 0;
        }
    }
    const char* wrow[B_ITERS];
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
        const int n = n0 + (tid >> 3) + 64 * i;               // W is padded to a multiple of BN rows
        wrow[i] = (const char*)a.W + (int64_t)n * a.K * 2 + chunk_src * 16;
    }

    const int nk = a.K / BK;

    auto stage = [&](int kt, int s) {
        char* sA = smem + s * STAGE_BYTES;
        char* sB = sA + BM * BK * 2;
        const int64_t koff = (int64_t)kt * (BK * 2);
        if constexpr (CONV) {
            const int k0 = kt * BK;
            const int tap = k0 / g.Cin;
            const int c0 = k0 - tap * g.Cin;
            const int dx = tap % g.kw;
            const int r2 = tap / g.kw;
            const int dy = r2 % g.kh;
            const int dt = r2 / g.kh;
#pragma unroll
            for (int i = 0; i < A_ITERS; ++i) {
                int ts = rows[i].t + dt;
                const int ys = rows[i].y + dy, xs = rows[i].x + dx;
                const char* src;
                if ((unsigned)ys >= (unsigned)g.H || (unsigned)xs >= (unsigned)g.W) {
                    src = (const char*)g.zeros;
                } else {
                    const char* basep = (const char*)a.A;
                    if (ts < 0) {
                        if (g.halo != nullptr) { basep = (const char*)g.halo; ts += g.halo_frames; }
                        else ts = 0;
                    }
                    const int64_t vox = ((int64_t)ts * g.H + ys) * g.W + xs;
                    src = basep + (vox * g.Cin + c0 + chunk_src * 8) * 2;
                }
                glds16(src, sA + (wave * 8 + 64 * i) * 128);
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_ITERS; ++i) glds16(rows[i].base + koff, sA + (wave * 8 + 64 * i) * 128);
        }
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) glds16(wrow[i] + koff, sB + (wave * 8 + 64 * i) * 128);
    };

    // ---- compute roles
    const int wm0 = (wave / WAVES_N) * WM;
    const int wn0 = (wave % WAVES_N) * WN;
    const int frow = lane & 15;
    // byte offset of this lane's 16-byte chunk inside a 128-byte LDS row, per k-step
    const int koffs0 = ((0 * 4 + (lane >> 4)) ^ (lane & 7)) << 4;
    const int koffs1 = ((1 * 4 + (lane >> 4)) ^ (lane & 7)) << 4;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    __syncthreads();          // drains the LDS-DMA queue (vmcnt(0)) before the barrier

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, cur ^ 1);
        const char* sA = smem + cur * STAGE_BYTES;
        const char* sB = sA + BM * BK * 2;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ko = ks == 0 ? koffs0 : koffs1;
            bf16x8 af[FM], bfr[FN];
#pragma unroll
            for (int j = 0; j < FN; ++j)
                bfr[j] = *(const bf16x8*)(sB + (wn0 + 16 * j + frow) * 128 + ko);
#pragma unroll
            for (int i = 0; i < FM; ++i)
                af[i] = *(const bf16x8*)(sA + (wm0 + 16 * i + frow) * 128 + ko);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();      // next tile landed (vmcnt(0)) + everyone done reading `cur`
    }

    // ---- epilogue.  Lane holds C[m = 16 i + (lane & 15)][n = 16 j + 4 (lane >> 4) + 0..3].
    const int ng = (lane >> 4) * 4;
    if constexpr (EPI_LDS) {
        // Through LDS (the K loop's stage buffers are free after its last barrier): passes of RI row fragments per wave
        // are parked as fp32 [rows][BN + 4] and leave row-contiguous, 8 columns (16 bytes of bf16) per thread, so every
        // global store / residual load instruction covers whole 128-byte lines.  For short-K problems (1x1 convs, the
        // pixel-shuffle upsamplers: 2 .. 8 K tiles per output tile) the direct epilogue's 8-byte scattered stores
        // -- 16 rows x 32 bytes per instruction -- were most of the kernel's time.
        constexpr int WAVES_M = BM / WM;
        constexpr int RI = 2;
        constexpr int PASS_ROWS = WAVES_M * RI * 16;
        constexpr int PITCH = BN * 4 + 16;
        constexpr int CH = BN / 8, ROWS_IT = THREADS / CH, ITERS = PASS_ROWS / ROWS_IT;
        static_assert(PASS_ROWS * PITCH <= 2 * STAGE_BYTES && FM % RI == 0 && PASS_ROWS % ROWS_IT == 0, "epilogue staging fits the stage buffers");
        const int c8 = tid % CH, r_it = tid / CH;
        const int n = n0 + c8 * 8;
        const bool swiglu = a.epilogue == SVR_EPI_SWIGLU;
        const bool col_ok = n < a.N && !(swiglu && (c8 & 2));      // SwiGLU: "in" blocks are consumed by their gate block's threads
        float bias8[8], gate8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { bias8[e] = 0.f; gate8[e] = 1.f; }
        if (col_ok && !swiglu) {
            if (a.bias) {
                const float4 b0 = *(const float4*)(a.bias + n), b1 = *(const float4*)(a.bias + n + 4);
                bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w;
                bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
            }
            if (a.gate && a.epilogue == SVR_EPI_RESID_GATE) {
                const float4 g0 = *(const float4*)(a.gate + n), g1 = *(const float4*)(a.gate + n + 4);
                gate8[0] = g0.x; gate8[1] = g0.y; gate8[2] = g0.z; gate8[3] = g0.w;
                gate8[4] = g1.x; gate8[5] = g1.y; gate8[6] = g1.z; gate8[7] = g1.w;
            }
        }
#pragma unroll
        for (int p = 0; p < FM / RI; ++p) {
            if (p > 0) __syncthreads();                     // the previous pass has been read out
#pragma unroll
            for (int ii = 0; ii < RI; ++ii) {
                char* row = smem + (((wave / WAVES_N) * RI + ii) * 16 + frow) * PITCH;
#pragma unroll
                for (int j = 0; j < FN; ++j) *(f32x4*)(row + (wn0 + 16 * j + ng) * 4) = acc[p * RI + ii][j];
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int lr = it * ROWS_IT + r_it;         // parked row -> (wave row, fragment, row in fragment)
                const int m = m0 + (lr / (RI * 16)) * WM + 16 * (p * RI + ((lr >> 4) % RI)) + (lr & 15);
                if (!col_ok || m >= a.M) continue;
                const char* src = smem + lr * PITCH + c8 * 32;
                const f32x4 lo = *(const f32x4*)src, hi = *(const f32x4*)(src + 16);
                const float v8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                float u8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (swiglu) {
                    const f32x4 ul = *(const f32x4*)(src + 64), uh = *(const f32x4*)(src + 80);
                    u8[0] = ul[0]; u8[1] = ul[1]; u8[2] = ul[2]; u8[3] = ul[3];
                    u8[4] = uh[0]; u8[5] = uh[1]; u8[6] = uh[2]; u8[7] = uh[3];
                }
                epilogue_store8(a, v8, u8, m, n, bias8, gate8);
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm0 + 16 * i + frow;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn0 + 16 * j + ng;
            // SWIGLU: 16-column blocks alternate gate | in; even j = gate block, odd j = in block
            const f32x4 u = acc[i][j | 1];
            if (m < a.M && n < a.N && !((j & 1) && a.epilogue == SVR_EPI_SWIGLU)) epilogue_store(a, acc[i][j], u, m, n);
        }
    }
}

template <int BM, int BN, int WM, int WN, bool CONV, bool EPI_LDS = false>
static int launch(const svr_gemm_args& a, hipStream_t s) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const size_t lds = 2 * (size_t)(BM + BN) * BK * 2;
    auto kern = gemm_kernel<BM, BN, WM, WN, CONV, EPI_LDS>;
    static uint64_t lds_attr_done = 0;               // per device (svr_common.h)
    {
        const int e = set_max_dynamic_lds((const void*)kern, (int)lds, lds_attr_done);
        if (e != 0) return e;
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(THREADS), lds, s, a);
    return (int)hipGetLastError();
}

// LDS-halo conv kernel (svr_conv_halo2.hip)
static bool conv_halo_eligible(const svr_gemm_args& a);
static int launch_conv_halo2(const svr_gemm_args& a, hipStream_t s);
static bool conv_halo2_eligible(const svr_gemm_args& a);
static bool conv_thin_eligible(const svr_gemm_args& a);
static int launch_conv_thin(const svr_gemm_args& a, hipStream_t s);
// thin-output conv kernel (svr_conv_thinout.hip)
static int launch_conv_thinout(const svr_gemm_args& a, hipStream_t s);
// sub-pixel upsampler conv kernel (svr_conv_sub.hip)
static bool conv_sub_eligible(const svr_gemm_args& a);
static int launch_conv_sub(const svr_gemm_args& a, hipStream_t s);
static int conv_sub_gn_blocks(const svr_gemm_args& a);
int g_conv_impl = 0;   // 0 auto, 1 generic implicit-GEMM kernel everywhere, 3 LDS-halo kernel ignoring W_frag (weights through LDS)

// measurement-only ablation selector of the conv kernels in -DSVR_ABLATIONS builds (svr_set_option("pipe_abl"))
int g_pipe_abl = 0;

// epilogue of gemm_kernel (svr_set_option("gemm_epi")): 0 auto, 1 always direct, 2 through LDS wherever the layout allows it.
// Auto = through LDS except for long-K SwiGLU (measured, profiles/r2_gemm_epilogue.txt: pixel-shuffle upsamplers x2.2-2.5,
// 1x1 convs x1.5, DiT attn-out x1.12, qkv / mlp-out x1.02-1.03, mlp-in SwiGLU x0.98).
int g_gemm_epi = 0;
constexpr int GEMM_EPI_LDS_MAX_K_SWIGLU = 1024;
static bool gemm_epi_lds(const svr_gemm_args& a) {
    if (g_gemm_epi == 1) return false;
    const bool aligned = (a.N % 8) == 0 && ((uintptr_t)a.C % 16) == 0 && (!a.resid || (((uintptr_t)a.resid % 16) == 0 && (a.ldr % 8) == 0)) &&
                         (!a.bias || ((uintptr_t)a.bias % 16) == 0) && (!a.gate || ((uintptr_t)a.gate % 16) == 0) &&
                         (a.ps.enabled ? (a.ps.C % 8) == 0 : a.phase.enabled ? true : (a.ldc % 8) == 0) &&
                         (!a.phase.bias_border || ((uintptr_t)a.phase.bias_border % 16) == 0);
    if (!aligned) return false;
    return g_gemm_epi == 2 || a.epilogue != SVR_EPI_SWIGLU || a.K <= GEMM_EPI_LDS_MAX_K_SWIGLU;
}

// per-frame partial blocks of fused GroupNorm statistics for this problem (0: not produced)
static int conv_gn_blocks(const svr_gemm_args& a);

int gemm_dispatch(const svr_gemm_args& a, hipStream_t s, const char** why) {
    *why = nullptr;
    if (a.M <= 0 || a.N <= 0) return 0;
    if (a.K <= 0 || (a.K % BK) != 0) { *why = "svr_gemm_bf16: K must be a positive multiple of 64"; return -1; }
    if (a.conv.enabled) {
        const svr_conv_geom& g = a.conv;
        if (!conv_thin_eligible(a)) {       // (thin input: Cin = 4, K = taps * 4 zero-padded to 128)
            if (g.Cin % BK != 0) { *why = "svr_gemm_bf16(conv): Cin must be a multiple of 64 (or the thin Cin = 4, 3x3, stride-1 geometry)"; return -1; }
            if (a.K != g.kt * g.kh * g.kw * g.Cin) { *why = "svr_gemm_bf16(conv): K != taps*Cin"; return -1; }
        }
        if (a.M != g.To * g.Ho * g.Wo) { *why = "svr_gemm_bf16(conv): M != To*Ho*Wo"; return -1; }
        if (!g.zeros) { *why = "svr_gemm_bf16(conv): zero page missing"; return -1; }
        if (g.halo && g.halo_frames < g.pt) { *why = "svr_gemm_bf16(conv): halo shorter than causal pad"; return -1; }
    }
    if (a.epilogue == SVR_EPI_SWIGLU && (a.N % 32) != 0) { *why = "svr_gemm_bf16: SWIGLU needs N % 32 == 0"; return -1; }
    if (a.ps.enabled && (a.N != 4 * a.ps.rz * a.ps.C || a.M != a.ps.F * a.ps.H * a.ps.W || (a.ps.C % 4) != 0)) {
        *why = "svr_gemm_bf16: bad pixel-shuffle geometry"; return -1;
    }
    if (a.phase.enabled) {
        const svr_conv_geom& g = a.conv;
        if (!g.enabled || g.st != 1 || g.sh != 1 || g.sw != 1 || g.Ho != g.H || g.Wo != g.W || a.ps.enabled || a.resid ||
            (a.gn_partial && conv_sub_gn_blocks(a) == 0) ||
            a.epilogue == SVR_EPI_SWIGLU || (unsigned)a.phase.py > 1u || (unsigned)a.phase.px > 1u ||
            (a.phase.t_stride != 1 && a.phase.t_stride != 2)) {
            *why = "svr_gemm_bf16: phase scatter needs a stride-1 same-size conv without ps / residual / SwiGLU (fused statistics: sub-pixel conv kernel only)"; return -1;
        }
    }
    if (a.gn_partial && conv_gn_blocks(a) == 0) { *why = "svr_gemm_bf16: gn_partial set but this launch cannot produce fused GroupNorm statistics"; return -1; }
    if (conv_thin_eligible(a)) return launch_conv_thin(a, s);
    if (conv_sub_eligible(a)) return launch_conv_sub(a, s);
    if ((g_conv_impl == 0 || g_conv_impl == 3) && conv_halo2_eligible(a)) return launch_conv_halo2(a, s);
    if (g_conv_impl != 1 && conv_halo_eligible(a) && a.N <= 32) return launch_conv_thinout(a, s);
    // 256-wide tiles when N allows it (otherwise W is padded to a multiple of 128 rows) -- unless they would leave CUs idle:
    // the VAE attention's P V product (16384 x 512 x 16384) has only 128 such tiles for 256 CUs
    const bool wide = (a.N % 256) == 0 &&
                      (int64_t)((a.M + 255) / 256) * (a.N / 256) >= (a.conv.enabled ? 0 : 256);
    if (gemm_epi_lds(a)) {
        if (a.conv.enabled) return wide ? launch<256, 256, 128, 64, true, true>(a, s) : launch<256, 128, 64, 64, true, true>(a, s);
        return wide ? launch<256, 256, 128, 64, false, true>(a, s) : launch<256, 128, 64, 64, false, true>(a, s);
    }
    if (a.conv.enabled) {
        return wide ? launch<256, 256, 128, 64, true>(a, s) : launch<256, 128, 64, 64, true>(a, s);
    }
    return wide ? launch<256, 256, 128, 64, false>(a, s) : launch<256, 128, 64, 64, false>(a, s);
}

}  // namespace svr
