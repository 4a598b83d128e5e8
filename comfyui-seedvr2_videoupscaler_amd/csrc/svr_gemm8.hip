// Plain bf16 GEMM for gfx950 with the weights streamed to registers: the structure of the 8-row conv kernel
// (svr_conv_halo2.hip, svr_conv_sub.hip) applied to C[M, N] = A[M, K] * W[N, K]^T for the NaDiT's large nn.Linear layers
// (mmattn.py:173,269; mlp.py:60-61).  DESIGN.md 3.2 prices the operand paths: a weight fragment that goes HBM/L2 -> LDS-DMA
// -> LDS -> register costs about three times what the same fragment costs straight from L1 into registers, and 256
// accumulators per wave halve the fragment reads per MFMA of gemm_kernel's 128 x 64 wave tile.
//
//   * workgroup = 256 x 256 output tile, four waves, ONE wave per SIMD, every wave 128 x 128 = 16 accumulators of
//     v_mfma_f32_32x32x16_bf16 (256 AGPRs);
//   * A (activations): 256 rows x 64 k per stage by 16-byte LDS-DMA into a ring of FOUR stage buffers (128 KiB), staged three
//     stages ahead, two pieces per k16 step; rows are 128 B with the 16-byte chunk index XORed with (row & 7) on the source
//     side and on the ds_read_b128 side (gemm_kernel's layout);
//   * W (weights): never in LDS.  svr_conv_pack_frag_taps(W, ., N, K, 1, 1, 1, K) stores them once per checkpoint as
//     [N / 32][K / 32][k-step 2][lane 64][8 bf16] = one coalesced 1 KiB load per MFMA operand; every wave streams the four
//     32-column blocks of its tile, FOUR k16 steps ahead, into eight rotating register sets;
//   * one continuous MFMA stream: per k16 step 4 fragment reads (for the NEXT step), 4 weight loads (four steps ahead), 2 LDS-DMA
//     pieces and 16 MFMAs in two groups of eight; every wait is counted; one workgroup barrier per stage (64 MFMAs), at the
//     end of step 2: by then every wave has finished ALL its reads of the current buffer's predecessor and of this stage
//     (lgkmcnt(0) in front of the barrier), so the pieces issued after the barrier may overwrite the buffer read one stage ago,
//     and the pieces a wave issued three stages ago for the buffer read next were retired by the counted vmcnt waits in between;
//   * epilogue: the fp32 tile is parked in LDS in four passes of 64 rows and leaves through epilogue_store8 of svr_gemm.hip
//     (bias / SiLU / GELU / gate + residual / SwiGLU), 16 bytes of bf16 per thread, whole 128-byte lines per instruction.
// Served: plain GEMMs with a fragment-ordered weight copy, N % 256 == 0, K % 128 == 0, K >= 256, bf16 output, no pixel shuffle;
// everything else stays on gemm_kernel.  svr_set_option("gemm_impl", 0 | 1).
#include "svr_common.h"
#include "../../include/seedvr2_hip.h"
#include <type_traits>

namespace svr {

constexpr int G8_BM = 256, G8_BN = 256, G8_NT = 256;
constexpr int G8_STAGE = G8_BM * 64 * 2;                     // 32 768 B: 256 rows x 64 k
constexpr int G8_NBUF = 4;
constexpr int G8_LDS = G8_NBUF * G8_STAGE;                   // 131 072 B (the epilogue parks in the same space)
constexpr int G8_EP_PITCH = G8_BN * 4 + 16;                  // 256 floats + 16 B pad
static_assert(64 * G8_EP_PITCH + 128 <= G8_LDS, "epilogue parking");

typedef bf16x8 g8_frag4[4];

SVR_DEVICE void g8_read4(g8_frag4& x, unsigned addr) {       // the four 32-row blocks of a wave's 128 rows, one k16 step
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\tds_read_b128 %2, %4 offset:8192\n\tds_read_b128 %3, %4 offset:12288"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]) : "v"(addr) : "memory");
}
template <int N> SVR_DEVICE void g8_wait_lds(g8_frag4& x) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "n"(N));
    __builtin_amdgcn_sched_barrier(0);
}
template <int N> SVR_DEVICE void g8_wait_vm(g8_frag4& x) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "n"(N));
    __builtin_amdgcn_sched_barrier(0);
}
SVR_DEVICE const char* g8_uniform(const char* p) {
    const uint64_t u = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return (const char*)(((uint64_t)hi << 32) | lo);
}

__global__ __launch_bounds__(G8_NT, 1) void gemm8_kernel(const svr_gemm_args a, const int abl_arg, const int stagger) {
#ifdef SVR_ABLATIONS   // measurement build (results invalid): 1 no weight loads | 2 no LDS-DMA pieces | 4 no fragment reads | 8 no stores
    const int abl = abl_arg;
#else
    constexpr int abl = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((ext_vector_type(16))) float f32x16_t;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // experiment (svr_set_option("gemm_stagger", n)): the workgroups of the FIRST wave on the chip start n * (XCD index) sleeps of
    // 8128 cycles late, so that the equal-length tiles that follow reach their store phases out of step instead of all at once
    if (stagger > 0 && blockIdx.x < 512u) {
        for (int i = 0; i < stagger * (int)(blockIdx.x & 7); ++i) __builtin_amdgcn_s_sleep(127);
    }
    const int wm = wave >> 1, wn = wave & 1;              // rows 128 wm .. + 127, columns 128 wn .. + 127 of the tile

    // ---- tile id: XCD-contiguous bands, then groups of 4 row panels x all column panels (gemm_kernel's order)
    const int tiles_m = (a.M + G8_BM - 1) / G8_BM;
    const int tiles_n = a.N / G8_BN;
    int t_;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, j = bid >> 3, q = nwg >> 3, r = nwg & 7;
        t_ = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    constexpr int GM = 4;
    const int group_size = GM * tiles_n;
    const int group = t_ / group_size;
    const int first_m = group * GM;
    const int gm = min(tiles_m - first_m, GM);
    const int tm = first_m + (t_ % group_size) % gm;
    const int tn = (t_ % group_size) / gm;
    const int m0 = tm * G8_BM, n0 = tn * G8_BN;

    const int nk = a.K >> 6;                              // stages of 64 k (even, >= 4: gemm8_eligible)

    // ---- A staging roles: piece i = rows (tid >> 3) + 32 i, this thread's 16-byte chunk (lane & 7), source-side XOR swizzle
    const char* const a_tile = (const char*)a.A + (int64_t)m0 * a.lda * 2 + (((lane & 7) ^ (lane >> 3)) << 4);
    uint32_t rowoff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rowoff[i] = (uint32_t)(min(m0 + (tid >> 3) + 32 * i, a.M - 1) - m0) * (uint32_t)(a.lda * 2);
    char* const wave_dst = smem + wave * 1024;
    auto stage_piece = [&](auto ic, int st, int buf) {
        constexpr int I = decltype(ic)::value;
        glds16(a_tile + rowoff[I] + (int64_t)st * 128, wave_dst + buf * G8_STAGE + I * 4096);
    };

    // ---- A fragment addressing: lane's 16-byte chunk of row (lane & 31) for k16 step 0 (step q: ^ (q << 5)), block 0 of the wave
    const int l31 = lane & 31, hi = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned fa = lds0 + (unsigned)((wm * 128 + l31) * 128 + ((hi ^ (l31 & 7)) << 4));

    // ---- W fragments: [32-column block][k32 slice][k-step][lane][8 bf16]; column block j of this wave, stage st = 4 KiB at st * 4096
    const int64_t nstride = (int64_t)(a.K >> 5) * 2048;
    const char* const wp = (const char*)a.W_frag + (int64_t)(n0 / 32 + wn * 4) * nstride;
    const int voff = lane * 16;

    f32x16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    g8_frag4 a0, a1;                                         // A fragments of the even / odd k16 steps
    g8_frag4 b0, b1, b2, b3, b4, b5, b6, b7;                 // W fragments of steps p = 0 .. 7 of the two-stage loop body

#define G8_WLOAD(B, P0, P1, P2, P3, Q) \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(B[0]) : "v"(voff), "s"(P0), "n"((Q) * 1024) : "memory"); \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(B[1]) : "v"(voff), "s"(P1), "n"((Q) * 1024) : "memory"); \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(B[2]) : "v"(voff), "s"(P2), "n"((Q) * 1024) : "memory"); \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(B[3]) : "v"(voff), "s"(P3), "n"((Q) * 1024) : "memory")
#define G8_MM(B, A, MI, NJ) acc[MI][NJ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B[NJ], A[MI], acc[MI][NJ], 0, 0, 0)
#define G8_MM8(B, A, MA, MB) \
    G8_MM(B, A, MA, 0); G8_MM(B, A, MA, 1); G8_MM(B, A, MA, 2); G8_MM(B, A, MA, 3); \
    G8_MM(B, A, MB, 0); G8_MM(B, A, MB, 1); G8_MM(B, A, MB, 2); G8_MM(B, A, MB, 3)

    // One k16 step: position Q of stage t.  ACUR / BCUR: this step's fragments; ANXT: the next step's (read now); BNXT: the set of
    // the step four ahead (= same Q of stage t + 1, pointers q0 .. q3).  VMEM issue order per step: 4 weight loads, 2 pieces.
    // Younger than BCUR's loads when this step waits for them: the 2 pieces of that step and 3 whole steps = 20 operations.
#define G8_STEP(Q, ACUR, ANXT, BCUR, BNXT) \
    { \
        if (!(abl & 4)) g8_read4(ANXT, (fa ^ (unsigned)((((Q) + 1) & 3) << 5)) + ((Q) == 3 ? nxt : cur)); \
        g8_wait_vm<20>(BCUR); \
        g8_wait_lds<4>(ACUR); \
        G8_MM8(BCUR, ACUR, 0, 1); \
        __builtin_amdgcn_sched_barrier(0); \
        if (!(abl & 1)) { G8_WLOAD(BNXT, q0, q1, q2, q3, Q); } \
        if (!(abl & 2)) { \
            stage_piece(std::integral_constant<int, 2 * (Q)>{}, sst, sbuf); \
            stage_piece(std::integral_constant<int, 2 * (Q) + 1>{}, sst, sbuf); \
        } \
        __builtin_amdgcn_sched_barrier(0); \
        G8_MM8(BCUR, ACUR, 2, 3); \
        __builtin_amdgcn_sched_barrier(0); \
        if constexpr ((Q) == 2) { \
            g8_wait_lds<0>(ANXT); \
            __builtin_amdgcn_s_barrier(); \
            __builtin_amdgcn_sched_barrier(0); \
        } \
    }

    // ---- prologue: A stages 0, 1, 2, the weights of stage 0
    {
        stage_piece(std::integral_constant<int, 0>{}, 0, 0); stage_piece(std::integral_constant<int, 1>{}, 0, 0);
        stage_piece(std::integral_constant<int, 2>{}, 0, 0); stage_piece(std::integral_constant<int, 3>{}, 0, 0);
        stage_piece(std::integral_constant<int, 4>{}, 0, 0); stage_piece(std::integral_constant<int, 5>{}, 0, 0);
        stage_piece(std::integral_constant<int, 6>{}, 0, 0); stage_piece(std::integral_constant<int, 7>{}, 0, 0);
        stage_piece(std::integral_constant<int, 0>{}, 1, 1); stage_piece(std::integral_constant<int, 1>{}, 1, 1);
        stage_piece(std::integral_constant<int, 2>{}, 1, 1); stage_piece(std::integral_constant<int, 3>{}, 1, 1);
        stage_piece(std::integral_constant<int, 4>{}, 1, 1); stage_piece(std::integral_constant<int, 5>{}, 1, 1);
        stage_piece(std::integral_constant<int, 6>{}, 1, 1); stage_piece(std::integral_constant<int, 7>{}, 1, 1);
        stage_piece(std::integral_constant<int, 0>{}, 2, 2); stage_piece(std::integral_constant<int, 1>{}, 2, 2);
        stage_piece(std::integral_constant<int, 2>{}, 2, 2); stage_piece(std::integral_constant<int, 3>{}, 2, 2);
        stage_piece(std::integral_constant<int, 4>{}, 2, 2); stage_piece(std::integral_constant<int, 5>{}, 2, 2);
        stage_piece(std::integral_constant<int, 6>{}, 2, 2); stage_piece(std::integral_constant<int, 7>{}, 2, 2);
        const char* q0 = g8_uniform(wp);
        const char* q1 = g8_uniform(wp + nstride);
        const char* q2 = g8_uniform(wp + 2 * nstride);
        const char* q3 = g8_uniform(wp + 3 * nstride);
        G8_WLOAD(b0, q0, q1, q2, q3, 0);
        G8_WLOAD(b1, q0, q1, q2, q3, 1);
        G8_WLOAD(b2, q0, q1, q2, q3, 2);
        G8_WLOAD(b3, q0, q1, q2, q3, 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    g8_read4(a0, fa);

    for (int t = 0; t < nk; t += 2) {
        {   // stage t: weights of stage t + 1 (same step positions), pieces of stage t + 3 (past the end: the last stage again,
            // cache-hot, into a buffer nobody reads any more -- the counted waits need the same number of operations in flight)
            const unsigned cur = (unsigned)((t & 3) * G8_STAGE), nxt = (unsigned)(((t + 1) & 3) * G8_STAGE);
            const int sst = min(t + 3, nk - 1), sbuf = (t + 3) & 3;
            const int64_t wo = (int64_t)min(t + 1, nk - 1) * 4096;
            const char* q0 = g8_uniform(wp + wo);
            const char* q1 = g8_uniform(wp + nstride + wo);
            const char* q2 = g8_uniform(wp + 2 * nstride + wo);
            const char* q3 = g8_uniform(wp + 3 * nstride + wo);
            G8_STEP(0, a0, a1, b0, b4);
            G8_STEP(1, a1, a0, b1, b5);
            G8_STEP(2, a0, a1, b2, b6);
            G8_STEP(3, a1, a0, b3, b7);
        }
        {   // stage t + 1
            const unsigned cur = (unsigned)(((t + 1) & 3) * G8_STAGE), nxt = (unsigned)(((t + 2) & 3) * G8_STAGE);
            const int sst = min(t + 4, nk - 1), sbuf = (t + 4) & 3;
            const int64_t wo = (int64_t)min(t + 2, nk - 1) * 4096;
            const char* q0 = g8_uniform(wp + wo);
            const char* q1 = g8_uniform(wp + nstride + wo);
            const char* q2 = g8_uniform(wp + 2 * nstride + wo);
            const char* q3 = g8_uniform(wp + 3 * nstride + wo);
            G8_STEP(0, a0, a1, b4, b0);
            G8_STEP(1, a1, a0, b5, b1);
            G8_STEP(2, a0, a1, b6, b2);
            G8_STEP(3, a1, a0, b7, b3);
        }
    }
    // drain the look-ahead loads / reads / pieces; every wave must be out of the K loop before the parking area is written
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
#undef G8_STEP
#undef G8_MM8
#undef G8_MM
#undef G8_WLOAD

    // ---- epilogue: four passes of 64 rows (each wave row parks one 32-row block per pass) as fp32 [row][256 + 4]; a thread then
    // finishes 8 consecutive columns of 8 rows per pass.  Lane holds C[m = lane & 31][n = 8 gq + 4 (lane >> 5) + 0 .. 3] per block.
    constexpr int CH = G8_BN / 8, ROWS_IT = G8_NT / CH, ITERS = 64 / ROWS_IT;
    const int c8 = tid % CH, r_it = tid / CH;
    const int n = n0 + c8 * 8;
    const bool swiglu = a.epilogue == SVR_EPI_SWIGLU;
    const bool col_ok = !(swiglu && (c8 & 2));            // SwiGLU: "in" blocks are consumed by their gate block's threads
    const bool with_resid = a.epilogue == SVR_EPI_RESID_GATE && a.resid != nullptr;
    float bias8[8], gate8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bias8[e] = 0.f; gate8[e] = 1.f; }
    if (!swiglu) {
        if (a.bias) {
            const float4 v0 = *(const float4*)(a.bias + n), v1 = *(const float4*)(a.bias + n + 4);
            bias8[0] = v0.x; bias8[1] = v0.y; bias8[2] = v0.z; bias8[3] = v0.w;
            bias8[4] = v1.x; bias8[5] = v1.y; bias8[6] = v1.z; bias8[7] = v1.w;
        }
        if (a.gate && a.epilogue == SVR_EPI_RESID_GATE) {
            const float4 v0 = *(const float4*)(a.gate + n), v1 = *(const float4*)(a.gate + n + 4);
            gate8[0] = v0.x; gate8[1] = v0.y; gate8[2] = v0.z; gate8[3] = v0.w;
            gate8[4] = v1.x; gate8[5] = v1.y; gate8[6] = v1.z; gate8[7] = v1.w;
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (p > 0) __syncthreads();                         // the previous pass has been read out
        char* row = smem + (wm * 32 + l31) * G8_EP_PITCH;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x16_t v = acc[p][j];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 o = {v[4 * gq], v[4 * gq + 1], v[4 * gq + 2], v[4 * gq + 3]};
                *(f32x4*)(row + (wn * 128 + j * 32 + 8 * gq + 4 * hi) * 4) = o;
            }
        }
        __syncthreads();
        // store side: branch-free sweeps of four rows, so their LDS reads and residual loads are in flight together (the same
        // arithmetic, in the same order, as epilogue_store8; rows past M read a clamped residual address and are masked at the store)
#pragma unroll
        for (int half = 0; half < ITERS / 4; ++half) {
            f32x4 lo[4], hi4[4], ul[4], uh[4];
            uint4 rr[4];
            int mrow[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int lr = (half * 4 + it) * ROWS_IT + r_it;   // parked row: wave row lr >> 5, row lr & 31 of its block p
                mrow[it] = m0 + (lr >> 5) * 128 + p * 32 + (lr & 31);
                const char* src = smem + lr * G8_EP_PITCH + c8 * 32;
                lo[it] = *(const f32x4*)src;
                hi4[it] = *(const f32x4*)(src + 16);
                if (swiglu) { ul[it] = *(const f32x4*)(src + 64); uh[it] = *(const f32x4*)(src + 80); }
            }
            if (with_resid) {
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    rr[it] = *(const uint4*)((const bf16_t*)a.resid + (int64_t)min(mrow[it], a.M - 1) * a.ldr + n);
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                float v[8] = {lo[it][0], lo[it][1], lo[it][2], lo[it][3], hi4[it][0], hi4[it][1], hi4[it][2], hi4[it][3]};
                const bool ok = col_ok && mrow[it] < a.M && !(abl & 8);
                if (swiglu) {
                    const float u[8] = {ul[it][0], ul[it][1], ul[it][2], ul[it][3], uh[it][0], uh[it][1], uh[it][2], uh[it][3]};
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = silu(v[e]) * u[e];
                    const int hid = ((n >> 5) << 4) + (n & 15);
                    if (ok) *(uint4*)((bf16_t*)a.C + (int64_t)mrow[it] * a.ldc + hid) = pack8(v);
                    continue;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bias8[e];
                if (a.epilogue == SVR_EPI_BIAS_SILU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = silu(v[e]);
                } else if (a.epilogue == SVR_EPI_BIAS_GELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = gelu_tanh(v[e]);
                } else if (a.epilogue == SVR_EPI_RESID_GATE) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= gate8[e];
                    if (with_resid) {
                        float r8[8];
                        unpack8(rr[it], r8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += r8[e];
                    }
                }
                if (ok) *(uint4*)((bf16_t*)a.C + (int64_t)mrow[it] * a.ldc + n) = pack8(v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// gemm4_kernel: the same scheme at HALF the accumulators so that TWO independent workgroups share a CU -- what
// profiles/r2_gemm8.txt asks for: gemm8's K loop beats gemm_kernel, but at one workgroup per CU nothing runs under a tile's
// epilogue (13 .. 45 % of the tile).  Here one workgroup stores while the other multiplies.
//   * workgroup = 256 x 128 tile, four waves of 128 x 64 (8 accumulators of v_mfma_f32_32x32x16_bf16 = 128 AGPRs), 64 KiB of LDS;
//   * A: 256 rows x 32 k per stage (16 KiB), ring of four, staged three stages ahead, 2 pieces per k16 step; rows are 64 B,
//     chunk position p of row r holds source chunk p ^ ((r >> 2) & 3) (conflict-free ds_read_b128 for 8 consecutive rows);
//   * W: two 32-column blocks per wave from the fragment-ordered copy, four k16 steps ahead, eight register sets;
//   * loop body = 4 stages = 8 k16 steps = one turn of the ring (buffer indices are constants); per step 4 fragment reads
//     (next step), 2 weight loads, 2 pieces, 8 MFMAs; VMEM order per step: 2 weight loads, 2 pieces -> a step's weights are
//     followed by 2 + 3 * 4 = 14 younger operations when they are awaited;
//   * one barrier per stage at the end of its FIRST step, after lgkmcnt(0) (the reads of the stage's second step are complete:
//     the buffer of the stage before is then free for the pieces issued from now on) and vmcnt(12) (this wave's pieces for the
//     stage read from the next step on -- issued two and a half stages ago, 12 operations younger than them -- have landed).
// NOT YET RUN ON A GPU (written after round 2's GPU budget was spent; compiles, resource usage checked).
// ------------------------------------------------------------------------------------------------------------------------
constexpr int G4_BM = 256, G4_BN = 128, G4_NT = 256;
constexpr int G4_STAGE = G4_BM * 32 * 2;                     // 16 384 B: 256 rows x 32 k
constexpr int G4_LDS = 4 * G4_STAGE;                         // 65 536 B
constexpr int G4_EP_PITCH = G4_BN * 4 + 16;
static_assert(64 * G4_EP_PITCH + 128 <= G4_LDS, "epilogue parking");

typedef bf16x8 g4_frag2[2];

SVR_DEVICE void g4_read4(g8_frag4& x, unsigned addr) {       // the four 32-row blocks of a wave's 128 rows (64-byte rows), one k16 step
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:2048\n\tds_read_b128 %2, %4 offset:4096\n\tds_read_b128 %3, %4 offset:6144"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]) : "v"(addr) : "memory");
}
template <int N> SVR_DEVICE void g4_wait_vm(g4_frag2& x) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(x[0]), "+v"(x[1]) : "n"(N));
    __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(G4_NT, 2) void gemm4_kernel(const svr_gemm_args a, const int stagger) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((ext_vector_type(16))) float f32x16_t;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // experiment (svr_set_option("gemm_stagger", n)): the workgroups of the FIRST wave on the chip start n * (XCD index) sleeps of
    // 8128 cycles late, so that the equal-length tiles that follow reach their store phases out of step instead of all at once
    if (stagger > 0 && blockIdx.x < 512u) {
        for (int i = 0; i < stagger * (int)(blockIdx.x & 7); ++i) __builtin_amdgcn_s_sleep(127);
    }
    const int wm = wave >> 1, wn = wave & 1;              // rows 128 wm .. + 127, columns 64 wn .. + 63 of the tile

    const int tiles_m = (a.M + G4_BM - 1) / G4_BM;
    const int tiles_n = a.N / G4_BN;
    int t_;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, j = bid >> 3, q = nwg >> 3, r = nwg & 7;
        t_ = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    constexpr int GM = 4;
    const int group_size = GM * tiles_n;
    const int group = t_ / group_size;
    const int first_m = group * GM;
    const int gm = min(tiles_m - first_m, GM);
    const int tm = first_m + (t_ % group_size) % gm;
    const int tn = (t_ % group_size) / gm;
    const int m0 = tm * G4_BM, n0 = tn * G4_BN;

    const int nk = a.K >> 5;                              // stages of 32 k (a multiple of 4, >= 8: gemm4_eligible)

    // ---- A staging roles: piece i = rows (tid >> 2) + 64 i, chunk position tid & 3 <- source chunk (tid & 3) ^ ((tid >> 4) & 3)
    const char* const a_tile = (const char*)a.A + (int64_t)m0 * a.lda * 2 + ((((tid & 3) ^ ((tid >> 4) & 3))) << 4);
    uint32_t rowoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rowoff[i] = (uint32_t)(min(m0 + (tid >> 2) + 64 * i, a.M - 1) - m0) * (uint32_t)(a.lda * 2);
    char* const wave_dst = smem + wave * 1024;
    auto stage_piece = [&](auto ic, int st, auto bufc) {
        constexpr int I = decltype(ic)::value, BUF = decltype(bufc)::value;
        glds16(a_tile + rowoff[I] + (int64_t)st * 64, wave_dst + BUF * G4_STAGE + I * 4096);
    };

    const int l31 = lane & 31, hi = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned fa = lds0 + (unsigned)((wm * 128 + l31) * 64 + ((hi ^ ((l31 >> 2) & 3)) << 4));   // k16 step 1 of a stage: ^ 32

    const int64_t nstride = (int64_t)(a.K >> 5) * 2048;
    const char* const wp = (const char*)a.W_frag + (int64_t)(n0 / 32 + wn * 2) * nstride;
    const int voff = lane * 16;

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    g8_frag4 a0, a1;
    g4_frag2 b0, b1, b2, b3, b4, b5, b6, b7;

#define G4_WLOAD(B, P0, P1, Q) \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(B[0]) : "v"(voff), "s"(P0), "n"((Q) * 1024) : "memory"); \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(B[1]) : "v"(voff), "s"(P1), "n"((Q) * 1024) : "memory")
#define G4_MM(B, A, MI, NJ) acc[MI][NJ] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B[NJ], A[MI], acc[MI][NJ], 0, 0, 0)
#define G4_MM4(B, A, MA, MB) G4_MM(B, A, MA, 0); G4_MM(B, A, MA, 1); G4_MM(B, A, MB, 0); G4_MM(B, A, MB, 1)
    // One k16 step P = 0 .. 7 of the body: stage P / 2 (buffer P / 2), position P & 1.  Reads now: step P + 1 (buffer ((P + 1) / 2) & 3,
    // position (P + 1) & 1); weights now: step P + 4 (pointers q0, q1 + (P & 3) KiB); pieces now: 2 (P & 1), + 1 of stage t + P / 2 + 3
    // into buffer (P / 2 + 3) & 3.
#define G4_STEP(P, ACUR, ANXT, BCUR, BNXT, SST) \
    { \
        g4_read4(ANXT, (fa ^ (unsigned)((((P) + 1) & 1) << 5)) + (unsigned)(((((P) + 1) >> 1) & 3) * G4_STAGE)); \
        g4_wait_vm<14>(BCUR); \
        g8_wait_lds<4>(ACUR); \
        G4_MM4(BCUR, ACUR, 0, 1); \
        __builtin_amdgcn_sched_barrier(0); \
        G4_WLOAD(BNXT, q0, q1, (P) & 3); \
        stage_piece(std::integral_constant<int, 2 * ((P) & 1)>{}, SST, std::integral_constant<int, (((P) >> 1) + 3) & 3>{}); \
        stage_piece(std::integral_constant<int, 2 * ((P) & 1) + 1>{}, SST, std::integral_constant<int, (((P) >> 1) + 3) & 3>{}); \
        __builtin_amdgcn_sched_barrier(0); \
        G4_MM4(BCUR, ACUR, 2, 3); \
        __builtin_amdgcn_sched_barrier(0); \
        if constexpr (((P) & 1) == 0) { \
            asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" : "+v"(ANXT[0]), "+v"(ANXT[1]), "+v"(ANXT[2]), "+v"(ANXT[3])); \
            __builtin_amdgcn_s_barrier(); \
            __builtin_amdgcn_sched_barrier(0); \
        } \
    }

    // ---- prologue: A stages 0, 1, 2 (buffers 0, 1, 2), the weights of steps 0 .. 3
    {
        stage_piece(std::integral_constant<int, 0>{}, 0, std::integral_constant<int, 0>{}); stage_piece(std::integral_constant<int, 1>{}, 0, std::integral_constant<int, 0>{});
        stage_piece(std::integral_constant<int, 2>{}, 0, std::integral_constant<int, 0>{}); stage_piece(std::integral_constant<int, 3>{}, 0, std::integral_constant<int, 0>{});
        stage_piece(std::integral_constant<int, 0>{}, 1, std::integral_constant<int, 1>{}); stage_piece(std::integral_constant<int, 1>{}, 1, std::integral_constant<int, 1>{});
        stage_piece(std::integral_constant<int, 2>{}, 1, std::integral_constant<int, 1>{}); stage_piece(std::integral_constant<int, 3>{}, 1, std::integral_constant<int, 1>{});
        stage_piece(std::integral_constant<int, 0>{}, 2, std::integral_constant<int, 2>{}); stage_piece(std::integral_constant<int, 1>{}, 2, std::integral_constant<int, 2>{});
        stage_piece(std::integral_constant<int, 2>{}, 2, std::integral_constant<int, 2>{}); stage_piece(std::integral_constant<int, 3>{}, 2, std::integral_constant<int, 2>{});
        const char* q0 = g8_uniform(wp);
        const char* q1 = g8_uniform(wp + nstride);
        G4_WLOAD(b0, q0, q1, 0);
        G4_WLOAD(b1, q0, q1, 1);
        G4_WLOAD(b2, q0, q1, 2);
        G4_WLOAD(b3, q0, q1, 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    g4_read4(a0, fa);

    for (int t = 0; t < nk; t += 4) {                     // stages t .. t + 3 in buffers 0 .. 3; k16 steps 2 t .. 2 t + 7
        {   // steps 0 .. 3: weights of steps 4 .. 7 of this body (always inside K)
            const char* q0 = g8_uniform(wp + (int64_t)(2 * t + 4) * 1024);
            const char* q1 = g8_uniform(wp + nstride + (int64_t)(2 * t + 4) * 1024);
            G4_STEP(0, a0, a1, b0, b4, t + 3);
            G4_STEP(1, a1, a0, b1, b5, t + 3);
            G4_STEP(2, a0, a1, b2, b6, min(t + 4, nk - 1));
            G4_STEP(3, a1, a0, b3, b7, min(t + 4, nk - 1));
        }
        {   // steps 4 .. 7: weights of steps 0 .. 3 of the NEXT body (past the end: the last four steps again, cache-hot)
            const int64_t so = (int64_t)min(2 * t + 8, 2 * nk - 4) * 1024;
            const char* q0 = g8_uniform(wp + so);
            const char* q1 = g8_uniform(wp + nstride + so);
            G4_STEP(4, a0, a1, b4, b0, min(t + 5, nk - 1));
            G4_STEP(5, a1, a0, b5, b1, min(t + 5, nk - 1));
            G4_STEP(6, a0, a1, b6, b2, min(t + 6, nk - 1));
            G4_STEP(7, a1, a0, b7, b3, min(t + 6, nk - 1));
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
#undef G4_STEP
#undef G4_MM4
#undef G4_MM
#undef G4_WLOAD

    // ---- epilogue: four passes of 64 rows (each wave row parks one 32-row block per pass) as fp32 [row][128 + 4]; one sweep of
    // four rows per thread and pass (16 threads per row, 8 columns each) -- gemm8_kernel's store side
    constexpr int CH = G4_BN / 8, ROWS_IT = G4_NT / CH;
    static_assert(64 / ROWS_IT == 4, "one sweep of four rows per pass");
    const int c8 = tid % CH, r_it = tid / CH;
    const int n = n0 + c8 * 8;
    const bool swiglu = a.epilogue == SVR_EPI_SWIGLU;
    const bool col_ok = !(swiglu && (c8 & 2));
    const bool with_resid = a.epilogue == SVR_EPI_RESID_GATE && a.resid != nullptr;
    float bias8[8], gate8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bias8[e] = 0.f; gate8[e] = 1.f; }
    if (!swiglu) {
        if (a.bias) {
            const float4 v0 = *(const float4*)(a.bias + n), v1 = *(const float4*)(a.bias + n + 4);
            bias8[0] = v0.x; bias8[1] = v0.y; bias8[2] = v0.z; bias8[3] = v0.w;
            bias8[4] = v1.x; bias8[5] = v1.y; bias8[6] = v1.z; bias8[7] = v1.w;
        }
        if (a.gate && a.epilogue == SVR_EPI_RESID_GATE) {
            const float4 v0 = *(const float4*)(a.gate + n), v1 = *(const float4*)(a.gate + n + 4);
            gate8[0] = v0.x; gate8[1] = v0.y; gate8[2] = v0.z; gate8[3] = v0.w;
            gate8[4] = v1.x; gate8[5] = v1.y; gate8[6] = v1.z; gate8[7] = v1.w;
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (p > 0) __syncthreads();
        char* row = smem + (wm * 32 + l31) * G4_EP_PITCH;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x16_t v = acc[p][j];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 o = {v[4 * gq], v[4 * gq + 1], v[4 * gq + 2], v[4 * gq + 3]};
                *(f32x4*)(row + (wn * 64 + j * 32 + 8 * gq + 4 * hi) * 4) = o;
            }
        }
        __syncthreads();
        f32x4 lo[4], hi4[4], ul[4], uh[4];
        uint4 rr[4];
        int mrow[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int lr = it * ROWS_IT + r_it;
            mrow[it] = m0 + (lr >> 5) * 128 + p * 32 + (lr & 31);
            const char* src = smem + lr * G4_EP_PITCH + c8 * 32;
            lo[it] = *(const f32x4*)src;
            hi4[it] = *(const f32x4*)(src + 16);
            if (swiglu) { ul[it] = *(const f32x4*)(src + 64); uh[it] = *(const f32x4*)(src + 80); }
        }
        if (with_resid) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
                rr[it] = *(const uint4*)((const bf16_t*)a.resid + (int64_t)min(mrow[it], a.M - 1) * a.ldr + n);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            float v[8] = {lo[it][0], lo[it][1], lo[it][2], lo[it][3], hi4[it][0], hi4[it][1], hi4[it][2], hi4[it][3]};
            const bool ok = col_ok && mrow[it] < a.M;
            if (swiglu) {
                const float u[8] = {ul[it][0], ul[it][1], ul[it][2], ul[it][3], uh[it][0], uh[it][1], uh[it][2], uh[it][3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = silu(v[e]) * u[e];
                const int hid = ((n >> 5) << 4) + (n & 15);
                if (ok) *(uint4*)((bf16_t*)a.C + (int64_t)mrow[it] * a.ldc + hid) = pack8(v);
                continue;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bias8[e];
            if (a.epilogue == SVR_EPI_BIAS_SILU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = silu(v[e]);
            } else if (a.epilogue == SVR_EPI_BIAS_GELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_tanh(v[e]);
            } else if (a.epilogue == SVR_EPI_RESID_GATE) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= gate8[e];
                if (with_resid) {
                    float r8[8];
                    unpack8(rr[it], r8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += r8[e];
                }
            }
            if (ok) *(uint4*)((bf16_t*)a.C + (int64_t)mrow[it] * a.ldc + n) = pack8(v);
        }
    }
}

// svr_set_option("gemm_impl"): 0 = gemm_kernel for every plain GEMM; 1 = gemm8_kernel for the plain GEMMs that bring a
// fragment-ordered weight copy and fit it; 2 = as 1, but such a GEMM that does NOT fit is an error (tests: no silent fallback);
// 3 / 4 = the same two meanings for gemm4_kernel (two workgroups per CU)
int g_gemm_impl = 0;
int g_gemm_stagger = 0;   // experiment knob, see gemm8_kernel

// what gemm8_kernel serves (everything else: gemm_kernel)
static bool gemm8_eligible(const svr_gemm_args& a) {
    return !a.conv.enabled && !a.ps.enabled && !a.phase.enabled && a.W_frag != nullptr && !a.out_f32 &&
           a.gn_partial == nullptr && (a.N % G8_BN) == 0 && (a.K % 128) == 0 && a.K >= 256 && a.M >= 1 &&
           (a.lda % 8) == 0 && (a.ldc % 8) == 0 && ((uintptr_t)a.A % 16) == 0 && ((uintptr_t)a.C % 16) == 0 &&
           ((uintptr_t)a.W_frag % 16) == 0 && (int64_t)256 * a.lda * 2 < (int64_t)1 << 31 &&
           (!a.resid || (((uintptr_t)a.resid % 16) == 0 && (a.ldr % 8) == 0)) &&
           (!a.bias || ((uintptr_t)a.bias % 16) == 0) && (!a.gate || ((uintptr_t)a.gate % 16) == 0) &&
           (a.epilogue == SVR_EPI_BIAS || a.epilogue == SVR_EPI_BIAS_SILU || a.epilogue == SVR_EPI_BIAS_GELU ||
            a.epilogue == SVR_EPI_RESID_GATE || a.epilogue == SVR_EPI_SWIGLU);
}

static bool gemm4_eligible(const svr_gemm_args& a) {
    svr_gemm_args b = a;
    b.N = a.N % G4_BN == 0 ? G8_BN : 1;                    // same conditions as gemm8 except the column tile: N % 128 == 0
    return gemm8_eligible(b);
}

static int launch_gemm4(const svr_gemm_args& a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles = ((a.M + G4_BM - 1) / G4_BM) * (a.N / G4_BN);
    hipLaunchKernelGGL(gemm4_kernel, dim3(tiles), dim3(G4_NT), G4_LDS, s, a, g_gemm_stagger);
    return (int)hipGetLastError();
}

static int launch_gemm8(const svr_gemm_args& a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles = ((a.M + G8_BM - 1) / G8_BM) * (a.N / G8_BN);
    hipLaunchKernelGGL(gemm8_kernel, dim3(tiles), dim3(G8_NT), G8_LDS, s, a, g_pipe_abl, g_gemm_stagger);
    return (int)hipGetLastError();
}

}  // namespace svr
