// Software-pipelined 256x256x64 bf16 MFMA GEMM / implicit-GEMM causal Conv3d for gfx950.
//
// Same contract as gemm_kernel (svr_gemm.hip): C[M, N] = A[M, K] * W[N, K]^T + fused epilogue, conv
// mode gathers the A tile per tap from NDHWC activations.  Used whenever N % 256 == 0.
//
// Why a second kernel: in the one-barrier-per-K-tile structure both waves of a SIMD wait for the same
// LDS-DMA at the same time, so the matrix pipe idles half of the time (PMC: MFMA busy 49 %, wave-parked
// 42 %).  Here the K tile is cut into four half-tiles (A0 | B0 | B1 | A1, 128 rows x 64 k each) that are
// streamed by global_load_lds five half-tiles ahead of their consumer with *counted* vmcnt waits (three
// half-tiles stay in flight across every barrier), and the two wave groups of the workgroup (waves 0-3
// / 4-7 = the two waves of each SIMD) run half a phase apart: while one group issues its 8 MFMAs
// (v_mfma_f32_32x32x16_bf16, one 64x32 quadrant x K=64) the other reads fragments from LDS and issues
// the next half-tile.  Four phases per K tile, two raw s_barriers per phase.
//
//   phase q of K tile kt      reads (ds_read_b128)     multiplies          stages (global_load_lds)
//     0                       A0 (8), B0 (4)           acc[0][0]           B0 of kt+1
//     1                       B1 (4)                   acc[0][1]           B1 of kt+1
//     2                       A1 (8)                   acc[1][1]           A1 of kt+1
//     3                       B0 (4)                   acc[1][0]           A0 of kt+2
//
// Hazards (MI355X_MICROARCH.md "Two waves per SIMD" item 7, cdna_hip_programming.md 8-phase rules):
//   RAW  a half-tile is read one phase after the counted vmcnt + barrier that retired it;
//   WAR  every wave retires its own ds_reads (lgkmcnt(0)) before the phase's first barrier, and a
//        buffer is re-staged no earlier than the phase after its last read.
// LDS rows are 128 B; 16-byte chunk c of row r lives at chunk position c ^ ((r >> 1) & 7) (applied on
// the global source address and on the ds_read side; the LDS-DMA destination stays lane-linear), which
// is conflict-free for the 32-row fragments of the 32x32x16 MFMA.
#include "svr_common.h"
#include "../../include/seedvr2_hip.h"
#include <type_traits>

namespace svr {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int P_HALF = 128 * 64 * 2;      // one half-tile: 128 rows x 64 k bf16 = 16 KiB
constexpr int P_KT = 4 * P_HALF;          // one K tile: A0 | B0 | B1 | A1
constexpr int P_TAB = 2 * P_KT;           // conv tap table lives behind the two K-tile buffers
constexpr int P_TAB_BYTES = 4096;         // 256 K tiles x int4
constexpr int P_LDS = P_TAB + P_TAB_BYTES;
constexpr int P_D = 5;                    // prefetch distance in half-tiles
constexpr int P_SUB_A0 = 0, P_SUB_B0 = 1, P_SUB_B1 = 2, P_SUB_A1 = 3;

template <int N> SVR_DEVICE void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
SVR_DEVICE void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ABL: measurement-only ablations (tools/kbench.py, SVR_PIPE_ABL): 0 = product kernel, 1 = no staging
// inside the K loop, 2 = no MFMA, 3 = no fragment reads, 4 = lgkmcnt wait after the barrier.  Results
// of ABL != 0 are garbage by construction.
template <bool CONV, int ABL>
__global__ __launch_bounds__(512) void gemm_pipe_kernel(const svr_gemm_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2;             // wave group: 0 = first wave of each SIMD, 1 = second
    const int wc = wave & 3;

    // ---- tile id: XCD-contiguous bands, then grouped (4 row panels x all column panels) order
    const int tiles_m = (a.M + 255) / 256;
    const int tiles_n = a.N / 256;
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
    const int m0 = tm * 256, n0 = tn * 256;

    const int nk = a.K / 64;
    const svr_conv_geom& g = a.conv;

    // ---- staging roles: thread stages chunk position (lane & 7) of rows srow, srow + 64 of a half-tile
    const int srow = tid >> 3;
    const int chunk_src = (lane & 7) ^ ((srow >> 1) & 7);
    const char* aptr[2][2];               // plain mode: row pointers (half h, i)
    int rt[2][2], ry[2][2], rx[2][2];     // conv mode: receptive-field origin of the row's output voxel
    const char* wptr[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = min(m0 + h * 128 + srow + 64 * i, a.M - 1);
            if constexpr (CONV) {
                const int xo = m % g.Wo;
                const int r2 = m / g.Wo;
                const int yo = r2 % g.Ho;
                const int to = r2 / g.Ho;
                rt[h][i] = to * g.st - g.pt;
                ry[h][i] = yo * g.sh - g.ph;
                rx[h][i] = xo * g.sw - g.pw;
                aptr[h][i] = nullptr;
            } else {
                aptr[h][i] = (const char*)a.A + (int64_t)m * a.lda * 2 + chunk_src * 16;
                rt[h][i] = ry[h][i] = rx[h][i] = 0;
            }
            const int n = n0 + h * 128 + srow + 64 * i;
            wptr[h][i] = (const char*)a.W + (int64_t)n * a.K * 2 + chunk_src * 16;
        }

    if constexpr (CONV) {                 // tap table: K tile -> (dt, dy, dx, first channel)
        if (tid < nk) {
            const int k0 = tid * 64;
            const int tap = k0 / g.Cin;
            const int dx = tap % g.kw;
            const int r2 = tap / g.kw;
            *(int4*)(smem + P_TAB + tid * 16) = make_int4(r2 / g.kh, r2 % g.kh, dx, k0 - tap * g.Cin);
        }
        __syncthreads();
    }

    char* const wave_dst = smem + wave * (8 * 128);       // + buffer + sub-slot + 64 rows * i

    auto stage_b = [&](int h, int ktp, int sub) {
        char* dst = wave_dst + (ktp & 1) * P_KT + sub * P_HALF;
        const int64_t koff = (int64_t)ktp * 128;
        glds16(wptr[h][0] + koff, dst);
        glds16(wptr[h][1] + koff, dst + 64 * 128);
    };
    auto stage_a = [&](int h, int ktp, int sub, int4 tap) {
        char* dst = wave_dst + (ktp & 1) * P_KT + sub * P_HALF;
        if constexpr (CONV) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int ts = rt[h][i] + tap.x;
                const int ys = ry[h][i] + tap.y, xs = rx[h][i] + tap.z;
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
                    src = basep + (vox * g.Cin + tap.w + chunk_src * 8) * 2;
                }
                glds16(src, dst + i * (64 * 128));
            }
        } else {
            const int64_t koff = (int64_t)ktp * 128;
            glds16(aptr[h][0] + koff, dst);
            glds16(aptr[h][1] + koff, dst + 64 * 128);
        }
    };
    auto tap_of = [&](int ktp) -> int4 {
        if constexpr (CONV) return *(const int4*)(smem + P_TAB + min(ktp, nk - 1) * 16);
        else return make_int4(0, 0, 0, 0);
    };

    // ---- fragment read offsets (bytes inside a half-tile)
    const int sw = (lane >> 1) & 7;
    const int rd_a = (wr * 64 + (lane & 31)) * 128;
    const int rd_b = (wc * 32 + (lane & 31)) * 128;
    int koffs[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koffs[ks] = ((2 * ks + (lane >> 5)) ^ sw) << 4;

    f32x16 acc[2][2][2];                  // [A half][B half][32-row block]
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int z = 0; z < 2; ++z)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[x][y][z][e] = 0.f;

    // ---- prologue: half-tiles 0 .. P_D-1 = A0 B0 B1 A1 of K tile 0, A0 of K tile 1
    stage_a(0, 0, P_SUB_A0, tap_of(0));
    stage_b(0, 0, P_SUB_B0);
    stage_b(1, 0, P_SUB_B1);
    stage_a(1, 0, P_SUB_A1, tap_of(0));
    if (nk > 1) {
        stage_a(0, 1, P_SUB_A0, tap_of(1));
        wait_vmcnt<2 * (P_D - 2)>();      // A0, B0 of K tile 0 landed (this wave's share)
    } else {
        wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1 && ABL != 6) __builtin_amdgcn_s_barrier();   // second group runs one barrier behind
    __builtin_amdgcn_sched_barrier(0);

    bf16x8 af[2][4], wf[4];
    if constexpr (ABL == 3 || ABL >= 5) {  // fragments never loaded: give them a defined value
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            wf[ks] = *(const bf16x8*)(smem + rd_b + koffs[ks]);
            af[0][ks] = af[1][ks] = *(const bf16x8*)(smem + rd_a + koffs[ks]);
        }
    }
    int4 tap = tap_of(1);                 // conv: tap of the next A half-tile to stage (A1 of kt+1 / A0 of kt+2)

    auto phase = [&](auto qc, int kt, bool last) {
        constexpr int Q = decltype(qc)::value;
        constexpr int HA = Q >> 1;
        constexpr int HB = (Q == 1 || Q == 2) ? 1 : 0;
        const char* buf = smem + (kt & 1) * P_KT;
        // 1. fragments of this phase
        if constexpr ((Q == 0 || Q == 2) && ABL != 3 && ABL < 5) {
            const char* pa = buf + (HA ? P_SUB_A1 : P_SUB_A0) * P_HALF + rd_a;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) af[mt][ks] = *(const bf16x8*)(pa + mt * (32 * 128) + koffs[ks]);
        }
        if constexpr (Q != 2 && ABL != 3 && ABL < 5) {
            const char* pb = buf + (HB ? P_SUB_B1 : P_SUB_B0) * P_HALF + rd_b;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wf[ks] = *(const bf16x8*)(pb + koffs[ks]);
        }
        // 2. stage half-tile 4 kt + Q + P_D
        constexpr int SUB = (Q + P_D) & 3;
        const int ktp = kt + (Q + P_D) / 4;
        if (ktp < nk && ABL != 1 && ABL < 5) {
            if constexpr (SUB == P_SUB_B0) stage_b(0, ktp, SUB);
            if constexpr (SUB == P_SUB_B1) stage_b(1, ktp, SUB);
            if constexpr (SUB == P_SUB_A1) stage_a(1, ktp, SUB, tap);
            if constexpr (SUB == P_SUB_A0) stage_a(0, ktp, SUB, tap);
            wait_vmcnt<2 * (P_D - 2)>();  // everything the next phase reads has landed (this wave's share)
        } else {
            wait_vmcnt<0>();
        }
        if constexpr (CONV) {             // tap for the A half-tile staged in the next phase
            if constexpr (Q == 1) tap = tap_of(kt + 1);
            if constexpr (Q == 2) tap = tap_of(kt + 2);
        }
        if constexpr (ABL != 4) wait_lgkm0();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ABL != 6) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ABL == 4) wait_lgkm0();
        // 3. one 64 x 32 quadrant x K = 64
        __builtin_amdgcn_s_setprio(1);
        if constexpr (ABL != 2) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[HA][HB][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], af[mt][ks], acc[HA][HB][mt], 0, 0, 0);
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                asm volatile("" ::"v"(wf[ks]));
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) asm volatile("" ::"v"(af[mt][ks]));
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ABL != 6) {
            if (!(last && wr == 1)) __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    for (int kt = 0; kt < nk; ++kt) {
        phase(std::integral_constant<int, 0>{}, kt, false);
        phase(std::integral_constant<int, 1>{}, kt, false);
        phase(std::integral_constant<int, 2>{}, kt, false);
        phase(std::integral_constant<int, 3>{}, kt, kt == nk - 1);
    }

    // ---- epilogue.  32x32 tile: lane holds C[m = lane & 31][n = 8 g + 4 (lane >> 5) + 0..3], g = 0..3.
    const int hi4 = (lane >> 5) * 4;
#pragma unroll
    for (int ha = 0; ha < 2; ++ha)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = m0 + ha * 128 + wr * 64 + 32 * mt + (lane & 31);
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                const f32x16 v = acc[ha][hb][mt];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int n = n0 + hb * 128 + wc * 32 + 8 * gq + hi4;
                    // SWIGLU: columns 0-15 of a 32-block are gate rows, 16-31 the matching "in" rows
                    const int gu = (gq + 2) & 3;
                    const f32x4 accv = {v[4 * gq], v[4 * gq + 1], v[4 * gq + 2], v[4 * gq + 3]};
                    const f32x4 u = {v[4 * gu], v[4 * gu + 1], v[4 * gu + 2], v[4 * gu + 3]};
                    if (m < a.M && n < a.N && !(gq >= 2 && a.epilogue == SVR_EPI_SWIGLU)) epilogue_store(a, accv, u, m, n);
                }
            }
        }
}

template <bool CONV, int ABL>
static int launch_pipe_abl(const svr_gemm_args& a, hipStream_t s) {
    const int tiles = ((a.M + 255) / 256) * (a.N / 256);
    auto kern = gemm_pipe_kernel<CONV, ABL>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), P_LDS, s, a);
    return (int)hipGetLastError();
}

template <bool CONV>
static int launch_pipe(const svr_gemm_args& a, hipStream_t s) {
    const int abl = g_pipe_abl;
    (void)abl;   // the ablation variants (ABL 1..6) are not instantiated in product builds
    return launch_pipe_abl<CONV, 0>(a, s);
}

// true when the pipelined kernel can take this problem
static bool pipe_eligible(const svr_gemm_args& a) {
    return (a.N % 256) == 0 && a.K >= 64 && (a.K / 64) <= 256;
}

}  // namespace svr
