// LDS-halo implicit-GEMM causal Conv3d, second generation: 16 x 32-voxel patches, 32-channel K slices.
//
// Same contract and idea as svr_conv_halo.hip (stride-1 3x3 spatial taps run out of an LDS halo image,
// only the weights stream per tap), re-proportioned after the round-1 measurements
// (profiles/r1_halo_experiments.txt): that kernel is limited by the LDS port -- per 16 MFMAs a wave
// reads 16 KiB of fragments and the DMA writes 2.6 KiB -- not by its schedule.  Here
//   * a workgroup owns a 16 x 32 patch (512 voxels) x 128 couts, every wave 128 voxels x 64 couts
//     (4 x 2 accumulators of v_mfma_f32_32x32x16_bf16): weight fragments are shared by 4 voxel rows;
//   * the K slice per interval is 32 channels, so the halo image (18 x 34 pixels x 64 B = 38 KiB) still
//     double-buffers next to an 8-deep ring of 8 KiB weight units: 1.5 KiB of DMA per wave-interval;
//   * taps run dx-major and the halo-row fragments are kept across dy (tap (dy, dx) multiplies output
//     row m with halo row m + dy): 6 row fragments per dx instead of 12;
//   => 6 KiB of fragment reads per 16 MFMAs (was 16), 43 % less LDS-DMA, and twice the work per
//      prologue / epilogue.
// LDS rows are 64 B (4 chunks).  Halo pixel (hy, hx): chunk c at position c ^ ((hx >> 2) & 3); weight row
// r: c ^ ((r >> 2) & 3).  Bank slot of a 16-byte read = ((row & 3) * 4 + position) mod 16, and for the 16
// lanes of a ds_read_b128 group (rows distinct mod 16, same chunk) the pair (row & 3, (row >> 2) & 3)
// is a bijection -> conflict-free for any halo column shift (halo rows are 34 apart, 34 = 2 mod 4 keeps it
// a bijection per image row).
// Pipeline as in the first kernel: one interval per tap, one raw s_barrier per interval, loads issued
// CG_D intervals ahead with counted vmcnt, the two waves of a SIMD in opposite order.
#include "svr_common.h"
#include "../../include/seedvr2_hip.h"
#include <type_traits>

namespace svr {

constexpr int CG_TY = 16, CG_TX = 32;
constexpr int CG_HX = CG_TX + 2, CG_HY = CG_TY + 2;
constexpr int CG_ROWS = CG_HX * CG_HY;                    // 612 halo pixels
constexpr int CG_ABUF = CG_ROWS * 64;                     // 39 168 B: 32 channels per pixel
constexpr int CG_ACHUNKS = CG_ROWS * 4;                   // 2448 16-byte chunks
constexpr int CG_PIECES = (CG_ACHUNKS + 511) / 512;       // 5 (the last one partial)
constexpr int CG_BUNIT = 128 * 64;                        // 128 couts x 32 k = 8 KiB
constexpr int CG_NB = 8;                                  // weight ring
constexpr int CG_BOFF = 2 * CG_ABUF;                      // 78 336
constexpr int CG_LDS = CG_BOFF + CG_NB * CG_BUNIT;        // 143 872 B (the epilogue reuses the first 135 168)
constexpr int CG_D = 4;                                   // weight prefetch distance (intervals)
static_assert(CG_D < CG_NB, "ring slot of unit k + D must not hold a unit still being read");
static_assert(CG_PIECES <= 7, "the next halo must have landed before interval 8");

template <int N> SVR_DEVICE void cg_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(512) void conv_halo2_kernel(const svr_gemm_args a) {
    constexpr int MTW = 4, NTW = 2;                       // 32-voxel rows / 32-cout blocks per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((ext_vector_type(16))) float f32x16_t;

    const svr_conv_geom& g = a.conv;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                            // the two waves of a SIMD are in different groups
    const int wm = wave >> 1, wn = wave & 1;              // rows 4 wm .. 4 wm + 3, couts 64 wn .. 64 wn + 63

    // ---- tile id -> (frame, patch row, patch column, cout tile); XCD-contiguous bands
    const int tiles_x = (g.W + CG_TX - 1) / CG_TX;
    const int tiles_y = (g.H + CG_TY - 1) / CG_TY;
    const int tiles_n = a.N / 128;
    int tl;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, j = bid >> 3, q = nwg >> 3, r = nwg & 7;
        tl = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tn = tl % tiles_n;
    int rr = tl / tiles_n;
    const int tx = rr % tiles_x; rr /= tiles_x;
    const int ty = rr % tiles_y;
    const int to = rr / tiles_y;
    const int y0 = ty * CG_TY, x0 = tx * CG_TX, n0 = tn * 128;

    const int cpk = g.Cin / 32;                           // 32-channel slices per tap
    const int nA = g.kt * cpk;                            // A steps
    const int P = nA * 9;                                 // intervals
    const int64_t frame_bytes = (int64_t)g.H * g.W * g.Cin * 2;

    // ---- staging roles: one 16-byte chunk per thread and piece / unit (row = id >> 2, position = id & 3)
    const int srow = tid >> 2, spos = tid & 3;
    const char* wbase = (const char*)a.W + (int64_t)(n0 + srow) * a.K * 2 + ((spos ^ ((srow >> 2) & 3)) * 16);
    uint32_t poff[CG_PIECES];                             // halo piece -> pixel index (or ~0: outside the image)
    uint32_t akeys = 0;                                   // halo piece -> swizzled source chunk (2 bits each)
#pragma unroll
    for (int q = 0; q < CG_PIECES; ++q) {
        const int row = q * 128 + srow;
        const int hy = row / CG_HX, hx = row - hy * CG_HX;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool ok = row < CG_ROWS && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
        poff[q] = ok ? (uint32_t)(y * g.W + x) : 0xffffffffu;
        akeys |= (uint32_t)(spos ^ ((hx >> 2) & 3)) << (2 * q);
    }
    char* const wave_dst = smem + wave * 1024;

    // input frame of A step s (+ channel slice): frame to + dt - pt, the halo tensor or frame 0 before the slice
    auto frame_ptr = [&](int s) -> const char* {
        const int dt = s / cpk;
        const int c0 = (s - dt * cpk) * 32;
        int f = to + dt - g.pt;
        const char* basep = (const char*)a.A;
        if (f < 0) {
            if (g.halo != nullptr) { basep = (const char*)g.halo; f += g.halo_frames; }
            else f = 0;
        }
        return basep + (int64_t)f * frame_bytes + c0 * 2;
    };
    auto stage_a_piece = [&](auto qc, const char* fptr, int buf) {
        constexpr int Q = decltype(qc)::value;
        if (Q * 512 + wave * 64 >= CG_ACHUNKS) return false;         // wave-uniform: nothing of this piece
        const int ck = (akeys >> (2 * Q)) & 3;
        const char* src = poff[Q] == 0xffffffffu ? (const char*)g.zeros
                                                 : fptr + ((int64_t)poff[Q] * g.Cin + ck * 8) * 2;
        if (Q * 512 + tid < CG_ACHUNKS) glds16(src, wave_dst + buf * CG_ABUF + Q * 8192);
        return true;
    };
    // weight unit (A step s, tap) -> ring slot: one chunk per thread
    auto stage_b = [&](int s, int tap, int slot) {
        const int dt = s / cpk;
        const int c0 = (s - dt * cpk) * 32;
        const int64_t koff = ((int64_t)(dt * 9 + tap) * g.Cin + c0) * 2;
        glds16(wbase + koff, wave_dst + CG_BOFF + slot * CG_BUNIT);
    };

    // ---- fragment addressing: byte offset of k-step 0 for halo column shift dx / for the weight rows;
    // k-step 1 is `^ 32` (chunk bits 4..5 never carry: buffer / slot bases are multiples of 64 B)
    const int l31 = lane & 31, hi = lane >> 5;
    int rd_a0[3], rd_b0;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
        rd_a0[dx] = (wm * MTW * CG_HX + l31) * 64 + ((hi ^ (((dx + l31) >> 2) & 3)) << 4);
    rd_b0 = CG_BOFF + (wn * 64 + l31) * 64 + ((hi ^ ((l31 >> 2) & 3)) << 4);

    f32x16_t acc[MTW][NTW];
#pragma unroll
    for (int y = 0; y < MTW; ++y)
#pragma unroll
        for (int z = 0; z < NTW; ++z)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[y][z][e] = 0.f;
    // halo-row fragments [k-step]; row r = wave row 0..5 (separate small arrays: a [6][2] array lands in scratch)
    bf16x8 ar0[2], ar1[2], ar2[2], ar3[2], ar4[2], ar5[2], wf[NTW][2];

    // Interval position J of an A step = spatial tap (dy = J % 3, dx = J / 3).  dy = 0 loads wave rows 0..3,
    // dy = 1 adds row 4, dy = 2 row 5.
    auto reads = [&](auto jc, int s, int k) {
        constexpr int J = decltype(jc)::value;
        constexpr int DY = J % 3, DX = J / 3;
        int ra = rd_a0[DX] + (s & 1) * CG_ABUF;
        int rb = rd_b0 + (k & (CG_NB - 1)) * CG_BUNIT;
        asm volatile("" : "+v"(ra), "+v"(rb));            // opaque: keeps hipcc from hoisting the address registers
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const char* pa = smem + (ra ^ (ks << 5));
            const char* pb = smem + (rb ^ (ks << 5));
            if constexpr (DY == 0) {
                ar0[ks] = *(const bf16x8*)(pa + (0 * CG_HX + DX) * 64);
                ar1[ks] = *(const bf16x8*)(pa + (1 * CG_HX + DX) * 64);
                ar2[ks] = *(const bf16x8*)(pa + (2 * CG_HX + DX) * 64);
                ar3[ks] = *(const bf16x8*)(pa + (3 * CG_HX + DX) * 64);
            } else if constexpr (DY == 1) {
                ar4[ks] = *(const bf16x8*)(pa + (4 * CG_HX + DX) * 64);
            } else {
                ar5[ks] = *(const bf16x8*)(pa + (5 * CG_HX + DX) * 64);
            }
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) wf[nt][ks] = *(const bf16x8*)(pb + nt * (32 * 64));
        }
    };
    auto mfma_core = [&](const bf16x8 (&r0)[2], const bf16x8 (&r1)[2], const bf16x8 (&r2)[2], const bf16x8 (&r3)[2]) {
#define SVR_MM(KS, R, MT, NT) \
        acc[MT][NT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[NT][KS], R[KS], acc[MT][NT], 0, 0, 0)
        __builtin_amdgcn_s_setprio(1);
        SVR_MM(0, r0, 0, 0); SVR_MM(0, r0, 0, 1); SVR_MM(0, r1, 1, 0); SVR_MM(0, r1, 1, 1);
        SVR_MM(0, r2, 2, 0); SVR_MM(0, r2, 2, 1); SVR_MM(0, r3, 3, 0); SVR_MM(0, r3, 3, 1);
        SVR_MM(1, r0, 0, 0); SVR_MM(1, r0, 0, 1); SVR_MM(1, r1, 1, 0); SVR_MM(1, r1, 1, 1);
        SVR_MM(1, r2, 2, 0); SVR_MM(1, r2, 2, 1); SVR_MM(1, r3, 3, 0); SVR_MM(1, r3, 3, 1);
        __builtin_amdgcn_s_setprio(0);
#undef SVR_MM
    };
    auto mfmas = [&](auto jc) {
        constexpr int DY = decltype(jc)::value % 3;       // output row mt x halo row mt + dy
        if constexpr (DY == 0) mfma_core(ar0, ar1, ar2, ar3);
        else if constexpr (DY == 1) mfma_core(ar1, ar2, ar3, ar4);
        else mfma_core(ar2, ar3, ar4, ar5);
    };

    // ---- prologue: halo of step 0, weight units 0 .. CG_D-1
    {
        const char* f0 = frame_ptr(0);
        stage_a_piece(std::integral_constant<int, 0>{}, f0, 0);
        stage_a_piece(std::integral_constant<int, 1>{}, f0, 0);
        stage_a_piece(std::integral_constant<int, 2>{}, f0, 0);
        stage_a_piece(std::integral_constant<int, 3>{}, f0, 0);
        stage_a_piece(std::integral_constant<int, 4>{}, f0, 0);
#pragma unroll
        for (int u = 0; u < CG_D; ++u) stage_b(0, (u % 3) * 3 + u / 3, u);    // 9 > CG_D: all in step 0
        cg_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
    }
    if (grp == 0) {                                       // group 0 reads one interval ahead
        reads(std::integral_constant<int, 0>{}, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);

    // one interval: position J of A step s
    bool a_prev = false;
    auto interval = [&](auto jc, int s, const char* fnext) {
        constexpr int J = decltype(jc)::value;
        const int k = s * 9 + J;
        if (grp == 0) {
            mfmas(jc);
            if constexpr (J + 1 < 9) reads(std::integral_constant<int, (J + 1) % 9>{}, s, k + 1);
            else if (s + 1 < nA) reads(std::integral_constant<int, 0>{}, s + 1, k + 1);
        } else {
            reads(jc, s, k);
        }
        // loads of this interval: halo piece J of step s+1, weight unit k + CG_D
        bool a_issued = false;
        if constexpr (J < CG_PIECES) {
            if (s + 1 < nA) a_issued = stage_a_piece(std::integral_constant<int, (J < CG_PIECES ? J : 0)>{}, fnext, (s + 1) & 1);
        }
        const bool b_issued = k + CG_D < P;
        if (b_issued) {
            constexpr int JU = (J + CG_D) % 9;
            stage_b(s + (J + CG_D) / 9, (JU % 3) * 3 + JU / 3, (k + CG_D) & (CG_NB - 1));
        }
        // weight unit k+2 (issued CG_D-2 = 2 intervals ago, last op of its interval) must have landed: the wave may
        // leave in flight exactly what it issued after it -- the weight chunks of the last two intervals plus
        // their halo chunks (vmcnt retires in issue order, so an exact count gives the halo two intervals)
        static_assert(CG_D == 4, "counts below are written for a prefetch distance of 4");
        if (b_issued) {
            const int n = 2 + (a_issued ? 1 : 0) + (a_prev ? 1 : 0);
            if (n == 4) cg_wait_vmcnt<4>(); else if (n == 3) cg_wait_vmcnt<3>(); else cg_wait_vmcnt<2>();
        } else {
            cg_wait_vmcnt<0>();
        }
        a_prev = a_issued;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 1) mfmas(jc);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    for (int s = 0; s < nA; ++s) {
        const char* fnext = frame_ptr(min(s + 1, nA - 1));
        interval(std::integral_constant<int, 0>{}, s, fnext);
        interval(std::integral_constant<int, 1>{}, s, fnext);
        interval(std::integral_constant<int, 2>{}, s, fnext);
        interval(std::integral_constant<int, 3>{}, s, fnext);
        interval(std::integral_constant<int, 4>{}, s, fnext);
        interval(std::integral_constant<int, 5>{}, s, fnext);
        interval(std::integral_constant<int, 6>{}, s, fnext);
        interval(std::integral_constant<int, 7>{}, s, fnext);
        interval(std::integral_constant<int, 8>{}, s, fnext);
    }

    // ---- epilogue through LDS, two passes of 256 voxels (patch rows 0-7: waves with wm < 2, rows 8-15: wm >= 2):
    // the fp32 tile (+ bias) is parked in LDS [256 voxels][132 floats] and written back row-contiguous
    // (16 lanes cover one voxel's 128 couts; every global access is a full 16-byte lane / 256-byte row).
    constexpr int EP_PITCH = 528;                         // 128 floats + 16 B pad: conflict-free b128 writes
    const int hi4 = hi * 4;
    f32x4 bv[NTW][4];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int n = n0 + wn * 64 + nt * 32 + 8 * gq + hi4;
            bv[nt][gq] = a.bias ? *(const f32x4*)(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    const bool resid_gate = a.epilogue == SVR_EPI_RESID_GATE;
    // fused GroupNorm statistics of the stored (bf16-rounded) output: this thread always stores the same
    // 8-cout chunk (tid & 15), so it keeps two quad sums over its 16 voxels
    float gs0 = 0.f, gq0 = 0.f, gs1 = 0.f, gq1 = 0.f;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if ((wm >> 1) == pass) {
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                char* row = smem + (((wm & 1) * MTW + mt) * 32 + l31) * EP_PITCH;
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const f32x16_t v = acc[mt][nt];
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        f32x4 o = {v[4 * gq] + bv[nt][gq][0], v[4 * gq + 1] + bv[nt][gq][1],
                                   v[4 * gq + 2] + bv[nt][gq][2], v[4 * gq + 3] + bv[nt][gq][3]};
                        if (a.epilogue == SVR_EPI_BIAS_SILU) { o[0] = silu(o[0]); o[1] = silu(o[1]); o[2] = silu(o[2]); o[3] = silu(o[3]); }
                        *(f32x4*)(row + (wn * 64 + nt * 32 + 8 * gq + hi4) * 4) = o;
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int id = it * 512 + tid;
            const int vox = id >> 4, ch = id & 15;        // voxel of the half patch, 8-cout chunk
            const int y = y0 + pass * 8 + (vox >> 5), x = x0 + (vox & 31);
            if (y >= g.H || x >= g.W) continue;
            const int64_t m = ((int64_t)to * g.H + y) * g.W + x;
            const int n = n0 + ch * 8;
            const f32x4 lo = *(const f32x4*)(smem + vox * EP_PITCH + ch * 32);
            const f32x4 hi_ = *(const f32x4*)(smem + vox * EP_PITCH + ch * 32 + 16);
            float f[8] = {lo[0], lo[1], lo[2], lo[3], hi_[0], hi_[1], hi_[2], hi_[3]};
            if (resid_gate) {
                if (a.gate) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] *= a.gate[n + e];
                }
                if (a.resid) {
                    const uint4 rr8 = *(const uint4*)((const bf16_t*)a.resid + m * a.ldr + n);
                    float r8[8];
                    unpack8(rr8, r8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += r8[e];
                }
            }
            if (a.out_f32) {
                float* cp = (float*)a.C + m * a.ldc + n;
                *(float4*)cp = make_float4(f[0], f[1], f[2], f[3]);
                *(float4*)(cp + 4) = make_float4(f[4], f[5], f[6], f[7]);
            } else {
                const uint4 pk = pack8(f);
                *(uint4*)((bf16_t*)a.C + m * a.ldc + n) = pk;
                if (a.gn_partial) {
                    float r[8];
                    unpack8(pk, r);
                    gs0 += r[0] + r[1] + r[2] + r[3];
                    gq0 += r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
                    gs1 += r[4] + r[5] + r[6] + r[7];
                    gq1 += r[4] * r[4] + r[5] * r[5] + r[6] * r[6] + r[7] * r[7];
                }
            }
        }
        if (pass == 0) __syncthreads();
    }
    if (a.gn_partial) {                                   // fixed-order reduction: thread -> quad -> group
        __syncthreads();
        float4* red = (float4*)smem;                      // [512]
        double2* qsum = (double2*)(smem + 8192);          // [32 quads]
        red[tid] = make_float4(gs0, gq0, gs1, gq1);
        __syncthreads();
        if (tid < 32) {                                   // quad = 2 * chunk + half; rows tid' with tid' & 15 == chunk
            const int c = tid >> 1, h = tid & 1;
            double s = 0.0, q = 0.0;
            for (int j = 0; j < 32; ++j) {
                const float4 v = red[(j << 4) | c];
                s += (double)(h ? v.z : v.x);
                q += (double)(h ? v.w : v.y);
            }
            qsum[tid] = make_double2(s, q);
        }
        __syncthreads();
        const int qpg = (a.N / a.gn_groups) >> 2;         // quads per group (channels per group / 4)
        if (tid < 32 / qpg) {
            double s = 0.0, q = 0.0;
            for (int i = 0; i < qpg; ++i) { s += qsum[tid * qpg + i].x; q += qsum[tid * qpg + i].y; }
            const int blk = ty * tiles_x + tx, nblk = tiles_y * tiles_x;
            ((double2*)a.gn_partial)[((int64_t)to * nblk + blk) * a.gn_groups + (n0 >> 2) / qpg + tid] = make_double2(s, q);
        }
    }
}

static int launch_conv_halo2(const svr_gemm_args& a, hipStream_t s) {
    const svr_conv_geom& g = a.conv;
    const int tiles = g.To * ((g.H + CG_TY - 1) / CG_TY) * ((g.W + CG_TX - 1) / CG_TX) * (a.N / 128);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_halo2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CG_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(conv_halo2_kernel, dim3(tiles), dim3(512), CG_LDS, s, a);
    return (int)hipGetLastError();
}

static int conv_gn_blocks(const svr_gemm_args& a) {
    if (g_conv_impl != 0 || !conv_halo_eligible(a) || (a.N % 128) != 0 || a.conv.Cin % 32 != 0 || a.out_f32) return 0;
    const int cpg = a.gn_groups > 0 ? a.N / a.gn_groups : 0;          // channels per group: 4, 8 or 16
    if (cpg < 4 || (cpg & 3) || a.N % a.gn_groups || 128 % cpg) return 0;
    return ((a.conv.H + CG_TY - 1) / CG_TY) * ((a.conv.W + CG_TX - 1) / CG_TX);
}

// what conv_halo_eligible() accepts with 128-cout tiles and channels in 32-slices
static bool conv_halo2_eligible(const svr_gemm_args& a) {
    return conv_halo_eligible(a) && (a.N % 128) == 0 && a.conv.Cin % 32 == 0;
}

}  // namespace svr
