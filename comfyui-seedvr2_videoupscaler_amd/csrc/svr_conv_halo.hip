// LDS-halo implicit-GEMM causal Conv3d for gfx950: the kernel of the VAE's 3x3x3 / 1x3x3 stride-1 convs
// (InflatedCausalConv3d.forward, causal_inflation_lib.py:213-305) -- 97 % of the VAE's FLOPs.
//
// The generic implicit-GEMM kernel (svr_gemm.hip, CONV=true) re-gathers the activation tile from L2
// for each of the 9 spatial taps, so its global->LDS traffic is 64 KiB (N tile 256) or 48 KiB (N tile
// 128) per 64-deep K tile and it is bound by that traffic (profiles/r1_kbench_ab.txt).  Here a
// workgroup owns an 8 x 32 patch of ONE output frame: for every (temporal tap dt, 64-channel slice)
// it stages the 10 x 34 input halo once (42.5 KiB, NDHWC rows of 128 B, 16-byte global_load_lds) and
// runs all 9 spatial taps out of LDS -- the shifted A fragments of tap (dy, dx) are just other rows of
// the same halo image.  Only the weights stream per tap (16 KiB per 128 output channels).  Global->LDS
// traffic per 256x256x64 MACs drops from 64 to ~37 KiB (N tile 256) and from 96 to ~42 KiB (N tile 128).
//
// Pipeline: "interval" k = one (tap, 128-cout weight unit); every interval each wave issues the weight
// unit k+3 (ring of 4 x 16 KiB) and, during the first six intervals of an A step, one 8 KiB piece of
// the NEXT step's halo (double buffered), with counted vmcnt so loads stay in flight across barriers.
// One raw s_barrier per interval.  The two waves of a SIMD run the interval in opposite order
// (group 0: MFMA, then fragment reads + loads; group 1: reads + loads, then MFMA) so one of them
// always feeds the matrix pipe.
//   RAW: unit k+2 / the next halo are waited for (own share) before barrier k, read after it.
//   WAR: a ring slot / halo buffer is re-staged in the interval after the barrier that follows its last
//        read, and every wave drains its ds_reads (lgkmcnt(0)) before each barrier.
// LDS rows are 128 B.  Halo image: chunk c of halo pixel (hy, hx) sits at chunk position
// c ^ ((hx >> 1) & 7); weight image: chunk c of row r at c ^ ((r >> 1) & 7) (source-side swizzle,
// matching XOR on the ds_read side).  Both are conflict-free for the 32 consecutive rows of a 32x32x16
// MFMA fragment starting at ANY halo column, which is what the shifted taps need, and the halo key
// depends on dx only, so all fragment addresses are (12 lane constants) + immediate offsets.
#include "svr_common.h"
#include "../../include/seedvr2_hip.h"
#include <type_traits>

namespace svr {

constexpr int CH_TY = 8, CH_TX = 32;
constexpr int CH_HX = CH_TX + 2, CH_HY = CH_TY + 2;
constexpr int CH_ROWS = CH_HX * CH_HY;                    // 340 halo pixels
constexpr int CH_CHUNKS = CH_ROWS * 8;                    // 2720 16-byte chunks
constexpr int CH_ABUF = CH_ROWS * 128;                    // 43 520 B
constexpr int CH_PIECES = (CH_CHUNKS + 511) / 512;        // 6 (the last one is partial)
constexpr int CH_BUNIT = 128 * 128;                       // 128 couts x 64 k = 16 KiB
constexpr int CH_NB = 4;                                  // weight ring
constexpr int CH_BOFF = 2 * CH_ABUF;
constexpr int CH_LDS = CH_BOFF + CH_NB * CH_BUNIT;        // 152 576 B
constexpr int CH_D = 3;                                   // weight prefetch distance (units)

template <int N> SVR_DEVICE void ch_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ABL: measurement-only ablations (svr_set_option("pipe_abl")): 1 = no staging inside the K loop,
// 2 = no MFMA, 3 = MFMA only (no staging, fragment reads or barriers in the loop).  Garbage results.
template <int BN, int ABL>
__global__ __launch_bounds__(512) void conv_halo_kernel(const svr_gemm_args a) {
    // BN = 128: 8 waves = 4 (row pairs) x 2 (64 couts);  BN = 32 (thin outputs: conv_out 128->3, 512->32):
    // 8 waves = 8 rows x 32 couts, the weight unit is 32 rows (4 KiB) staged by waves 0-3 only.
    static_assert(BN == 128 || BN == 32, "N tile");
    constexpr int HBN = 1;                                // weight units per tap
    constexpr int PPS = 9 * HBN;                          // intervals per A step
    constexpr int MTW = (BN == 128) ? 2 : 1;              // 32-voxel rows per wave
    constexpr int NTW = (BN == 128) ? 2 : 1;              // 32-cout blocks per wave and unit
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((ext_vector_type(16))) float f32x16_t;

    const svr_conv_geom& g = a.conv;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                            // the two waves of a SIMD are in different groups
    const int wm = (BN == 128) ? (wave >> 1) : wave;
    const int wn = (BN == 128) ? (wave & 1) : 0;

    // ---- tile id -> (frame, patch row, patch column, cout tile); XCD-contiguous bands
    const int tiles_x = (g.W + CH_TX - 1) / CH_TX;
    const int tiles_y = (g.H + CH_TY - 1) / CH_TY;
    const int tiles_n = (a.N + BN - 1) / BN;
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
    const int y0 = ty * CH_TY, x0 = tx * CH_TX, n0 = tn * BN;

    const int cpk = g.Cin / 64;                           // 64-channel slices per tap
    const int nA = g.kt * cpk;                            // A steps
    const int P = nA * PPS;                               // intervals
    const int64_t frame_bytes = (int64_t)g.H * g.W * g.Cin * 2;

    // ---- staging roles
    const int srow = tid >> 3;
    const int csrc = (lane & 7) ^ ((tid >> 4) & 7);       // weights: swizzled source chunk for rows srow + 64 i
    uint32_t poff[CH_PIECES];                             // halo piece -> pixel index (or ~0: outside the image)
    uint32_t akeys = 0;                                   // halo piece -> swizzled source chunk (3 bits each)
#pragma unroll
    for (int q = 0; q < CH_PIECES; ++q) {
        const int row = q * 64 + srow;
        const int hy = row / CH_HX, hx = row - hy * CH_HX;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool ok = row < CH_ROWS && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
        poff[q] = ok ? (uint32_t)(y * g.W + x) : 0xffffffffu;
        akeys |= (uint32_t)((lane & 7) ^ ((hx >> 1) & 7)) << (3 * q);
    }
    const char* wbase[2];                                 // weight rows srow, srow + 64 of cout unit 0
#pragma unroll
    for (int i = 0; i < 2; ++i)
        wbase[i] = (const char*)a.W + (int64_t)(n0 + srow + 64 * i) * a.K * 2 + csrc * 16;
    const int64_t wunit = (int64_t)128 * a.K * 2;         // cout unit 1 = 128 rows further
    char* const wave_dst = smem + wave * 1024;

    // input frame of A step s: frame to + dt - pt of the slice, the halo tensor or frame 0 before it
    auto frame_ptr = [&](int s) -> const char* {
        const int dt = s / cpk;
        const int c0 = (s - dt * cpk) * 64;
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
        if (Q * 512 + wave * 64 >= CH_CHUNKS) return false;          // wave-uniform: nothing of this piece
        char* dst = wave_dst + buf * CH_ABUF + Q * 8192;
        const int ck = (akeys >> (3 * Q)) & 7;
        const char* src = poff[Q] == 0xffffffffu ? (const char*)g.zeros
                                                 : fptr + ((int64_t)poff[Q] * g.Cin + ck * 8) * 2;
        if (Q * 512 + tid < CH_CHUNKS) glds16(src, dst);
        return true;
    };
    // weight unit (A step s, tap, cout unit hb) -> ring slot
    auto stage_b = [&](int s, int tap, int hb, int slot) {
        const int dt = s / cpk;
        const int c0 = (s - dt * cpk) * 64;
        const int64_t koff = ((int64_t)(dt * 9 + tap) * g.Cin + c0) * 2 + hb * wunit;
        char* dst = wave_dst + CH_BOFF + slot * CH_BUNIT;
        if constexpr (BN == 128) {
            glds16(wbase[0] + koff, dst);
            glds16(wbase[1] + koff, dst + 64 * 128);
        } else {
            if (wave < 4) glds16(wbase[0] + koff, dst);   // 32 weight rows = the first 256 threads
        }
    };

    // ---- fragment addressing
    const int l31 = lane & 31, hi = lane >> 5;
    int rd_a[3][4], rd_b[4];                              // lane constants: [dx][ks] halo / [ks] weight byte offsets
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            rd_a[dx][ks] = (wm * MTW * CH_HX + l31) * 128 + (((2 * ks + hi) ^ (((dx + l31) >> 1) & 7)) << 4);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)                        // weight rows of a wave start at multiples of 32
        rd_b[ks] = CH_BOFF + (wn * 64 + l31) * 128 + (((2 * ks + hi) ^ ((lane >> 1) & 7)) << 4);

    f32x16_t acc[HBN][MTW][NTW];
#pragma unroll
    for (int x = 0; x < HBN; ++x)
#pragma unroll
        for (int y = 0; y < MTW; ++y)
#pragma unroll
            for (int z = 0; z < NTW; ++z)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[x][y][z][e] = 0.f;
    bf16x8 af[MTW][4], wf[NTW][4];

    // fragment reads of interval position J (tap J / HBN, cout unit J % HBN) of A step s
    auto reads = [&](auto jc, int s, int k) {
        constexpr int J = decltype(jc)::value;
        constexpr int TAP = J / HBN;
        constexpr int DY = TAP / 3, DX = TAP % 3;
        if constexpr (J % HBN == 0) {
            const char* ab = smem + (s & 1) * CH_ABUF;
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)            // halo row (y + dy, x + dx): immediate offset
                    af[mt][ks] = *(const bf16x8*)(ab + rd_a[DX][ks] + ((mt + DY) * CH_HX + DX) * 128);
        }
        const char* bb = smem + (k & (CH_NB - 1)) * CH_BUNIT;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wf[nt][ks] = *(const bf16x8*)(bb + rd_b[ks] + nt * (32 * 128));
    };
    auto mfmas = [&](auto jc) {
        constexpr int HB = decltype(jc)::value % HBN;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    if constexpr (ABL != 2)
                        acc[HB][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt][ks], af[mt][ks], acc[HB][mt][nt], 0, 0, 0);
                    else { asm volatile("" ::"v"(wf[nt][ks])); asm volatile("" ::"v"(af[mt][ks])); }
                }
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: halo of step 0, weight units 0 .. CH_D-1
    {
        const char* f0 = frame_ptr(0);
        stage_a_piece(std::integral_constant<int, 0>{}, f0, 0);
        stage_a_piece(std::integral_constant<int, 1>{}, f0, 0);
        stage_a_piece(std::integral_constant<int, 2>{}, f0, 0);
        stage_a_piece(std::integral_constant<int, 3>{}, f0, 0);
        stage_a_piece(std::integral_constant<int, 4>{}, f0, 0);
        stage_a_piece(std::integral_constant<int, 5>{}, f0, 0);
#pragma unroll
        for (int u = 0; u < CH_D; ++u)                    // PPS >= 9 > CH_D: all in step 0
            stage_b(0, u / HBN, u % HBN, u);
        ch_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
    }
    if (grp == 0 || ABL == 3) {                           // group 0 reads one interval ahead
        reads(std::integral_constant<int, 0>{}, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);

    // one interval: position J of A step s
    auto interval = [&](auto jc, int s, const char* fnext) {
        constexpr int J = decltype(jc)::value;
        const int k = s * PPS + J;
        if (grp == 0) {
            mfmas(jc);
            if constexpr (ABL != 3) {
                if constexpr (J + 1 < PPS) reads(std::integral_constant<int, J + 1>{}, s, k + 1);
                else if (s + 1 < nA) reads(std::integral_constant<int, 0>{}, s + 1, k + 1);
            }
        } else {
            if constexpr (ABL != 3) reads(jc, s, k);
        }
        // loads of this interval: halo piece J of step s+1, weight unit k + CH_D
        bool a_issued = false;
        if constexpr (J < CH_PIECES && ABL != 1 && ABL != 3) {
            if (s + 1 < nA) a_issued = stage_a_piece(std::integral_constant<int, (J < CH_PIECES ? J : 0)>{}, fnext, (s + 1) & 1);
        }
        const bool b_issued = k + CH_D < P && ABL != 1 && ABL != 3;
        if (b_issued) {
            constexpr int JU = (J + CH_D) % PPS;
            stage_b(s + (J + CH_D) / PPS, JU / HBN, JU % HBN, (k + CH_D) & (CH_NB - 1));
        }
        // loads this wave may leave in flight: what it issued in this interval (weight unit k+2 is older)
        const int nb = !b_issued ? 0 : (BN == 128 ? 2 : (wave < 4 ? 1 : 0));
        if (nb == 2)      { if (a_issued) ch_wait_vmcnt<3>(); else ch_wait_vmcnt<2>(); }
        else if (nb == 1) { if (a_issued) ch_wait_vmcnt<2>(); else ch_wait_vmcnt<1>(); }
        else              { if (a_issued) ch_wait_vmcnt<1>(); else ch_wait_vmcnt<0>(); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 1) mfmas(jc);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ABL != 3) __builtin_amdgcn_s_barrier();
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
        if constexpr (PPS == 18) {
            interval(std::integral_constant<int, 9 % PPS>{}, s, fnext);
            interval(std::integral_constant<int, 10 % PPS>{}, s, fnext);
            interval(std::integral_constant<int, 11 % PPS>{}, s, fnext);
            interval(std::integral_constant<int, 12 % PPS>{}, s, fnext);
            interval(std::integral_constant<int, 13 % PPS>{}, s, fnext);
            interval(std::integral_constant<int, 14 % PPS>{}, s, fnext);
            interval(std::integral_constant<int, 15 % PPS>{}, s, fnext);
            interval(std::integral_constant<int, 16 % PPS>{}, s, fnext);
            interval(std::integral_constant<int, 17 % PPS>{}, s, fnext);
        }
    }

    if constexpr (BN != 128) {
        // thin outputs (N <= 32, e.g. 3 RGB channels with ldc = 3): per-lane epilogue, the byte volume is tiny.
        // 32x32 tile: lane holds C[voxel = lane & 31][cout = 8 q + 4 (lane >> 5) + 0..3]
        const int y = y0 + wm, x = x0 + l31;
        const bool ok = y < g.H && x < g.W;
        const int m = (to * g.H + y) * g.W + x;
        const f32x16_t v = acc[0][0][0];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int n = n0 + 8 * gq + hi * 4;
            const f32x4 accv = {v[4 * gq], v[4 * gq + 1], v[4 * gq + 2], v[4 * gq + 3]};
            if (ok && n < a.N) epilogue_store(a, accv, accv, m, n);
        }
        return;
    } else {
    // ---- epilogue through LDS (BN == 128): the MFMA layout gives each lane 4 couts of 32 different
    // voxels (8-byte stores scattered over 32 cache lines, and the same for the residual loads), so the
    // fp32 tile (+ bias) is parked in LDS [256 voxels][132 floats] and written back row-contiguous:
    // 16 lanes cover one voxel's 128 couts, every global access is a full 16-byte lane / 256-byte row.
    constexpr int EP_PITCH = 528;                         // 128 floats + 16 B pad: conflict-free b128 writes
    const int hi4 = hi * 4;
    {
        f32x4 bv[NTW][4];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = n0 + wn * 64 + nt * 32 + 8 * gq + hi4;
                bv[nt][gq] = a.bias ? *(const f32x4*)(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            char* row = smem + ((wm * MTW + mt) * 32 + l31) * EP_PITCH;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const f32x16_t v = acc[0][mt][nt];
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
    const bool resid_gate = a.epilogue == SVR_EPI_RESID_GATE;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int id = it * 512 + tid;
        const int vox = id >> 4, ch = id & 15;            // voxel of the patch, 8-cout chunk
        const int y = y0 + (vox >> 5), x = x0 + (vox & 31);
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
                const uint4 rr = *(const uint4*)((const bf16_t*)a.resid + m * a.ldr + n);
                float r8[8];
                unpack8(rr, r8);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += r8[e];
            }
        }
        if (a.out_f32) {
            float* cp = (float*)a.C + m * a.ldc + n;
            *(float4*)cp = make_float4(f[0], f[1], f[2], f[3]);
            *(float4*)(cp + 4) = make_float4(f[4], f[5], f[6], f[7]);
        } else {
            *(uint4*)((bf16_t*)a.C + m * a.ldc + n) = pack8(f);
        }
    }
    }   // BN == 128
}

template <int BN, int ABL>
static int launch_conv_halo_abl(const svr_gemm_args& a, hipStream_t s) {
    const svr_conv_geom& g = a.conv;
    const int tiles = g.To * ((g.H + CH_TY - 1) / CH_TY) * ((g.W + CH_TX - 1) / CH_TX) * ((a.N + BN - 1) / BN);
    auto kern = conv_halo_kernel<BN, ABL>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), CH_LDS, s, a);
    return (int)hipGetLastError();
}

template <int BN>
static int launch_conv_halo(const svr_gemm_args& a, hipStream_t s) {
#ifdef SVR_ABLATIONS
    switch (g_pipe_abl) {
        case 1: return launch_conv_halo_abl<BN, 1>(a, s);
        case 2: return launch_conv_halo_abl<BN, 2>(a, s);
        case 3: return launch_conv_halo_abl<BN, 3>(a, s);
        default: break;
    }
#endif
    return launch_conv_halo_abl<BN, 0>(a, s);
}

// stride-1 "same" 3x3 spatial kernel (1 or 3 temporal taps), channels in 64-slices, N in 128-tiles,
// plain bias / residual epilogue, bf16 or fp32 store
static bool conv_halo_eligible(const svr_gemm_args& a) {
    const svr_conv_geom& g = a.conv;
    return g.enabled && g.kh == 3 && g.kw == 3 && g.sh == 1 && g.sw == 1 && g.st == 1 && g.ph == 1 && g.pw == 1 &&
           g.Ho == g.H && g.Wo == g.W && g.Cin % 64 == 0 && g.kt >= 1 && g.kt <= 3 &&
           g.To == g.T + g.pt - g.kt + 1 && !a.ps.enabled && !a.phase.enabled && a.epilogue != SVR_EPI_SWIGLU &&
           (a.N <= 32 || ((a.N % 128) == 0 && (a.ldc % 8) == 0 && (!a.resid || (a.ldr % 8) == 0))) &&
           (int64_t)g.H * g.W * g.Cin * 2 < (int64_t)1 << 32;
}

}  // namespace svr
