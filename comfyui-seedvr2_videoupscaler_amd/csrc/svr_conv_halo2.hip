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
//
// WREG variant (args.W_frag set): the weights do not pass through LDS at all.  svr_conv_pack_frag() stores them
// once in MFMA-fragment order -- [32-cout block][interval][k-step][lane][8 bf16], 1 KiB per wave instruction --
// and every wave streams its own 64 couts x 32 k per interval straight into registers with four coalesced
// global_load_dwordx4, two intervals ahead (three rotating register sets).  The four waves that share a cout
// half hit the same lines in L1/L2.  That halves the fragment reads on the LDS port, removes the weight ring,
// its LDS-DMA issues and the per-interval workgroup barrier: the only synchronisation left is one barrier per
// A step (9 intervals) for the halo double buffer, so the two waves of a SIMD drift into opposite phases on their
// own (group 0 runs its MFMA burst at a higher priority to break the tie after each barrier).
#include "svr_common.h"
#include "../../include/seedvr2_hip.h"
#include <type_traits>

namespace svr {

constexpr int CG_TX = 32, CG_HX = CG_TX + 2;
constexpr int CG_BUNIT = 128 * 64;                        // 128 couts x 32 k = 8 KiB
constexpr int CG_NB = 8;                                  // weight ring
constexpr int CG_D = 4;                                   // weight prefetch distance (intervals)
static_assert(CG_D < CG_NB, "ring slot of unit k + D must not hold a unit still being read");

// Patch geometry: TY rows x 32 columns, TY * 32 threads (one wave per 4 rows x 64 couts).
//   TY = 16, weights through the LDS ring   : 512 threads, 140.5 KiB LDS, one workgroup per CU
//   TY =  8, weights in registers (WREG)    : 256 threads, 66 KiB LDS, 228 VGPRs -> TWO independent workgroups per CU:
//            the prologue / epilogue of one overlaps the K loop of the other (what limited 128-channel layers)
template <int TY, int MODE> struct cg_geom {          // MODE 0 weights through the LDS ring, 1 in registers, 2 thin input,
                                                      // 3 in registers, 8 rows per wave, one wave per SIMD (see below)
    static constexpr int MTW = MODE == 3 ? 8 : 4;         // 32-voxel patch rows per wave
    static constexpr int NT = MODE == 3 ? 256 : TY * 32;  // threads (MODE 3: TY = 16 rows = 2 wave rows x 8)
    static constexpr int HY = TY + 2;
    static constexpr int ROWS = CG_HX * HY;               // halo pixels (612 / 340)
    static constexpr int ACHUNKS = ROWS * 4;              // 16-byte chunks
    static constexpr int PIECES = (ACHUNKS + NT - 1) / NT;
    // MODE 3 stages whole pieces from every wave (the tail reads the zero page) so that its vmcnt counts are compile-time
    static constexpr int ABUF = MODE == 3 ? PIECES * NT * 16 : ROWS * 64;     // 32 channels per pixel
    static constexpr int BOFF = 2 * ABUF;
    static constexpr int EP_ROWS = MODE ? 4 : 8;          // patch rows per epilogue pass
    static constexpr int EP_BYTES = EP_ROWS * 32 * 528;
    static constexpr int THIN_PITCH = 272;                // im2col row of the thin-input variant: 128 k + 16 B pad
    static constexpr int MAIN = MODE == 2 ? TY * 32 * THIN_PITCH : (MODE == 1 || MODE == 3 ? BOFF : BOFF + CG_NB * CG_BUNIT);
    static constexpr int LDS = MAIN > EP_BYTES ? MAIN : EP_BYTES;
    static_assert(MODE == 3 || PIECES <= 7, "the next halo must have landed before interval 8");
};

template <int N> SVR_DEVICE void cg_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// MODE 3 helpers.  Fragment reads are issued from inline asm so that they can be in flight under MFMAs (hipcc's own schedule is
// read -> s_waitcnt -> MFMA); hipcc does not count asm loads, so every consumer sits behind a counted wait that names the
// destination registers "+v" (cdna_hip_programming.md 5.7 form (ii)) followed by a sched_barrier.
template <int OFF> SVR_DEVICE void cg_rd2(bf16x8 (&r)[2], unsigned a0) {   // both k-steps of one halo row: chunk c and c ^ 2
    const unsigned a1 = a0 ^ 32u;
    asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%4"
                 : "=&v"(r[0]), "=&v"(r[1]) : "v"(a0), "v"(a1), "n"(OFF) : "memory");
}
template <int N> SVR_DEVICE void cg_wait_rows(bf16x8 (&a)[2]) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a[0]), "+v"(a[1]) : "n"(N));
    __builtin_amdgcn_sched_barrier(0);
}
template <int N> SVR_DEVICE void cg_wait_rows(bf16x8 (&a)[2], bf16x8 (&b)[2]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]) : "n"(N));
    __builtin_amdgcn_sched_barrier(0);
}
template <int N> SVR_DEVICE void cg_wait_rows(bf16x8 (&a)[2], bf16x8 (&b)[2], bf16x8 (&c)[2], bf16x8 (&d)[2], bf16x8 (&e)[2],
                                              bf16x8 (&f)[2], bf16x8 (&g)[2], bf16x8 (&h)[2]) {
    asm volatile("s_waitcnt lgkmcnt(%16)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(c[0]), "+v"(c[1]), "+v"(d[0]), "+v"(d[1]),
                   "+v"(e[0]), "+v"(e[1]), "+v"(f[0]), "+v"(f[1]), "+v"(g[0]), "+v"(g[1]), "+v"(h[0]), "+v"(h[1]) : "n"(N));
    __builtin_amdgcn_sched_barrier(0);
}
template <int N> SVR_DEVICE void cg_wait_w(bf16x8 (&w)[2][2]) {               // counted vmcnt naming one interval's weights
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]) : "n"(N));
    __builtin_amdgcn_sched_barrier(0);
}

#ifdef SVR_ABLATIONS
__device__ unsigned long long g_conv_tl[4096][4];
__device__ unsigned long long g_conv_ep[4096][16];      // DBG 256: stamps inside the epilogue (thread 0)       // DBG 256: s_memtime at kernel start / after prologue / after K loop / end
#endif
// DBG (builds with -DSVR_ABLATIONS only; results invalid): 1 no weight loads, 2 no halo LDS-DMA, 4 no global stores,
// 8 no workgroup barrier in the K loop, 16 halo staged once in the prologue (real data) and never again,
// 32 weights always from the same four (L1-hot) units, 64 halo-row fragments read from LDS in the first A step only, 128 halo re-staged every step but always from the same (cache-hot) addresses
template <int TY, int MODE, int DBG = 0>
__global__ __launch_bounds__((cg_geom<TY, MODE>::NT), (MODE == 3 ? 1 : 2)) void conv_halo2_kernel(const svr_gemm_args a, const int band_rows) {
#ifndef SVR_ABLATIONS
    static_assert(DBG == 0, "measurement variants (results invalid on purpose) exist only in -DSVR_ABLATIONS builds; the product library instantiates DBG = 0");
#endif
    typedef cg_geom<TY, MODE> G;
    constexpr bool WREG = MODE == 1 || MODE == 3, W8 = MODE == 3, THIN = MODE == 2;
    constexpr int CG_TY = TY, CG_ROWS = G::ROWS, CG_ABUF = G::ABUF, CG_ACHUNKS = G::ACHUNKS, CG_PIECES = G::PIECES,
                  CG_BOFF = G::BOFF, NT = G::NT;
    constexpr int MTW = G::MTW, NTW = 2;                  // 32-voxel rows / 32-cout blocks per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((ext_vector_type(16))) float f32x16_t;

    const svr_conv_geom& g = a.conv;
    const int tid = threadIdx.x;
#ifdef SVR_ABLATIONS
    if constexpr ((DBG & 256) != 0) { if (tid == 0 && blockIdx.x < 4096) g_conv_tl[blockIdx.x][0] = __builtin_amdgcn_s_memtime(); }
#endif
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                            // (TY = 16) the two waves of a SIMD are in different groups
    const int wm = wave >> 1, wn = wave & 1;              // rows MTW wm .. MTW wm + MTW - 1, couts 64 wn .. 64 wn + 63

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
    // Order: cout tile fastest, then x, then y inside a band of band_rows tile rows, then the OUTPUT FRAME, then the band: the
    // three temporal consumers of an input region run within one band's worth of traffic of each other, so two of the three
    // fetches of an input frame find it in the memory-side cache instead of HBM (band_rows = 0: frame outermost, round 1's
    // order).  Measured: 1 row per band +1..2.4 % on the kernel, wider bands and column groups nothing (profiles/r2_conv_experiments.txt).
    const int tn = tl % tiles_n;
    int rr = tl / tiles_n;
    const int tx = rr % tiles_x; rr /= tiles_x;
    int ty, to;
    if (band_rows > 0 && band_rows < tiles_y) {
        const int per_band = band_rows * g.To;            // (tile rows x frames) of a full band
        const int b = rr / per_band;                      // band index; the last band may be shorter
        const int rows_b = min(band_rows, tiles_y - b * band_rows);
        const int r2 = rr - b * per_band;
        to = r2 / rows_b;
        ty = b * band_rows + (r2 - to * rows_b);
    } else {
        ty = rr % tiles_y;
        to = rr / tiles_y;
    }
    const int y0 = ty * CG_TY, x0 = tx * CG_TX, n0 = tn * 128;

    const int cpk = THIN ? 1 : g.Cin / 32;                // 32-channel slices per tap
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
        const int row = q * (NT / 4) + srow;
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
        if (Q * NT + wave * 64 >= CG_ACHUNKS) return false;          // wave-uniform: nothing of this piece
        const int ck = (akeys >> (2 * Q)) & 3;
        const char* src = poff[Q] == 0xffffffffu ? (const char*)g.zeros
                                                 : fptr + ((int64_t)poff[Q] * g.Cin + ck * 8) * 2;
        if constexpr (!(DBG & 2))
            if (Q * NT + tid < CG_ACHUNKS) glds16(src, wave_dst + buf * CG_ABUF + Q * (NT * 16));
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
    bf16x8 ar6[2], ar7[2], ar8[2], ar9[2];                // MODE 3: ten halo rows per column shift

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

    if constexpr (THIN) {
        // ---- thin input (Cin = 4: RGB padded; encoder conv_in): the whole K = kt * 9 * 4 <= 128 fits one im2col image of
        // the patch in LDS ([256 voxels][128 k], 272-byte rows); one thread gathers one voxel's taps (8 bytes each, the
        // input is tiny and L2-resident), then 4 x 2 x (K / 16) MFMAs per wave with the weights read straight from
        // their [N, 128] rows.  HBM-bound on the 128-channel output, which leaves through the shared epilogue below
        // (fused GroupNorm statistics included) -- replaces an im2col pass + a K = 128 GEMM.
        constexpr int TP = G::THIN_PITCH;
        static_assert(NT == 256, "one voxel per thread");
        const int taps = g.kt * 9;
        {
            const int y = y0 + (tid >> 5), x = x0 + (tid & 31);
            char* rowp = smem + tid * TP;
            const int64_t fpix = (int64_t)g.H * g.W;
            // (all 27 gathers of a voxel in flight at once: with four at a time the gather phase was a chain of seven exposed L2
            // latencies per workgroup -- kbench "thin": see profiles/r4_kbench_thin.txt)
            uint2 vv[32];
#pragma unroll
            for (int tap = 0; tap < 32; ++tap) {
                uint2 v = make_uint2(0u, 0u);
                if (tap < taps) {
                    const int dt = tap / 9, r = tap - dt * 9, dy = r / 3, dx = r - dy * 3;
                    const int ys = y + dy - 1, xs = x + dx - 1;
                    if ((unsigned)ys < (unsigned)g.H && (unsigned)xs < (unsigned)g.W) {
                        int f = to + dt - g.pt;
                        const char* basep = (const char*)a.A;
                        if (f < 0) {
                            if (g.halo != nullptr) { basep = (const char*)g.halo; f += g.halo_frames; }
                            else f = 0;
                        }
                        v = *(const uint2*)(basep + ((int64_t)f * fpix + (int64_t)ys * g.W + xs) * 8);
                    }
                }
                vv[tap] = v;
            }
#pragma unroll
            for (int tap = 0; tap < 32; tap += 2)
                *(uint4*)(rowp + tap * 8) = make_uint4(vv[tap].x, vv[tap].y, vv[tap + 1].x, vv[tap + 1].y);
        }
        __syncthreads();
        const int nks = (taps * 4 + 15) >> 4;             // 16-wide k steps that hold real taps (7 for 3x3x3)
        const char* wrow0 = (const char*)a.W + ((int64_t)(n0 + wn * 64 + l31) * a.K + hi * 8) * 2;
        const char* arow0 = smem + ((wm * MTW) * 32 + l31) * TP + hi * 16;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks < nks) {
                const bf16x8 b0 = *(const bf16x8*)(wrow0 + ks * 32);
                const bf16x8 b1 = *(const bf16x8*)(wrow0 + (int64_t)32 * a.K * 2 + ks * 32);
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    const bf16x8 av = *(const bf16x8*)(arow0 + mt * 32 * TP + ks * 32);
                    acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, av, acc[mt][0], 0, 0, 0);
                    acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, av, acc[mt][1], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                  // the epilogue reuses the im2col image's LDS
    } else
    if constexpr (WREG) {
        // ---- weights straight from global memory in fragment order
        const int64_t nstride = (int64_t)P * 2048;                                          // bytes per 32-cout block
        const char* wp0 = (const char*)a.W_frag + (int64_t)(n0 / 32 + wn * 2) * nstride;   // wave-uniform (SGPR pair)
        const char* wp1 = wp0 + nstride;
        const int voff = lane * 16;
        bf16x8 w0[NTW][2], w1[NTW][2], w2[NTW][2];          // weights of intervals k = 0, 1, 2 (mod 3)
        // issued as inline asm: hipcc's own waitcnt insertion falls back to vmcnt(0) around the conditional LDS-DMA
        // issues, which would expose the full load latency; the waits below are counted by hand instead
        // (vmcnt retires in issue order)
        auto wload = [&](bf16x8 (&w)[NTW][2], int k) {
            if constexpr ((DBG & 32) != 0) k &= 3;                  // measurement: weights always from the same 4 L1-hot units
            // (the last two intervals re-load the final unit; readfirstlane pins the wave-uniform addresses in SGPRs for
            // the "s" constraints below -- it folds away wherever hipcc already proves them uniform)
            auto uniform_ptr = [](const char* p) {
                const uint64_t u = (uint64_t)p;
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
                return (const char*)(((uint64_t)hi << 32) | lo);
            };
            const char* p0 = uniform_ptr(wp0 + (int64_t)min(k, P - 1) * 2048);
            const char* p1 = uniform_ptr(wp1 + (int64_t)min(k, P - 1) * 2048);
            if constexpr ((DBG & 1) != 0 && W8) {
                if (k > 2) return;                                  // 8-row kernel: the three register sets keep their first (real) contents
            } else if constexpr (DBG & 1) {
                const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                w[0][0] = z; w[0][1] = z; w[1][0] = z; w[1][1] = z;
                return;
            }
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(w[0][0]) : "v"(voff), "s"(p0) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(w[0][1]) : "v"(voff), "s"(p0) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(w[1][0]) : "v"(voff), "s"(p1) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(w[1][1]) : "v"(voff), "s"(p1) : "memory");
        };
        auto reads_a = [&](auto jc, int s) {
            constexpr int J = decltype(jc)::value;
            constexpr int DY = J % 3, DX = J / 3;
            if constexpr ((DBG & 64) != 0) { if (s > 0) return; }  // measurement: halo-row fragments read in the first step only
            int ra = rd_a0[DX] + (s & 1) * CG_ABUF;
            asm volatile("" : "+v"(ra));
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const char* pa = smem + (ra ^ (ks << 5));
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
            }
        };
        auto mfma_w = [&](const bf16x8 (&w)[NTW][2], const bf16x8 (&r0)[2], const bf16x8 (&r1)[2],
                          const bf16x8 (&r2)[2], const bf16x8 (&r3)[2]) {
#define SVR_MM(KS, R, MT, NT) \
            acc[MT][NT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[NT][KS], R[KS], acc[MT][NT], 0, 0, 0)
            SVR_MM(0, r0, 0, 0); SVR_MM(0, r0, 0, 1); SVR_MM(0, r1, 1, 0); SVR_MM(0, r1, 1, 1);
            SVR_MM(0, r2, 2, 0); SVR_MM(0, r2, 2, 1); SVR_MM(0, r3, 3, 0); SVR_MM(0, r3, 3, 1);
            SVR_MM(1, r0, 0, 0); SVR_MM(1, r0, 0, 1); SVR_MM(1, r1, 1, 0); SVR_MM(1, r1, 1, 1);
            SVR_MM(1, r2, 2, 0); SVR_MM(1, r2, 2, 1); SVR_MM(1, r3, 3, 0); SVR_MM(1, r3, 3, 1);
#undef SVR_MM
        };
        if constexpr (W8) {
        // ---- MODE 3: 8 patch rows x 64 couts per wave (16 accumulators = 256 registers), ONE wave per SIMD.
        // Why: the kernel is bound by the power-managed clock, not by issue slots (profiles/r2_conv_experiments.txt: half the
        // resident waves or a stall-free schedule change nothing), and tools/ubench/operand_cost prices the operand paths of
        // the 4-row kernel at 24 % of a bare MFMA loop -- weight fragments from L1 7 %, fragment reads from LDS 5 %, halo
        // staging 4-12 %.  Eight rows per wave reuse every weight fragment twice as often (half the L1 bytes per MFMA), need
        // 10 halo rows per 8 output rows instead of 6 per 4 (-17 % LDS bytes) and an 18 x 34 halo per 16 x 32 patch (-10 %
        // staged bytes).
        // With no second wave on the SIMD to cover latencies the wave runs ONE continuous MFMA stream: a halo row's fragments
        // are re-read into the registers of a row the running burst has finished with (rows 8, 9 of this column shift at the
        // start of dy = 0; row 0 of the next shift under dy = 0, row 1 under dy = 1, rows 2..7 under dy = 2 as their pairs
        // retire), the weight loads and the LDS-DMA pieces of the next A step sit between MFMA groups, and every wait is
        // counted and retires operations issued at least a quarter burst (256 cycles) earlier.  One workgroup barrier per A
        // step, at the end of interval 6: by then every wave has finished reading this step's buffer (rows 8, 9 of dx = 2
        // were the last) and its pieces of the next halo have landed; the first reads of the next buffer follow in interval 7.
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
        unsigned fa[3];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) fa[dx] = lds0 + (unsigned)rd_a0[dx];
        auto stage_full = [&](auto qc, const char* fptr, int buf) {
            constexpr int Q = decltype(qc)::value;
            const int ck = (akeys >> (2 * Q)) & 3;
            const char* src = (poff[Q] == 0xffffffffu || fptr == nullptr) ? (const char*)g.zeros
                                                                          : fptr + ((int64_t)poff[Q] * g.Cin + ck * 8) * 2;
            glds16(src, wave_dst + buf * CG_ABUF + Q * (NT * 16));
        };
        static_assert(CG_PIECES == 10, "piece schedule below is written for ten pieces, two per interval 0..4");
        // (measurement builds, results invalid: DBG 16 no halo staging in the loop, 1 no weight loads, 64 no fragment reads in
        // the loop -- the counted waits that would then be wrong are dropped with them)
        constexpr bool ABL_NO_DMA = (DBG & 16) != 0, ABL_NO_W = (DBG & 1) != 0, ABL_NO_RD = (DBG & 64) != 0;
#define SVR_RD(ROW, R, DXV, BUFOFF) do { if constexpr (!ABL_NO_RD) cg_rd2<((R) * CG_HX + (DXV)) * 64>(ROW, fa[DXV] + (BUFOFF)); } while (0)
#define SVR_MM(W, KS, ROW, MT, NTI) \
        acc[MT][NTI] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[NTI][KS], ROW[KS], acc[MT][NTI], 0, 0, 0)
#define SVR_MM8(W, RA, MA, RB, MB) \
        SVR_MM(W, 0, RA, MA, 0); SVR_MM(W, 0, RA, MA, 1); SVR_MM(W, 0, RB, MB, 0); SVR_MM(W, 0, RB, MB, 1); \
        SVR_MM(W, 1, RA, MA, 0); SVR_MM(W, 1, RA, MA, 1); SVR_MM(W, 1, RB, MB, 0); SVR_MM(W, 1, RB, MB, 1)
        // (The last A step runs the same code: its "next halo" pieces stage the zero page into the idle buffer and its
        // look-ahead reads fetch those zeros -- one code path keeps every vmcnt count a compile-time constant, and a peeled
        // copy made hipcc shuffle the 256 accumulators between register files at its entry.)
        auto interval8 = [&](auto jc, int s, const char* fnext, bf16x8 (&wc)[NTW][2], bf16x8 (&wn_)[NTW][2]) {
            constexpr int J = decltype(jc)::value;
            constexpr int DY = J % 3, DX = J / 3;
            const int k = s * 9 + J;
            const unsigned cur = (unsigned)((s & 1) * CG_ABUF), nxt = (unsigned)(((s + 1) & 1) * CG_ABUF);
            // VMEM issue order per interval: 4 weight loads, then (J < 5) 2 halo pieces.  Younger than the
            // weights of interval J (issued in J - 2): the pieces of J - 2, the weights and pieces of J - 1.
            constexpr int NPC[9] = {2, 2, 2, 2, 2, 0, 0, 0, 0};
            constexpr int NV = 4 + NPC[(J + 8) % 9] + NPC[(J + 7) % 9];
            if constexpr (DY == 0) {
                cg_wait_rows<0>(ar0, ar1, ar2, ar3, ar4, ar5, ar6, ar7);
                SVR_RD(ar8, 8, DX, cur);
                SVR_RD(ar9, 9, DX, cur);
            } else if constexpr (DY == 1 && J != 7) {
                cg_wait_rows<4>(ar8);                    // in flight behind it: row 9, row 0 of the next shift
            } else if constexpr (DY == 2 && J != 8) {
                cg_wait_rows<4>(ar9);                    // row 0 and row 1 of the next shift
            }
            if constexpr (!ABL_NO_DMA && !ABL_NO_W) cg_wait_w<NV>(wc);
            else if constexpr (ABL_NO_DMA && !ABL_NO_W) cg_wait_w<4>(wc);
            if constexpr (J == 7) SVR_RD(ar0, 0, 0, nxt);
            __builtin_amdgcn_sched_barrier(0);
            // group 1
            if constexpr (DY == 0) { SVR_MM8(wc, ar0, 0, ar1, 1); }
            else if constexpr (DY == 1) { SVR_MM8(wc, ar1, 0, ar2, 1); }
            else { SVR_MM8(wc, ar2, 0, ar3, 1); }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DX < 2) {
                if constexpr (DY == 0) SVR_RD(ar0, 0, DX + 1, cur);
                else if constexpr (DY == 1) SVR_RD(ar1, 1, DX + 1, cur);
                else { SVR_RD(ar2, 2, DX + 1, cur); SVR_RD(ar3, 3, DX + 1, cur); }
            } else {
                if constexpr (DY == 1) SVR_RD(ar1, 1, 0, nxt);
                else if constexpr (DY == 2) { SVR_RD(ar2, 2, 0, nxt); SVR_RD(ar3, 3, 0, nxt); }
            }
            wload(wn_, k + 2);
            __builtin_amdgcn_sched_barrier(0);
            // group 2
            if constexpr (DY == 0) { SVR_MM8(wc, ar2, 2, ar3, 3); }
            else if constexpr (DY == 1) { SVR_MM8(wc, ar3, 2, ar4, 3); }
            else { SVR_MM8(wc, ar4, 2, ar5, 3); }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DY == 2) {
                if constexpr (DX < 2) { SVR_RD(ar4, 4, DX + 1, cur); SVR_RD(ar5, 5, DX + 1, cur); }
                else { SVR_RD(ar4, 4, 0, nxt); SVR_RD(ar5, 5, 0, nxt); }
            }
            if constexpr (J < 5 && !ABL_NO_DMA) {
                stage_full(std::integral_constant<int, (J < 5 ? 2 * J : 0)>{}, fnext, (s + 1) & 1);
                stage_full(std::integral_constant<int, (J < 5 ? 2 * J + 1 : 0)>{}, fnext, (s + 1) & 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            // group 3
            if constexpr (DY == 0) { SVR_MM8(wc, ar4, 4, ar5, 5); }
            else if constexpr (DY == 1) { SVR_MM8(wc, ar5, 4, ar6, 5); }
            else { SVR_MM8(wc, ar6, 4, ar7, 5); }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DY == 2) {
                if constexpr (DX < 2) { SVR_RD(ar6, 6, DX + 1, cur); SVR_RD(ar7, 7, DX + 1, cur); }
                else { SVR_RD(ar6, 6, 0, nxt); SVR_RD(ar7, 7, 0, nxt); }
            }
            __builtin_amdgcn_sched_barrier(0);
            // group 4
            if constexpr (DY == 0) { SVR_MM8(wc, ar6, 6, ar7, 7); }
            else if constexpr (DY == 1) { SVR_MM8(wc, ar7, 6, ar8, 7); }
            else { SVR_MM8(wc, ar8, 6, ar9, 7); }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (J == 6) {
                cg_wait_rows<0>(ar8, ar9);                // the last reads of this step's buffer
                if constexpr (!ABL_NO_DMA && !ABL_NO_W) cg_wait_vmcnt<8>();   // the pieces of interval 4 (younger: the weight loads of 5 and 6)
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        {
            const char* f0 = frame_ptr(0);
            stage_full(std::integral_constant<int, 0>{}, f0, 0);
            stage_full(std::integral_constant<int, 1>{}, f0, 0);
            stage_full(std::integral_constant<int, 2>{}, f0, 0);
            stage_full(std::integral_constant<int, 3>{}, f0, 0);
            stage_full(std::integral_constant<int, 4>{}, f0, 0);
            stage_full(std::integral_constant<int, 5>{}, f0, 0);
            stage_full(std::integral_constant<int, 6>{}, f0, 0);
            stage_full(std::integral_constant<int, 7>{}, f0, 0);
            stage_full(std::integral_constant<int, 8>{}, f0, 0);
            stage_full(std::integral_constant<int, 9>{}, f0, 0);
            wload(w0, 0);
            wload(w1, 1);
            if constexpr (ABL_NO_W) wload(w2, 2);
            // hipcc materialises the 256 zeroed accumulators where the first MFMA needs them -- 249 v_accvgpr_write BEHIND this wait
            // and barrier, ~1 000 issue cycles per tile with nothing in flight.  Pinning them here puts the writes under the latency
            // of the first halo and weight loads (behind the sched_barrier, so that they do not delay the loads' issue): bit-identical,
            // -0.9 / -1.2 % on the 128-channel forms, -0.2 % at 256, 0 at 512 channels (profiles/r5_conv_queue_ab.txt).
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int y = 0; y < MTW; ++y)
#pragma unroll
                for (int z = 0; z < NTW; ++z) asm volatile("" : "+a"(acc[y][z]));
            cg_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#define SVR_RD0(ROW, R) cg_rd2<((R) * CG_HX) * 64>(ROW, fa[0])
            SVR_RD0(ar0, 0); SVR_RD0(ar1, 1); SVR_RD0(ar2, 2); SVR_RD0(ar3, 3);
            SVR_RD0(ar4, 4); SVR_RD0(ar5, 5); SVR_RD0(ar6, 6); SVR_RD0(ar7, 7);
            if constexpr (ABL_NO_RD) { SVR_RD0(ar8, 8); SVR_RD0(ar9, 9); }
#undef SVR_RD0
        }
#ifdef SVR_ABLATIONS
        if constexpr ((DBG & 256) != 0) { if (tid == 0 && blockIdx.x < 4096) g_conv_tl[blockIdx.x][1] = __builtin_amdgcn_s_memtime(); }
#endif
        for (int s = 0; s < nA; ++s) {
            const char* fnext = s + 1 < nA ? frame_ptr((DBG & 128) ? 0 : s + 1) : nullptr;     // (128: always the same, cache-hot halo)
            interval8(std::integral_constant<int, 0>{}, s, fnext, w0, w2);
            interval8(std::integral_constant<int, 1>{}, s, fnext, w1, w0);
            interval8(std::integral_constant<int, 2>{}, s, fnext, w2, w1);
            interval8(std::integral_constant<int, 3>{}, s, fnext, w0, w2);
            interval8(std::integral_constant<int, 4>{}, s, fnext, w1, w0);
            interval8(std::integral_constant<int, 5>{}, s, fnext, w2, w1);
            interval8(std::integral_constant<int, 6>{}, s, fnext, w0, w2);
            interval8(std::integral_constant<int, 7>{}, s, fnext, w1, w0);
            interval8(std::integral_constant<int, 8>{}, s, fnext, w2, w1);
        }
        // drain: the hand-issued weight loads (the epilogue reuses their registers) and the look-ahead fragment reads
        cg_wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#undef SVR_RD
#undef SVR_MM
#undef SVR_MM8
        } else {
        bool a_prev3 = false;
        auto interval3 = [&](auto jc, int s, const char* fnext, const bf16x8 (&wc)[NTW][2], bf16x8 (&wn_)[NTW][2]) {
            constexpr int J = decltype(jc)::value;
            constexpr int DY = J % 3;
            const int k = s * 9 + J;
            bool a_issued = false;
            if constexpr (J < CG_PIECES && !(DBG & 16)) {
                if (s + 1 < nA) a_issued = stage_a_piece(std::integral_constant<int, (J < CG_PIECES ? J : 0)>{}, fnext, (s + 1) & 1);
            }
            wload(wn_, k + 2);
            reads_a(jc, s);
            // weights of this interval were issued two intervals ago; younger: 2 x 4 weight loads + the halo pieces of
            // this and the previous interval
            {
                const int n = 8 + (a_issued ? 1 : 0) + (a_prev3 ? 1 : 0);
                if (n == 10) cg_wait_vmcnt<10>(); else if (n == 9) cg_wait_vmcnt<9>(); else cg_wait_vmcnt<8>();
            }
            a_prev3 = a_issued;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (grp == 0) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
            if constexpr (DY == 0) mfma_w(wc, ar0, ar1, ar2, ar3);
            else if constexpr (DY == 1) mfma_w(wc, ar1, ar2, ar3, ar4);
            else mfma_w(wc, ar2, ar3, ar4, ar5);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (J == 8) {
                // the halo of step s+1 (issued in intervals 0..4, before the last 8 weight loads) must have landed for
                // every wave, and every wave must be done reading this step's buffer before it is refilled
                // (last step: drain the hand-issued weight loads -- hipcc does not know they are in flight and would
                // reuse their destination registers in the epilogue while the data is still on its way)
                if (s + 1 < nA) cg_wait_vmcnt<8>(); else cg_wait_vmcnt<0>();
                if constexpr (!(DBG & 8)) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        {
            const char* f0 = frame_ptr(0);
            stage_a_piece(std::integral_constant<int, 0>{}, f0, 0);
            stage_a_piece(std::integral_constant<int, 1>{}, f0, 0);
            stage_a_piece(std::integral_constant<int, 2>{}, f0, 0);
            stage_a_piece(std::integral_constant<int, 3>{}, f0, 0);
            stage_a_piece(std::integral_constant<int, 4>{}, f0, 0);
            if constexpr (CG_PIECES > 5) stage_a_piece(std::integral_constant<int, (CG_PIECES > 5 ? 5 : 0)>{}, f0, 0);
            static_assert(CG_PIECES <= 6, "prologue stages at most six pieces");
            if constexpr ((DBG & 16) != 0) {          // measurement: both halo buffers hold real data, no staging in the loop
                stage_a_piece(std::integral_constant<int, 0>{}, f0, 1);
                stage_a_piece(std::integral_constant<int, 1>{}, f0, 1);
                stage_a_piece(std::integral_constant<int, 2>{}, f0, 1);
                stage_a_piece(std::integral_constant<int, 3>{}, f0, 1);
                stage_a_piece(std::integral_constant<int, 4>{}, f0, 1);
                if constexpr (CG_PIECES > 5) stage_a_piece(std::integral_constant<int, (CG_PIECES > 5 ? 5 : 0)>{}, f0, 1);
            }
            wload(w0, 0);
            wload(w1, 1);
            cg_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef SVR_ABLATIONS
        if constexpr ((DBG & 256) != 0) { if (tid == 0 && blockIdx.x < 4096) g_conv_tl[blockIdx.x][1] = __builtin_amdgcn_s_memtime(); }
#endif
        for (int s = 0; s < nA; ++s) {
            const char* fnext = frame_ptr((DBG & 128) ? 0 : min(s + 1, nA - 1));     // (128: always the same, cache-hot halo)
            interval3(std::integral_constant<int, 0>{}, s, fnext, w0, w2);
            interval3(std::integral_constant<int, 1>{}, s, fnext, w1, w0);
            interval3(std::integral_constant<int, 2>{}, s, fnext, w2, w1);
            interval3(std::integral_constant<int, 3>{}, s, fnext, w0, w2);
            interval3(std::integral_constant<int, 4>{}, s, fnext, w1, w0);
            interval3(std::integral_constant<int, 5>{}, s, fnext, w2, w1);
            interval3(std::integral_constant<int, 6>{}, s, fnext, w0, w2);
            interval3(std::integral_constant<int, 7>{}, s, fnext, w1, w0);
            interval3(std::integral_constant<int, 8>{}, s, fnext, w2, w1);
        }
        }
    } else {
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

    }

#ifdef SVR_ABLATIONS
    if constexpr ((DBG & 256) != 0) { if (tid == 0 && blockIdx.x < 4096) g_conv_tl[blockIdx.x][2] = __builtin_amdgcn_s_memtime(); }
#endif
    // ---- epilogue through LDS, two passes of TY / 2 patch rows (every wave parks two of its four rows per pass): the fp32
    // tile is parked in LDS [voxels][132 floats] and written back row-contiguous, bias / residual added on the way out
    // (16 lanes cover one voxel's 128 couts; every global access is a full 16-byte lane / 256-byte row).
    constexpr int EP_PITCH = 528;                         // 128 floats + 16 B pad: conflict-free b128 writes
    const int hi4 = hi * 4;
    // bias of the 8 couts this thread stores (every store iteration has the same chunk tid & 15): loaded now, used after
    // the first barrier -- the latency hides under the LDS write phase
    f32x4 bias_lo = {0.f, 0.f, 0.f, 0.f}, bias_hi = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
        bias_lo = *(const f32x4*)(a.bias + n0 + (tid & 15) * 8);
        bias_hi = *(const f32x4*)(a.bias + n0 + (tid & 15) * 8 + 4);
    }
#ifdef SVR_ABLATIONS
#define SVR_EP_STAMP(i) if constexpr ((DBG & 256) != 0) { if (tid == 0 && blockIdx.x < 4096) g_conv_ep[blockIdx.x][i] = __builtin_amdgcn_s_memtime(); }
#else
#define SVR_EP_STAMP(i)
#endif
    SVR_EP_STAMP(0)
    const bool resid_gate = a.epilogue == SVR_EPI_RESID_GATE;
    // fused GroupNorm statistics of the stored (bf16-rounded, or fp32) output: this thread always stores the same
    // 8-cout chunk (tid & 15), so it keeps two quad sums over its 16 voxels
    float gs0 = 0.f, gq0 = 0.f, gs1 = 0.f, gq1 = 0.f;
    constexpr int EP_ROWS = G::EP_ROWS;                             // patch rows per pass
    constexpr int NPASS = MTW / 2;                                  // every wave parks two of its MTW rows per pass
    static_assert(TY / EP_ROWS == NPASS && EP_ROWS * 32 * 16 == 8 * NT, "MTW / 2 passes, eight store iterations each");
    // The body is instantiated once per option set of the production calls (bf16 output, no SiLU, no gate; residual and
    // fused statistics on / off) so that the loops carry no per-element option branches -- a taken scalar branch costs a
    // wave ~40 cycles of instruction refetch, and with one per LDS write / five per store iteration they made up a third
    // of the epilogue (s_memtime).  F < 0: every option decided at run time (fp32 output, SiLU, gate: tests, DiT-style use).
    auto ep_body = [&](auto fc) {
        constexpr int F = decltype(fc)::value;
        constexpr bool RT = F < 0, F_RESID = !RT && (F & 1), F_GN = !RT && (F & 2);
        const bool with_resid = RT ? (resid_gate && a.resid != nullptr) : F_RESID;
        // wide residual trunk: the compiled option sets store / read it as h16 (F & 4 output, F & 8 residual; ABI v6 -- round 3's
        // fp32 trunk cost 2.4 % of the step for the same parity); fp32 tensors take the run-time body.  The fused statistics are
        // those of the values as stored.
        const int ko = RT ? a.out_f32 : ((F & 4) != 0 ? SVR_STORE_H16 : SVR_STORE_BF16);
        const int kr = RT ? a.resid_f32 : ((F & 8) != 0 ? SVR_STORE_H16 : SVR_STORE_BF16);
        const bool o32 = ko == SVR_STORE_FP32, r32 = kr == SVR_STORE_FP32;
        const bool with_gn = RT ? a.gn_partial != nullptr : F_GN;
        // 8-row kernel, compiled option sets with a (bf16 / h16) residual: the residual chunks of a pass are loaded ONE PASS AHEAD --
        // pass 0's before the first parking, pass p + 1's right behind pass p's barrier -- instead of inside the store sweeps that
        // consume them, so their latency runs under the LDS parking, the barrier and the previous pass's sweeps.  Same values, same
        // arithmetic: bit-identical, -1.9 % on the 128 -> 128 conv2 form, -0.8 % at 256 / 512 channels (same box, standalone harness:
        // profiles/r4_conv_epilogue_resid_prefetch_ab.txt).  64 more registers in the epilogue (456 of 512): the kernels that run two
        // workgroups per CU keep the loads inside their sweeps.
        constexpr bool PRE = W8 && !RT && F_RESID;
        uint4 rpre[2][8];
        auto resid_prefetch = [&](auto pc, uint4 (&dst)[8]) {
            constexpr int P = decltype(pc)::value;
            const int n_ = n0 + (tid & 15) * 8;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int vox = (q * NT + tid) >> 4;
                const int r = vox >> 5;
                const int y = y0 + (r >> 1) * MTW + 2 * P + (r & 1), x = x0 + (vox & 31);
                const int64_t m_ = ((int64_t)to * g.H + min(y, g.H - 1)) * g.W + min(x, g.W - 1);
                dst[q] = *(const uint4*)((const bf16_t*)a.resid + m_ * a.ldr + n_);
            }
        };
        if constexpr (PRE) resid_prefetch(std::integral_constant<int, 0>{}, rpre[0]);
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            // every wave parks two of its MTW rows per pass (all waves write in every pass): LDS slot row wm * 2 + j
            // holds patch row wm * MTW + 2 * pass + j
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                char* row = smem + ((wm * 2 + j) * 32 + l31) * EP_PITCH;
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const f32x16_t v = acc[2 * pass + j][nt];
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const f32x4 o = {v[4 * gq], v[4 * gq + 1], v[4 * gq + 2], v[4 * gq + 3]};
                        *(f32x4*)(row + (wn * 64 + nt * 32 + 8 * gq + hi4) * 4) = o;
                    }
                }
            }
            SVR_EP_STAMP(1 + 3 * pass)                     // LDS writes issued
            if constexpr ((DBG & 512) != 0) __syncthreads(); else lds_barrier();      // (512: round 2's barriers, for the A/B)
            SVR_EP_STAMP(2 + 3 * pass)                     // barrier passed
            if constexpr (PRE) {
                if (pass + 1 < NPASS) {
                    if (pass == 0) resid_prefetch(std::integral_constant<int, 1>{}, rpre[1]);
                    else if (pass == 1) resid_prefetch(std::integral_constant<int, 2>{}, rpre[0]);
                    else if (pass == 2) resid_prefetch(std::integral_constant<int, 3>{}, rpre[1]);
                }
            }
            // store side, branch-free sweeps of four iterations so their LDS reads and residual loads are in flight
            // together (out-of-image voxels read a clamped address and are masked at the store)
            const int n = n0 + (tid & 15) * 8;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
            f32x4 lo[4], hi_[4];
            uint4 rr8[4];
            f32x4 rf0[4], rf1[4];
            int64_t mrow[4];
            bool ok[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int vox = ((half * 4 + it) * NT + tid) >> 4;      // voxel slot of this pass
                const int r = vox >> 5;
                const int y = y0 + (r >> 1) * MTW + 2 * pass + (r & 1), x = x0 + (vox & 31);
                ok[it] = y < g.H && x < g.W;
                if constexpr ((DBG & 4) != 0) ok[it] = false;
                mrow[it] = ((int64_t)to * g.H + min(y, g.H - 1)) * g.W + min(x, g.W - 1);
                lo[it] = *(const f32x4*)(smem + vox * EP_PITCH + (tid & 15) * 32);
                hi_[it] = *(const f32x4*)(smem + vox * EP_PITCH + (tid & 15) * 32 + 16);
            }
            if (with_resid) {
                if constexpr (RT) {                       // (run-time option set: tests / rare epilogues -- fp32 residuals are
                    if (!r32) {                           //  loaded where they are used instead of four iterations ahead)
#pragma unroll
                        for (int it = 0; it < 4; ++it) rr8[it] = *(const uint4*)((const bf16_t*)a.resid + mrow[it] * a.ldr + n);
                    }
                } else if (r32) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const float* rp = (const float*)a.resid + mrow[it] * a.ldr + n;
                        rf0[it] = *(const f32x4*)rp;
                        rf1[it] = *(const f32x4*)(rp + 4);
                    }
                } else {
                    if constexpr (PRE) {
#pragma unroll
                        for (int it = 0; it < 4; ++it) rr8[it] = rpre[pass & 1][half * 4 + it];
                    } else {
#pragma unroll
                        for (int it = 0; it < 4; ++it) rr8[it] = *(const uint4*)((const bf16_t*)a.resid + mrow[it] * a.ldr + n);
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                float f[8] = {lo[it][0] + bias_lo[0], lo[it][1] + bias_lo[1], lo[it][2] + bias_lo[2], lo[it][3] + bias_lo[3],
                              hi_[it][0] + bias_hi[0], hi_[it][1] + bias_hi[1], hi_[it][2] + bias_hi[2], hi_[it][3] + bias_hi[3]};
                if constexpr (RT) {
                    if (a.epilogue == SVR_EPI_BIAS_SILU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = silu(f[e]);
                    }
                    if (resid_gate && a.gate) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] *= a.gate[n + e];
                    }
                }
                if (with_resid) {
                    float r8[8];
                    if (r32) {
                        if constexpr (RT) {
                            const float* rp = (const float*)a.resid + mrow[it] * a.ldr + n;
                            rf0[0] = *(const f32x4*)rp;
                            rf1[0] = *(const f32x4*)(rp + 4);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) { r8[e] = rf0[RT ? 0 : it][e]; r8[4 + e] = rf1[RT ? 0 : it][e]; }
                    } else if (kr == SVR_STORE_H16) {
                        unpack8h(rr8[it], r8);
                    } else {
                        unpack8(rr8[it], r8);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += r8[e];
                }
                if (o32) {
                    if (ok[it]) {
                        float* cp = (float*)a.C + mrow[it] * a.ldc + n;
                        *(float4*)cp = make_float4(f[0], f[1], f[2], f[3]);
                        *(float4*)(cp + 4) = make_float4(f[4], f[5], f[6], f[7]);
                        if (with_gn) {
                            gs0 += f[0] + f[1] + f[2] + f[3];
                            gq0 += f[0] * f[0] + f[1] * f[1] + f[2] * f[2] + f[3] * f[3];
                            gs1 += f[4] + f[5] + f[6] + f[7];
                            gq1 += f[4] * f[4] + f[5] * f[5] + f[6] * f[6] + f[7] * f[7];
                        }
                    }
                } else {
                    const uint4 pk = ko == SVR_STORE_H16 ? pack8h(f) : pack8(f);
                    if (ok[it]) *(uint4*)((bf16_t*)a.C + mrow[it] * a.ldc + n) = pk;
                    if (with_gn && ok[it]) {
                        float r[8];
                        if (ko == SVR_STORE_H16) unpack8h(pk, r); else unpack8(pk, r);
                        gs0 += r[0] + r[1] + r[2] + r[3];
                        gq0 += r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
                        gs1 += r[4] + r[5] + r[6] + r[7];
                        gq1 += r[4] * r[4] + r[5] * r[5] + r[6] * r[6] + r[7] * r[7];
                    }
                }
            }
            }
            SVR_EP_STAMP(3 + 3 * pass)                     // stores issued
            if (pass + 1 < NPASS) { if constexpr ((DBG & 512) != 0) __syncthreads(); else lds_barrier(); }
        }
    };
    // (measurement builds' DBG variants: the round-2 option sets only -- they are never launched with a wide trunk)
    const bool wr = resid_gate && a.resid != nullptr;
    if (a.epilogue != SVR_EPI_BIAS_SILU && a.gate == nullptr && a.out_f32 != SVR_STORE_FP32 && !(wr && a.resid_f32 == SVR_STORE_FP32) &&
        (DBG == 0 || (!a.out_f32 && !a.resid_f32))) {
        const int F = (wr ? 1 : 0) | (a.gn_partial != nullptr ? 2 : 0) | (a.out_f32 == SVR_STORE_H16 ? 4 : 0) |
                      (wr && a.resid_f32 == SVR_STORE_H16 ? 8 : 0);
        switch (F) {
#define SVR_EP_CASE(v) case v: ep_body(std::integral_constant<int, v>{}); break;
            SVR_EP_CASE(0) SVR_EP_CASE(1) SVR_EP_CASE(2) SVR_EP_CASE(3)
            default:
                if constexpr (DBG == 0) switch (F) {
                    SVR_EP_CASE(4) SVR_EP_CASE(5) SVR_EP_CASE(6) SVR_EP_CASE(7) SVR_EP_CASE(9) SVR_EP_CASE(11) SVR_EP_CASE(13) SVR_EP_CASE(15)
                    default: break;
                }
                break;
#undef SVR_EP_CASE
        }
    } else {
        ep_body(std::integral_constant<int, -1>{});
    }
    if (a.gn_partial) {                                   // fixed-order reduction: thread -> quad -> group
        __syncthreads();
        float4* red = (float4*)smem;                      // [NT]
        double2* qsum = (double2*)(smem + 8192);          // [32 quads]
        red[tid] = make_float4(gs0, gq0, gs1, gq1);
        __syncthreads();
        if (tid < 32) {                                   // quad = 2 * chunk + half; rows tid' with tid' & 15 == chunk
            const int c = tid >> 1, h = tid & 1;
            double s = 0.0, q = 0.0;
            for (int j = 0; j < NT / 16; ++j) {
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
#ifdef SVR_ABLATIONS
    if constexpr ((DBG & 256) != 0) {     // (no wait for the stores: when the last store has been ISSUED)
        if (tid == 0 && blockIdx.x < 4096) g_conv_tl[blockIdx.x][3] = __builtin_amdgcn_s_memtime();
    }
#endif
}

int g_conv_band = 1;       // tile rows per band of the frame-inner tile order (0: frame outermost); svr_set_option("conv_band")
int g_conv_rows = 8;   // rows per wave of the register-streamed kernel: 4 | 8
int g_conv_lds_dbg = 0;    // measurement knob: dynamic LDS bytes to request (forces one workgroup per CU when > 80 KiB)
static bool conv_halo2_wreg(const svr_gemm_args& a) { return a.W_frag != nullptr && g_conv_impl == 0; }

template <int TY, int MODE, int DBG = 0> static int launch_conv_halo2_t(const svr_gemm_args& a, hipStream_t s) {
    typedef cg_geom<TY, MODE> G;
    const svr_conv_geom& g = a.conv;
    const int tiles = g.To * ((g.H + TY - 1) / TY) * ((g.W + CG_TX - 1) / CG_TX) * (a.N / 128);
    static uint64_t lds_attr_done = 0;               // per device (svr_common.h)
    {
        const int e = set_max_dynamic_lds((const void*)conv_halo2_kernel<TY, MODE, DBG>, 160 * 1024, lds_attr_done);
        if (e != 0) return e;
    }
    hipLaunchKernelGGL((conv_halo2_kernel<TY, MODE, DBG>), dim3(tiles), dim3(G::NT), g_conv_lds_dbg > G::LDS ? g_conv_lds_dbg : G::LDS, s, a,
                       g_conv_band);
    return (int)hipGetLastError();
}

static int launch_conv_halo2(const svr_gemm_args& a, hipStream_t s) {
#ifdef SVR_ABLATIONS
#include "measure/svr_conv_halo2_measure_1.inc"
#endif
    if (conv_halo2_wreg(a)) return g_conv_rows == 8 ? launch_conv_halo2_t<16, 3>(a, s) : launch_conv_halo2_t<8, 1>(a, s);
    return launch_conv_halo2_t<16, 0>(a, s);
}

// thin-input variant: Cin = 4 (RGB padded), 3x3 spatial taps, stride 1, the whole K in one 128-wide image
static bool conv_thin_eligible(const svr_gemm_args& a) {
    const svr_conv_geom& g = a.conv;
    return g.enabled && g.Cin == 4 && g.kh == 3 && g.kw == 3 && g.sh == 1 && g.sw == 1 && g.st == 1 &&
           g.ph == 1 && g.pw == 1 && g.Ho == g.H && g.Wo == g.W && g.kt >= 1 && g.kt <= 3 && a.K == 128 &&
           g.To == g.T + g.pt - g.kt + 1 && !a.ps.enabled && a.epilogue != SVR_EPI_SWIGLU && (a.N % 128) == 0 &&
           (a.ldc % 8) == 0 && (!a.resid || (a.ldr % 8) == 0);
}
static int launch_conv_thin(const svr_gemm_args& a, hipStream_t s) { return launch_conv_halo2_t<8, 2>(a, s); }

static int conv_gn_blocks(const svr_gemm_args& a) {
    if (a.gn_groups > 0 && conv_sub_eligible(a)) return conv_sub_gn_blocks(a);
    if (conv_thin_eligible(a) && a.gn_groups > 0) {
        const int cpg = a.N / a.gn_groups;
        if (cpg < 4 || (cpg & 3) || a.N % a.gn_groups || 128 % cpg) return 0;
        return ((a.conv.H + 7) / 8) * ((a.conv.W + CG_TX - 1) / CG_TX);
    }
    if ((g_conv_impl != 0 && g_conv_impl != 3) || !conv_halo_eligible(a) || (a.N % 128) != 0 || a.conv.Cin % 32 != 0) return 0;
    const int cpg = a.gn_groups > 0 ? a.N / a.gn_groups : 0;          // channels per group: 4, 8 or 16
    if (cpg < 4 || (cpg & 3) || a.N % a.gn_groups || 128 % cpg) return 0;
    const int ty = conv_halo2_wreg(a) && g_conv_rows != 8 ? 8 : 16;
    return ((a.conv.H + ty - 1) / ty) * ((a.conv.W + CG_TX - 1) / CG_TX);
}

// stride-1 "same" 3x3 spatial kernel (1 or 3 temporal taps), plain bias / residual epilogue: the geometry of the LDS-halo kernels
// (N <= 32: the thin-output kernel, svr_conv_thinout.hip; N % 128 == 0: conv_halo2_kernel)
static bool conv_halo_eligible(const svr_gemm_args& a) {
    const svr_conv_geom& g = a.conv;
    return g.enabled && g.kh == 3 && g.kw == 3 && g.sh == 1 && g.sw == 1 && g.st == 1 && g.ph == 1 && g.pw == 1 &&
           g.Ho == g.H && g.Wo == g.W && g.Cin % 64 == 0 && g.kt >= 1 && g.kt <= 3 &&
           g.To == g.T + g.pt - g.kt + 1 && !a.ps.enabled && !a.phase.enabled && a.epilogue != SVR_EPI_SWIGLU &&
           (a.N <= 32 || ((a.N % 128) == 0 && (a.ldc % 8) == 0 && (!a.resid || (a.ldr % 8) == 0))) &&
           (int64_t)g.H * g.W * g.Cin * 2 < (int64_t)1 << 32;
}

// what conv_halo_eligible() accepts with 128-cout tiles and channels in 32-slices
static bool conv_halo2_eligible(const svr_gemm_args& a) {
    return conv_halo_eligible(a) && (a.N % 128) == 0 && a.conv.Cin % 32 == 0;
}

// ------------------------------------------------------------------------------------------------
// Weight re-pack for the WREG variant: W [N, K = (dt, dy, dx, c)] -> [N / 32][k = (dt * Cin/32 + slice) * 9 + dx * 3 + dy]
// [k-step 2][lane 64][8 bf16]; lane l of a v_mfma_f32_32x32x16_bf16 operand holds cout (l & 31), k (l >> 5) * 8 .. + 8.
// One thread per 16-byte chunk; done once per checkpoint.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_pack_frag_kernel(const bf16_t* __restrict__ W, uint4* __restrict__ out, int N, int K,
                                                             int kt, int Cin) {
    const int cpk = Cin / 32, P = kt * cpk * 9;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)(N / 32) * P * 128) return;
    const int lane = (int)(idx & 63), ks = (int)((idx >> 6) & 1);
    const int kk = (int)((idx >> 7) % P), n32 = (int)((idx >> 7) / P);
    const int s = kk / 9, J = kk - s * 9;
    const int dt = s / cpk, cs = s - dt * cpk, dy = J % 3, dx = J / 3;
    const int n = n32 * 32 + (lane & 31), c = cs * 32 + ks * 16 + (lane >> 5) * 8;
    out[idx] = *(const uint4*)(W + (int64_t)n * K + ((dt * 9 + dy * 3 + dx) * Cin + c));
}

}  // namespace svr
