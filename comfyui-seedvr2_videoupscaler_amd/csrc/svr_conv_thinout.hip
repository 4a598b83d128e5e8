// Thin-output 3x3 causal conv for gfx950: N <= 32 output channels (decoder conv_out 128 -> 3, encoder conv_out 512 -> 32) at
// full resolution.  Almost no FLOPs (the 32-cout MFMA tile is mostly padding) but a full read of the widest activations, so
// the budget is the input's HBM time; the first LDS-halo kernel (svr_conv_halo.hip, 32-cout variant) spent 37 k cycles per A
// step on it -- one weight unit and one workgroup barrier per TAP, each interval four MFMAs long and waiting out an L2 latency.
//
// Here the interval is the whole A step (temporal tap, 64-channel slice): the 10 x 34 halo of an 8 x 32 patch (42.5 KiB) AND
// the step's nine weight units (9 x 32 couts x 128 B = 36 KiB) are staged together by LDS-DMA into double buffers while the
// previous step computes; one `vmcnt(0)` + barrier per step.  Eight waves, one patch row each: 36 MFMAs
// (v_mfma_f32_32x32x16_bf16, couts x voxels) per wave and step straight out of LDS.  Halo and weight images use the
// first kernel's 128-byte rows and source-side XOR swizzles (conflict-free 32-row fragment reads at any column shift).
#include "svr_common.h"
#include "../../include/seedvr2_hip.h"
#include <type_traits>

namespace svr {

constexpr int CT_TY = 8, CT_TX = 32, CT_HX = CT_TX + 2, CT_HY = CT_TY + 2;
constexpr int CT_NT = 512;
constexpr int CT_AROWS = CT_HX * CT_HY;                  // 340 halo pixels, 128 B each
constexpr int CT_ACHUNKS = CT_AROWS * 8;                 // 2720 16-byte chunks
constexpr int CT_APIECES = (CT_ACHUNKS + CT_NT - 1) / CT_NT;   // 6 (the last one partial)
constexpr int CT_ABUF = CT_AROWS * 128;                  // 43 520 B
constexpr int CT_WROWS = 9 * 32;                         // (tap, cout) rows, 128 B each
constexpr int CT_WCHUNKS = CT_WROWS * 8;                 // 2304
constexpr int CT_WPIECES = (CT_WCHUNKS + CT_NT - 1) / CT_NT;   // 5 (the last one half)
constexpr int CT_WBUF = CT_WROWS * 128;                  // 36 864 B
constexpr int CT_STEP = CT_ABUF + CT_WBUF;               // one step's operands
constexpr int CT_LDS = 2 * CT_STEP;                      // 160 768 B
static_assert(CT_LDS <= 160 * 1024, "two steps of operands fit the LDS");

__global__ __launch_bounds__(CT_NT) void conv_thinout_kernel(const svr_gemm_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((ext_vector_type(16))) float f32x16_t;
    const svr_conv_geom& g = a.conv;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = patch row of this wave

    const int tiles_x = (g.W + CT_TX - 1) / CT_TX;
    const int tiles_y = (g.H + CT_TY - 1) / CT_TY;
    int tl;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, j = bid >> 3, q = nwg >> 3, r = nwg & 7;
        tl = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tx = tl % tiles_x;
    const int rr = tl / tiles_x;
    const int ty = rr % tiles_y;
    const int to = rr / tiles_y;
    const int y0 = ty * CT_TY, x0 = tx * CT_TX;

    const int cpk = g.Cin / 64;                           // 64-channel slices per temporal tap
    const int nA = g.kt * cpk;                            // A steps
    const int64_t frame_bytes = (int64_t)g.H * g.W * g.Cin * 2;

    // ---- staging roles: chunk id = piece * 512 + tid -> (row = id >> 3, position = id & 7); the source chunk is the position
    // XORed with the row's key (halo: (hx >> 1) & 7, weights: (row >> 1) & 7)
    const int srow = tid >> 3, spos = tid & 7;
    uint32_t poff[CT_APIECES];                            // halo piece -> pixel index (or ~0: outside the image)
    uint32_t akeys = 0;                                   // halo piece -> source chunk (3 bits each)
#pragma unroll
    for (int q = 0; q < CT_APIECES; ++q) {
        const int row = q * 64 + srow;
        const int hy = row / CT_HX, hx = row - hy * CT_HX;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool ok = (row < CT_AROWS) & ((unsigned)y < (unsigned)g.H) & ((unsigned)x < (unsigned)g.W);
        poff[q] = ok ? (uint32_t)(y * g.W + x) : 0xffffffffu;
        akeys |= (uint32_t)(spos ^ ((hx >> 1) & 7)) << (3 * q);
    }
    // weight piece q: row = q * 64 + srow = tap * 32 + cout; element offset of the thread's source chunk inside W for tap-slice 0
    int64_t woff[CT_WPIECES];
#pragma unroll
    for (int q = 0; q < CT_WPIECES; ++q) {
        const int row = q * 64 + srow;
        const int tap = row >> 5, n = row & 31;
        const int ck = spos ^ ((row >> 1) & 7);
        woff[q] = (int64_t)n * a.K + (int64_t)tap * g.Cin + ck * 8;
    }
    char* const wave_dst = smem + wave * 1024;

    auto stage_step = [&](int s, int buf) {
        const int dt = s / cpk;
        const int c0 = (s - dt * cpk) * 64;
        int f = to + dt - g.pt;
        const char* basep = (const char*)a.A;
        if (f < 0) {
            if (g.halo != nullptr) { basep = (const char*)g.halo; f += g.halo_frames; }
            else f = 0;
        }
        const char* fptr = basep + (int64_t)f * frame_bytes + c0 * 2;
        char* dst = wave_dst + buf * CT_STEP;
#pragma unroll
        for (int q = 0; q < CT_APIECES; ++q) {
            const int ck = (akeys >> (3 * q)) & 7;
            const char* src = poff[q] == 0xffffffffu ? (const char*)g.zeros : fptr + ((int64_t)poff[q] * g.Cin + ck * 8) * 2;
            if (q * CT_NT + tid < CT_ACHUNKS) glds16(src, dst + q * 8192);
        }
        const bf16_t* wsl = (const bf16_t*)a.W + (int64_t)dt * 9 * g.Cin + c0;
#pragma unroll
        for (int q = 0; q < CT_WPIECES; ++q)
            if (q * CT_NT + tid < CT_WCHUNKS) glds16(wsl + woff[q], dst + CT_ABUF + q * 8192);
    };

    // ---- fragment addressing (lane constants): halo pixel (wave + dy, dx + l31), weight row tap * 32 + l31; k-step ks reads
    // chunk 2 ks + hi
    const int l31 = lane & 31, hi = lane >> 5;
    int rd_a[3][4], rd_b[4];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            rd_a[dx][ks] = (wave * CT_HX + dx + l31) * 128 + (((2 * ks + hi) ^ (((dx + l31) >> 1) & 7)) << 4);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        rd_b[ks] = CT_ABUF + l31 * 128 + (((2 * ks + hi) ^ ((l31 >> 1) & 7)) << 4);

    f32x16_t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    // Fragment reads are issued from inline asm: hipcc puts `s_waitcnt vmcnt(0)` in front of every plain LDS load while an
    // LDS-DMA is in flight (the next step's operands -- it would serialise the double buffer) and schedules read -> wait ->
    // MFMA one fragment at a time; here the eight reads of tap t + 1 are in flight under the four MFMAs of tap t.
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    bf16x8 hA[4], wA[4], hB[4], wB[4];
    unsigned ra[3][4], rb[4];                             // this step's fragment addresses
#define CT_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
#define CT_ISSUE(TAP, H, W_) \
    CT_RD(H[0], ra[(TAP) % 3][0], ((TAP) / 3) * (CT_HX * 128)); CT_RD(W_[0], rb[0], (TAP) * (32 * 128)); \
    CT_RD(H[1], ra[(TAP) % 3][1], ((TAP) / 3) * (CT_HX * 128)); CT_RD(W_[1], rb[1], (TAP) * (32 * 128)); \
    CT_RD(H[2], ra[(TAP) % 3][2], ((TAP) / 3) * (CT_HX * 128)); CT_RD(W_[2], rb[2], (TAP) * (32 * 128)); \
    CT_RD(H[3], ra[(TAP) % 3][3], ((TAP) / 3) * (CT_HX * 128)); CT_RD(W_[3], rb[3], (TAP) * (32 * 128))
    auto mm = [&](bf16x8 (&h)[4], bf16x8 (&w)[4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[ks], h[ks], acc, 0, 0, 0);
    };
#define CT_WAIT(N, H, W_) \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(H[0]), "+v"(H[1]), "+v"(H[2]), "+v"(H[3]), "+v"(W_[0]), "+v"(W_[1]), "+v"(W_[2]), "+v"(W_[3])); \
    __builtin_amdgcn_sched_barrier(0)

    stage_step(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s < nA; ++s) {
        if (s + 1 < nA) stage_step(s + 1, (s + 1) & 1);
        const unsigned bufoff = lds0 + (unsigned)((s & 1) * CT_STEP);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ra[dx][ks] = bufoff + (unsigned)rd_a[dx][ks];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) rb[ks] = bufoff + (unsigned)rd_b[ks];
        __builtin_amdgcn_sched_barrier(0);
        CT_ISSUE(0, hA, wA);
        CT_ISSUE(1, hB, wB); CT_WAIT(8, hA, wA); mm(hA, wA); __builtin_amdgcn_sched_barrier(0);
        CT_ISSUE(2, hA, wA); CT_WAIT(8, hB, wB); mm(hB, wB); __builtin_amdgcn_sched_barrier(0);
        CT_ISSUE(3, hB, wB); CT_WAIT(8, hA, wA); mm(hA, wA); __builtin_amdgcn_sched_barrier(0);
        CT_ISSUE(4, hA, wA); CT_WAIT(8, hB, wB); mm(hB, wB); __builtin_amdgcn_sched_barrier(0);
        CT_ISSUE(5, hB, wB); CT_WAIT(8, hA, wA); mm(hA, wA); __builtin_amdgcn_sched_barrier(0);
        CT_ISSUE(6, hA, wA); CT_WAIT(8, hB, wB); mm(hB, wB); __builtin_amdgcn_sched_barrier(0);
        CT_ISSUE(7, hB, wB); CT_WAIT(8, hA, wA); mm(hA, wA); __builtin_amdgcn_sched_barrier(0);
        CT_ISSUE(8, hA, wA); CT_WAIT(8, hB, wB); mm(hB, wB); __builtin_amdgcn_sched_barrier(0);
        CT_WAIT(0, hA, wA); mm(hA, wA);
        __builtin_amdgcn_sched_barrier(0);
        // next step landed (this wave's share: vmcnt(0)) + everyone done with this buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
#undef CT_WAIT
#undef CT_ISSUE
#undef CT_RD

    // ---- epilogue (tiny byte volume): lane holds C[voxel = lane & 31][cout = 8 q + 4 (lane >> 5) + 0..3]
    const int y = y0 + wave, x = x0 + l31;
    if (y < g.H && x < g.W) {
        const int m = (to * g.H + y) * g.W + x;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int n = 8 * gq + hi * 4;
            const f32x4 accv = {acc[4 * gq], acc[4 * gq + 1], acc[4 * gq + 2], acc[4 * gq + 3]};
            if (n < a.N) epilogue_store(a, accv, accv, m, n);
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// The N <= 4 form (decoder conv_out, 128 -> 3 at full resolution: 0.8 % of the BASELINE config 3 step in round 4).  The kernel above
// is LDS-bound -- eight waves of ONE patch row each read 72 KiB of fragments per step, every wave its own copy of the step's nine
// weight units: 576 KiB per step and CU = 4608 cycles at 128 B/clk for 2304 cycles of (mostly padding) MFMA work -- and behind
// that it is bound by its staging: 3.6 staged bytes per input byte (halo x three temporal taps) at almost no arithmetic.  Here:
//   * v_mfma_f32_16x16x32_bf16 with the couts as the 16-row operand, and only FOUR cout rows per (channel slice, temporal tap, tap)
//     unit in LDS (256 B): lanes 4..15 of the operand read the rows of lanes 0..3 again -- their accumulator rows are never stored,
//     and a row of D depends on its own row of the operand only -- so ALL weights of the conv (Cin x kt x 9 x 4 x 2 B = 27 KiB for
//     128 x 3) are staged ONCE per workgroup and stay resident;
//   * four waves, FOUR patch rows each (16 x 32 patch): a wave's weight fragments serve 8 MFMAs each, its six halo rows are
//     shared by the three vertical taps;
//   * FRAME STREAMING: a workgroup keeps its patch and walks the INPUT frames of its chunk of the clip; a staged halo
//     (input frame f, 32-channel slice) is multiplied with the weights of ALL temporal taps and accumulated into the up to
//     three output frames it belongs to (to = f - dt + pt; three accumulator sets of 8 f32x4 in a static ring) -- the thin
//     output is what makes three live output frames affordable, and the staging drops from 3.6 to 1.2 bytes per input byte;
//   * a THREE-deep halo ring (18 x 34 pixels x 64 B per step): steps k + 1 and k + 2 are in flight under the MFMAs of step k
//     (`vmcnt(10)`: every wave stages whole pieces, the tail of the tenth reads the zero page, so the count is a constant), one
//     barrier per step.  Round 4's double buffer -- one step, 17 MB on the chip, in flight when a step starts and nothing when it
//     ends -- was half of what the HBM latency needs;
//   * per column shift dx: 12 halo + 3 weight fragment reads (15 = the most lgkmcnt can count) issued under the MFMAs of the
//     previous shift, the other temporal taps' 6 weight fragments under the first 24 MFMAs of their shift;
//   * the step body exists twice: without branches when all three temporal taps are live (all input frames but the first and
//     last kt - 1 of a chunk) -- behind a wave-uniform branch hipcc copies the 32 accumulators of the not-taken side, 9 x 32
//     v_accvgpr_mov per step with one wave per SIMD -- and with them for the head and tail frames;
//   * a completed output frame leaves through LDS: the MFMA layout puts one voxel's N <= 4 couts in one lane, i.e. N scalar stores of
//     2 or 4 bytes per voxel at a stride of N elements -- 600 of the 1 680 us of the 9 x 1024^2 launch were those partial writes.
//     Each wave parks its four patch rows (32 voxels x N couts, contiguous in the [T, H, W, N] output) in a private 2 KiB of LDS
//     and writes them back one element per lane: full lines.  Taken for the plain bias epilogue into a dense output (conv_out);
//     anything else goes through epilogue_store().
// Round 5 measurements (profiles/r5_conv_thinout4_ab.txt): 1 885 -> 1 131 us on 9 x 1024^2 x 128 -> 3, bit-identical to round 4's
// kernel (N <= 16 couts per unit, weights re-staged with every halo, double buffer, scalar stores), which it replaces.
// Same swizzle of the halo image as the LDS-halo kernels (64-byte rows, chunk c at position c ^ ((hx >> 2) & 3)).
// ------------------------------------------------------------------------------------------------------------------------
constexpr int C4_TY = 16, C4_TX = 32, C4_HX = C4_TX + 2, C4_HY = C4_TY + 2;
constexpr int C4_NT = 256;
constexpr int C4_AROWS = C4_HX * C4_HY;                  // 612 halo pixels, 64 B each
constexpr int C4_ACHUNKS = C4_AROWS * 4;                 // 2448 16-byte chunks
constexpr int C4_APIECES = (C4_ACHUNKS + C4_NT - 1) / C4_NT;   // 10 (the tail of the last one reads the zero page)
constexpr int C4_ABUF = C4_APIECES * C4_NT * 16;         // 40 960 B: ten whole pieces
constexpr int C4_WOFF = 3 * C4_ABUF;                     // 122 880: the resident weights follow the ring
constexpr int C4_FLUSH = 4 * 4 * 128 * 4;                // 8 KiB: per wave four patch rows x (32 voxels x <= 4 couts) dwords
static_assert(C4_APIECES == 10 && C4_ABUF >= C4_AROWS * 64, "piece schedule: vmcnt(10) names one step's staging");

// DBG (builds with -DSVR_ABLATIONS only; results invalid): 1 no halo staging after the first two steps, 4 no fragment reads, 8 no global
// stores (2: no MFMAs -- not launched: the accumulators then travel through VGPRs around the empty asm and the time says nothing)
template <int DBG = 0>
__global__ __launch_bounds__(C4_NT, 1) void conv_thinout4_kernel(const svr_gemm_args a, const int frames_per_chunk, const int flush_off) {
#ifndef SVR_ABLATIONS
    static_assert(DBG == 0, "measurement variants (results invalid on purpose) exist only in -DSVR_ABLATIONS builds; the product library instantiates DBG = 0");
#endif

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const svr_conv_geom& g = a.conv;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // patch rows 4 wave .. 4 wave + 3

    const int tiles_x = (g.W + C4_TX - 1) / C4_TX;
    const int tiles_y = (g.H + C4_TY - 1) / C4_TY;
    int tl;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, j = bid >> 3, q = nwg >> 3, r = nwg & 7;
        tl = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tx = tl % tiles_x;
    const int rr = tl / tiles_x;
    const int ty = rr % tiles_y;
    const int chunk = rr / tiles_y;
    const int y0 = ty * C4_TY, x0 = tx * C4_TX;
    const int t0 = chunk * frames_per_chunk, t1 = min(g.To, t0 + frames_per_chunk);   // output frames of this workgroup

    const int kt = g.kt;
    const int cpk = g.Cin / 32;                           // 32-channel slices per input frame
    const int n_in = (t1 - t0) + kt - 1;                  // input frames walked: i = 0 .. n_in - 1 is frame f = t0 - pt + i
    const int nsteps = n_in * cpk;
    const int64_t frame_bytes = (int64_t)g.H * g.W * g.Cin * 2;
    if (nsteps <= 0) return;

    // ---- halo staging roles: chunk id = piece * 256 + tid -> (row = id >> 2, position = id & 3); source chunk = position ^ key(row).
    // Every thread stages all ten pieces: rows past the halo image or outside the frame read the zero page
    const int srow = tid >> 2, spos = tid & 3;
    uint32_t poff[C4_APIECES];
    uint32_t akeys = 0;
#pragma unroll
    for (int q = 0; q < C4_APIECES; ++q) {
        const int row = q * 64 + srow;
        const int hy = row / C4_HX, hx = row - hy * C4_HX;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool ok = (row < C4_AROWS) & ((unsigned)y < (unsigned)g.H) & ((unsigned)x < (unsigned)g.W);
        poff[q] = ok ? (uint32_t)(y * g.W + x) : 0xffffffffu;
        akeys |= (uint32_t)(spos ^ ((hx >> 2) & 3)) << (2 * q);
    }
    char* const wave_dst = smem + wave * 1024;

    auto stage_halo = [&](int k, int buf) {               // step k = (input frame i = k / cpk, channel slice k % cpk)
        const int i = k / cpk;
        const int c0 = (k - i * cpk) * 32;
        int f = t0 - g.pt + i;
        const char* basep = (const char*)a.A;
        if (f < 0) {
            if (g.halo != nullptr) { basep = (const char*)g.halo; f += g.halo_frames; }
            else f = 0;
        }
        const char* fptr = basep + (int64_t)f * frame_bytes + c0 * 2;
        char* dst = wave_dst + buf * C4_ABUF;
#pragma unroll
        for (int q = 0; q < C4_APIECES; ++q) {
            const int ck = (akeys >> (2 * q)) & 3;
            const char* src = poff[q] == 0xffffffffu ? (const char*)g.zeros : fptr + ((int64_t)poff[q] * g.Cin + ck * 8) * 2;
            glds16(src, dst + q * 4096);
        }
    };

    // ---- the resident weights: unit u = (slice * kt + dt) * 9 + tap holds cout rows 0..3 x 32 channels (4 x 64 B); LDS slot id =
    // piece * 256 + tid -> (u = id >> 4, n = (id >> 2) & 3, chunk = id & 3).  W is [n][K = (dt, dy, dx, c)], N <= 4 rows.
    {
        const int units = cpk * kt * 9;
        const int wpieces = (units * 16 + C4_NT - 1) / C4_NT;
        for (int p = 0; p < wpieces; ++p) {
            const int id = p * C4_NT + tid;
            const int u = id >> 4, n = (id >> 2) & 3, ck = id & 3;
            const int sl = u / (kt * 9), dtap = u - sl * (kt * 9);
            // (rows n >= N of the 4-row unit come from the zero page: W needs no row padding here, a [3, K] weight is read as is)
            const bf16_t* src = u < units && n < a.N ? (const bf16_t*)a.W + (int64_t)n * a.K + (int64_t)dtap * g.Cin + sl * 32 + ck * 8
                                                     : (const bf16_t*)g.zeros;
            glds16(src, wave_dst + C4_WOFF + p * 4096);
        }
    }

    // ---- fragment addressing (lane constants).  v_mfma_f32_16x16x32_bf16: lane (l15, kq) holds row / column l15, k = 8 kq .. 8 kq + 7.
    // halo fragment (row hy, column shift dx, half h): pixel hx = dx + 16 h + l15; weights: row (l15 & 3) of the unit, chunk kq
    const int l15 = lane & 15, kq = lane >> 4;
    int rd_a[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int hx = dx + 16 * h + l15;
            rd_a[dx][h] = (wave * 4 * C4_HX + hx) * 64 + ((kq ^ ((hx >> 2) & 3)) << 4);
        }
    const int rd_b = C4_WOFF + (l15 & 3) * 64 + kq * 16;
    const int wslice = kt * 9 * 256;                      // bytes of one channel slice's units

    f32x4 acc[3][4][2];                                   // [ring slot of the output frame][patch row][half]
#pragma unroll
    for (int sl = 0; sl < 3; ++sl)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[sl][r][h] = f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    bf16x8 hA[6][2], hB[6][2], w0A[3], w0B[3], w1[3], w2[3];
    unsigned ra[3][2], rb;
#define C4_RD(DST, ADDR, OFF) do { if constexpr ((DBG & 4) == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF) : "memory"); \
                                else asm volatile("" : "=v"(DST) : "v"(ADDR)); } while (0)
#define C4_ISSUE_H(DX, H, W_) \
    C4_RD(H[0][0], ra[DX][0], 0 * C4_HX * 64); C4_RD(H[0][1], ra[DX][1], 0 * C4_HX * 64); \
    C4_RD(H[1][0], ra[DX][0], 1 * C4_HX * 64); C4_RD(H[1][1], ra[DX][1], 1 * C4_HX * 64); \
    C4_RD(H[2][0], ra[DX][0], 2 * C4_HX * 64); C4_RD(H[2][1], ra[DX][1], 2 * C4_HX * 64); \
    C4_RD(W_[0], rb, (0 * 3 + DX) * 256); C4_RD(W_[1], rb, (1 * 3 + DX) * 256); C4_RD(W_[2], rb, (2 * 3 + DX) * 256); \
    C4_RD(H[3][0], ra[DX][0], 3 * C4_HX * 64); C4_RD(H[3][1], ra[DX][1], 3 * C4_HX * 64); \
    C4_RD(H[4][0], ra[DX][0], 4 * C4_HX * 64); C4_RD(H[4][1], ra[DX][1], 4 * C4_HX * 64); \
    C4_RD(H[5][0], ra[DX][0], 5 * C4_HX * 64); C4_RD(H[5][1], ra[DX][1], 5 * C4_HX * 64)
#define C4_ISSUE_W12(DX) \
    C4_RD(w1[0], rb, (9 + 0 * 3 + DX) * 256); C4_RD(w1[1], rb, (9 + 1 * 3 + DX) * 256); C4_RD(w1[2], rb, (9 + 2 * 3 + DX) * 256); \
    C4_RD(w2[0], rb, (18 + 0 * 3 + DX) * 256); C4_RD(w2[1], rb, (18 + 1 * 3 + DX) * 256); C4_RD(w2[2], rb, (18 + 2 * 3 + DX) * 256)
#define C4_LANDED_H(H, W_) \
    asm volatile("s_waitcnt lgkmcnt(0)" \
                 : "+v"(H[0][0]), "+v"(H[0][1]), "+v"(H[1][0]), "+v"(H[1][1]), "+v"(H[2][0]), "+v"(H[2][1]), "+v"(H[3][0]), "+v"(H[3][1]), \
                   "+v"(H[4][0]), "+v"(H[4][1]), "+v"(H[5][0]), "+v"(H[5][1]), "+v"(W_[0]), "+v"(W_[1]), "+v"(W_[2])); \
    __builtin_amdgcn_sched_barrier(0)
#define C4_LANDED_W12() \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w1[0]), "+v"(w1[1]), "+v"(w1[2]), "+v"(w2[0]), "+v"(w2[1]), "+v"(w2[2])); \
    __builtin_amdgcn_sched_barrier(0)
    auto mm = [&](auto slc, bf16x8 (&h)[6][2], bf16x8 (&w)[3]) {
        constexpr int SL = decltype(slc)::value;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    if constexpr ((DBG & 2) == 0)
                        acc[SL][r][hh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[dy], h[r + dy][hh], acc[SL][r][hh], 0, 0, 0);
                    else { f32x4& av = acc[SL][r][hh]; const bf16x8 wv = w[dy], hv = h[r + dy][hh]; asm volatile("" : "+v"(av) : "v"(wv), "v"(hv)); }
                }
    };
    // one step (input frame i with i % 3 == Q, channel slice c) out of ring buffer `buf`; ok0 / ok1 / ok2: does temporal tap dt land in
    // an output frame of this chunk (wave-uniform).  Output frame of tap dt: t0 + i - dt -> ring slot (Q - dt) mod 3.
    // ALL: every temporal tap lands in an output frame of this chunk (all input frames but the first and last kt - 1): no branches
    // around the MFMAs, so the accumulators stay in place (behind a branch hipcc copies all 32 of them on the not-taken side)
    auto step_body = [&](auto qc, auto allc, int buf, int c, bool ok0_, bool ok1_, bool ok2_) {
        constexpr int Q = decltype(qc)::value;
        constexpr bool ALL = decltype(allc)::value;
        const bool ok0 = ALL || ok0_, ok1 = ALL || ok1_, ok2 = ALL || ok2_;
        using S0 = std::integral_constant<int, Q % 3>;
        using S1 = std::integral_constant<int, (Q + 2) % 3>;
        using S2 = std::integral_constant<int, (Q + 1) % 3>;
        const unsigned bufoff = lds0 + (unsigned)(buf * C4_ABUF);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int h = 0; h < 2; ++h) ra[dx][h] = bufoff + (unsigned)rd_a[dx][h];
        rb = lds0 + (unsigned)(rd_b + c * wslice);
        __builtin_amdgcn_sched_barrier(0);
        const bool any12 = ok1 | ok2;
        C4_ISSUE_H(0, hA, w0A);
        C4_LANDED_H(hA, w0A);
        if (any12) { C4_ISSUE_W12(0); }
        if (ok0) mm(S0{}, hA, w0A);
        __builtin_amdgcn_sched_barrier(0);
        C4_LANDED_W12();
        C4_ISSUE_H(1, hB, w0B);
        if (ok1) mm(S1{}, hA, w1);
        if (ok2) mm(S2{}, hA, w2);
        __builtin_amdgcn_sched_barrier(0);
        C4_LANDED_H(hB, w0B);
        if (any12) { C4_ISSUE_W12(1); }
        if (ok0) mm(S0{}, hB, w0B);
        __builtin_amdgcn_sched_barrier(0);
        C4_LANDED_W12();
        C4_ISSUE_H(2, hA, w0A);
        if (ok1) mm(S1{}, hB, w1);
        if (ok2) mm(S2{}, hB, w2);
        __builtin_amdgcn_sched_barrier(0);
        C4_LANDED_H(hA, w0A);
        if (any12) { C4_ISSUE_W12(2); }
        if (ok0) mm(S0{}, hA, w0A);
        __builtin_amdgcn_sched_barrier(0);
        C4_LANDED_W12();
        if (ok1) mm(S1{}, hA, w1);
        if (ok2) mm(S2{}, hA, w2);
        __builtin_amdgcn_sched_barrier(0);
    };
    // dense [T, H, W, N] output behind the plain bias epilogue: through LDS (see the header); wave-uniform
    const bool dense = a.epilogue == SVR_EPI_BIAS && !a.ps.enabled && !a.phase.enabled && a.ldc == a.N;
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (dense && a.bias) {
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) if (ch < a.N) bias4[ch] = a.bias[ch];
    }
    const unsigned fl_base = lds0 + (unsigned)(flush_off + wave * 2048);
    const int row_px = min(C4_TX, g.W - x0);              // voxels of a patch row inside the image
    auto flush = [&](auto slc, int to) {
        constexpr int SL = decltype(slc)::value;
        if (dense) {
            const int N = a.N;
            const int kind = a.out_f32;
            // park: lane (l15, kq == 0) holds voxel 16 h + l15 of row r, couts 0..3 -> dword slot (r * 32 + voxel) * N + cout
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned slot = fl_base + (unsigned)(((r * 32 + 16 * h + l15) * N) << 2);
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch) {
                        const float v = acc[SL][r][h][ch] + bias4[ch];
                        const unsigned bits = kind == SVR_STORE_FP32 ? __float_as_uint(v)
                                            : kind == SVR_STORE_H16 ? (pack2h_raw(v * H16_SCALE, 0.f) & 0xffffu) : (unsigned)f2bf(v);
                        const unsigned wa = slot + 4u * ch;
                        if (kq == 0 && ch < N) asm volatile("ds_write_b32 %0, %1" :: "v"(wa), "v"(bits) : "memory");
                    }
                    acc[SL][r][h] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // write back: element e = lane, lane + 64 of every row (32 voxels x N couts, contiguous in the output)
            unsigned e0[4], e1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned rd = fl_base + (unsigned)((r * 32 * N + lane) << 2);
                asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:256" : "=&v"(e0[r]), "=&v"(e1[r]) : "v"(rd) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e0[0]), "+v"(e0[1]), "+v"(e0[2]), "+v"(e0[3]), "+v"(e1[0]), "+v"(e1[1]), "+v"(e1[2]), "+v"(e1[3]));
            const int nel = row_px * N;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int y = y0 + wave * 4 + r;
                if (y < g.H && ((DBG & 8) == 0 || a.N < 0)) {
                    const int64_t el = ((int64_t)(to * g.H + y) * g.W + x0) * N;
                    if (kind == SVR_STORE_FP32) {
                        unsigned* cp = (unsigned*)a.C + el;
                        if (lane < nel) cp[lane] = e0[r];
                        if (lane + 64 < nel) cp[lane + 64] = e1[r];
                    } else {
                        unsigned short* cp = (unsigned short*)a.C + el;
                        if (lane < nel) cp[lane] = (unsigned short)e0[r];
                        if (lane + 64 < nel) cp[lane + 64] = (unsigned short)e1[r];
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int y = y0 + wave * 4 + r, x = x0 + 16 * h + l15;
                if (y < g.H && x < g.W && kq == 0 && ((DBG & 8) == 0 || a.N < 0)) {   // couts 4 kq .. 4 kq + 3: only lanes kq == 0 hold real ones
                    const int m = (to * g.H + y) * g.W + x;
                    epilogue_store(a, acc[SL][r][h], acc[SL][r][h], m, 0);
                }
                acc[SL][r][h] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    };

    stage_halo(0, 0);
    if (nsteps > 1) stage_halo(1, 1);
    int k = 0, buf = 0, buf2 = 2;                         // ring slots of step k and of step k + 2
    for (int i = 0; i < n_in; ++i) {
        const int q3 = i % 3;
        const bool ok0 = (t0 + i < t1);
        const bool ok1 = kt > 1 && i >= 1 && (t0 + i - 1 < t1);
        const bool ok2 = kt > 2 && i >= 2 && (t0 + i - 2 < t1);
        for (int c = 0; c < cpk; ++c, ++k) {
            // step k's operands (and, at k = 0, the weights) have landed: this wave's share by the counted wait -- the ten pieces of
            // step k + 1 may stay in flight -- everyone's by the barrier, which also says that ring slot buf2 (step k - 1) is free
            if (k + 1 < nsteps && (DBG & 1) == 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((DBG & 1) == 0) { if (k + 2 < nsteps) stage_halo(k + 2, buf2); }
            if (ok0 & ok1 & ok2) {
                if (q3 == 0) step_body(std::integral_constant<int, 0>{}, std::true_type{}, buf, c, true, true, true);
                else if (q3 == 1) step_body(std::integral_constant<int, 1>{}, std::true_type{}, buf, c, true, true, true);
                else step_body(std::integral_constant<int, 2>{}, std::true_type{}, buf, c, true, true, true);
            } else {
                if (q3 == 0) step_body(std::integral_constant<int, 0>{}, std::false_type{}, buf, c, ok0, ok1, ok2);
                else if (q3 == 1) step_body(std::integral_constant<int, 1>{}, std::false_type{}, buf, c, ok0, ok1, ok2);
                else step_body(std::integral_constant<int, 2>{}, std::false_type{}, buf, c, ok0, ok1, ok2);
            }
            buf2 = buf;
            buf = buf == 2 ? 0 : buf + 1;
        }
        const int to = t0 + i - (kt - 1);
        if (to >= t0 && to < t1) {
            const int sl = (q3 + 3 - ((kt - 1) % 3)) % 3;
            if (sl == 0) flush(std::integral_constant<int, 0>{}, to);
            else if (sl == 1) flush(std::integral_constant<int, 1>{}, to);
            else flush(std::integral_constant<int, 2>{}, to);
        }
    }
#undef C4_LANDED_W12
#undef C4_LANDED_H
#undef C4_ISSUE_W12
#undef C4_ISSUE_H
#undef C4_RD
}

int g_conv_thinout4 = 1;  // svr_set_option("conv_thinout4"): 1 (default) N <= 4 on conv_thinout4_kernel | 0 on the 32-cout kernel above

static int launch_conv_thinout(const svr_gemm_args& a, hipStream_t s) {
    const svr_conv_geom& g = a.conv;
    const int w4_bytes = ((g.Cin / 32) * g.kt * 9 * 256 + 4095) / 4096 * 4096;      // resident 4-cout weight units, whole staging pieces
    if (g_conv_thinout4 && a.N <= 4 && (g.Cin % 32) == 0 && C4_WOFF + w4_bytes + C4_FLUSH <= 160 * 1024) {
        // a workgroup walks the frames of its chunk of the clip; the clip is cut into chunks only when the patches alone would leave
        // the chip under-filled (every chunk re-stages kt - 1 input frames)
        const int patches = ((g.H + C4_TY - 1) / C4_TY) * ((g.W + C4_TX - 1) / C4_TX);
        const int want_chunks = std::max(1, std::min(g.To, (4 * device_cu_count() + patches - 1) / patches));
        const int fpc = (g.To + want_chunks - 1) / want_chunks;
        const int chunks = (g.To + fpc - 1) / fpc;
#ifdef SVR_ABLATIONS
#define SVR_T4_ABL(D) case D: { static uint64_t done = 0; const int e = set_max_dynamic_lds((const void*)conv_thinout4_kernel<D>, 160 * 1024, done); \
        if (e != 0) return e; hipLaunchKernelGGL(conv_thinout4_kernel<D>, dim3(patches * chunks), dim3(C4_NT), C4_WOFF + w4_bytes + C4_FLUSH, s, a, fpc, C4_WOFF + w4_bytes); \
        return (int)hipGetLastError(); }
        switch (g_pipe_abl) { SVR_T4_ABL(1) SVR_T4_ABL(4) SVR_T4_ABL(8) default: break; }
#undef SVR_T4_ABL
#endif
        static uint64_t lds_attr_done4 = 0;
        const int e4 = set_max_dynamic_lds((const void*)conv_thinout4_kernel<0>, 160 * 1024, lds_attr_done4);
        if (e4 != 0) return e4;
        hipLaunchKernelGGL(conv_thinout4_kernel<0>, dim3(patches * chunks), dim3(C4_NT), C4_WOFF + w4_bytes + C4_FLUSH, s, a, fpc, C4_WOFF + w4_bytes);
        return (int)hipGetLastError();
    }
    const int tiles = g.To * ((g.H + CT_TY - 1) / CT_TY) * ((g.W + CT_TX - 1) / CT_TX);
    static uint64_t lds_attr_done = 0;               // per device (svr_common.h)
    {
        const int e = set_max_dynamic_lds((const void*)conv_thinout_kernel, 160 * 1024, lds_attr_done);
        if (e != 0) return e;
    }
    hipLaunchKernelGGL(conv_thinout_kernel, dim3(tiles), dim3(CT_NT), CT_LDS, s, a);
    return (int)hipGetLastError();
}

}  // namespace svr
