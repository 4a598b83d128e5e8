// Sub-pixel upsampler conv for gfx950: the (kt', 2, 2)-tap stride-1 convs over the LOW-resolution input that replace
// Upsample3D's upscale_conv + pixel shuffle + 3x3x3 conv (subpixel.py), each launch computing one output phase
// of the upsampled grid (svr_gemm_args.phase).
//
// Same structure as the 8-row LDS-halo kernel of svr_conv_halo2.hip (conv_halo2_kernel<16, 3>), re-derived for a 2 x 2 window:
//   * a workgroup owns a 16 x 32 patch of the LOW-resolution grid x 128 couts, four waves, ONE wave per SIMD, every wave
//     8 patch rows x 64 couts = 16 accumulators of v_mfma_f32_32x32x16_bf16 (256 AGPRs);
//   * per A step (temporal source frame, 32-channel slice) the 17 x 33 halo of the patch (origin y0 - ph, x0 - pw: the window
//     of a phase starts one row / column earlier when py / px = 0) is staged once by LDS-DMA and all four taps run out of it:
//     interval J = dx * 2 + dy multiplies output row m with halo row m + dy at column shift dx;
//   * the weights stream to registers from the fragment-ordered copy (svr_conv_pack_frag_taps), two intervals ahead;
//   * one continuous MFMA stream per wave: the 9 halo rows of a column shift live in registers; a row's fragments are re-read
//     into the registers of a row the running burst has finished with (row 0 of the next shift under dy = 0, rows 1..7 under
//     dy = 1 as their pairs retire, row 8 at the start of dy = 0), every wait is counted and retires reads issued at least
//     three MFMA groups (768 cycles) earlier.
// With only four tap intervals per A step the next halo cannot be staged one step ahead and still land in time, so the
// halo is staged THREE steps ahead into a ring of four buffers (9 pieces of 4 KiB each, issued under intervals 2 and 3);
// one workgroup barrier per A step, at the end of interval 1: every wave has then finished the reads of the buffer that
// interval 2 starts to refill (its last reads -- row 8 of dx = 1 -- were issued in interval 2 of the PREVIOUS step), and the
// pieces it staged two steps ago for the buffer that interval 2 starts to read have long landed (the counted vmcnt waits
// of the intervals in between retire them).
// Epilogue: the fp32 tile is parked in LDS in four passes and leaves row-contiguous into its phase of the upsampled tensor,
// voxels on the image border taking their bias from the border table.
#include "svr_common.h"
#include "../../include/seedvr2_hip.h"
#include <type_traits>

namespace svr {

constexpr int CS_TY = 16, CS_TX = 32, CS_HX = CS_TX + 1, CS_HY = CS_TY + 1;
constexpr int CS_NT = 256;
constexpr int CS_ROWS = CS_HX * CS_HY;                       // 561 halo pixels
constexpr int CS_PIECES = (CS_ROWS * 4 + CS_NT - 1) / CS_NT; // 9 (the tail reads the zero page)
constexpr int CS_ABUF = CS_PIECES * CS_NT * 16;              // 36 864 B per buffer
constexpr int CS_NBUF = 4;
constexpr int CS_EP_PITCH = 528;                             // 128 floats + 16 B pad
constexpr int CS_EP_BYTES = 4 * 32 * CS_EP_PITCH;            // four patch rows per epilogue pass
constexpr int CS_LDS = CS_NBUF * CS_ABUF;                    // 147 456 B (the epilogue parks in the same space)
static_assert(CS_PIECES == 9 && CS_EP_BYTES <= CS_LDS, "piece schedule / epilogue parking");

template <int OFF> SVR_DEVICE void cs_rd2(bf16x8 (&r)[2], unsigned a0) {   // both k-steps of one halo row: chunk c and c ^ 2
    const unsigned a1 = a0 ^ 32u;
    asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%4"
                 : "=&v"(r[0]), "=&v"(r[1]) : "v"(a0), "v"(a1), "n"(OFF) : "memory");
}
template <int N> SVR_DEVICE void cs_wait_rows(bf16x8 (&a)[2]) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a[0]), "+v"(a[1]) : "n"(N));
    __builtin_amdgcn_sched_barrier(0);
}
template <int N> SVR_DEVICE void cs_wait_rows(bf16x8 (&a)[2], bf16x8 (&b)[2]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]) : "n"(N));
    __builtin_amdgcn_sched_barrier(0);
}
template <int N> SVR_DEVICE void cs_wait_rows(bf16x8 (&a)[2], bf16x8 (&b)[2], bf16x8 (&c)[2], bf16x8 (&d)[2], bf16x8 (&e)[2]) {
    asm volatile("s_waitcnt lgkmcnt(%10)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(c[0]), "+v"(c[1]), "+v"(d[0]), "+v"(d[1]), "+v"(e[0]), "+v"(e[1])
                 : "n"(N));
    __builtin_amdgcn_sched_barrier(0);
}
template <int N> SVR_DEVICE void cs_wait_w(bf16x8 (&w)[2][2]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]) : "n"(N));
    __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(CS_NT, 1) void conv_sub_kernel(const svr_gemm_args a, const int band_rows) {
    constexpr int MTW = 8, NTW = 2, NT = CS_NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((ext_vector_type(16))) float f32x16_t;
    const svr_conv_geom& g = a.conv;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;              // patch rows 8 wm .. 8 wm + 7, couts 64 wn .. 64 wn + 63

    // ---- tile id -> (frame, patch row, patch column, cout tile): XCD-contiguous bands of the banded order of svr_conv_halo2.hip
    const int tiles_x = (g.W + CS_TX - 1) / CS_TX;
    const int tiles_y = (g.H + CS_TY - 1) / CS_TY;
    const int tiles_n = a.N / 128;
    int tl;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, j = bid >> 3, q = nwg >> 3, r = nwg & 7;
        tl = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    // quad launch (svr_phase_scatter.quad): the spatial phase is the fastest tile index; everything that depends on the phase --
    // pads, weights, biases, output position -- is taken from the per-phase arrays (wave-uniform: scalar loads)
    const bool quad = a.phase.enabled && a.phase.quad != 0;
    int qp = 0;
    if (quad) { qp = tl & 3; tl >>= 2; }
    const int ph_py = quad ? (qp >> 1) : a.phase.py, ph_px = quad ? (qp & 1) : a.phase.px;
    const int pad_h = quad ? 1 - ph_py : g.ph, pad_w = quad ? 1 - ph_px : g.pw;
    const void* const w_frag = quad ? a.phase.W_frag4[qp] : a.W_frag;
    const float* const bias_p = quad ? a.phase.bias4[qp] : a.bias;
    const float* const bias_border_p = quad ? a.phase.bias_border4[qp] : a.phase.bias_border;
    const int tn = tl % tiles_n;
    int rr = tl / tiles_n;
    const int tx = rr % tiles_x; rr /= tiles_x;
    int ty, to;
    if (band_rows > 0 && band_rows < tiles_y) {
        const int per_band = band_rows * g.To;
        const int b = rr / per_band;
        const int rows_b = min(band_rows, tiles_y - b * band_rows);
        const int r2 = rr - b * per_band;
        to = r2 / rows_b;
        ty = b * band_rows + (r2 - to * rows_b);
    } else {
        ty = rr % tiles_y;
        to = rr / tiles_y;
    }
    const int y0 = ty * CS_TY, x0 = tx * CS_TX, n0 = tn * 128;

    const int cpk = g.Cin / 32;                           // 32-channel slices per temporal tap
    const int nA = g.kt * cpk;                            // A steps
    const int P = nA * 4;                                 // tap intervals
    const int64_t frame_bytes = (int64_t)g.H * g.W * g.Cin * 2;
    // ---- staging roles: one 16-byte chunk per thread and piece (halo pixel = id >> 2, chunk position = id & 3)
    const int srow = tid >> 2, spos = tid & 3;
    uint32_t poff[CS_PIECES];                             // halo piece -> pixel index (or ~0: outside the image / past the halo)
    uint32_t akeys = 0;                                   // halo piece -> swizzled source chunk (2 bits each)
#pragma unroll
    for (int q = 0; q < CS_PIECES; ++q) {
        const int row = q * (NT / 4) + srow;
        const int hy = row / CS_HX, hx = row - hy * CS_HX;
        const int y = y0 - pad_h + hy, x = x0 - pad_w + hx;
        const bool ok = (row < CS_ROWS) & ((unsigned)y < (unsigned)g.H) & ((unsigned)x < (unsigned)g.W);
        poff[q] = ok ? (uint32_t)(y * g.W + x) : 0xffffffffu;
        akeys |= (uint32_t)(spos ^ ((hx >> 2) & 3)) << (2 * q);
    }
    char* const wave_dst = smem + wave * 1024;
    // input frame of A step s (+ channel slice): frame to + dt - pt, the halo tensor or frame 0 before the slice; nullptr past the end
    auto frame_ptr = [&](int s) -> const char* {
        if (s >= nA) return nullptr;
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
    auto stage = [&](auto qc, const char* fptr, int buf) {
        constexpr int Q = decltype(qc)::value;
        const int ck = (akeys >> (2 * Q)) & 3;
        const char* src = (poff[Q] == 0xffffffffu || fptr == nullptr) ? (const char*)g.zeros
                                                                      : fptr + ((int64_t)poff[Q] * g.Cin + ck * 8) * 2;
        glds16(src, wave_dst + buf * CS_ABUF + Q * (NT * 16));
    };
    auto stage_all = [&](const char* fptr, int buf) {
        stage(std::integral_constant<int, 0>{}, fptr, buf); stage(std::integral_constant<int, 1>{}, fptr, buf);
        stage(std::integral_constant<int, 2>{}, fptr, buf); stage(std::integral_constant<int, 3>{}, fptr, buf);
        stage(std::integral_constant<int, 4>{}, fptr, buf); stage(std::integral_constant<int, 5>{}, fptr, buf);
        stage(std::integral_constant<int, 6>{}, fptr, buf); stage(std::integral_constant<int, 7>{}, fptr, buf);
        stage(std::integral_constant<int, 8>{}, fptr, buf);
    };

    // ---- fragment addressing: byte offset of k-step 0 for halo column shift dx (k-step 1 is `^ 32`)
    const int l31 = lane & 31, hi = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    unsigned fa[2];
#pragma unroll
    for (int dx = 0; dx < 2; ++dx)
        fa[dx] = lds0 + (unsigned)((wm * MTW * CS_HX + l31) * 64 + ((hi ^ (((dx + l31) >> 2) & 3)) << 4));

    f32x16_t acc[MTW][NTW];
#pragma unroll
    for (int y = 0; y < MTW; ++y)
#pragma unroll
        for (int z = 0; z < NTW; ++z)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[y][z][e] = 0.f;
    bf16x8 a0[2], a1[2], a2[2], a3[2], a4[2], a5[2], a6[2], a7[2], a8[2];      // halo rows of the current column shift
    bf16x8 w0[NTW][2], w1[NTW][2], w2[NTW][2], w3[NTW][2];     // weights of intervals k = 0 .. 3 (mod 4): two in flight, one in use
                                                               // (a fourth set makes the rotation one A step long: a single loop body)

    // ---- weights straight from global memory in fragment order ([32-cout block][interval][k-step][lane][8 bf16])
    const int64_t nstride = (int64_t)P * 2048;
    const char* wp0 = (const char*)w_frag + (int64_t)(n0 / 32 + wn * 2) * nstride;         // wave-uniform
    const char* wp1 = wp0 + nstride;
    const int voff = lane * 16;
    auto wload = [&](bf16x8 (&w)[NTW][2], int k) {
        auto uniform_ptr = [](const char* p) {
            const uint64_t u = (uint64_t)p;
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi_ = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
            return (const char*)(((uint64_t)hi_ << 32) | lo);
        };
        const char* p0 = uniform_ptr(wp0 + (int64_t)min(k, P - 1) * 2048);               // (the last two intervals re-load the final unit)
        const char* p1 = uniform_ptr(wp1 + (int64_t)min(k, P - 1) * 2048);
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(w[0][0]) : "v"(voff), "s"(p0) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(w[0][1]) : "v"(voff), "s"(p0) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(w[1][0]) : "v"(voff), "s"(p1) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(w[1][1]) : "v"(voff), "s"(p1) : "memory");
    };

#define CS_RD(ROW, R, DXV, BUFOFF) cs_rd2<((R) * CS_HX + (DXV)) * 64>(ROW, fa[DXV] + (BUFOFF))
#define CS_MM(W, KS, ROW, MT, NTI) \
    acc[MT][NTI] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[NTI][KS], ROW[KS], acc[MT][NTI], 0, 0, 0)
#define CS_MM8(W, RA, MA, RB, MB) \
    CS_MM(W, 0, RA, MA, 0); CS_MM(W, 0, RA, MA, 1); CS_MM(W, 0, RB, MB, 0); CS_MM(W, 0, RB, MB, 1); \
    CS_MM(W, 1, RA, MA, 0); CS_MM(W, 1, RA, MA, 1); CS_MM(W, 1, RB, MB, 0); CS_MM(W, 1, RB, MB, 1)

    // One tap interval: position J = dx * 2 + dy of A step s.  VMEM issue order per interval: 4 weight loads (after group 1),
    // then (J = 2: 5, J = 3: 4) pieces of the halo of step s + 3.  Younger than the weights of interval J (issued in J - 2):
    // the pieces of J - 2, the weights and pieces of J - 1.
    auto interval = [&](auto jc, int s, const char* fstage, bf16x8 (&wc)[NTW][2], bf16x8 (&wn_)[NTW][2]) {
        constexpr int J = decltype(jc)::value;
        constexpr int DX = J >> 1, DY = J & 1;
        constexpr int NPC[4] = {0, 0, 5, 4};
        constexpr int NV = 4 + NPC[(J + 3) & 3] + NPC[(J + 2) & 3];
        const int k = s * 4 + J;
        const unsigned cur = (unsigned)((s & 3) * CS_ABUF), nxt = (unsigned)(((s + 1) & 3) * CS_ABUF);
        const int sbuf = (s + 3) & 3;
        // the column shift after this one: dx = 1 of the same step, or dx = 0 of the next step's buffer
        constexpr int NDX = DX ^ 1;
        const unsigned nb = DX == 0 ? cur : nxt;
        if constexpr (DY == 0) {
            CS_RD(a8, 8, DX, cur);
            // in flight (oldest first): rows 0 .. 7 of this shift (issued under the two previous intervals), row 8 just now:
            // at most 8 reads outstanding = rows 0 .. 4 have landed (issued at least two MFMA groups ago)
            cs_wait_rows<8>(a0, a1, a2, a3, a4);
            cs_wait_w<NV>(wc);
            CS_MM8(wc, a0, 0, a1, 1);
            __builtin_amdgcn_sched_barrier(0);
            CS_RD(a0, 0, NDX, nb);
            wload(wn_, k + 2);
            __builtin_amdgcn_sched_barrier(0);
            CS_MM8(wc, a2, 2, a3, 3);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (J == 2) {
                stage(std::integral_constant<int, 0>{}, fstage, sbuf); stage(std::integral_constant<int, 1>{}, fstage, sbuf);
                stage(std::integral_constant<int, 2>{}, fstage, sbuf);
            }
            cs_wait_rows<8>(a4, a5);                       // younger than row 5: rows 6, 7, 8 and the next shift's row 0
            CS_MM8(wc, a4, 4, a5, 5);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (J == 2) { stage(std::integral_constant<int, 3>{}, fstage, sbuf); stage(std::integral_constant<int, 4>{}, fstage, sbuf); }
            cs_wait_rows<4>(a6, a7);                       // younger than row 7: row 8 and the next shift's row 0
            CS_MM8(wc, a6, 6, a7, 7);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            cs_wait_w<NV>(wc);
            CS_MM8(wc, a1, 0, a2, 1);
            __builtin_amdgcn_sched_barrier(0);
            CS_RD(a1, 1, NDX, nb); CS_RD(a2, 2, NDX, nb);
            wload(wn_, k + 2);
            __builtin_amdgcn_sched_barrier(0);
            CS_MM8(wc, a3, 2, a4, 3);
            __builtin_amdgcn_sched_barrier(0);
            CS_RD(a3, 3, NDX, nb); CS_RD(a4, 4, NDX, nb);
            if constexpr (J == 3) { stage(std::integral_constant<int, 5>{}, fstage, sbuf); stage(std::integral_constant<int, 6>{}, fstage, sbuf); }
            __builtin_amdgcn_sched_barrier(0);
            CS_MM8(wc, a5, 4, a6, 5);
            __builtin_amdgcn_sched_barrier(0);
            CS_RD(a5, 5, NDX, nb); CS_RD(a6, 6, NDX, nb);
            if constexpr (J == 3) { stage(std::integral_constant<int, 7>{}, fstage, sbuf); stage(std::integral_constant<int, 8>{}, fstage, sbuf); }
            // row 8 of this shift was read at the start of the dy = 0 interval; younger: the next shift's rows 0 .. 6 = 14 reads
            cs_wait_rows<14>(a8);
            CS_MM8(wc, a7, 6, a8, 7);
            __builtin_amdgcn_sched_barrier(0);
            CS_RD(a7, 7, NDX, nb);
            if constexpr (J == 1) {
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- prologue: halos of steps 0, 1, 2 (zero page past the last step), the first two weight units
    stage_all(frame_ptr(0), 0);
    stage_all(frame_ptr(1), 1);
    stage_all(frame_ptr(2), 2);
    wload(w0, 0);
    wload(w1, 1);
    __builtin_amdgcn_sched_barrier(0);                   // (see svr_conv_halo2.hip: the zeroed accumulators are materialised
#pragma unroll                                            //  under the latency of the first loads, not behind the wait)
    for (int y = 0; y < MTW; ++y)
#pragma unroll
        for (int z = 0; z < NTW; ++z) asm volatile("" : "+a"(acc[y][z]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    CS_RD(a0, 0, 0, 0u); CS_RD(a1, 1, 0, 0u); CS_RD(a2, 2, 0, 0u); CS_RD(a3, 3, 0, 0u);
    CS_RD(a4, 4, 0, 0u); CS_RD(a5, 5, 0, 0u); CS_RD(a6, 6, 0, 0u); CS_RD(a7, 7, 0, 0u);

    for (int s = 0; s < nA; ++s) {
        const char* fs = frame_ptr(s + 3);
        interval(std::integral_constant<int, 0>{}, s, fs, w0, w2);
        interval(std::integral_constant<int, 1>{}, s, fs, w1, w3);
        interval(std::integral_constant<int, 2>{}, s, fs, w2, w0);
        interval(std::integral_constant<int, 3>{}, s, fs, w3, w1);
    }
    // drain: the hand-issued weight loads and look-ahead fragment reads, the zero-page pieces of the last steps; every wave
    // must be out of the K loop before the parking area (the halo buffers' space) is written
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
#undef CS_RD
#undef CS_MM
#undef CS_MM8

    // ---- epilogue: four passes of four patch rows (every wave parks two of its eight rows per pass) through LDS; one thread
    // finishes 8 couts of a voxel: bias (border voxels: from the table), bf16, one 16-byte store into the voxel's phase position
    const int hi4 = hi * 4;
    const int n = n0 + (tid & 15) * 8;
    f32x4 bias_lo = {0.f, 0.f, 0.f, 0.f}, bias_hi = {0.f, 0.f, 0.f, 0.f};
    if (bias_p) {
        bias_lo = *(const f32x4*)(bias_p + n);
        bias_hi = *(const f32x4*)(bias_p + n + 4);
    }
    const int py = a.phase.enabled ? ph_py : 0, px = a.phase.enabled ? ph_px : 0;
    const int up = a.phase.enabled ? 2 : 1, ts = a.phase.enabled ? a.phase.t_stride : 1;
    const int yb = py ? g.H - 1 : 0, xb = px ? g.W - 1 : 0;
    const float* btab = a.phase.enabled ? bias_border_p : nullptr;
    float gs0 = 0.f, gq0 = 0.f, gs1 = 0.f, gq1 = 0.f;      // fused GroupNorm statistics of the stored values (columns n .. n + 3, n + 4 .. n + 7)
    // (the body is instantiated per output kind -- bf16 | fp32 | h16, the wide residual trunk -- so its loops carry no option branches)
    auto ep_body = [&](auto o32c) {
    constexpr int OKIND = decltype(o32c)::value;
    constexpr bool O32 = OKIND == SVR_STORE_FP32;
#pragma unroll
    for (int pass = 0; pass < MTW / 2; ++pass) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            char* row = smem + ((wm * 2 + j) * 32 + l31) * CS_EP_PITCH;
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
        lds_barrier();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 lo[4], hi_[4];
            int64_t off[4];
            bool ok[4];
            int border[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int vox = ((half * 4 + it) * NT + tid) >> 4;          // voxel slot of this pass
                const int r = vox >> 5;
                const int y = y0 + (r >> 1) * MTW + 2 * pass + (r & 1), x = x0 + (vox & 31);
                ok[it] = y < g.H && x < g.W;
                const int yc = min(y, g.H - 1), xc = min(x, g.W - 1);
                border[it] = (yc == yb ? 1 : 0) | (xc == xb ? 2 : 0);
                off[it] = (((int64_t)to * ts * (up * g.H) + up * yc + py) * (up * g.W) + up * xc + px) * a.N + n;
                lo[it] = *(const f32x4*)(smem + vox * CS_EP_PITCH + (tid & 15) * 32);
                hi_[it] = *(const f32x4*)(smem + vox * CS_EP_PITCH + (tid & 15) * 32 + 16);
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                f32x4 bl = bias_lo, bh = bias_hi;
                if (btab != nullptr && border[it] != 0) {                   // (rare: the one-voxel frame of the image)
                    const float* bb = btab + (int64_t)(border[it] - 1) * a.N + n;
                    bl = *(const f32x4*)bb;
                    bh = *(const f32x4*)(bb + 4);
                }
                const float f[8] = {lo[it][0] + bl[0], lo[it][1] + bl[1], lo[it][2] + bl[2], lo[it][3] + bl[3],
                                    hi_[it][0] + bh[0], hi_[it][1] + bh[1], hi_[it][2] + bh[2], hi_[it][3] + bh[3]};
                float r[8];
                if constexpr (O32) {
                    if (ok[it]) {
                        float* cp = (float*)a.C + off[it];
                        *(float4*)cp = make_float4(f[0], f[1], f[2], f[3]);
                        *(float4*)(cp + 4) = make_float4(f[4], f[5], f[6], f[7]);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) r[e] = f[e];
                } else {
                    const uint4 pk = OKIND == SVR_STORE_H16 ? pack8h(f) : pack8(f);
                    if (ok[it]) *(uint4*)((bf16_t*)a.C + off[it]) = pk;
                    if constexpr (OKIND == SVR_STORE_H16) unpack8h(pk, r); else unpack8(pk, r);
                }
                if (a.gn_partial != nullptr && ok[it]) {
                    gs0 += r[0] + r[1] + r[2] + r[3];
                    gq0 += r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
                    gs1 += r[4] + r[5] + r[6] + r[7];
                    gq1 += r[4] * r[4] + r[5] * r[5] + r[6] * r[6] + r[7] * r[7];
                }
            }
        }
        if (pass + 1 < MTW / 2) lds_barrier();
    }
    };
    if (a.out_f32 == SVR_STORE_FP32) ep_body(std::integral_constant<int, SVR_STORE_FP32>{});
    else if (a.out_f32 == SVR_STORE_H16) ep_body(std::integral_constant<int, SVR_STORE_H16>{});
    else ep_body(std::integral_constant<int, SVR_STORE_BF16>{});
    if (a.gn_partial) {
        // fixed-order reduction thread -> quad -> group (svr_conv_halo2.hip's); the (sum, sum of squares) of this patch go to
        // gn_partial[output frame][phase (py, px)][block][group]: the four phase launches of an upsampled frame fill one row of
        // 4 x blocks entries, which svr_groupnorm_reduce() adds up (a dense launch: [frame][block][group])
        __syncthreads();
        float4* red = (float4*)smem;                      // [NT]
        double2* qsum = (double2*)(smem + 8192);          // [32 quads]
        red[tid] = make_float4(gs0, gq0, gs1, gq1);
        __syncthreads();
        if (tid < 32) {                                   // quad = 2 * chunk + half; rows tid' with tid' & 15 == chunk
            const int c = tid >> 1, h = tid & 1;
            double s_ = 0.0, q_ = 0.0;
            for (int j = 0; j < NT / 16; ++j) {
                const float4 v = red[(j << 4) | c];
                s_ += (double)(h ? v.z : v.x);
                q_ += (double)(h ? v.w : v.y);
            }
            qsum[tid] = make_double2(s_, q_);
        }
        __syncthreads();
        const int qpg = (a.N / a.gn_groups) >> 2;         // quads per group (channels per group / 4)
        if (tid < 32 / qpg) {
            double s_ = 0.0, q_ = 0.0;
            for (int i = 0; i < qpg; ++i) { s_ += qsum[tid * qpg + i].x; q_ += qsum[tid * qpg + i].y; }
            const int nblk = tiles_y * tiles_x;
            const int per_frame = a.phase.enabled ? 4 * nblk : nblk;
            const int slot = (a.phase.enabled ? (py * 2 + px) * nblk : 0) + ty * tiles_x + tx;
            ((double2*)a.gn_partial)[((int64_t)to * ts * per_frame + slot) * a.gn_groups + (n0 >> 2) / qpg + tid] = make_double2(s_, q_);
        }
    }
}

int g_conv_sub = 1;        // 0: the sub-pixel convs run on the generic implicit-GEMM kernel (svr_set_option("conv_sub"))

// what conv_sub_kernel serves: (kt, 2, 2) taps, stride 1, same-size output, pads 0 | 1, fragment-ordered weights, plain bias epilogue
static bool conv_sub_gn_ok(const svr_gemm_args& a) {         // fused statistics: 4, 8 or 16 channels per group inside a 128-cout tile
    const int cpg = a.gn_groups > 0 ? a.N / a.gn_groups : 0;
    return cpg >= 4 && !(cpg & 3) && a.N % a.gn_groups == 0 && 128 % cpg == 0;
}

static bool conv_sub_eligible(const svr_gemm_args& a) {
    const svr_conv_geom& g = a.conv;
    const bool quad = a.phase.enabled && a.phase.quad != 0;
    bool ptrs = true;
    if (quad) {
        for (int p = 0; p < 4; ++p)
            ptrs = ptrs && a.phase.W_frag4[p] != nullptr && (!a.phase.bias4[p] || ((uintptr_t)a.phase.bias4[p] % 16) == 0) &&
                   (!a.phase.bias_border4[p] || ((uintptr_t)a.phase.bias_border4[p] % 16) == 0);
    } else {
        ptrs = a.W_frag != nullptr && (unsigned)g.ph <= 1u && (unsigned)g.pw <= 1u && (!a.bias || ((uintptr_t)a.bias % 16) == 0) &&
               (!a.phase.bias_border || ((uintptr_t)a.phase.bias_border % 16) == 0);
    }
    return g_conv_sub && g.enabled && ptrs && g.kh == 2 && g.kw == 2 && g.sh == 1 && g.sw == 1 && g.st == 1 &&
           g.Ho == g.H && g.Wo == g.W && g.Cin % 32 == 0 && g.kt >= 1 && g.kt <= 3 &&
           g.To == g.T + g.pt - g.kt + 1 && !a.ps.enabled && a.epilogue == SVR_EPI_BIAS && !a.resid && !a.gate &&
           (!a.gn_partial || conv_sub_gn_ok(a)) && (a.N % 128) == 0 && (a.phase.enabled || (a.ldc == a.N)) &&
           (int64_t)g.H * g.W * g.Cin * 2 < (int64_t)1 << 32 && ((uintptr_t)a.C % 16) == 0;
}

// partial blocks per OUTPUT frame of the fused statistics (0: not produced); a phase launch fills a quarter of them
static int conv_sub_gn_blocks(const svr_gemm_args& a) {
    if (!conv_sub_eligible(a) || !conv_sub_gn_ok(a)) return 0;
    const int nblk = ((a.conv.H + CS_TY - 1) / CS_TY) * ((a.conv.W + CS_TX - 1) / CS_TX);
    return a.phase.enabled ? 4 * nblk : nblk;
}

static int launch_conv_sub(const svr_gemm_args& a, hipStream_t s) {
    const svr_conv_geom& g = a.conv;
    const int tiles = g.To * ((g.H + CS_TY - 1) / CS_TY) * ((g.W + CS_TX - 1) / CS_TX) * (a.N / 128) *
                      (a.phase.enabled && a.phase.quad ? 4 : 1);
    static uint64_t lds_attr_done = 0;               // per device (svr_common.h)
    {
        const int e = set_max_dynamic_lds((const void*)conv_sub_kernel, 160 * 1024, lds_attr_done);
        if (e != 0) return e;
    }
    hipLaunchKernelGGL(conv_sub_kernel, dim3(tiles), dim3(CS_NT), CS_LDS, s, a, g_conv_band);
    return (int)hipGetLastError();
}

// Weight re-pack for the register-streamed conv kernels, any spatial tap grid: W [N, K = (dt, dy, dx, c)] ->
// [N / 32][k = (dt * Cin/32 + slice) * kh * kw + dx * kh + dy][k-step 2][lane 64][8 bf16]
__global__ __launch_bounds__(256) void conv_pack_frag_taps_kernel(const bf16_t* __restrict__ W, uint4* __restrict__ out, int N, int K,
                                                                  int kt, int kh, int kw, int Cin) {
    const int cpk = Cin / 32, taps = kh * kw, P = kt * cpk * taps;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)(N / 32) * P * 128) return;
    const int lane = (int)(idx & 63), ks = (int)((idx >> 6) & 1);
    const int kk = (int)((idx >> 7) % P), n32 = (int)((idx >> 7) / P);
    const int s = kk / taps, J = kk - s * taps;
    const int dt = s / cpk, cs = s - dt * cpk, dy = J % kh, dx = J / kh;
    const int nn = n32 * 32 + (lane & 31), c = cs * 32 + ks * 16 + (lane >> 5) * 8;
    out[idx] = *(const uint4*)(W + (int64_t)nn * K + ((dt * kh + dy) * kw + dx) * Cin + c);
}

}  // namespace svr
