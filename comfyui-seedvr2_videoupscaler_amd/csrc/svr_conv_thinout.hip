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


static int launch_conv_thinout(const svr_gemm_args& a, hipStream_t s) {
    const svr_conv_geom& g = a.conv;
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
