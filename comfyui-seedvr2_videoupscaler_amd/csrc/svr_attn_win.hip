// Window attention of the NaDiT for gfx950, second generation (head_dim 128, windows of <= AW_MAXL rows).
//
// Replaces pytorch_varlen_attention (src/models/dit_3b/attention.py:27-64) + the window gather / text concat /
// scatter around it (mmattn.py:199,245-264), like svr_attn.hip; that first kernel stays for head_dim 512 (VAE)
// and for longer sequences.  What round 1's profile said about it (profiles/r1_cfg3_pmc_traffic.json): 18 % of the
// MFMA peak, K/V fetched 5.5x from HBM, because (a) every 64-key tile was staged synchronously behind TWO dependent
// global loads (row index, then the row) with two barriers per tile and nothing in flight meanwhile, (b) V went
// through registers and 8-byte transposing LDS stores, (c) the q-tiles of one (window, head) were consecutive block
// ids, i.e. round-robin over the 8 XCDs, so each private L2 fetched the window's K/V again.  Here:
//
//   * block = 8 waves x 32 queries (256-query tile, one workgroup per CU; a 4-wave / two-workgroup build is kept behind
//     svr_set_option("attn_variant") -- 72 KiB LDS, <= 256 registers either way);
//   * the window's rows are resolved ONCE into an LDS table of qkv byte offsets (16-byte units, 32 bits: qkv < 64 GiB),
//     so the tile loop has no dependent global load and no 64-bit multiply;
//   * K and V tiles (64 keys x 128 d, 16 KiB each) are double-buffered and BOTH filled by 16-byte LDS-DMA
//     (global_load_lds) for tile t+1 while tile t is computed -- one barrier per tile, no VGPR staging;
//   * v_mfma_f32_32x32x16_bf16 throughout, everything transposed so that softmax is lane-local:
//       S^T[key, q] = K Q^T    A = K rows (ds_read_b128, chunk index XOR key&15 on the DMA source side: conflict-free),
//                              B = Q held in registers; lane (q = lane&31, hi = lane>>5) gets keys (r&3)+8(r>>2)+4hi
//       O^T[d,   q] = V^T P^T  B = P^T straight from the S^T accumulators (registers 8u..8u+7 of a 32-key block are
//                              the 8 key slots of k-step u), A = V^T built by ds_read_b64_tr_b16 from the row-major V
//                              tile: two transposing reads fetch exactly the two 4-key runs a lane's slots stand for
//                              (keys 16u+4hi+{0..3} and +8), so P never crosses lanes and V is never transposed in
//                              software.  V chunk index is XORed with (key&3)<<2 (source side) so that the 32 lanes of
//                              a tr-read group (2 d-halves x 4 keys x 4 column quads) cover all 64 banks once;
//   * the two halves of a query (lane, lane+32) exchange the running max with one v_permlane32_swap per tile and
//     the row sum once; the bf16 output rows are widened to 16-byte stores with the same instruction;
//   * block ids are remapped so that all q-tiles of a (window, head) pair -- and consecutive pairs -- run on ONE XCD:
//     K/V of a pair is fetched from HBM once and re-read from that XCD's L2 by its other q-tiles.
#include "svr_common.h"
#include "../../include/seedvr2_hip.h"
#include <cstdlib>
#include <type_traits>

namespace svr {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int AW_D = 128;                        // head dim
constexpr int AW_KT = 64;                        // keys per tile
constexpr int AW_MAXL = 2048;                    // longest window (rows incl. text) the LDS row table holds
constexpr int AW_TILE = AW_KT * AW_D * 2;        // bytes of one K (or V) tile
constexpr int AW_LDS = 4 * AW_TILE + AW_MAXL * 4;
constexpr float AW_DEFER = 6.0f;                 // log2 units: P <= 64 between rescales

// v_permlane32_swap_b32 vdst, src exchanges lanes 32-63 of vdst with lanes 0-31 of src.  Fed the same value in both
// operands, one of the two results is this lane's own value and the other the value of lane ^ 32 -- in BOTH halves.
// NOTE: the two results are copied into scalars before any __builtin_bit_cast: bit-casting the vector-element lvalue
// r[1] directly reads element 0 (clang 22 / ROCm 7.2: found on the GPU as a 17 % error, each half of a query
// normalising with its own max and sum; the -O0 IR shows both loads at offset 0).
SVR_DEVICE void aw_swap_halves(float v, float& own_or_other, float& other_or_own) {
    const unsigned a = __builtin_bit_cast(unsigned, v);
    const u32x2 r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    const unsigned r0 = r.x, r1 = r.y;
    own_or_other = __builtin_bit_cast(float, r0);
    other_or_own = __builtin_bit_cast(float, r1);
}
SVR_DEVICE float aw_other_half_max(float v) {    // max(v, value held by lane ^ 32)
    float a, b;
    aw_swap_halves(v, a, b);
    return fmaxf(a, b);
}
SVR_DEVICE float aw_other_half_sum(float v) {
    float a, b;
    aw_swap_halves(v, a, b);
    return a + b;
}

// Eight transposing reads = the V^T fragments (A operands) of the four 32-d blocks for ONE 16-key k-step: rows OFF/256
// .. + 3 and + 8 .. + 11 of the V tile.  Issued from inline asm so that (a) hipcc does not serialise them behind the
// LDS-DMA of the next tile (it puts s_waitcnt vmcnt(0) in front of the ds_read_tr builtin, not in front of plain LDS
// loads) and (b) the reads of k-step g+1 are in flight under the MFMAs of k-step g (the compiler's own schedule was
// read -> wait -> MFMA, one read at a time).  hipcc does not count asm loads: aw_wait_lgkm<N> is the counted wait, it
// names the destinations "+v" so no consumer can be scheduled above it (cdna_hip_programming.md 5.7 form (ii)).
template <int OFF>
SVR_DEVICE void aw_tr8(bf16x4 (&v)[8], unsigned a0, unsigned a1, unsigned a2, unsigned a3) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8 offset:%12\n\t"
        "ds_read_b64_tr_b16 %1, %8 offset:%13\n\t"
        "ds_read_b64_tr_b16 %2, %9 offset:%12\n\t"
        "ds_read_b64_tr_b16 %3, %9 offset:%13\n\t"
        "ds_read_b64_tr_b16 %4, %10 offset:%12\n\t"
        "ds_read_b64_tr_b16 %5, %10 offset:%13\n\t"
        "ds_read_b64_tr_b16 %6, %11 offset:%12\n\t"
        "ds_read_b64_tr_b16 %7, %11 offset:%13"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(OFF), "n"(OFF + 2048)
        : "memory");
}
// Four K fragments (A operands of S^T = K Q^T) by ds_read_b128, same discipline.
template <int OFF>
SVR_DEVICE void aw_k4(bf16x8 (&k)[4], unsigned a0, unsigned a1, unsigned a2, unsigned a3) {
    asm volatile(
        "ds_read_b128 %0, %4 offset:%8\n\t"
        "ds_read_b128 %1, %5 offset:%8\n\t"
        "ds_read_b128 %2, %6 offset:%8\n\t"
        "ds_read_b128 %3, %7 offset:%8"
        : "=&v"(k[0]), "=&v"(k[1]), "=&v"(k[2]), "=&v"(k[3])
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(OFF)
        : "memory");
}
template <int N>
SVR_DEVICE void aw_wait_k(bf16x8 (&k)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(k[0]), "+v"(k[1]), "+v"(k[2]), "+v"(k[3]) : "n"(N));
    __builtin_amdgcn_sched_barrier(0);
}
template <int N>
SVR_DEVICE void aw_wait_lgkm(bf16x4 (&v)[8]) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                 : "n"(N));
    __builtin_amdgcn_sched_barrier(0);
}

// NW: waves per workgroup (4: 128-query tiles, two workgroups per CU; 8: 256-query tiles, one workgroup per CU -- half the
// LDS-DMA instructions and L2->LDS bytes per MFMA, coarser tiles for ragged windows).  PRIO: s_setprio 1 around MFMA groups.
// QW: 32-query blocks per wave.  QW = 2 (with NW = 4: the same 256-query tile, ONE wave per SIMD) is the structural step the
// round-2 / round-3 reviews name: every K fragment (ds_read_b128) and every V^T fragment (ds_read_b64_tr_b16) feeds two MFMAs
// instead of one, so the LDS fragment bytes per MFMA halve while the LDS-DMA bytes per MFMA stay those of the 8-wave build;
// the two blocks' MFMAs alternate on independent accumulators.  Per query row the arithmetic (MFMA order, rescale decisions
// -- taken per 32-query block, and the blocks are the same 32-aligned blocks in every build) is unchanged, so all builds
// agree bit for bit.  192 accumulator + 64 Q registers per wave: one wave per SIMD (launch bound 1).
// PIPE (QW = 2 only; attn_variant 7 / 8): K Q^T runs ONE TILE AHEAD of the softmax.  Iteration t holds S(t) with its statistics done
// and computes, in this order,  K(t+1) Q^T -> S(t+1)  with the exponentials of S(t) between its MFMA groups,  V(t)^T P(t)^T,  the
// row sums, then the statistics (max, deferred rescale) of S(t+1).  LDS protocol: K is staged TWO tiles ahead, V one; with the one
// barrier at the end of an iteration that is still two buffers each -- in iteration t the waves READ K buffer (t+1)&1 and V buffer
// t&1 and the LDS-DMA WRITES K(t+2) into K buffer t&1 (last read in iteration t-1, before that iteration's barrier) and V(t+1)
// into V buffer (t+1)&1 (ditto); the prologue needs one extra barrier, between K(0) Q^T and the first refill of K buffer 0.
// The operations on O, l and the running max happen in the same order as in every other build (rescale(t), PV(t), rescale(t+1),
// PV(t+1), ...), so the result is again bit-identical.  The accumulating score set is copied into the one the VALU reads at the end of an iteration.
// ASYM (attn_variant 9 / 10; QW = 1): the instruction mix of this kernel is balanced -- per wave and tile 1 024 cycles
// of MFMAs (K Q^T, then V^T P^T) around ~960 cycles of softmax VALU work -- so two waves per SIMD could keep the MFMA pipe busy all the
// time, IF they ran in opposite phases.  They do not: every wave leaves the tile barrier at the same moment, both waves of a SIMD
// issue K Q^T together (sharing the pipe), then both sit in the softmax (sharing the VALU port): ~3 970 cycles per tile pair, which is
// the 46 % MFMA-busy the counters show.  With one wave of each SIMD at a higher priority that wave's K Q^T goes first and its softmax
// runs under the other wave's K Q^T, its V^T P^T under the other's softmax: ~2 500 cycles by the same arithmetic.  Nothing but the
// issue order changes.  ASYM = 1: waves 0 .. NW/2-1 are the favoured ones (waves w and w + NW/2 share a SIMD if the dispatcher deals
// waves round-robin over the four SIMDs), ASYM = 2: the even waves (if it fills SIMD by SIMD).
template <int NW, bool PRIO, int QW = 1, bool PIPE = false, int ASYM = 0>
__global__ __launch_bounds__(NW * 64, (QW == 2 ? 1 : 2)) void attn_win_kernel(
    const bf16_t* __restrict__ qkv, int64_t ld_qkv, bf16_t* __restrict__ out, int64_t ld_out,
    const int32_t* __restrict__ seq_rows, const int32_t* __restrict__ out_rows, const int32_t* __restrict__ cu,
    int heads, int n_pairs, int qt_per_pair, float scale_log2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* srow = (unsigned*)(smem + 4 * AW_TILE);     // row table: byte offset of every window row in qkv, in 16-byte units

    // ---- work item: XCD x owns pairs x, x + 8, ...; the q-tiles of a pair are consecutive in its dispatch order
    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int pair = (j / qt_per_pair) * 8 + xcd;
    if (pair >= n_pairs) return;
    const int seq = pair / heads, head = pair - seq * heads;
    const int beg = cu[seq];
    const int L = cu[seq + 1] - beg;
    constexpr int AW_QB = NW * 32 * QW, NP = 16 / NW; // queries per workgroup; LDS-DMA pieces (1 KiB of K + 1 KiB of V) per wave per tile
    const int q0 = (j % qt_per_pair) * AW_QB;
    if (q0 >= L) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int nk = (L + AW_KT - 1) / AW_KT;
    if constexpr (ASYM != 0) {
        if (ASYM == 1 ? (wave < NW / 2) : ((wave & 1) == 0)) __builtin_amdgcn_s_setprio(2);
    }

    const int64_t ld_bytes = ld_qkv * 2;                  // (multiple of 16: checked by the launcher)
    for (int i = tid; i < nk * AW_KT; i += NW * 64)
        srow[i] = (unsigned)(((int64_t)seq_rows[beg + min(i, L - 1)] * ld_bytes) >> 4);
    __syncthreads();

    const char* qbase = (const char*)qkv + (int64_t)head * (AW_D * 2);
    const int64_t k_off = (int64_t)heads * (AW_D * 2), v_off = 2 * k_off;

    // ---- LDS-DMA roles: wave instruction (it, wave) fills chunk positions (it*NW + wave)*64 + lane = 4 keys x 16 slots
    const int st_key = wave * 4 + (lane >> 4);                       // key within the (4 NW)-key group of piece `it`
    const int st_k = (((lane & 15) ^ (st_key & 15)) << 4);           // source chunk of K: slot ^ (key & 15)
    const int st_v = (((lane & 15) ^ ((lane >> 4) << 2)) << 4);      // source chunk of V: slot ^ ((key & 3) << 2)
    const char* nsrc[NP];                                            // source rows of the NEXT tile to stage (this lane's NP keys)
    auto next_rows = [&](int t) {                                    // (plain LDS reads + 64-bit mads, waited for by hipcc right here)
        const int tt = min(t, nk - 1);
#pragma unroll
        for (int it = 0; it < NP; ++it) nsrc[it] = qbase + ((uint64_t)srow[tt * AW_KT + it * (4 * NW) + st_key] << 4);
    };
    auto stage_piece = [&](int it, int buf) {                        // 1 KiB of K and 1 KiB of V per wave instruction
        glds16(nsrc[it] + k_off + st_k, smem + buf * AW_TILE + wave * 1024 + it * (NW * 1024));
        glds16(nsrc[it] + v_off + st_v, smem + (2 + buf) * AW_TILE + wave * 1024 + it * (NW * 1024));
    };
    next_rows(0);
#pragma unroll
    for (int it = 0; it < NP; ++it) stage_piece(it, 0);

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q = q0 + 32 (QW wave + b) + l31][16 ds + 8 hi .. + 8]
    int qpos[QW];
    bf16x8 qf[QW][8];
#pragma unroll
    for (int b = 0; b < QW; ++b) {
        qpos[b] = q0 + (wave * QW + b) * 32 + l31;
        const char* qp = qbase + ((uint64_t)srow[min(qpos[b], L - 1)] << 4) + hi * 16;
#pragma unroll
        for (int ds = 0; ds < 8; ++ds) qf[b][ds] = *(const bf16x8*)(qp + ds * 32);
    }
    next_rows(1);

    // ---- per-lane LDS byte addresses (buffer 0; the tile loop adds +-AW_TILE)
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // K: row = key (32 kb + l31), 16-byte slot = (2 ds + hi) ^ (key & 15); kb = 1 is the +8192 immediate
    unsigned ka_[8];
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) ka_[ds] = lds_base + l31 * 256 + (((2 * ds + hi) ^ (l31 & 15)) << 4);
    // V (ds_read_b64_tr_b16): this lane supplies the address of key row jr (of its group's 4 keys), columns 4 c4 .. + 3 of
    // the 16-d half G of 32-d block m; the instruction hands lane (G, i) the 4 keys of column i.  hi selects keys + 4.
    const int i16 = lane & 15, G = (lane >> 4) & 1, jr = i16 >> 2, c4 = i16 & 3;
    unsigned va_[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
        va_[m] = lds_base + 2 * AW_TILE + hi * 1024 + jr * 256 + (((((m ^ jr) << 2) | (2 * G + (c4 >> 1))) << 4)) + (c4 & 1) * 8;

    f32x16 o[QW][4];
    float m_run[QW], l_run[QW];
#pragma unroll
    for (int b = 0; b < QW; ++b) {
        m_run[b] = -INFINITY;
        l_run[b] = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[b][m][r] = 0.f;
    }

    __syncthreads();                                   // tile 0 landed (vmcnt(0) + barrier)

    if constexpr (PIPE) {
        static_assert(QW == 2 && NW == 4 && NP == 4, "the pipelined loop is written for four waves x 64 queries");
        const char* nsk[NP];                           // source rows of the K tile to stage (two tiles ahead); nsrc: of the V tile (one ahead)
        auto k_rows = [&](int t) {
            const int tt = min(t, nk - 1);
#pragma unroll
            for (int it = 0; it < NP; ++it) nsk[it] = qbase + ((uint64_t)srow[tt * AW_KT + it * (4 * NW) + st_key] << 4);
        };
        auto stage_k = [&](int it, int buf) { glds16(nsk[it] + k_off + st_k, smem + buf * AW_TILE + wave * 1024 + it * (NW * 1024)); };
        auto stage_v = [&](int it, int buf) { glds16(nsrc[it] + v_off + st_v, smem + (2 + buf) * AW_TILE + wave * 1024 + it * (NW * 1024)); };
        float mcv[QW];
        f32x2 psv[QW];
        bf16x8 kA[4], kB[4];
        bf16x4 vA[8], vB[8];
        bf16x8 pf[QW][2][2];
        f32x16 sA[QW][2], sB[QW][2];
        // statistics of one score set (tile `tile`): ragged mask, row max, deferred rescale of O and l (per 32-query block)
        auto stats = [&](f32x16 (&S)[QW][2], int tile) __attribute__((always_inline)) {
#pragma unroll
            for (int b = 0; b < QW; ++b) {
                if ((tile + 1) * AW_KT > L) {          // ragged last tile (wave-uniform): mask keys >= L
                    const int kbase = tile * AW_KT + 4 * hi;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (kbase + kb * 32 + (r & 3) + 8 * (r >> 2) >= L) S[b][kb][r] = -INFINITY;
                }
                float mx = S[b][0][0];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[b][kb][r]);
                mx = aw_other_half_max(mx);
                if (!__all((mx - m_run[b]) * scale_log2 <= AW_DEFER)) {
                    const float m_new = fmaxf(m_run[b], mx);
                    const float alpha = fast_exp2((m_run[b] - m_new) * scale_log2);
                    m_run[b] = m_new;
                    l_run[b] *= alpha;
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[b][m][r] *= alpha;
                }
                mcv[b] = m_run[b] * scale_log2;
                psv[b] = f32x2{0.f, 0.f};
            }
        };
        auto pexp = [&](f32x16 (&S)[QW][2], auto bc, auto kbc, auto uc) __attribute__((always_inline)) {
            constexpr int B = decltype(bc)::value, KB = decltype(kbc)::value, U = decltype(uc)::value;
            const f32x2 c2 = {scale_log2, scale_log2}, mc2 = {-mcv[B], -mcv[B]};
            float p[8];
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f32x2 sv = {S[B][KB][8 * U + e], S[B][KB][8 * U + e + 1]};
                const f32x2 a = __builtin_elementwise_fma(sv, c2, mc2);
                p[e] = fast_exp2(a[0]);
                p[e + 1] = fast_exp2(a[1]);
                const f32x2 pv = {p[e], p[e + 1]};
                psv[B] += pv;
            }
            const uint4 pk = pack8(p);
            pf[B][KB][U] = __builtin_bit_cast(bf16x8, pk);
        };
#define AW_I(v) std::integral_constant<int, (v)>{}
#define AW_P2(S, KB, U) pexp(S, AW_I(0), AW_I(KB), AW_I(U)); pexp(S, AW_I(1), AW_I(KB), AW_I(U))
#define AW_QK2(K, Q0, S, KB)                                                                                       \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                              \
            S[0][KB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(K[e], qf[0][Q0 + e], S[0][KB], 0, 0, 0);              \
            S[1][KB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(K[e], qf[1][Q0 + e], S[1][KB], 0, 0, 0);              \
        }
#define AW_PV2(V, KB, U)                                                                                           \
        _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                                              \
            const bf16x8 vf = __builtin_shufflevector(V[2 * m], V[2 * m + 1], 0, 1, 2, 3, 4, 5, 6, 7);               \
            o[0][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[0][KB][U], o[0][m], 0, 0, 0);                   \
            o[1][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[1][KB][U], o[1][m], 0, 0, 0);                   \
        }
#define AW_SGB()                                                                                                   \
        if (PRIO) { _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                  \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                       \
            __builtin_amdgcn_sched_group_barrier(0x402, 5, 0); } }
        auto zero_s = [&](f32x16 (&S)[QW][2]) __attribute__((always_inline)) {
#pragma unroll
            for (int b = 0; b < QW; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) { S[b][0][r] = 0.f; S[b][1][r] = 0.f; }
        };

        // ---- prologue: K(1) -> K buffer 1 (rows of tile 1 are in nsrc), S(0) = K(0) Q^T, its statistics; the rows of K(2)
#pragma unroll
        for (int it = 0; it < NP; ++it) glds16(nsrc[it] + k_off + st_k, smem + AW_TILE + wave * 1024 + it * (NW * 1024));
        zero_s(sA);
        aw_k4<0>(kA, ka_[0], ka_[1], ka_[2], ka_[3]);
        aw_k4<0>(kB, ka_[4], ka_[5], ka_[6], ka_[7]);
        aw_wait_k<4>(kA);
        AW_QK2(kA, 0, sA, 0)
        aw_k4<8192>(kA, ka_[0], ka_[1], ka_[2], ka_[3]);
        aw_wait_k<4>(kB);
        AW_QK2(kB, 4, sA, 0)
        aw_k4<8192>(kB, ka_[4], ka_[5], ka_[6], ka_[7]);
        aw_wait_k<4>(kA);
        AW_QK2(kA, 0, sA, 1)
        aw_wait_k<0>(kB);
        AW_QK2(kB, 4, sA, 1)
        stats(sA, 0);
        k_rows(2);
#pragma unroll
        for (int ds = 0; ds < 8; ++ds) ka_[ds] += AW_TILE;      // iteration 0 reads K(1) from K buffer 1
        __syncthreads();                               // K(1) landed (vmcnt(0)); every wave is done reading K buffer 0, which iteration 0 refills

        // ---- one tile.  Sc = S(t), statistics done; Sn (not LAST) receives S(t+1).
        auto iter = [&](auto lastc, f32x16 (&Sc)[QW][2], f32x16 (&Sn)[QW][2], const int t) __attribute__((always_inline)) {
            constexpr bool LAST = decltype(lastc)::value != 0;
            const int kw = t & 1, vw = kw ^ 1;         // buffers being refilled: K(t+2) -> kw, V(t+1) -> vw
            if constexpr (!LAST) {
                // (K(t+2) is staged unconditionally: past the last tile k_rows() clamps to it and the copy lands in a buffer nobody
                // reads again -- a wave-uniform branch here would cut the MFMA groups and the exponentials into separate blocks)
                zero_s(Sn);
                aw_k4<0>(kA, ka_[0], ka_[1], ka_[2], ka_[3]);
                aw_k4<0>(kB, ka_[4], ka_[5], ka_[6], ka_[7]);
                aw_wait_k<4>(kA);
                AW_QK2(kA, 0, Sn, 0)
                AW_P2(Sc, 0, 0);
                AW_SGB()
                stage_k(0, kw);
                stage_v(0, vw);
                aw_k4<8192>(kA, ka_[0], ka_[1], ka_[2], ka_[3]);
                aw_wait_k<4>(kB);
                AW_QK2(kB, 4, Sn, 0)
                AW_P2(Sc, 0, 1);
                AW_SGB()
                stage_k(1, kw);
                stage_v(1, vw);
                aw_k4<8192>(kB, ka_[4], ka_[5], ka_[6], ka_[7]);
                aw_wait_k<4>(kA);
                AW_QK2(kA, 0, Sn, 1)
                AW_P2(Sc, 1, 0);
                AW_SGB()
                stage_k(2, kw);
                stage_v(2, vw);
                aw_tr8<0>(vA, va_[0], va_[1], va_[2], va_[3]);         // V^T fragments of the first k-step
                aw_wait_k<8>(kB);
                AW_QK2(kB, 4, Sn, 1)
                stage_k(3, kw);
                stage_v(3, vw);
            } else {
                aw_tr8<0>(vA, va_[0], va_[1], va_[2], va_[3]);
                AW_P2(Sc, 0, 0);
                AW_P2(Sc, 0, 1);
                AW_P2(Sc, 1, 0);
            }
            aw_tr8<4096>(vB, va_[0], va_[1], va_[2], va_[3]);
            aw_wait_lgkm<8>(vA);
            AW_PV2(vA, 0, 0)
            AW_P2(Sc, 1, 1);
            AW_SGB()
            aw_tr8<8192>(vA, va_[0], va_[1], va_[2], va_[3]);
            aw_wait_lgkm<8>(vB);
            AW_PV2(vB, 0, 1)
            aw_tr8<12288>(vB, va_[0], va_[1], va_[2], va_[3]);
            aw_wait_lgkm<8>(vA);
            AW_PV2(vA, 1, 0)
            aw_wait_lgkm<0>(vB);
            AW_PV2(vB, 1, 1)
#pragma unroll
            for (int b = 0; b < QW; ++b) l_run[b] += psv[b][0] + psv[b][1];
            if constexpr (!LAST) {
                stats(Sn, t + 1);
                next_rows(t + 2);                      // V(t+2): staged in iteration t + 1
                k_rows(t + 3);                         // K(t+3)
                const int flip = (t & 1) ? -AW_TILE : AW_TILE;
#pragma unroll
                for (int ds = 0; ds < 8; ++ds) ka_[ds] -= flip;        // K(t+2) sits in K buffer t & 1
#pragma unroll
                for (int m = 0; m < 4; ++m) va_[m] += flip;            // V(t+1) in V buffer (t+1) & 1
                __syncthreads();                       // K(t+2), V(t+1) landed (vmcnt(0)); everyone is done with K buffer (t+1)&1 and V buffer t&1
            }
        };
        int t = 0;
        for (; t + 1 < nk; ++t) {
            iter(AW_I(0), sA, sB, t);
#pragma unroll
            for (int b = 0; b < QW; ++b) { sA[b][0] = sB[b][0]; sA[b][1] = sB[b][1]; }   // (the accumulating set lives in AGPRs: this is the read-out the VALU needs anyway)
        }
        iter(AW_I(1), sA, sA, t);
#undef AW_SGB
#undef AW_PV2
#undef AW_QK2
#undef AW_P2
#undef AW_I
    } else
    for (int t = 0; t < nk; ++t) {
        const int nxt = (t & 1) ^ 1;
        const bool more = t + 1 < nk;                  // wave-uniform

        // ---- S^T = K Q^T : 2 key blocks x 8 d-steps.  Fragment reads (asm, counted waits) run 4 k-steps ahead of the
        //      MFMAs; the 8 LDS-DMA pieces of tile t+1 are issued between the MFMA groups, not as a burst.
        f32x16 sacc[QW][2];
#pragma unroll
        for (int b = 0; b < QW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[b][0][r] = 0.f; sacc[b][1][r] = 0.f; }
        bf16x8 kA[4], kB[4];
        bf16x4 vA[8], vB[8];
        aw_k4<0>(kA, ka_[0], ka_[1], ka_[2], ka_[3]);
        aw_k4<0>(kB, ka_[4], ka_[5], ka_[6], ka_[7]);
#define AW_QK(K, Q0, S)                                                                                               \
        if (PRIO) __builtin_amdgcn_s_setprio(1);                                                                      \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                                 \
            _Pragma("unroll") for (int b = 0; b < QW; ++b)                                                            \
                sacc[b][S] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(K[e], qf[b][Q0 + e], sacc[b][S], 0, 0, 0);       \
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        aw_wait_k<4>(kA);
        AW_QK(kA, 0, 0)
        if (more) stage_piece(0, nxt);
        aw_k4<8192>(kA, ka_[0], ka_[1], ka_[2], ka_[3]);
        aw_wait_k<4>(kB);
        AW_QK(kB, 4, 0)
        if (more && NP == 4) stage_piece(1, nxt);
        aw_k4<8192>(kB, ka_[4], ka_[5], ka_[6], ka_[7]);
        aw_wait_k<4>(kA);
        AW_QK(kA, 0, 1)
        if (more) stage_piece(NP == 4 ? 2 : 1, nxt);
        aw_tr8<0>(vA, va_[0], va_[1], va_[2], va_[3]);         // V^T fragments of the first k-step fly under the softmax
        aw_wait_k<8>(kB);
        AW_QK(kB, 4, 1)
        if (more && NP == 4) stage_piece(3, nxt);
#undef AW_QK

        bf16x8 pf[QW][2][2];
        if constexpr (QW == 1) {
            // ---- online softmax, lane-local over this lane's 32 keys; the other 32 keys of the tile live in lane ^ 32
#pragma unroll
            for (int b = 0; b < QW; ++b) {
                if ((t + 1) * AW_KT > L) {                 // ragged last tile (wave-uniform): mask keys >= L
                    const int kbase = t * AW_KT + 4 * hi;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (kbase + kb * 32 + (r & 3) + 8 * (r >> 2) >= L) sacc[b][kb][r] = -INFINITY;
                }
                float mx = sacc[b][0][0];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[b][kb][r]);
                mx = aw_other_half_max(mx);
                // Deferred rescale (cdna_hip_programming.md T13): while no row's maximum grew by more than 2^AW_DEFER over the
                // reference value m_run, keep m_run -- P is then bounded by 2^AW_DEFER instead of 1 (same RELATIVE bf16 precision,
                // fp32 accumulators) and the O / l rescale is skipped for the whole 32-query block.  Everything exponentiated in
                // this tile uses the m_run decided HERE, and O, l are rescaled in the same place, so nothing is ever at a stale scale.
                if (!__all((mx - m_run[b]) * scale_log2 <= AW_DEFER)) {       // also true for the first tile (m_run = -inf)
                    const float m_new = fmaxf(m_run[b], mx);                  // finite: every tile holds at least one valid key
                    const float alpha = fast_exp2((m_run[b] - m_new) * scale_log2);
                    m_run[b] = m_new;
                    l_run[b] *= alpha;
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[b][m][r] *= alpha;
                }
                const float mc = m_run[b] * scale_log2;
                f32x2 ps = {0.f, 0.f};
                const f32x2 c2 = {scale_log2, scale_log2}, mc2 = {-mc, -mc};
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        float p[8];
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            const f32x2 sv = {sacc[b][kb][8 * u + e], sacc[b][kb][8 * u + e + 1]};
                            const f32x2 a = __builtin_elementwise_fma(sv, c2, mc2);            // v_pk_fma_f32
                            p[e] = fast_exp2(a[0]);
                            p[e + 1] = fast_exp2(a[1]);
                            const f32x2 pv = {p[e], p[e + 1]};
                            ps += pv;                                                           // v_pk_add_f32
                        }
                        const uint4 pk = pack8(p);
                        pf[b][kb][u] = __builtin_bit_cast(bf16x8, pk);
                    }
                l_run[b] += ps[0] + ps[1];
            }

            // ---- O^T += V^T P^T : 4 k-steps (16 keys each) x 4 d blocks; the reads of k-step g+1 fly under the MFMAs of g
#define AW_PV(V, KB, U)                                                                                            \
            if (PRIO) __builtin_amdgcn_s_setprio(1);                                                                     \
            _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                                              \
                const bf16x8 vf = __builtin_shufflevector(V[2 * m], V[2 * m + 1], 0, 1, 2, 3, 4, 5, 6, 7);               \
                _Pragma("unroll") for (int b = 0; b < QW; ++b)                                                           \
                    o[b][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[b][KB][U], o[b][m], 0, 0, 0);               \
            }                                                                                                            \
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            aw_tr8<4096>(vB, va_[0], va_[1], va_[2], va_[3]);
            aw_wait_lgkm<8>(vA);
            AW_PV(vA, 0, 0)
            aw_tr8<8192>(vA, va_[0], va_[1], va_[2], va_[3]);
            aw_wait_lgkm<8>(vB);
            AW_PV(vB, 0, 1)
            aw_tr8<12288>(vB, va_[0], va_[1], va_[2], va_[3]);
            aw_wait_lgkm<8>(vA);
            AW_PV(vA, 1, 0)
            aw_wait_lgkm<0>(vB);
            AW_PV(vB, 1, 1)
#undef AW_PV
        } else {
            // ---- QW = 2.  One wave per SIMD: nothing but this wave's own instruction stream can run VALU work under its MFMAs, so
            // the exponentials are cut into four slices (k-step (kb, u): 8 scores per lane and block) and every slice but the
            // first is placed inside the MFMA group BEFORE the one that consumes it:
            //   statistics of both blocks (max, rescale decision)  ->  P(0,0)  |  PV(0,0) || P(0,1)  |  PV(0,1) || P(1,0)  |
            //   PV(1,0) || P(1,1)  |  PV(1,1)
            // Same values in the same order as the QW = 1 path (the row sums are accumulated slice by slice in its order), so
            // the result is bit-identical; PRIO builds add sched_group_barrier hints (one MFMA : five VALU / TRANS issues).
            float mcv[QW];
            f32x2 psv[QW];
#pragma unroll
            for (int b = 0; b < QW; ++b) {
                if ((t + 1) * AW_KT > L) {             // ragged last tile (wave-uniform): mask keys >= L
                    const int kbase = t * AW_KT + 4 * hi;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (kbase + kb * 32 + (r & 3) + 8 * (r >> 2) >= L) sacc[b][kb][r] = -INFINITY;
                }
                float mx = sacc[b][0][0];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[b][kb][r]);
                mx = aw_other_half_max(mx);
                if (!__all((mx - m_run[b]) * scale_log2 <= AW_DEFER)) {   // deferred rescale, per 32-query block (see the QW = 1 path)
                    const float m_new = fmaxf(m_run[b], mx);
                    const float alpha = fast_exp2((m_run[b] - m_new) * scale_log2);
                    m_run[b] = m_new;
                    l_run[b] *= alpha;
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[b][m][r] *= alpha;
                }
                mcv[b] = m_run[b] * scale_log2;
                psv[b] = f32x2{0.f, 0.f};
            }
            auto pexp = [&](auto bc, auto kbc, auto uc) {      // P slice (kb, u) of block b: 8 exponentials, packed to bf16
                constexpr int B = decltype(bc)::value, KB = decltype(kbc)::value, U = decltype(uc)::value;
                const f32x2 c2 = {scale_log2, scale_log2}, mc2 = {-mcv[B], -mcv[B]};
                float p[8];
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const f32x2 sv = {sacc[B][KB][8 * U + e], sacc[B][KB][8 * U + e + 1]};
                    const f32x2 a = __builtin_elementwise_fma(sv, c2, mc2);
                    p[e] = fast_exp2(a[0]);
                    p[e + 1] = fast_exp2(a[1]);
                    const f32x2 pv = {p[e], p[e + 1]};
                    psv[B] += pv;
                }
                const uint4 pk = pack8(p);
                pf[B][KB][U] = __builtin_bit_cast(bf16x8, pk);
            };
#define AW_I(v) std::integral_constant<int, (v)>{}
#define AW_P2(KB, U) pexp(AW_I(0), AW_I(KB), AW_I(U)); pexp(AW_I(1), AW_I(KB), AW_I(U))
#define AW_PV2(V, KB, U)                                                                                           \
            _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                                          \
                const bf16x8 vf = __builtin_shufflevector(V[2 * m], V[2 * m + 1], 0, 1, 2, 3, 4, 5, 6, 7);           \
                o[0][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[0][KB][U], o[0][m], 0, 0, 0);               \
                o[1][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[1][KB][U], o[1][m], 0, 0, 0);               \
            }
#define AW_SGB()                                                                                                   \
            if (PRIO) { _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                              \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      /* one MFMA */                               \
                __builtin_amdgcn_sched_group_barrier(0x402, 5, 0); } }  /* five VALU / TRANS issues in its shadow */
            AW_P2(0, 0);
            aw_tr8<4096>(vB, va_[0], va_[1], va_[2], va_[3]);
            aw_wait_lgkm<8>(vA);
            AW_PV2(vA, 0, 0)
            AW_P2(0, 1);
            AW_SGB()
            aw_tr8<8192>(vA, va_[0], va_[1], va_[2], va_[3]);
            aw_wait_lgkm<8>(vB);
            AW_PV2(vB, 0, 1)
            AW_P2(1, 0);
            AW_SGB()
            aw_tr8<12288>(vB, va_[0], va_[1], va_[2], va_[3]);
            aw_wait_lgkm<8>(vA);
            AW_PV2(vA, 1, 0)
            AW_P2(1, 1);
            AW_SGB()
            aw_wait_lgkm<0>(vB);
            AW_PV2(vB, 1, 1)
#undef AW_SGB
#undef AW_PV2
#undef AW_P2
#undef AW_I
#pragma unroll
            for (int b = 0; b < QW; ++b) l_run[b] += psv[b][0] + psv[b][1];
        }
        next_rows(t + 2);                              // (after the last counted wait: hipcc's own lgkmcnt(0) here is harmless)
        const int flip = (t & 1) ? -AW_TILE : AW_TILE;   // the other buffer of the K pair / V pair
#pragma unroll
        for (int ds = 0; ds < 8; ++ds) ka_[ds] += flip;
#pragma unroll
        for (int m = 0; m < 4; ++m) va_[m] += flip;
        __syncthreads();                               // tile t+1 landed (vmcnt(0)); everyone is done reading this tile's buffers
    }

    // ---- normalise and scatter.  Lane (q, hi) holds O[q][32 m + 8 rq + 4 hi + 0..3] in o[b][m][4 rq .. 4 rq + 3]
#pragma unroll
    for (int b = 0; b < QW; ++b) {
        const float l = aw_other_half_sum(l_run[b]);
        const float inv = 1.0f / l;
        const bool valid = qpos[b] < L;
        const int orow = valid ? out_rows[beg + qpos[b]] : 0;
        char* op = (char*)out + ((int64_t)orow * ld_out + (int64_t)head * AW_D) * 2 + hi * 16;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int rq = 0; rq < 4; rq += 2) {
                uint32_t ax = pack2bf(o[b][m][4 * rq + 0] * inv, o[b][m][4 * rq + 1] * inv);
                uint32_t ay = pack2bf(o[b][m][4 * rq + 2] * inv, o[b][m][4 * rq + 3] * inv);
                uint32_t bx = pack2bf(o[b][m][4 * rq + 4] * inv, o[b][m][4 * rq + 5] * inv);
                uint32_t by = pack2bf(o[b][m][4 * rq + 6] * inv, o[b][m][4 * rq + 7] * inv);
                // half exchange: lower lanes end up with d = 32 m + 8 rq + 0..7, upper lanes with 32 m + 8 (rq + 1) + 0..7
                const u32x2 sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                const u32x2 sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                if (valid) *(uint4*)(op + (32 * m + 8 * rq) * 2) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
            }
    }
}

// svr_set_option("attn_variant", v): A/B knob over the build variants -- 0 = default (8 waves; measured best on both window
// families, profiles/r2_attn_kbench.jsonl), 1 = 4 waves + s_setprio, 2 = 8 waves, 3 = 8 waves + s_setprio, 4 = 4 waves,
// 5 = 4 waves x 64 queries (one wave per SIMD; 5 .. 8 were written at the end of round 4 with no GPU minutes left: opt-in until measured),
// 6 = 5 + sched_group_barrier hints, 7 = 5 with K Q^T one tile ahead of the softmax (PIPE), 8 = 7 + hints,
// 9 / 10 = the default build with one wave of each SIMD at a higher issue priority (waves 0 .. 3 / the even waves)
int g_attn_variant = 0;

template <int NW, bool PRIO, int QW = 1, bool PIPE = false, int ASYM = 0>
static int launch_attn_win_t(const void* qkv, int64_t ld_qkv, void* out, int64_t ld_out, const int32_t* seq_rows,
                             const int32_t* out_rows, const int32_t* cu, int n_seq, int max_len, int heads, float scale,
                             hipStream_t s) {
    auto kern = attn_win_kernel<NW, PRIO, QW, PIPE, ASYM>;
    static uint64_t lds_attr_done = 0;               // per device (svr_common.h)
    {
        const int e = set_max_dynamic_lds((const void*)kern, AW_LDS, lds_attr_done);
        if (e != 0) return e;
    }
    constexpr int QB = NW * 32 * QW;
    if ((ld_qkv * 2) % 16 != 0) return -3;            // (attn_dispatch only sends 16-byte aligned row pitches here)
    const int qt = (max_len + QB - 1) / QB;
    const int64_t n_pairs = (int64_t)n_seq * heads;
    const int64_t blocks = 8 * ((n_pairs + 7) / 8) * qt;
    if (blocks > 0x7fffffff) return -2;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NW * 64), AW_LDS, s, (const bf16_t*)qkv, ld_qkv,
                       (bf16_t*)out, ld_out, seq_rows, out_rows, cu, heads, (int)n_pairs, qt,
                       scale * 1.4426950408889634f);
    return (int)hipGetLastError();
}

static int launch_attn_win(const void* qkv, int64_t ld_qkv, void* out, int64_t ld_out, const int32_t* seq_rows,
                           const int32_t* out_rows, const int32_t* cu, int n_seq, int max_len, int heads, float scale,
                           hipStream_t s) {
#define AW_ARGS qkv, ld_qkv, out, ld_out, seq_rows, out_rows, cu, n_seq, max_len, heads, scale, s
    switch (g_attn_variant) {
        case 1: return launch_attn_win_t<4, true>(AW_ARGS);
        case 3: return launch_attn_win_t<8, true>(AW_ARGS);
        case 4: return launch_attn_win_t<4, false>(AW_ARGS);
        case 5: return launch_attn_win_t<4, false, 2>(AW_ARGS);
        case 6: return launch_attn_win_t<4, true, 2>(AW_ARGS);
        case 7: return launch_attn_win_t<4, false, 2, true>(AW_ARGS);
        case 8: return launch_attn_win_t<4, true, 2, true>(AW_ARGS);
        case 9: return launch_attn_win_t<8, false, 1, false, 1>(AW_ARGS);
        case 10: return launch_attn_win_t<8, false, 1, false, 2>(AW_ARGS);
        default: return launch_attn_win_t<8, false>(AW_ARGS);
    }
#undef AW_ARGS
}

}  // namespace svr
