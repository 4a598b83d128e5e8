// gemm_w4r_kernel (round 4, ABI v7): gemm_w4q_kernel with the WEIGHTS streamed from a fragment-ordered copy straight into registers and
// the ACTIVATIONS brought into LDS by LDS-DMA.  (Included by svr_gemm.hip, inside namespace svr, behind gemm_w4q_kernel, whose tile
// walk, LDS row layout, XOR key and epilogues it keeps.)
//
// Why.  Round 3's phase counts said gemm_w4q_kernel's K loop is limited by the LDS (profiles/r3_gemm_w4_ablations.txt section 11).
// Priced with the rates of MI355X_MICROARCH.md: per K tile a workgroup moved 64 KiB into LDS with ds_write_b128 (~79 B/clk: ~830 LDS
// cycles, issued from the SIMDs' VGPR read path) and read 128 KiB of fragments back (256 B/clk: 512 cycles) under 2 048 cycles of MFMAs.
//   * The weights are static, so they do not need LDS to get into fragment order: svr_gemm_pack_frag() stores them once as
//     [128-row panel][k32 step][16-row block J][lane][8 bf16] -- 1 KiB per wave instruction, 8 KiB contiguous per (panel, k32 step) --
//     and every wave loads the 8 + 8 fragments of its 128 columns per K tile with 16 coalesced buffer_load_dwordx4 (the two waves
//     that share a column half ask for the same lines within a K tile of each other: L1).
//   * The activations still need the transposition, but not the registers: buffer_load_dwordx4 ... lds writes a piece where
//     ds_write_b128 put it (the XOR is on the source side), without a VGPR round trip, a counted wait per piece or a store issue.
//   LDS per K tile: 32 KiB of DMA writes + 64 KiB of fragment reads; 64 staging VGPRs and 16 ds_writes per wave and K tile are gone.
// Measured (MI355X, same box, the NaDiT's four shapes at M = 291 600; profiles/r4_gemm_w4r_ablations.txt): qkv -4.6 %, attn-out into
// the fp32 stream -7.2 %, mlp-in SwiGLU -3.8 %, mlp-out -1.3 % against gemm_w4q_kernel; results bit-identical (same MFMAs, same k
// order).  The ablations in the same file say what is left: with the A path removed altogether the K loop runs 22 % faster, with the
// weight loads removed 11 % -- global loads themselves, not the LDS, are now what the MFMA pipe waits for (one wave per SIMD: nothing
// else to issue) -- and a version with NO LDS at all (A fragments loaded row-wise, 64 B per row and instruction) ran 35 % SLOWER: sixteen
// half-used lines per instruction are more than the L1 takes (section 3; kernel withdrawn).
//
// Schedule of K tile f (stage s = f & 1); registers at its start: AX = A fragments of k32 half 0 (read from stage s in K tile f - 1),
// BX / BY = the weight fragments of both halves of K tile f.  MFMA order is J outer (slot S: J = S >> 3, I = S & 7,
// acc[I][J] += B[J] x A[I]), so a weight fragment is live for eight slots and its registers are reloaded with K tile f + 1's fragment
// right behind its last use -- a prefetch distance of two halves minus eight slots (~1 900 cycles, about a microsecond).
//   half 0 (X):  slot 8 J: counted vmcnt -> BX[J] has landed;  slot 8 J + 7: reload BX[J];  AY reads (half 1 of stage s) in slots 10 + 6 i.
//   half 1 (Y):  same for BY;  slot 6: vmcnt(12) (this wave's pieces of K tile f + 1 are in LDS) + lgkmcnt(0) + THE barrier (every
//                wave's pieces are, and every wave's reads of stage s are done);  AX reads of K tile f + 1 in slots 10 + 4 i;  the eight
//                pieces of K tile f + 2 -> stage s in slots 8 + 4 q;  stage flip and the A cursor behind.
//   LDS: [stage 0: 32 KiB][stage 1: 32 KiB][epilogue parking: 96 KiB] (at a tile's end one stage holds the next tile's first K tile and
//   the other is being written with its second).
// The 24 loads of a K tile are issued in the same order every time, so every wait is a counted vmcnt naming the register it is for:
//   BX0..7 (slots 8 J + 7 of half 0) | half 1: Y0 (7) D0 (8) D1 (12) Y1 (15) D2 (16) D3 (20) Y2 (23) D4 (24) D5 (28) Y3 (31) D6 (32)
//   D7 (36) Y4 (39) Y5 (47) Y6 (55) Y7 (63)
//   -> loads issued behind the awaited one: BX[J] 23; BY[J] 23 21 21 21 21 23 23 23; the pieces before the barrier: 12 (Y4..Y7 and the
//   eight BX reloads were issued behind D7).  The prologue issues one whole period in that order, so the counts hold from the first
//   K tile on; nothing else touches vmcnt inside the K loop (the previous tile's stores are drained behind its epilogue).
// (The first version of this kernel staged the activations through 32 VGPRs and ds_write_b128 like gemm_w4q_kernel -- svr_set_option
// ("gemm_w4r", 2) in round 4; 1-3 % slower, and its bit-equality test failed once in round 5's full GPU run: removed.)

// W [N, K] (row-major bf16; N % 128 == 0, K % 32 == 0) -> fragment order: 16-byte unit
//   (((n / 128) * (K / 32) + k32) * 8 + (n % 128) / 16) * 64 + lane  =  W[(n & ~15) + (lane & 15)][k32 * 32 + (lane >> 4) * 8 .. + 7]
__global__ __launch_bounds__(256) void gemm_pack_frag_kernel(const bf16_t* __restrict__ W, uint4* __restrict__ out, int N, int K) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)N * K / 8) return;
    const int lane = (int)(idx & 63);
    const int64_t u = idx >> 6;
    const int J = (int)(u & 7);
    const int64_t v = u >> 3;
    const int nks = K / 32;
    const int ks = (int)(v % nks), p = (int)(v / nks);
    const int n = p * 128 + J * 16 + (lane & 15), k = ks * 32 + (lane >> 4) * 8;
    out[idx] = *(const uint4*)(W + (int64_t)n * K + k);
}

template <int OFF> SVR_DEVICE void w4r_bload(bf16x8& r, const w4p_u32x4& rsrc, uint32_t voff, uint32_t soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(r) : "v"(voff), "s"(rsrc), "s"(soff), "n"(OFF) : "memory");
}
template <int N> SVR_DEVICE void w4r_wait_frag(bf16x8& r) {           // counted vmcnt naming the fragment whose data must be there
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(N));
}

// DBG (builds with -DSVR_ABLATIONS only; results invalid): 1 no K-loop barrier | 2 no weight loads in the K loop | 4 no A loads (LDS-DMA pieces)
// in the K loop | 8 no fragment reads in the K loop | 32 A loads always from the same (cache-hot) K tile |
// 64 weight loads always from the same K tile   (svr_set_option("pipe_abl", 500 + bits); profiles/r4_gemm_w4r_ablations.txt)
constexpr int w4s_by_count(int J) { return J >= 1 && J <= 4 ? 21 : 23; }
SVR_DEVICE void w4s_dma(unsigned lds_wave_base, const w4p_u32x4& rsrc, uint32_t voff, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_wave_base), "v"(voff), "s"(rsrc), "s"(soff) : "memory");     // (m0 is reserved: hipcc neither allocates it nor, on gfx950, keeps a value of its own in it)
}
// (Other placements of the LDS operations -- AY reads in the first 16 slots of half 0; barrier in slot 1 of half 1 with the AX reads right
// behind it and the pieces in slots 20 .. 48 -- measured within +-0.5 % of this one: profiles/r4_gemm_w4r_ablations.txt section 4.)
template <int DBG = 0>
__global__ __launch_bounds__(W4_THREADS, 1) void gemm_w4r_kernel(const svr_gemm_args a) {
#ifndef SVR_ABLATIONS
    static_assert(DBG == 0, "measurement variants (results invalid on purpose) exist only in -DSVR_ABLATIONS builds; the product library instantiates DBG = 0");
#endif
    constexpr int BAR_SLOT = 6;
    constexpr int S1 = 32768;                             // byte offset of stage 1
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_m = (a.M + W4_T - 1) / W4_T;
    const int tiles_n = a.N / W4_T;
    const int tiles = tiles_m * tiles_n;
    const int nwg = gridDim.x;
    constexpr int GM = 4;
    const int group_size = GM * tiles_n;
    auto tile_origin = [&](int t, int& m0, int& n0) {
        const int group = t / group_size;
        const int first_m = group * GM;
        const int gm = min(tiles_m - first_m, GM);
        m0 = __builtin_amdgcn_readfirstlane((first_m + (t % group_size) % gm) * W4_T);
        n0 = __builtin_amdgcn_readfirstlane(((t % group_size) / gm) * W4_T);
    };
    int t = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
    int m0, n0;
    tile_origin(t, m0, n0);
    const int wm = wave >> 1, wn = wave & 1;
    const int srow = wave * 8 + (lane >> 3);
    const int chunk_src = (lane & 7) ^ ((srow >> 1) & 7);
    const int nk = a.K / BK;

    // ---- A cursor (K tile kl of output tile tl: two K tiles ahead of the MFMAs, across tile boundaries), as in gemm_w4q_kernel
    int tl = t, kl = 0;
    w4p_u32x4 Arsrc;
    uint32_t Akoff = 0;
    const uint32_t aoff0 = (uint32_t)((int64_t)srow * a.lda * 2) + chunk_src * 16;
    const uint32_t arowblk = (uint32_t)(32 * a.lda * 2);
    auto point_a = [&](int lm0) {
        const uint64_t base = (uint64_t)(uintptr_t)a.A + (uint64_t)lm0 * (uint64_t)a.lda * 2;
        const uint64_t left = (uint64_t)(a.M - lm0) * (uint64_t)a.lda * 2;
        Arsrc[0] = __builtin_amdgcn_readfirstlane((uint32_t)base);
        Arsrc[1] = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32) & 0xffffu);
        Arsrc[2] = __builtin_amdgcn_readfirstlane(left > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)left);
        Arsrc[3] = 0x00020000u;
        Akoff = 0;
    };
    auto advance_a = [&]() {
        ++kl;
        const bool same = kl < nk;
        Akoff = same ? Akoff + BK * 2 : Akoff;
        if (__builtin_expect(!same, 0)) {
            if (tl + nwg < tiles) {
                tl += nwg; kl = 0;
                int lm0, ln0;
                tile_origin(tl, lm0, ln0);
                point_a(lm0);
            } else {
                kl = nk - 1;
            }
        }
    };
    // ---- B cursor (K tile kb of output tile tb: ONE K tile ahead): this wave's 128-row panel of the fragment-ordered copy
    int tb = t, kb = 0;
    w4p_u32x4 Brsrc;
    uint32_t Bsoff = 0;                                   // byte offset of the cursor's K tile, k32 half 0 (half 1: + 8 KiB)
    const uint32_t panel_bytes = (uint32_t)a.K * 256u;    // 128 rows x K x 2 B
    const uint32_t boff = (uint32_t)lane * 16u;
    auto point_b = [&](int ln0) {
        const uint64_t bb = (uint64_t)(uintptr_t)a.W_frag + (uint64_t)((ln0 >> 7) + wn) * (uint64_t)panel_bytes;
        Brsrc[0] = __builtin_amdgcn_readfirstlane((uint32_t)bb);
        Brsrc[1] = __builtin_amdgcn_readfirstlane((uint32_t)(bb >> 32) & 0xffffu);
        Brsrc[2] = __builtin_amdgcn_readfirstlane(panel_bytes);
        Brsrc[3] = 0x00020000u;
        Bsoff = 0;
    };
    auto advance_b = [&]() {
        ++kb;
        const bool same = kb < nk;
        Bsoff = same ? Bsoff + 16384u : Bsoff;
        if (__builtin_expect(!same, 0)) {
            if (tb + nwg < tiles) {
                tb += nwg; kb = 0;
                int lm0, ln0;
                tile_origin(tb, lm0, ln0);
                point_b(ln0);
            } else {
                kb = nk - 1;                              // (behind the last tile: the same K tile again, never used)
            }
        }
    };
    point_a(m0);
    point_b(n0);

    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    unsigned dmaS = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(wave * 1024));     // this wave's slice of the CURRENT stage
    const int l15 = lane & 15, kq = lane >> 4;
    const unsigned key = (unsigned)((l15 >> 1) & 7);
    const unsigned rA = lds0 + (unsigned)((wm * 128 + l15) * 128);
    unsigned rdA[2];                                      // [k32 half] of the CURRENT stage; fragment i (16 rows = 2048 bytes) is an immediate
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) rdA[kh] = rA + ((((unsigned)(4 * kh + kq)) ^ key) << 4);
    f32x4 acc[8][8];
    bf16x8 AX[8], AY[8], BX[8], BY[8];

#define W4_FENCE() __builtin_amdgcn_sched_barrier(0)
#define W4R_C(v) std::integral_constant<int, (v)>{}
#define W4S_DMA(BASE, Q) w4s_dma((BASE) + (Q) * 4096u, Arsrc, aoff0, Akoff + (Q) * arowblk)
#define W4R_LDBX(J) w4r_bload<((J) & 3) * 1024>(BX[J], Brsrc, boff, Bsoff + ((J) >> 2) * 4096u)
#define W4R_LDBY(J) w4r_bload<((J) & 3) * 1024>(BY[J], Brsrc, boff, Bsoff + 8192u + ((J) >> 2) * 4096u)
#define W4S_LANDED() asm volatile("s_waitcnt vmcnt(0)" : "+v"(BX[0]), "+v"(BX[1]), "+v"(BX[2]), "+v"(BX[3]), "+v"(BX[4]), "+v"(BX[5]), \
                                  "+v"(BX[6]), "+v"(BX[7]), "+v"(BY[0]), "+v"(BY[1]), "+v"(BY[2]), "+v"(BY[3]), "+v"(BY[4]), "+v"(BY[5]), \
                                  "+v"(BY[6]), "+v"(BY[7]))
#define W4R_READ_X_ALL() do { w4_rd<0 * 2048>(AX[0], rdA[0]); w4_rd<1 * 2048>(AX[1], rdA[0]); w4_rd<2 * 2048>(AX[2], rdA[0]); \
        w4_rd<3 * 2048>(AX[3], rdA[0]); w4_rd<4 * 2048>(AX[4], rdA[0]); w4_rd<5 * 2048>(AX[5], rdA[0]); w4_rd<6 * 2048>(AX[6], rdA[0]); \
        w4_rd<7 * 2048>(AX[7], rdA[0]); } while (0)

    // ---- prologue (once per workgroup): A K tile 0 -> registers -> stage 0; then ONE PERIOD of the steady-state load order (weights
    // of K tile 0, A pieces of K tile 1), so that the counted waits of the first K tile find the sequence they count in
    // A K tile 0 -> stage 0; then one period of the steady-state order: weights of K tile 0, A K tile 1 -> stage 1
    W4S_DMA(dmaS, 0); W4S_DMA(dmaS, 1); W4S_DMA(dmaS, 2); W4S_DMA(dmaS, 3); W4S_DMA(dmaS, 4); W4S_DMA(dmaS, 5); W4S_DMA(dmaS, 6); W4S_DMA(dmaS, 7);
    advance_a();
    W4_FENCE();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    W4_FENCE();
    W4R_LDBX(0); W4R_LDBX(1); W4R_LDBX(2); W4R_LDBX(3); W4R_LDBX(4); W4R_LDBX(5); W4R_LDBX(6); W4R_LDBX(7);
    W4R_LDBY(0); W4S_DMA(dmaS + S1, 0); W4S_DMA(dmaS + S1, 1); W4R_LDBY(1); W4S_DMA(dmaS + S1, 2); W4S_DMA(dmaS + S1, 3);
    W4R_LDBY(2); W4S_DMA(dmaS + S1, 4); W4S_DMA(dmaS + S1, 5); W4R_LDBY(3); W4S_DMA(dmaS + S1, 6); W4S_DMA(dmaS + S1, 7);
    W4R_LDBY(4); W4R_LDBY(5); W4R_LDBY(6); W4R_LDBY(7);
    advance_a();
    W4_FENCE();
    w4_wait_lgkm_n<0>();
    __builtin_amdgcn_s_barrier();
    W4_FENCE();
    W4R_READ_X_ALL();
    W4_FENCE();

    int st = 0;
    int kt = 0;
    // one slot = one MFMA + at most one LDS store, one LDS read, one A load and one weight load behind it
    auto slot0 = [&](auto sc) {                           // half 0: sets AX, BX
        constexpr int S = decltype(sc)::value, J = S >> 3, I = S & 7;
        if constexpr (I == 0 && !(DBG & 2)) { w4r_wait_frag<23>(BX[J]); W4_FENCE(); }
        w4q_mfma(acc[I][J], BX[J], AX[I]);
        W4_FENCE();
        if constexpr (S == 1 && !(DBG & 64)) advance_b(); // the reloads of this K tile fetch the next one's weights
        constexpr int RY = (S >= 10 && S <= 52 && (S - 10) % 6 == 0) ? (S - 10) / 6 : -1;
        if constexpr (RY >= 0 && !(DBG & 8)) w4_rd<RY * 2048>(AY[RY], rdA[1]);
        if constexpr (I == 7 && !(DBG & 2)) W4R_LDBX(J);
        W4_FENCE();
    };
    auto slot1 = [&](auto sc) {                           // half 1: sets AY, BY
        constexpr int S = decltype(sc)::value, J = S >> 3, I = S & 7;
        if constexpr (S == BAR_SLOT) {                    // THE barrier of the K tile
            if constexpr (!(DBG & 4)) w4_wait_vmcnt<12>();    // this wave's pieces of the next K tile are in LDS
            w4_wait_lgkm_n<0>();
            if constexpr (!(DBG & 1)) __builtin_amdgcn_s_barrier();
            W4_FENCE();
            rdA[0] += st ? (unsigned)-S1 : (unsigned)S1;              // the AX reads below come from the other stage
            W4_FENCE();
        }
        if constexpr (I == 0 && !(DBG & 2)) { w4r_wait_frag<w4s_by_count(J)>(BY[J]); W4_FENCE(); }
        w4q_mfma(acc[I][J], BY[J], AY[I]);
        W4_FENCE();
        constexpr int RX = (S >= 10 && S <= 38 && (S - 10) % 4 == 0) ? (S - 10) / 4 : -1;
        constexpr int DQ = (S >= 8 && S <= 36 && (S - 8) % 4 == 0) ? (S - 8) / 4 : -1;      // piece of K tile f + 2 -> the stage just freed
        if constexpr (RX >= 0 && !(DBG & 8)) w4_rd<RX * 2048>(AX[RX], rdA[0]);
        if constexpr (DQ >= 0 && !(DBG & 4)) W4S_DMA(dmaS, DQ);
        if constexpr (S == 54) {
            const unsigned d = st ? (unsigned)-S1 : (unsigned)S1;
            rdA[1] += d; dmaS += d;
            st ^= 1;
        }
        if constexpr (S == 58 && !(DBG & 32)) advance_a();
        if constexpr (I == 7 && !(DBG & 2)) W4R_LDBY(J);
        if constexpr (S == 63) { if (kt + 1 == nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        W4_FENCE();
    };
#define W4R_8(F, B) F(W4R_C(B)); F(W4R_C(B + 1)); F(W4R_C(B + 2)); F(W4R_C(B + 3)); F(W4R_C(B + 4)); F(W4R_C(B + 5)); F(W4R_C(B + 6)); F(W4R_C(B + 7))
#define W4R_64(F) W4R_8(F, 0); W4R_8(F, 8); W4R_8(F, 16); W4R_8(F, 24); W4R_8(F, 32); W4R_8(F, 40); W4R_8(F, 48); W4R_8(F, 56)

    for (;;) {                                            // the output tiles of this workgroup
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (kt = 0; kt < nk; ++kt) {
            w4_wait_lgkm_n<0>();                          // AX (read in the previous half 1) is there
            W4_FENCE();
            W4R_64(slot0);
            w4_wait_lgkm_n<0>();                          // AY and this half's A stores (the last one ten slots back)
            W4_FENCE();
            W4R_64(slot1);
        }
        W4_FENCE();
        W4S_LANDED();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        {
            int lane_e;                                   // (rebuilt here, by an asm hipcc cannot hoist: no register kept for it across the K loop)
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
            const int wave_e = wave, tid_e = wave * 64 + lane_e;
            char* const park = smem + 65536;
            if (a.epilogue == SVR_EPI_SWIGLU && !a.out_f32)
                epilogue_swiglu_bf16_m16<W4_THREADS, W4P_EPI>(a, acc, park, m0, n0, tid_e, lane_e, wave_e);
            else
                epilogue_through_lds<W4_T, W4_T, 128, 128, W4_THREADS, W4P_EPI, false, 2, 2, 0, true, 16, false>(a, acc, park, m0, n0, tid_e, lane_e, wave_e);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);               // (hipcc's own vmcnt(0): the stores of this tile are out before the K loop counts again)
        t += nwg;
        if (t >= tiles) break;
        tile_origin(t, m0, n0);
        __syncthreads();
        W4_FENCE();
        W4R_READ_X_ALL();
        W4_FENCE();
    }
#undef W4_FENCE
#undef W4R_C
#undef W4S_DMA
#undef W4S_LANDED
#undef W4R_LDBX
#undef W4R_LDBY
#undef W4R_READ_X_ALL
#undef W4R_8
#undef W4R_64
}

