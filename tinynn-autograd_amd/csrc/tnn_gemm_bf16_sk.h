// 256-row tiles + split-K for the SKINNY bf16 products of config E (M = 512: forward z = a W and dX = dz W^T,
// core/ops.py:151,157 at bf16).  Included by tnn_gemm_bf16.hip inside its anonymous namespace, after tnn_gemm_bf16_types.h.
// (The geometries, hand-off protocols and ablation switches this form was chosen from live in the probe's own copy,
// tools/probes/gemm_bf16_sk_probe_kernel.h; this header is the shipped path only.)
//
// Why: with 128 x 128 tiles a 512 x 8192 x 8192 product has exactly one tile per CU and every CU pulls 2 x 2 MB of operands
// through the L2 -> LDS path (1 GB per product, 12 TB/s at the measured 87 us) — the operand stream INTO the CU bounds the
// shape.  A 256 x 128 tile cuts the bytes per flop by a quarter; to still fill 256 CUs the K range is split over TWO
// workgroups per tile, which hand HALF a tile to each other inside the launch:
//   * each of the two workgroups keeps the accumulator blocks of half the rows (slice s: the row blocks mi with
//     mi / (MI / 2) == s), stores the other half as a slab (register order: float4 i of thread t at [i][t] — every store
//     instruction writes 8 KB contiguous; write-through `sc1`, so no L2 write-back fence is needed), drains, bumps its flag,
//     waits for the partner's flag, adds what it received (p0 + p1: IEEE addition commutes, so the same bits whichever side
//     adds) and finishes its half of the output tile — half the slab bytes per workgroup, both directions in flight at once,
//     half an epilogue each;
//   * the two flags of a tile count launches in lockstep (own flag + 1 is the value to wait for in the partner's): nothing
//     is ever reset;
//   * each side waits for a workgroup that may not have FINISHED its K loop yet; it has been dispatched, though — the
//     partners are adjacent block indices and the host launches this kernel only when the whole grid fits the chip at once —
//     so the wait ends unless the device stops running dispatched workgroups.  The spin is BOUNDED all the same, and a wait
//     that runs out does not consume the slab: the workgroup stores TNN_FAULT_SPLITK_HANDOFF in the process's sticky fault
//     word (host-visible: tnn_stream_sync / tnn_memcpy_d2h fail from then on) and fills its half of the output with NaN.
// K loop: the LDS-DMA ring + XOR swizzle of tnn_gemm_bf16_dma.h (16-B chunk c of row r lives in slot c ^ ((r >> 1) & 7)),
// separate rings for the two operands — NSA stages of A (activations, L2-resident) and NSB stages of B (weights, streamed
// from HBM) — 8 waves, wave tile (256 / WM) x (BN / WN), fragments double-buffered per 16-deep k-step (the 128 accumulator
// registers of a 128 x 64 wave tile leave room for two fragment sets only), ONE raw s_barrier per K-tile: it publishes tile
// kt + 1 and frees the stages of tile kt, whose refill DMAs are issued BETWEEN the MFMAs of the tile's last k-step.
// Block -> (tile, slice) map (speed only): XCD x = block % 8 works on slice x % 2, so an XCD's L2 holds ONE K-slice of the
// activation panel, and the M-tiles that share a weight tile sit next to each other on the same XCD.
// Epilogue (interior bf16 tiles): through an LDS image of the finished half tile, which leaves as whole 512-B row segments;
// with g.CT set the SAME image is read a second time column-wise and the half tile's TRANSPOSE goes to CT [N][ldct] — the
// K-contiguous operand the dW product of the backward pass wants (core/ops.py:159-160: a^T and dz^T), so no transpose launch.
namespace sk {

template <int N>
__device__ __forceinline__ void wait_vm_lgkm0() {
    static_assert(N >= 0 && N <= 24 && N % 2 == 0, "unexpected DMA count");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 22) asm volatile("s_waitcnt vmcnt(22) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");
}

// one LDS-DMA instruction (64 lanes x 16 B -> lds_dst + lane * 16); a NON-template function, see tnn_gemm_bf16_dma.h
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_dst, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
}

typedef __attribute__((address_space(1))) unsigned gu32;

template <int BN, int WM, int WN, int NSA, int NSB>
__global__ __launch_bounds__(512) void gemm_bf16_sk_kernel(BfArgs g) {
    constexpr int BM = 256, ROWB = 128, KK = 4, S = 2;
    constexpr int A_TILE_B = BM * ROWB, B_TILE_B = BN * ROWB;
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
    constexpr int DJA = BM / 64, DJB = BN / 64;          // DMA instructions per wave, operand and K-tile (1 KB each)
    static_assert(WM * WN == 8, "eight waves");
    static_assert(NSA >= 2 && NSB >= 2, "double buffering at least");
    static_assert(MI % 2 == 0, "the hand-off splits the row blocks of a wave in two");
    constexpr int NSMIN = NSA < NSB ? NSA : NSB, NSMAX = NSA < NSB ? NSB : NSA;
    constexpr int LDS_B = NSA * A_TILE_B + NSB * B_TILE_B;
    static_assert(LDS_B <= 163840, "LDS");
    // DMAs issued after the last one tile kt + 1 needs, seen from the barrier of iteration kt (issue order per iteration:
    // A(j + NSA) then B(j + NSB)): the B share of that iteration when the rings differ, then NSA - 2 whole iterations
    constexpr int C_STEADY = (NSMIN - 2) * (DJA + DJB) + (NSA < NSB ? DJB : 0);
    constexpr int C_PROLOGUE = C_STEADY + DJA + DJB;
    __shared__ __attribute__((aligned(1024))) char lds[LDS_B];      // ONE shared object (tnn_gemm_bf16_dma.h)

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int l31 = lane & 31, lhi = lane >> 5;

    // ---- block -> (tile, slice)
    const int tiles = g.tiles_m * g.tiles_n, nb = tiles * S;
    int tile, slice;
    {
        const int b = (int)blockIdx.x;
        constexpr int G = 8 / S;
        if (nb % 8 == 0 && tiles % G == 0) {
            const int x = b & 7, j = b >> 3;
            slice = x % S;
            tile = (x / S) * (tiles / G) + j;
        } else {
            slice = b % S;
            tile = b / S;
        }
    }
    const int64_t m0 = (int64_t)(tile % g.tiles_m) * BM, n0 = (int64_t)(tile / g.tiles_m) * BN;
    const int nk = (int)(g.K / 64) / S;                 // K-tiles of this slice (> max(NSA, NSB), host-checked)
    const uint32_t k_byte0 = (uint32_t)slice * (uint32_t)nk * ROWB;

    // ---- DMA geometry: instruction gi = wid + 8 j fills rows 8 gi .. 8 gi + 7 of an operand tile; lane L writes slot L % 8
    // of row 8 gi + L / 8 and therefore LOADS chunk slot ^ swz(row)
    uint32_t a_voff[DJA], b_voff[DJB];
#pragma unroll
    for (int j = 0; j < DJA; ++j) {
        const int row = 8 * (wid + 8 * j) + (lane >> 3), chunk = (lane & 7) ^ ((row >> 1) & 7);
        const int64_t gm = m0 + row;
        a_voff[j] = (uint32_t)(((gm < g.M ? gm : 0) * g.lda + chunk * 8) * 2);
    }
#pragma unroll
    for (int j = 0; j < DJB; ++j) {
        const int row = 8 * (wid + 8 * j) + (lane >> 3), chunk = (lane & 7) ^ ((row >> 1) & 7);
        const int64_t gn = n0 + row;
        b_voff[j] = (uint32_t)(((gn < g.N ? gn : 0) * g.ldb + chunk * 8) * 2);
    }
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(g.A), 0, 0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(g.B), 0, 0xffffffffu, 0x00020000);
    char* const a_ring = lds;
    char* const b_ring = lds + NSA * A_TILE_B;
    auto issue_a1 = [&](int kt, int slot, int j) {
        dma16(a_rsrc, a_ring + slot * A_TILE_B + (wid + 8 * j) * 1024, a_voff[j], k_byte0 + (uint32_t)kt * ROWB);
    };
    auto issue_b1 = [&](int kt, int slot, int j) {
        dma16(b_rsrc, b_ring + slot * B_TILE_B + (wid + 8 * j) * 1024, b_voff[j], k_byte0 + (uint32_t)kt * ROWB);
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses: lane (row l31, k-group lhi) reads chunk 2 kk + lhi of its row; (row >> 1) & 7 == (l31 >> 1) & 7
    // for every block row offset used (multiples of 32)
    const int swz = (l31 >> 1) & 7;
    int foff[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) foff[kk] = ((2 * kk + lhi) ^ swz) * 16;
    const int a_base = (wm * TM + l31) * ROWB;
    const int b_base = (wn * TN + l31) * ROWB;

    bf16x8 fa[2][MI], fb[2][NI];
    auto read_frag = [&](int set, const char* a_st, const char* b_st, int kk) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
            fb[set][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(b_st + b_base + j * 32 * ROWB + foff[kk]));
#pragma unroll
        for (int i = 0; i < MI; ++i)
            fa[set][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(a_st + a_base + i * 32 * ROWB + foff[kk]));
    };
    auto mfma_set = [&](int set) {
        // operands swapped: the accumulator block is (A B^T)^T — lane l31 owns an output ROW, a register quad 4 consecutive
        // columns (8-B bf16 stores in the epilogue)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[set][j], fa[set][i], acc[i][j], 0, 0, 0);
    };

    // ---- prologue: "iterations" -NSB .. -1 of the steady-state issue order
#pragma unroll
    for (int j = -NSMAX; j < 0; ++j) {
        if (j + NSA >= 0) {
#pragma unroll
            for (int q = 0; q < DJA; ++q) issue_a1(j + NSA, j + NSA, q);
        }
        if (j + NSB >= 0) {
#pragma unroll
            for (int q = 0; q < DJB; ++q) issue_b1(j + NSB, j + NSB, q);
        }
    }
    wait_vm_lgkm0<C_PROLOGUE>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int a_cur = 0, b_cur = 0;                            // ring slots of tile kt
    read_frag(0, a_ring, b_ring, 0);
    // One K-tile.  STEADY: tiles kt + 1 .. kt + NSB all exist — no conditionals, the refill DMAs of the two freed stages are
    // spread between the MFMAs of k-step 3.  The accumulators are written at ONE place per k-step in both forms (MFMAs in
    // different branches of a conditional made hipcc copy the 128 accumulator registers through scratch every iteration).
    auto k_tile = [&](int kt, auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        const char* a_st = a_ring + a_cur * A_TILE_B;
        const char* b_st = b_ring + b_cur * B_TILE_B;
        read_frag(1, a_st, b_st, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_set(0);
        __builtin_amdgcn_sched_barrier(0);
        read_frag(0, a_st, b_st, 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_set(1);
        __builtin_amdgcn_sched_barrier(0);
        read_frag(1, a_st, b_st, 3);
        __builtin_amdgcn_sched_barrier(0);
        mfma_set(0);
        __builtin_amdgcn_sched_barrier(0);
        const int a_nxt = a_cur + 1 == NSA ? 0 : a_cur + 1, b_nxt = b_cur + 1 == NSB ? 0 : b_cur + 1;
        if (STEADY || kt + 1 < nk) {
            // every DMA counted in C_STEADY exists as long as iteration kt - 1 issued its B share
            if constexpr (STEADY) wait_vm_lgkm0<C_STEADY>();
            else wait_vm_lgkm0<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            read_frag(0, a_ring + a_nxt * A_TILE_B, b_ring + b_nxt * B_TILE_B, 0);
            if (STEADY || kt + NSA < nk) {
#pragma unroll
                for (int q = 0; q < DJA; ++q) issue_a1(kt + NSA, a_cur, q);
            }
            if (STEADY || kt + NSB < nk) {
#pragma unroll
                for (int q = 0; q < DJB; ++q) issue_b1(kt + NSB, b_cur, q);
            }
        }
        mfma_set(1);
        if constexpr (STEADY) {
            __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);        // the next tile's first fragments
#pragma unroll
            for (int q = 0; q < DJA + DJB; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);          // one VMEM read (the DMA)
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        a_cur = a_nxt;
        b_cur = b_nxt;
    };
    int kt = 0;
    // two K-tiles per trip: hipcc waits lgkmcnt(0) at a loop head (the fragments just requested included), inside a trip it
    // counts exactly
    for (; kt + NSMAX + 1 < nk; kt += 2) {
        k_tile(kt, std::true_type{});
        k_tile(kt + 1, std::true_type{});
    }
    for (; kt + NSMAX < nk; ++kt) k_tile(kt, std::true_type{});
    for (; kt < nk; ++kt) k_tile(kt, std::false_type{});

    // ---- split-K hand-off (symmetric, see the header)
    constexpr int MH = MI / 2;                             // accumulator row blocks this workgroup finishes ...
    const int mi0 = slice * MH;                            // ... starting at this one
    bool partner_ok = true;                                // block-uniform
    {
        gu32* flag = (gu32*)(g.sk_cnt + 2 * tile);
        const __amdgpu_buffer_rsrc_t ws_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            g.sk_ws + (int64_t)tile * S * (BM * BN), 0, 0xffffffffu, 0x00020000);
        constexpr uint32_t SLAB_B = BM * BN * 4;
        const int give0 = (1 - slice) * MH;               // first row block of the half the partner finishes
        const uint32_t base = (uint32_t)slice * SLAB_B + (uint32_t)tid * 16;
#pragma unroll
        for (int i = 0; i < MH; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ch = (i * NI + j) * 4 + q;
                    // both halves are read with compile-time indices (a run-time block index spills the accumulators)
                    const f32x4 lo = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                    const f32x4 hi = {acc[MH + i][j][4 * q], acc[MH + i][j][4 * q + 1], acc[MH + i][j][4 * q + 2], acc[MH + i][j][4 * q + 3]};
                    const u32x4 v = __builtin_bit_cast(u32x4, give0 == 0 ? lo : hi);
                    __builtin_amdgcn_raw_buffer_store_b128(v, ws_rsrc, base + (uint32_t)ch * 8192, 0, 16 /* sc1: write-through */);
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // EVERY storing wave drains its write-through stores
        __syncthreads();
        int* const ok_lds = reinterpret_cast<int*>(lds);                 // (every wave is past the ring: the barrier above)
        if (tid == 0) {
            const unsigned mine = __hip_atomic_load(flag + slice, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            __hip_atomic_store(flag + slice, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            bool there = false;
            while (!(there = (int)(__hip_atomic_load(flag + (1 - slice), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - mine) >= 0) &&
                   ++spins < (1u << 24))
                __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (!there && g.fault != nullptr)
                __hip_atomic_store(g.fault, (int)TNN_FAULT_SPLITK_HANDOFF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            ok_lds[0] = there ? 1 : 0;
        }
        __syncthreads();
        partner_ok = ok_lds[0] != 0;
        __syncthreads();                                    // ok_lds sits where the output image is about to be built
        constexpr int NCHH = MH * NI * 4, CB = 4;
        if (partner_ok) {
#pragma unroll
            for (int c0 = 0; c0 < NCHH; c0 += CB) {
                __builtin_amdgcn_sched_barrier(0);
                f32x4 in[CB];
#pragma unroll
                for (int c = 0; c < CB; ++c)
                    in[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                ws_rsrc, (uint32_t)tid * 16 + (uint32_t)(c0 + c) * 8192, (uint32_t)(1 - slice) * SLAB_B, 16 /* sc1 */));
#pragma unroll
                for (int c = 0; c < CB; ++c) {
                    const int ch = c0 + c, i = ch / (NI * 4), j = (ch / 4) % NI, q = ch % 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // p0 + p1 on both sides (IEEE addition commutes: the same bits)
                        if (slice == 0) acc[i][j][4 * q + e] = acc[i][j][4 * q + e] + in[c][e];
                        else acc[MH + i][j][4 * q + e] = in[c][e] + acc[MH + i][j][4 * q + e];
                    }
                }
            }
        } else {
            // the partner's slab never arrived: it is NOT consumed; this half of the output is poisoned
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = __builtin_nanf("");
        }
    }

    // ---- epilogue, interior bf16 tiles: through an LDS image of the output tile.  In the accumulator layout a lane owns a ROW
    // (8-B pieces of 32 different rows per store instruction: 64 partial lines — measured 12 us for the 8 MB of a
    // 512 x 8192 output); the image [128][BN] bf16 (the idle ring) is written as 8-B items, slot s of row r at
    // s ^ (r & 15) (16 lanes that write 16 rows at one s hit 16 different bank pairs), and read back as 16-B pieces, 32 lanes
    // per 512-B row segment: whole lines, and the ReLU mask of dX (BEPI_MASK) is read with the same coalesced geometry.
    // The image holds this workgroup's half of the rows only, compactly: wave row wm's MH * 32 rows at wm * MH * 32.
    const bool staged = g.c_bf16 && m0 + BM <= g.M && n0 + BN <= g.N && g.ldc % 8 == 0 &&
                        (reinterpret_cast<uintptr_t>(g.C) & 15) == 0 &&
                        (g.epi != BEPI_MASK || (g.ldy % 8 == 0 && (reinterpret_cast<uintptr_t>(g.Y) & 15) == 0)) &&
                        (g.CT == nullptr || (g.ldct % 8 == 0 && (reinterpret_cast<uintptr_t>(g.CT) & 15) == 0));
    if (staged) {                                           // block-uniform
        constexpr int IMG_ROWB = BN * 2;
        static_assert(BM * IMG_ROWB <= LDS_B, "the output image lives in the ring");
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            if ((mi / MH) * MH != mi0) continue;            // block-uniform
            const int row = wm * (MH * 32) + (mi % MH) * 32 + l31;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = wn * TN + ni * 32 + 8 * q + 4 * lhi;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float x = acc[mi][ni][4 * q + j];
                        if (g.epi == BEPI_BIAS_ACT) {
                            x += g.bias ? g.bias[n0 + col + j] : 0.f;
                            if (g.act == TNN_ACT_RELU) x = x < 0.f ? (g.relu_sign ? -0.0f : 0.f) : fabsf(x);
                        }
                        v[j] = x;
                    }
                    const u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    *reinterpret_cast<u32x2*>(lds + row * IMG_ROWB + (((col >> 2) ^ (row & 15)) << 3)) = pk;
                }
        }
        __syncthreads();
        constexpr int PPR = BN / 8;                          // 16-B pieces per row
        constexpr int HROWS = BM / 2;                        // rows of the half tile
        constexpr int PIECES = HROWS * PPR / 512;            // per thread
        bf16_t* cout = reinterpret_cast<bf16_t*>(g.C);
        const bool want_t = g.CT != nullptr;                 // block-uniform
#pragma unroll
        for (int it = 0; it < PIECES; ++it) {
            const int p = tid + it * 512, irow = p / PPR, j = p % PPR;        // irow: row of the image
            const int row = (irow / (MH * 32)) * TM + mi0 * 32 + irow % (MH * 32);      // row of the tile
            char* const src = lds + irow * IMG_ROWB + ((j ^ ((irow & 15) >> 1)) << 4);
            u32x4 v = *reinterpret_cast<const u32x4*>(src);
            if (irow & 1) v = u32x4{v[2], v[3], v[0], v[1]};  // odd rows: the two 8-B slots of the pair are swapped
            if (g.epi == BEPI_MASK) {
                const u32x4 y = *reinterpret_cast<const u32x4*>(g.Y + (m0 + row) * g.ldy + n0 + 8 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // bf16 pair e: element 2e in the low half; a set sign bit in Y (the -0.0 of a negative pre-activation) zeroes it
                    const uint32_t keep = ((y[e] & 0x8000u) ? 0u : 0xffffu) | ((y[e] & 0x80000000u) ? 0u : 0xffff0000u);
                    v[e] &= keep;
                }
                // the transposed copy below reads the image again: the masked piece goes back where it came from
                if (want_t) *reinterpret_cast<u32x4*>(src) = (irow & 1) ? u32x4{v[2], v[3], v[0], v[1]} : v;
            }
            *reinterpret_cast<u32x4*>(cout + (m0 + row) * g.ldc + n0 + 8 * j) = v;
        }
        if (want_t) {
            // CT [n0 + c][m0 + row]: the half tile is four runs of 32 consecutive rows (wave row wm: tile rows wm TM + mi0 32 ..
            // + 31 = image rows wm 32 .. + 31), i.e. per column four 64-B runs.  A thread builds 16 B = 8 consecutive rows of one
            // column from eight 2-B image reads; lanes run along the rows first (4 lanes = one 64-B run), then along the columns.
            if (g.epi == BEPI_MASK) __syncthreads();
            static_assert(MH == 1, "the transposed read-back is written for one 32-row block per wave row");
            constexpr int T_PIECES = BN * (HROWS / 8) / 512;  // per thread
#pragma unroll
            for (int it = 0; it < T_PIECES; ++it) {
                const int p = tid + it * 512, rg = p & 3, c = (p >> 2) % BN, run = (p >> 2) / BN;   // run = wave row wm
                const int irow0 = run * 32 + rg * 8;
                uint32_t w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r0 = irow0 + 2 * e, r1 = r0 + 1;
                    const uint32_t lo = *reinterpret_cast<const uint16_t*>(lds + r0 * IMG_ROWB + (((c >> 2) ^ (r0 & 15)) << 3) + (c & 3) * 2);
                    const uint32_t hi = *reinterpret_cast<const uint16_t*>(lds + r1 * IMG_ROWB + (((c >> 2) ^ (r1 & 15)) << 3) + (c & 3) * 2);
                    w[e] = lo | (hi << 16);
                }
                const int64_t trow = m0 + run * TM + mi0 * 32 + rg * 8;
                *reinterpret_cast<u32x4*>(g.CT + (n0 + c) * g.ldct + trow) = u32x4{w[0], w[1], w[2], w[3]};
            }
        }
        return;
    }
    // ---- epilogue, edge tiles / fp32 outputs: block (mi, ni) is rows m0 + wm TM + mi 32 + l31, register r the column
    // (r & 3) + 8 (r >> 2) + 4 lhi
    const bool vec_out = g.ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(g.C) & 15) == 0 &&
                         (g.epi != BEPI_MASK || (g.ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(g.Y) & 7) == 0));
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        if ((mi / MH) * MH != mi0) continue;                // the partner finishes the other half
        const int64_t row = m0 + wm * TM + mi * 32 + l31;
        if (row >= g.M) continue;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t col = n0 + wn * TN + ni * 32 + 8 * q + 4 * lhi;
                if (col >= g.N) continue;
                const bool full = vec_out && col + 3 < g.N;
                float v[4];
                uint16_t ymask[4] = {0, 0, 0, 0};
                if (g.epi == BEPI_MASK) {
                    if (full) {
                        const u32x2 yv = *reinterpret_cast<const u32x2*>(g.Y + row * g.ldy + col);
                        ymask[0] = (uint16_t)yv.x; ymask[1] = (uint16_t)(yv.x >> 16);
                        ymask[2] = (uint16_t)yv.y; ymask[3] = (uint16_t)(yv.y >> 16);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (col + j < g.N) ymask[j] = g.Y[row * g.ldy + col + j];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = acc[mi][ni][4 * q + j];
                    if (g.epi == BEPI_BIAS_ACT) {
                        x += (g.bias && col + j < g.N) ? g.bias[col + j] : 0.f;
                        if (g.act == TNN_ACT_RELU) x = x < 0.f ? (g.relu_sign ? -0.0f : 0.f) : fabsf(x);
                    } else if (g.epi == BEPI_MASK) {
                        if (ymask[j] & 0x8000u) x = 0.f;
                    }
                    v[j] = x;
                }
                if (g.c_bf16) {
                    bf16_t* dst = reinterpret_cast<bf16_t*>(g.C) + row * g.ldc + col;
                    if (full) {
                        u32x2 pk;
                        pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
                        pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
                        *reinterpret_cast<u32x2*>(dst) = pk;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (col + j < g.N) dst[j] = f2bf(v[j]);
                    }
                    if (g.CT != nullptr) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (col + j < g.N) g.CT[(col + j) * g.ldct + row] = f2bf(v[j]);
                    }
                } else {
                    float* dst = reinterpret_cast<float*>(g.C) + row * g.ldc + col;
                    if (full) {
                        *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (col + j < g.N) dst[j] = v[j];
                    }
                }
            }
    }
}

}  // namespace sk
