// LDS-DMA variant of the bf16 K-contiguous GEMM (included by tnn_gemm_bf16.hip inside its anonymous namespace; uses its
// BfArgs, bf16x8, f32x16, u32x4, u32x2, f32x4, f2bf, xcd_remap16 and the BEPI_* codes).
//
// Same product and the same 128 x 128 x 64 tile as the register-staged kernel — the M = 512 shapes of config E need
// >= 256 tiles to fill 256 CUs, so the tile cannot grow without split-K — but:
//   * operands go global -> LDS directly (buffer_load_dwordx4 ... lds): no staging VGPRs, no ds_write pass;
//   * the DMA writes lane-linear (wave-uniform base + lane * 16 B), so LDS rows cannot be padded; the bank conflicts of
//     the ds_read_b128 fragment reads are avoided with an XOR swizzle applied on BOTH sides: the 16-B chunk c of row r
//     lives in slot c ^ ((r >> 1) & 7) — the loading lane picks its global chunk accordingly, the reading lane its slot;
//   * an NS-stage LDS ring (32 KB per stage): the DMA of tile kt + NS is issued while tile kt is multiplied, each wave
//     waits only for its OWN share of the tile it is about to publish (counted vmcnt, never 0 in steady state when
//     NS > 2) and ONE raw s_barrier per K-tile both publishes tile kt + 1 and frees the stage of tile kt;
//   * SWAP (bf16 outputs): MFMA operands swapped like the fp32 kernel — the accumulator block is (A B^T)^T, lane l31
//     owns an output ROW and its registers 4 consecutive columns -> 8-B stores instead of 2-B ones.  fp32 outputs keep
//     lane = column: a store instruction then writes whole 128-B row segments, which the 268 MB dW outputs need (with the
//     swapped form every store touches 32 rows x 32 B and the dW product fell from 597 to 358 TFLOP/s).
// MEASURED (MI355X, tools/gemm_bf16_sweep.py, TFLOP/s on 512x8192x8192 -> bf16 / 8192x8192x512 -> f32 / 4096^3 -> f32 /
// 8192^3 -> bf16, uniform [-1, 1) operands):
//     register-staged, 4 waves, 2 workgroups per CU                      810 / 503 / 934 / 862
//     DMA, 8 waves of 64x32, 2 stages (2 workgroups per CU)   "dma8s"    794 / 597 / 978 / 900
//     DMA, 4 waves of 64x64, 2 stages (2 workgroups per CU)   "dma4s"    714 / 477 / 977 / 871
//     DMA, 8 waves of 64x32, 4 stages (1 workgroup per CU)    "dma8"     806 / 353 / 745 / 771
// Two resident workgroups matter more than the depth of the ring: one workgroup cannot overlap its own barrier, prologue
// and epilogue.  The M = 512 shapes have exactly one 128 x 128 tile per CU whatever the kernel (~800 TFLOP/s, 32 % of
// peak): more needs 256-wide tiles + split-K over this kernel's DMA / swizzle machinery — built in round 4: tnn_gemm_bf16_sk.h (JOURNAL.md).
namespace g8 {
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int ROWB = BK * 2;                 // bytes per LDS row (no padding)
constexpr int TILE_B = BM * ROWB;            // 16 KB per operand tile
constexpr int STAGE_B = 2 * TILE_B;          // A tile + B tile
constexpr int KK = BK / 16;
}  // namespace g8

#define TNN_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define TNN_VMCNT_LGKM0(n) asm volatile("s_waitcnt vmcnt(" #n ") lgkmcnt(0)" ::: "memory")

// one LDS-DMA instruction: 64 lanes x 16 B, global (buffer resource + per-lane byte offset + scalar offset) -> LDS at
// lds_dst + lane * 16.  A NON-template function on purpose: inside the kernel template the cast to the LDS address space
// is a dependent expression and hipcc's host pass then silently drops the whole instantiation (undefined __device_stub__)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_dst, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
}

// NW = 8: eight waves of 64 x 32 (2 per SIMD);  NW = 4: four waves of 64 x 64 (one per SIMD, a third less fragment traffic)
// NS = LDS stages: 4 -> one workgroup per CU, tile kt + 4 in flight; 2 -> two workgroups per CU, tile kt + 2 in flight
// SWAP: MFMA operands swapped -> lane = output row, 4 consecutive columns per register quad (8-B / 16-B stores, but every
// store instruction touches 32 rows x 32 B: partial lines).  !SWAP: lane = output column, a store instruction writes whole
// 128-B row segments (fp32) — what the 268 MB dW outputs want.
// ADAM (with !SWAP, fp32 product): the tile is a weight gradient and Adam's update of the matching 128 x 128 block of the
// fp32 master weights / moments, the bf16 working copy and its transpose happens in the epilogue (tnn_gemm_bf16_nt_adam).
template <int NW, int NS, bool SWAP, bool ADAM = false>
__global__ __launch_bounds__(NW * 64) void gemm_bf16_dma_kernel(BfArgs g) {
    static_assert(!(ADAM && SWAP), "the Adam epilogue is written for lane = column");
    constexpr int BM = g8::BM, BN = g8::BN, BK = g8::BK;
    constexpr int WN = NW == 8 ? 4 : 2, TM = 64, TN = BN / WN, MI = 2, NI = TN / 32;
    constexpr int ROWB = g8::ROWB, TILE_B = g8::TILE_B, STAGE_B = g8::STAGE_B, KK = g8::KK;
    constexpr int DJ = 16 / NW;                  // DMA instructions per wave, operand and tile (1 KB each)
    static_assert((NS & (NS - 1)) == 0 && NS >= 2, "stage count must be a power of two");
    // ADAM: + 16 KB behind the ring for the transposed bf16 image of half a tile (see the epilogue)
    __shared__ __attribute__((aligned(1024))) char lds[NS * STAGE_B + (ADAM ? 16384 : 0)];   // ONE shared object (a second one
                                                                             // makes hipcc drain the DMA queue before every ds_read)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int l31 = lane & 31, lhi = lane >> 5;

    const int nb = g.tiles_m * g.tiles_n;
    const int t = xcd_remap16((int)blockIdx.x, nb);
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * g.tiles_n;
    const int first_m = (t / per_group) * GROUP_M;
    const int gsz = min(g.tiles_m - first_m, GROUP_M);
    const int64_t m0 = (int64_t)(first_m + (t % per_group) % gsz) * BM;
    const int64_t n0 = (int64_t)((t % per_group) / gsz) * BN;
    const int nk = (int)(g.K / BK);

    // ---- DMA geometry: instruction gi = wid + NW j fills LDS bytes [gi * 1024, + 1024) of an operand tile = rows
    // 8 gi .. 8 gi + 7; lane L writes slot L % 8 of row 8 gi + L / 8 and therefore LOADS chunk slot ^ swz(row)
    uint32_t a_voff[DJ], b_voff[DJ];
#pragma unroll
    for (int j = 0; j < DJ; ++j) {
        const int row = 8 * (wid + NW * j) + (lane >> 3), slot = lane & 7;
        const int chunk = slot ^ ((row >> 1) & 7);
        const int64_t gm = m0 + row, gn = n0 + row;
        a_voff[j] = (uint32_t)(((gm < g.M ? gm : 0) * g.lda + chunk * 8) * 2);
        b_voff[j] = (uint32_t)(((gn < g.N ? gn : 0) * g.ldb + chunk * 8) * 2);
    }
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(g.A), 0, 0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(g.B), 0, 0xffffffffu, 0x00020000);
    auto issue_tile = [&](int kt) {
        char* stage = lds + (kt & (NS - 1)) * STAGE_B;
        const uint32_t koff = (uint32_t)kt * ROWB;
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            dma16(a_rsrc, stage + (wid + NW * j) * 1024, a_voff[j], koff);
            dma16(b_rsrc, stage + TILE_B + (wid + NW * j) * 1024, b_voff[j], koff);
        }
    };
    // "at most `later` whole tiles issued after the one I need may still be in flight" (2 DJ DMAs per tile and wave)
    auto wait_tiles = [&](int later, bool lgkm) {
        const int n = later * 2 * DJ;
        if (lgkm) {
            switch (n) {
                case 0: TNN_VMCNT_LGKM0(0); break;
                case 4: TNN_VMCNT_LGKM0(4); break;
                case 8: TNN_VMCNT_LGKM0(8); break;
                case 12: TNN_VMCNT_LGKM0(12); break;
                case 16: TNN_VMCNT_LGKM0(16); break;
                default: TNN_VMCNT_LGKM0(24); break;
            }
        } else {
            switch (n) {
                case 0: TNN_VMCNT(0); break;
                case 4: TNN_VMCNT(4); break;
                case 8: TNN_VMCNT(8); break;
                case 12: TNN_VMCNT(12); break;
                case 16: TNN_VMCNT(16); break;
                default: TNN_VMCNT(24); break;
            }
        }
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
    const int b_base = TILE_B + (wn * TN + l31) * ROWB;

    // ---- software pipeline (the fp32 kernel's, with the DMA in place of the register staging).  Fragment sets F(kt, kk),
    // kk = 0..3, are read half a tile ahead of their MFMAs:
    //     read F(kt,2), F(kt,3)            | MFMA F(kt,0), F(kt,1)
    //     wait own DMA share of tile kt+1, s_barrier   (tile kt+1 published; every wave has read all of tile kt)
    //     DMA tile kt+NS -> the stage of tile kt
    //     read F(kt+1,0), F(kt+1,1)        | MFMA F(kt,2), F(kt,3)
    bf16x8 fa[KK][MI], fb[KK][NI];
    auto read_frag = [&](int tile, int kk) {
        const char* stage = lds + (tile & (NS - 1)) * STAGE_B;
#pragma unroll
        for (int j = 0; j < NI; ++j)
            fb[kk][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(stage + b_base + j * 32 * ROWB + foff[kk]));
#pragma unroll
        for (int i = 0; i < MI; ++i)
            fa[kk][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(stage + a_base + i * 32 * ROWB + foff[kk]));
    };
    auto mfma_set = [&](int kk) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                if constexpr (SWAP)           // the block is (A B^T)^T, lane = output row
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kk][j], fa[kk][i], acc[i][j], 0, 0, 0);
                else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
            }
    };
#pragma unroll
    for (int p = 0; p < NS; ++p)
        if (p < nk) issue_tile(p);
    wait_tiles(min(nk - 1, NS - 1), false);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_frag(0, 0);
    read_frag(0, 1);
    for (int kt = 0; kt < nk; ++kt) {
        read_frag(kt, 2);
        read_frag(kt, 3);
        __builtin_amdgcn_sched_barrier(0);
        mfma_set(0);
        mfma_set(1);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
            wait_tiles(min(nk - 2 - kt, NS - 2), true);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + NS < nk) issue_tile(kt + NS);
            read_frag(kt + 1, 0);
            read_frag(kt + 1, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_set(2);
        mfma_set(3);
        __builtin_amdgcn_sched_barrier(0);
    }

    if constexpr (ADAM) {
        // Lane = column (whole 128-B row segments per instruction for the fp32 arrays).  Per 32 x 32 block a lane owns 16
        // rows of its column: p / m / v are requested for a quad of rows (r = 4 q .. 4 q + 3: 12 loads in flight), updated,
        // stored; the four bf16 results of the quad are consecutive in the TRANSPOSED copy's row -> one 8-B store there.
        // The gradient itself goes to C only when asked for (g.C != NULL): 4 of the 32 bytes per parameter that the
        // separate GEMM + Adam launches move are the write of the gradient and 4 its re-read.
        if (g.guard != nullptr && *g.guard != 0) return;
        const float ic1 = (float)(1.0 / (1.0 - g.pows[0])), ic2 = (float)(1.0 / (1.0 - g.pows[1]));
        const float omb1 = 1.f - g.b1, omb2 = 1.f - g.b2, lr = g.lr, eps = g.eps;
        float* gout = reinterpret_cast<float*>(g.C);
        if constexpr (NS * STAGE_B >= BM * BN * 4) {
            // INTERIOR tiles: the epilogue moves 28 B per element against the 2 x 2 B x K / 128 of the operands — it IS the
            // kernel (7 of every 8 bytes).  Lane = column means 4-B accesses (4.9 TB/s measured, the stand-alone optimizer
            // kernel reaches 6.7 with 16-B ones), so the accumulators go through the now idle LDS ring (128 x 128 fp32 =
            // its 64 KB exactly) and come back as 4 x 4 sub-blocks: thread (a, b) owns rows 4a..4a+3 x columns 4b..4b+3,
            // a wave instruction covers two whole 512-B tile rows of p / m / v (16-B loads and stores, non-temporal), the
            // bf16 copy gets 8-B stores and the transposed copy 8-B stores of 4 consecutive rows of one column.
            const bool interior = m0 + BM <= g.M && n0 + BN <= g.N && g.ldc % 4 == 0 && g.ldt % 4 == 0 &&
                                  ((reinterpret_cast<uintptr_t>(g.ap) | reinterpret_cast<uintptr_t>(g.am) |
                                    reinterpret_cast<uintptr_t>(g.av) | reinterpret_cast<uintptr_t>(gout)) & 15) == 0 &&
                                  (reinterpret_cast<uintptr_t>(g.aw16) & 7) == 0 && (reinterpret_cast<uintptr_t>(g.awT16) & 15) == 0 &&
                                  g.ldt % 8 == 0;
            if (interior) {                       // block-uniform
                float* tile = reinterpret_cast<float*>(lds);
                __syncthreads();                  // every wave is done with the last stage's fragments
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            tile[(wm * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * BN + wn * TN + ni * 32 + l31] = acc[mi][ni][r];
                __syncthreads();
                constexpr int PASSES = (BM / 4) * (BN / 4) / (NW * 64);
                const int cb = tid % (BN / 4);
                // Software pipeline over the 2 PASSES x 2 row-pairs of this thread's sub-blocks: the 6 x 16-B loads of stage
                // s + 1 are in flight while stage s is computed and stored (12 loads per thread, 24 KB per wave of 64 lanes:
                // the epilogue is the memory-bound part of the kernel and lives on the number of bytes in flight), within
                // 128 VGPRs so that TWO workgroups stay resident (all 4 rows of a sub-block at once: 134 VGPRs, one
                // workgroup, 460 us instead of 384 on 8192 x 8192 x 512).
                static_assert(PASSES == 2, "the stage schedule below is written for two passes");
                f32x4 pv[2][2], mv[2][2], vv[2][2];
                uint32_t hp[4][2];                         // the current sub-block's bf16 results, packed by row
                auto base_of = [&](int stage) -> int64_t {
                    const int ra = tid / (BN / 4) + (stage >> 1) * (NW * 64 / (BN / 4));
                    return (m0 + 4 * ra) * g.ldc + n0 + 4 * cb;
                };
                auto request = [&](int stage, int buf) {
                    const int64_t o0 = base_of(stage);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int64_t o = o0 + (2 * (stage & 1) + jj) * g.ldc;
                        pv[buf][jj] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g.ap + o));
                        mv[buf][jj] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g.am + o));
                        vv[buf][jj] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g.av + o));
                    }
                };
                request(0, 0);
#pragma unroll
                for (int stage = 0; stage < 4; ++stage) {
                    const int buf = stage & 1, half = stage & 1;
                    if (stage + 1 < 4) request(stage + 1, buf ^ 1);
                    const int ra = tid / (BN / 4) + (stage >> 1) * (NW * 64 / (BN / 4));
                    const int64_t o0 = base_of(stage);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = 2 * half + jj;
                        const int64_t o = o0 + j * g.ldc;
                        const f32x4 gv = *reinterpret_cast<const f32x4*>(tile + (4 * ra + j) * BN + 4 * cb);
                        uint32_t h[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float gi = gv[c];
                            mv[buf][jj][c] = mv[buf][jj][c] + omb1 * (gi - mv[buf][jj][c]);
                            vv[buf][jj][c] = vv[buf][jj][c] + omb2 * (gi * gi - vv[buf][jj][c]);
                            pv[buf][jj][c] = pv[buf][jj][c] + (-lr * (mv[buf][jj][c] * ic1) / (sqrtf(vv[buf][jj][c] * ic2) + eps));
                            h[c] = (uint32_t)f2bf(pv[buf][jj][c]);
                        }
                        __builtin_nontemporal_store(mv[buf][jj], reinterpret_cast<f32x4*>(g.am + o));
                        __builtin_nontemporal_store(vv[buf][jj], reinterpret_cast<f32x4*>(g.av + o));
                        __builtin_nontemporal_store(pv[buf][jj], reinterpret_cast<f32x4*>(g.ap + o));
                        hp[j][0] = h[0] | (h[1] << 16);
                        hp[j][1] = h[2] | (h[3] << 16);
                        if (g.aw16) *reinterpret_cast<u32x2*>(g.aw16 + o) = u32x2{hp[j][0], hp[j][1]};
                        if (gout) *reinterpret_cast<f32x4*>(gout + o) = gv;
                    }
                    if (half == 1 && g.awT16 != nullptr) {
                        // The transposed bf16 copy of this pass' 64 rows leaves through an LDS image [128 columns][64 rows]
                        // (16 KB behind the ring): written here as 8-B items (4 consecutive rows of one column), read back
                        // after a barrier as 16-B pieces so that 8 lanes store one whole 128-B segment of a W^T row.  (The
                        // direct form — every lane an 8-B store into a different W^T row, 16 KB apart — cost 49 of the
                        // kernel's 370 us for 7 % of its bytes.)  XOR swizzle on the 8-B slot index: row n keeps logical
                        // slot q at physical slot q ^ ((n >> 2) & 15) — the 16 lanes that write 16 different rows at the
                        // same q hit 16 different bank pairs.
                        char* timg = lds + NS * STAGE_B;
                        const int ral = tid / (BN / 4);                       // this pass' row quad 0..15
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int w = c >> 1, sh = 16 * (c & 1);
                            const uint32_t lo = ((hp[0][w] >> sh) & 0xffffu) | (((hp[1][w] >> sh) & 0xffffu) << 16);
                            const uint32_t hi = ((hp[2][w] >> sh) & 0xffffu) | (((hp[3][w] >> sh) & 0xffffu) << 16);
                            const int n = 4 * cb + c;
                            *reinterpret_cast<u32x2*>(timg + n * 128 + 8 * (ral ^ (cb & 15))) = u32x2{lo, hi};
                        }
                        __syncthreads();
                        {
                            // 128 rows x 8 pieces of 16 B = 1024 pieces, 2 per thread
#pragma unroll
                            for (int it = 0; it < 2; ++it) {
                                const int piece = tid + it * (NW * 64);
                                const int n = piece >> 3, k = piece & 7;          // physical slots 2k, 2k + 1 of row n
                                const int key = (n >> 2) & 15;
                                typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
                                u32x4_ v = *reinterpret_cast<const u32x4_*>(timg + n * 128 + 16 * k);
                                if (key & 1) v = u32x4_{v[2], v[3], v[0], v[1]};  // odd key: the two logical slots are swapped
                                const int q0 = (2 * k) ^ (key & ~1);              // first logical slot of the pair
                                *reinterpret_cast<u32x4_*>(g.awT16 + (n0 + n) * g.ldt + m0 + (stage >> 1) * 64 + 4 * q0) = v;
                            }
                        }
                        if (stage + 1 < 4) __syncthreads();                   // the next pass overwrites the image
                    }
                }
                return;
            }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int64_t col = n0 + wn * TN + ni * 32 + l31;
                const bool col_ok = col < g.N;
                const int64_t cc = col_ok ? col : 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // edge tiles only (interior ones left through the LDS path above): a quad of rows at a time
                    const int64_t rbase = m0 + wm * TM + mi * 32 + 8 * q + 4 * lhi;
                    float pv[4], mv[4], vv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int64_t o = (rbase + j < g.M ? rbase + j : 0) * g.ldc + cc;
                        pv[j] = g.ap[o];
                        mv[j] = g.am[o];
                        vv[j] = g.av[o];
                    }
                    uint32_t packed[2] = {0u, 0u};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float gi = acc[mi][ni][4 * q + j];
                        const float m1 = mv[j] + omb1 * (gi - mv[j]);
                        const float v1 = vv[j] + omb2 * (gi * gi - vv[j]);
                        const float p1 = pv[j] + (-lr * (m1 * ic1) / (sqrtf(v1 * ic2) + eps));
                        const bf16_t h = f2bf(p1);
                        packed[j >> 1] |= (uint32_t)h << (16 * (j & 1));
                        if (col_ok && rbase + j < g.M) {
                            const int64_t o = (rbase + j) * g.ldc + col;
                            __builtin_nontemporal_store(m1, g.am + o);
                            __builtin_nontemporal_store(v1, g.av + o);
                            __builtin_nontemporal_store(p1, g.ap + o);
                            if (g.aw16) g.aw16[o] = h;
                            if (gout) gout[o] = gi;
                        }
                    }
                    if (g.awT16 != nullptr && col_ok) {
                        bf16_t* dst = g.awT16 + col * g.ldt + rbase;
                        if (rbase + 3 < g.M) {
                            *reinterpret_cast<u32x2*>(dst) = u32x2{packed[0], packed[1]};
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (rbase + j < g.M) dst[j] = (bf16_t)(packed[j >> 1] >> (16 * (j & 1)));
                        }
                    }
                }
            }
        return;
    }
    if constexpr (!SWAP) {
        if constexpr (NS * STAGE_B >= BM * BN * 4) {
            // Plain fp32 output (the weight gradient of the data-parallel / keep-gradients step: 268 MB per 8192 x 8192
            // product), interior tiles: through the idle ring like the Adam form — the tile comes back row-major and leaves
            // as 16-B stores, a wave instruction covering two whole 512-B tile rows (lane = column wrote 128-B segments of
            // two rows per instruction, 4 B per lane: 143-147 us for the 8192 x 8192 x 512 product).
            const bool staged = !g.c_bf16 && g.epi == BEPI_PLAIN && m0 + BM <= g.M && n0 + BN <= g.N && g.ldc % 4 == 0 &&
                                (reinterpret_cast<uintptr_t>(g.C) & 15) == 0;
            if (staged) {                         // block-uniform
                float* tile = reinterpret_cast<float*>(lds);
                __syncthreads();                  // every wave is done with the last stage's fragments
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            tile[(wm * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * BN + wn * TN + ni * 32 + l31] = acc[mi][ni][r];
                __syncthreads();
                float* cout = reinterpret_cast<float*>(g.C);
                constexpr int PIECES = BM * (BN / 4) / (NW * 64);          // 16-B pieces per thread
#pragma unroll
                for (int it = 0; it < PIECES; ++it) {
                    const int p = tid + it * (NW * 64), row = p / (BN / 4), c4 = p % (BN / 4);
                    const f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * BN + 4 * c4);
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(cout + (m0 + row) * g.ldc + n0 + 4 * c4));
                }
                return;
            }
        }
        // epilogue, lane = column: col = l31, row = (r & 3) + 8 (r >> 2) + 4 lhi of each 32 x 32 block
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int64_t col = n0 + wn * TN + ni * 32 + l31;
                if (col >= g.N) continue;
                const float bias = (g.epi == BEPI_BIAS_ACT && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = m0 + wm * TM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (row >= g.M) continue;
                    float v = acc[mi][ni][r];
                    if (g.epi == BEPI_BIAS_ACT) {
                        v += bias;
                        if (g.act == TNN_ACT_RELU) v = v < 0.f ? (g.relu_sign ? -0.0f : 0.f) : fabsf(v);
                    } else if (g.epi == BEPI_MASK) {
                        if (g.Y[row * g.ldy + col] & 0x8000u) v = 0.f;
                    }
                    if (g.c_bf16) reinterpret_cast<bf16_t*>(g.C)[row * g.ldc + col] = f2bf(v);
                    else reinterpret_cast<float*>(g.C)[row * g.ldc + col] = v;
                }
            }
        return;
    }
    // ---- epilogue, bf16 output, interior tiles: through an LDS image of the output tile (tnn_gemm_bf16_sk.h has the account:
    // in the accumulator layout a store instruction writes 8-B pieces of 32 different rows).  Image [128][128] bf16 in the idle
    // ring, 8-B slot s of row r at s ^ (r & 15); read back as 16-B pieces, 16 lanes per 256-B row segment.
    {
        const bool staged = g.c_bf16 && m0 + BM <= g.M && n0 + BN <= g.N && g.ldc % 8 == 0 &&
                            (reinterpret_cast<uintptr_t>(g.C) & 15) == 0 &&
                            (g.epi != BEPI_MASK || (g.ldy % 8 == 0 && (reinterpret_cast<uintptr_t>(g.Y) & 15) == 0));
        if (staged) {                                       // block-uniform
            constexpr int IMG_ROWB = BN * 2;
            __syncthreads();                                // every wave is done with the ring
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int row = wm * TM + mi * 32 + l31;
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
            constexpr int PPR = BN / 8, PIECES = BM * PPR / (NW * 64);
            bf16_t* cout = reinterpret_cast<bf16_t*>(g.C);
#pragma unroll
            for (int it = 0; it < PIECES; ++it) {
                const int p = tid + it * (NW * 64), row = p / PPR, j = p % PPR;
                u32x4 v = *reinterpret_cast<const u32x4*>(lds + row * IMG_ROWB + ((j ^ ((row & 15) >> 1)) << 4));
                if (row & 1) v = u32x4{v[2], v[3], v[0], v[1]};      // odd rows: the two 8-B slots of the pair are swapped
                if (g.epi == BEPI_MASK) {
                    const u32x4 y = *reinterpret_cast<const u32x4*>(g.Y + (m0 + row) * g.ldy + n0 + 8 * j);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] &= ((y[e] & 0x8000u) ? 0u : 0xffffu) | ((y[e] & 0x80000000u) ? 0u : 0xffff0000u);
                }
                *reinterpret_cast<u32x4*>(cout + (m0 + row) * g.ldc + n0 + 8 * j) = v;
            }
            return;
        }
    }
    // ---- epilogue, edge tiles / fp32 outputs: block (mi, ni) is rows m0 + wm 64 + mi 32 + l31, register r the column
    // (r & 3) + 8 (r >> 2) + 4 lhi
    const bool vec_out = g.ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(g.C) & 15) == 0 &&
                         (g.epi != BEPI_MASK || (g.ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(g.Y) & 7) == 0));
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
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
#undef TNN_VMCNT
#undef TNN_VMCNT_LGKM0
