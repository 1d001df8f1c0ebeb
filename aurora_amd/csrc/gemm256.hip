// 256x256x64 MFMA GEMM for gfx950 with a staggered two-group schedule (large-M shapes: ViT layers, Llama prefill).
//
// Same contract, operand formats and epilogues as gemm.hip (C = A[M,K] x W[N,K]^T, W fragment-packed, A row-major);
// the difference is the pipeline:
//  * 8 waves (512 threads) as 2 (M) x 4 (N); each wave owns 128 x 64 outputs = 8 x 4 accumulator tiles.
//  * LDS (all 160 KiB): activation half-tiles double-buffered (2 x 2 x 16 KiB), weight half-tiles TRIPLE-buffered
//    (3 x 2 x 16 KiB) - weights stream from HBM / MALL and need the longer flight time, activations hit in L2.
//  * A K-tile is 4 phases; a phase = {LDS reads of one register sub-tile, issue ONE half-tile of LDS-DMA prefetch}
//    -> s_barrier -> {16 MFMAs on one accumulator quadrant} -> s_barrier.
//  * Waves 4-7 run ONE barrier behind waves 0-3, so on every SIMD one wave's MFMA burst coincides with its partner's
//    read/DMA-issue segment (guide: 8-phase template, T3/T4/T5).
//  * Prefetch issue order A1(t+1), W0(t+2), W1(t+2), A0(t+2) in phases 0..3 of K-tile t: every half-tile is in flight
//    for >= 4 phases (weights 6-7), the only wait is a COUNTED s_waitcnt vmcnt(6) in phase 3 (three half-tiles stay in
//    flight across the K-tile boundary), and every slot is re-filled >= 2 barriers after its last ds_read.
//      RAW: the wait sits before phase 3's first barrier, the first read of the new tile is two barriers later
//           (one for the staggered group).  WAR: see the table in DESIGN.md section 4.
//  * PERSISTENT workgroups (one per CU) walk the output tiles, and the NEXT tile's prologue (its first 7 half-tiles of LDS-DMA)
//    is issued before the CURRENT tile's epilogue: a 256x256 tile costs ~20 us on top of its K loop when every tile is its own
//    workgroup (dispatch + cold prologue round trip + store drain, one workgroup per CU because of the 160 KiB of LDS; measured
//    with a K sweep: 23 us at K = 128, +1.45 us per 64 of K - tools/gemm_probe.py), i.e. 40 % of a ViT tile (K = 1280) and 18 %
//    of a prefill tile (K = 4096).  With the overlap only the store drain (vmcnt(0) before the next tile's first barrier) is
//    left exposed.  LDS is free for the next prologue as soon as the K loop's last, group-aligning barrier has been passed.
#include "gemm_epilogue.h"
#include "tile_order.h"

// Which phase of K-tile t issues which half-tile of LDS-DMA prefetch: none | A1(t+1), W0(t+2) | W1(t+2) | A0(t+2) - the read-heavy phase 0
// (12 fragment reads) issues nothing, the issue ORDER A1, W0, W1, A0 fixes the counted wait vmcnt(6) of phase 3.  Round 4 measured the
// segments with in-kernel stamps (profiles/r04_gemm_segments.log): with one half-tile per phase (rounds 1-3) the load segments of phases 0
// and 3 outlast the partner group's 16-MFMA burst (329 / 222-287 cycles against 292) while phases 1 and 2 idle at the barrier; this
// placement gained 3-6 % on every shape of the path, same bits.  An LDS-DMA instruction costs its wave 60-90 cycles wherever it sits (issuing
// it inside the MFMA bursts lengthens the bursts by what it takes off the load segments); without any operand DMA the same kernel runs 1.28x
// faster - the price of feeding a 256 x 256 tile.  Also built, measured and removed in rounds 3-5 (lab variants of this kernel, in the
// history up to commit 282d3cc; results in profiles/r03_contention_lab.log, r04_gemm_segments.log, r04_gemm_store_policy.log,
// r04_gemm_epilogue_*.log): nt / sc1 operand DMAs, a K-tile-major activation image, a start-up stagger of the persistent workgroups,
// and a continuous pipeline across tile boundaries (bit-exact, 1-2 % slower: the prologue burst was not what the epilogue waits for).
__device__ __forceinline__ void g2_glds16(const void* gsrc_lane, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GPTR(gsrc_lane), LPTR(lds_wave_base), 16, 0, 0);
}

#define G2_LDS (160 * 1024)
#define G2_SLOT 16384
#define G2_ABUF (2 * G2_SLOT)          // one activation buffer (2 half-tiles)
#define G2_WBASE (2 * G2_ABUF)         // weight region starts after the two activation buffers
#define G2_WBUF (2 * G2_SLOT)          // one weight buffer (2 half-tiles), three of them

#define G2_FENCE() asm volatile("" ::: "memory")
#define G2_BARRIER()                         \
    do {                                     \
        __builtin_amdgcn_sched_barrier(0);   \
        G2_FENCE();                          \
        __builtin_amdgcn_s_barrier();        \
        G2_FENCE();                          \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)

// per-thread LDS-DMA sources of one output tile: 2 instructions per half-tile
struct G2Src {
    const half_t* a[2][2];
    const half_t* w[2][2];
};
__device__ __forceinline__ void g2_sources(const GemmArgs& a, int bm, int bn, int w, int lane, G2Src& src) {
    const int K32 = a.K >> 5;
    const int m0 = bm * 256;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int grp = w * 2 + i;
            const int row = grp * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            int m = m0 + h * 128 + row;
            m = m < a.M ? m : a.M - 1;
            src.a[h][i] = a.A + (int64_t)m * a.lda + c * 8;
            const int f = grp;                      // fragment of the half-tile: n16 = f >> 1, kk = f & 1
            src.w[h][i] = a.W + ((int64_t)(bn * 16 + h * 8 + (f >> 1)) * K32 + (f & 1)) * AUR_FRAG_HALVES + lane * 8;
        }
}
// K-tile kt of the tile lands in A buffer kt % 2 / W buffer kt % 3
__device__ __forceinline__ void g2_stage_a(const G2Src& src, char* smem, int w, int h, int kt) {
    char* dst = smem + (kt & 1) * G2_ABUF + h * G2_SLOT;
    g2_glds16(src.a[h][0] + (int64_t)kt * 64, dst + (w * 2 + 0) * 1024);
    g2_glds16(src.a[h][1] + (int64_t)kt * 64, dst + (w * 2 + 1) * 1024);
}
__device__ __forceinline__ void g2_stage_w(const G2Src& src, char* smem, int w, int h, int kt, int wb) {
    char* dst = smem + G2_WBASE + wb * G2_WBUF + h * G2_SLOT;
    g2_glds16(src.w[h][0] + (int64_t)kt * 2 * AUR_FRAG_HALVES, dst + (w * 2 + 0) * 1024);
    g2_glds16(src.w[h][1] + (int64_t)kt * 2 * AUR_FRAG_HALVES, dst + (w * 2 + 1) * 1024);
}
// prologue of a tile: K-tile 0 completely (8 instructions) into A buffer 0 / W buffer 0, plus W0, W1, A0 of K-tile 1 (6 instructions that may
// stay in flight) into W buffer 1 / A buffer 1
__device__ __forceinline__ void g2_prologue(const GemmArgs& a, const G2Src& src, char* smem, int w) {
    g2_stage_a(src, smem, w, 0, 0);
    g2_stage_a(src, smem, w, 1, 0);
    g2_stage_w(src, smem, w, 0, 0, 0);
    g2_stage_w(src, smem, w, 1, 0, 0);
    if ((a.K >> 6) > 1) {
        g2_stage_w(src, smem, w, 0, 1, 1);
        g2_stage_w(src, smem, w, 1, 1, 1);
        g2_stage_a(src, smem, w, 0, 1);
    }
}

// K loop of one tile.  `first`: this is the workgroup's first tile - its prologue was issued just before and K-tile 1's three half-tiles
// may stay in flight (counted wait); later tiles' prologues were issued before the previous tile's epilogue: drain everything.
template <int EPI, bool VMODE>
__device__ __forceinline__ void g2_mainloop(const GemmArgs& a, const G2Src& src, char* smem, f4 (&acc)[4][8], int w, int lane, bool first) {
    const int wr = w >> 2, wc = w & 3;
    const int r = lane & 15, g = lane >> 4;
    const int nkt = a.K >> 6;
    auto stage_a = [&](int h, int kt) { g2_stage_a(src, smem, w, h, kt); };
    auto stage_w = [&](int h, int kt, int wb) { g2_stage_w(src, smem, w, h, kt, wb); };

    // fragment read offsets inside a buffer
    const int sw = (r >> 1) & 7;
    const int a_off0 = wr * G2_SLOT + r * 128 + (((0 * 4 + g) ^ sw) << 4);       // kk = 0
    const int a_off1 = wr * G2_SLOT + r * 128 + (((1 * 4 + g) ^ sw) << 4);       // kk = 1
    const int w_off = (wc >> 1) * G2_SLOT + (wc & 1) * 8192 + lane * 16;

    // af[uu*2+kk]: 4 m-tiles of the current M half; wf0 / wf1 [tt*2+kk]: 2 n-tiles of N half 0 / 1.  N half 0 is used by the
    // first and the last quadrant of a K-tile and stays in registers in between (LDS read bandwidth is the co-limiter:
    // 24 KiB per wave per K-tile instead of 28)
    h8 af[8], wf0[4], wf1[4];
    auto read_a = [&](const char* buf, int mh) {
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
            af[uu * 2 + 0] = *(const h8*)(buf + a_off0 + (mh * 4 + uu) * 2048);
            af[uu * 2 + 1] = *(const h8*)(buf + a_off1 + (mh * 4 + uu) * 2048);
        }
    };
    auto read_w = [&](const char* buf, int nh, h8 (&wf)[4]) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            wf[tt * 2 + 0] = *(const h8*)(buf + w_off + (nh * 2 + tt) * 2048);
            wf[tt * 2 + 1] = *(const h8*)(buf + w_off + (nh * 2 + tt) * 2048 + 1024);
        }
    };
#define G2_COMPUTE(NH, MH, wf)                                                                                \
    do {                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                      \
            _Pragma("unroll") for (int tt = 0; tt < 2; ++tt)                                                  \
                _Pragma("unroll") for (int uu = 0; uu < 4; ++uu) {                                            \
                    if (VMODE)                                                                               \
                        acc[(NH) * 2 + tt][(MH) * 4 + uu] = mfma16(af[uu * 2 + kk], wf[tt * 2 + kk], acc[(NH) * 2 + tt][(MH) * 4 + uu]); \
                    else                                                                                     \
                        acc[(NH) * 2 + tt][(MH) * 4 + uu] = mfma16(wf[tt * 2 + kk], af[uu * 2 + kk], acc[(NH) * 2 + tt][(MH) * 4 + uu]); \
                }                                                                                            \
        __builtin_amdgcn_s_setprio(0);                                                                       \
    } while (0)

    // ---- the prologue was issued by the caller: K-tile 0 must have landed, W0, W1, A0 of K-tile 1 may stay in flight
    if (first && nkt > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G2_BARRIER();
    if (wr == 1) G2_BARRIER();                    // stagger: waves 4-7 run one barrier behind

    int wb = 0;                                            // weight buffer of K-tile t: t % 3
    for (int t = 0; t < nkt; ++t) {
        const char* abuf = smem + (t & 1) * G2_ABUF;
        const char* wbuf = smem + G2_WBASE + wb * G2_WBUF;
        const int wb2 = wb == 0 ? 2 : wb - 1;              // W buffer of K-tile t + 2
        const bool n1 = t + 1 < nkt, n2 = t + 2 < nkt;     // block-uniform
        // phase 0: quadrant (n half 0, m half 0)
        read_a(abuf, 0);
        read_w(wbuf, 0, wf0);
        G2_BARRIER();
        G2_COMPUTE(0, 0, wf0);
        G2_BARRIER();
        // phase 1: (n half 1, m half 0)
        read_w(wbuf, 1, wf1);
        if (n1) stage_a(1, t + 1);
        if (n2) stage_w(0, t + 2, wb2);
        G2_BARRIER();
        G2_COMPUTE(1, 0, wf1);
        G2_BARRIER();
        // phase 2: (n half 1, m half 1)
        read_a(abuf, 1);
        if (n2) stage_w(1, t + 2, wb2);
        G2_BARRIER();
        G2_COMPUTE(1, 1, wf1);
        G2_BARRIER();
        // phase 3: (n half 0, m half 1); retire K-tile t+1's half-tiles, keep W0, W1, A0 of t+2 in flight
        if (n2) {
            stage_a(0, t + 2);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        G2_BARRIER();
        G2_COMPUTE(0, 1, wf0);
        G2_BARRIER();
        wb = wb == 2 ? 0 : wb + 1;
    }
    if (wr == 0) G2_BARRIER();                    // match the stagger barrier of waves 4-7
#undef G2_COMPUTE
}

// EPI_ROW epilogue through LDS: full-line stores.
// In the accumulator layout a lane owns 4 consecutive columns of one row per 16x16 tile, so the direct epilogue issues 32
// 8-byte stores per lane whose wave-instructions each touch 16 rows x 32 B - store-ISSUE-bound (guide T21): measured 16-18 us per
// 256x256 tile on top of a K loop of 1.6 us per 64 of K (tools/gemm_probe.py with the stores stubbed out: 4.6 us vs 22.6 us per tile
// at K = 128, 33.7 vs 49.9 at K = 1280), i.e. a third of a ViT tile.  Here every wave transposes its 128 x 64 block through a
// private 16-row fp32 slab in LDS (row stride 272 B: conflict-free ds_write_b128, 16-byte aligned ds_read_b128) and a lane then
// owns 8 consecutive columns: ONE coalesced 16-byte residual load and ONE 16-byte store per lane, 8 rows x 128 B per
// wave-instruction (16 instead of 32 store instructions, whole 128-byte lines).  Arithmetic is unchanged - bias, activation and
// residual are added in fp32 in the same order before the single rounding to fp16 - so the result stays bit-identical to the
// 128x128 kernel's epilogue.  The slabs live in the LDS the NEXT tile's prologue does not touch (weight buffer 2 and the second
// half of activation buffer 1): the prologue's DMA is already in flight while this runs.
#define G2_SLAB_STRIDE 272
#define G2_SLAB_BYTES (16 * G2_SLAB_STRIDE)
// One 16-row block (u) of the wave's 128 x 64 region.  U is a TEMPLATE parameter on purpose: as `#pragma unroll for (int u ..)` the loop
// sits at the edge of LLVM's pragma-unroll threshold - round 4's third activation branch (act_other_f) pushed the body over it, the loop
// stayed rolled, `acc[t][u]` became a runtime index and hipcc moved the whole accumulator array to SCRATCH (`.private_segment_fixed_size
// 528`: 32 + 32 stores and 32 loads of 1 KiB per wave and tile; ViT fc1 938 -> 590 TF/s, found in the .s and in tools/gemm_probe.py before
// it shipped).  With a compile-time U every accumulator index is static whatever the body grows to (tests/test_static_asm.py pins 0
// bytes of scratch for every kernel of this file).
// ACT is a template parameter too (dispatched once per tile in g2_epilogue_row): the epilogue is VALU-issue-bound - both waves of
// every SIMD run it at the same time, ~43-46 k cycles per tile in rounds 2-3 against a 61 k-cycle K loop at K = 1280
// (in-kernel stamps, profiles/r04_gemm_segments.log) - so the activation ladder is resolved outside the 32 (u, t) blocks instead of inside each.
// ACT == -2 (GT_OTHER: projector, patch embedding, microbenchmarks): the activation is resolved at run time inside the blocks.
// One output vector of the wide epilogue: a non-temporal store when the launch asks for it (GemmArgs.nt_out: outputs larger
// than the L2s together; engine.hip ctx_gemm).  Every round of tiles writes 256 x 128 KiB - the capacity of
// the eight L2s - in one burst, because all CUs reach their epilogue together; written with the default policy the output evicts the operand
// panels the next tiles share through L2 and the K loops AFTER the epilogue pay for it (round 4, profiles/r04_gemm_store_policy.log: the per-tile fixed cost is 9.7 us with up to 128 CUs running and 16.4 us with 256; a K = 128 sweep 500 -> 310 us, K = 1280
// 1180 -> 1085 us, prefill gate/up -3 %, with `nt`; sc1 / sc0 sc1 write-through stores gain less).  Nothing on this path re-reads C from L2:
// the consumer is the next launch, and C is 40-800 MB.
template <typename V>
__device__ __forceinline__ void g2_store_c(bool nt, half_t* dst, const V& v) {
    if (nt) __builtin_nontemporal_store(v, (V*)dst);
    else *(V*)dst = v;
}
template <int U, int ACT>
__device__ __forceinline__ void g2_epilogue_row_u(const GemmArgs& a, f4 (&acc)[4][8], const f4 (&bias)[4], const h8 (&res)[8][2], int mb, int nb, int lane, char* slab) {
    const int r = lane & 15, g = lane >> 4;
    const int row0 = lane >> 3, chunk = lane & 7;
    const int n = nb + chunk * 8;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[t][U][i] + bias[t][i];
        if constexpr (ACT == ACT_QUICK_GELU) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = quick_gelu_f(v[i]);
        } else if constexpr (ACT == ACT_GELU) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = gelu_erf_f(v[i]);
        } else if constexpr (ACT == -2) {                 // GT_OTHER: resolved at run time
            if (a.act == ACT_QUICK_GELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = quick_gelu_f(v[i]);
            } else if (a.act == ACT_GELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = gelu_erf_f(v[i]);
            } else if (a.act >= ACT_SILU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = act_other_f(v[i], a.act);
            }
        }
        *(f4*)(slab + r * G2_SLAB_STRIDE + (t * 16 + 4 * g) * 4) = v;
    }
    asm volatile("" ::: "memory");                       // LDS writes above, reads below: same wave, program order
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int row = hh * 8 + row0;
        const int m = mb + U * 16 + row;
        const f4 x0 = *(const f4*)(slab + row * G2_SLAB_STRIDE + chunk * 32);
        const f4 x1 = *(const f4*)(slab + row * G2_SLAB_STRIDE + chunk * 32 + 16);
        if (m >= a.M || n >= a.n_real) continue;
        if (ACT == ACT_SILU_MUL || (ACT == -2 && a.act == ACT_SILU_MUL)) {      // (gate, up) interleaved: 4 outputs at columns n / 2 .. n / 2 + 3
            h4 o;
            o[0] = (half_t)(silu_f(x0[0]) * x0[1]);
            o[1] = (half_t)(silu_f(x0[2]) * x0[3]);
            o[2] = (half_t)(silu_f(x1[0]) * x1[1]);
            o[3] = (half_t)(silu_f(x1[2]) * x1[3]);
            half_t* dst = a.C + (int64_t)m * a.ldc + (n >> 1);
            if (n + 8 <= a.n_real) g2_store_c(a.nt_out != 0, dst, o);
            else g2_store_c(a.nt_out != 0, dst, h2{o[0], o[1]});             // n_real % 4 == 0: the chunk holds 4 real columns
        } else {
            float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
            const bool full = n + 8 <= a.n_real;
            if (a.resid) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += (float)res[U][hh][i];      // lanes of a partial chunk hold zeros in 4 .. 7: those lanes are not stored
            }
            h8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (half_t)v[i];
            half_t* dst = a.C + (int64_t)m * a.ldc + n;
            if (full) g2_store_c(a.nt_out != 0, dst, o);
            else g2_store_c(a.nt_out != 0, dst, h4{o[0], o[1], o[2], o[3]});
        }
    }
    asm volatile("" ::: "memory");
}
// The wide epilogue issues NO load after its first store.  Rounds 2-3 loaded the residual row (and looked up `out_rows[m]`) inside each of
// the 16 row blocks: hipcc answers a load next to in-flight LDS-DMA with `s_waitcnt vmcnt(0)` (guide, "three .s-level traps" (b)), and
// because stores and loads share vmcnt each block then waited for the PREVIOUS block's stores to be acknowledged by memory - 16 dependent
// store round trips per tile = 38-46 k cycles, whatever the activation (round 4, profiles/r04_gemm_segments.log: 40 % of a K = 1280 tile).
// Now every residual vector of the tile is fetched up front into the registers the dead operand fragments leave free (16 x 16 bytes per
// lane), one wait, and the 16 stores go out back to back.  (`out_rows` launches - the projector's last GEMM - take the direct epilogue.)
template <int ACT>
__device__ __forceinline__ void g2_epilogue_row_act(const GemmArgs& a, f4 (&acc)[4][8], int mb, int nb, int lane, int w, char* smem, int wfree, int afree) {
    const int g = lane >> 4;
    // per-wave transposition slabs in LDS no staged K-tile occupies: W buffer `wfree` (waves 0-5), second half of A buffer `afree` (6-7)
    char* slab = w < 6 ? smem + G2_WBASE + wfree * G2_WBUF + w * G2_SLAB_BYTES : smem + afree * G2_ABUF + G2_SLOT + (w - 6) * G2_SLAB_BYTES;
    f4 bias[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bias[t] = a.bias ? *(const f4*)(a.bias + nb + t * 16 + 4 * g) : f4{0.f, 0.f, 0.f, 0.f};
    h8 res[8][2];
    if (ACT != ACT_SILU_MUL && a.resid) {
        const int row0 = lane >> 3, n = nb + (lane & 7) * 8;
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int m = mb + u * 16 + hh * 8 + row0;
                h8 rr = h8{0, 0, 0, 0, 0, 0, 0, 0};
                if (m < a.M && n < a.n_real) {
                    const half_t* rp = a.resid + (int64_t)m * a.ldr + n;
                    if (n + 8 <= a.n_real) rr = *(const h8*)rp;
                    else {
                        const h4 q = *(const h4*)rp;
                        rr = h8{q[0], q[1], q[2], q[3], 0, 0, 0, 0};
                    }
                }
                res[u][hh] = rr;
            }
    }
    g2_epilogue_row_u<0, ACT>(a, acc, bias, res, mb, nb, lane, slab);
    g2_epilogue_row_u<1, ACT>(a, acc, bias, res, mb, nb, lane, slab);
    g2_epilogue_row_u<2, ACT>(a, acc, bias, res, mb, nb, lane, slab);
    g2_epilogue_row_u<3, ACT>(a, acc, bias, res, mb, nb, lane, slab);
    g2_epilogue_row_u<4, ACT>(a, acc, bias, res, mb, nb, lane, slab);
    g2_epilogue_row_u<5, ACT>(a, acc, bias, res, mb, nb, lane, slab);
    g2_epilogue_row_u<6, ACT>(a, acc, bias, res, mb, nb, lane, slab);
    g2_epilogue_row_u<7, ACT>(a, acc, bias, res, mb, nb, lane, slab);
}
// Which activation a projection TAG implies (launch_gemm256 only picks a tagged instantiation when the arguments agree; anything else
// runs GT_OTHER, whose epilogue resolves the activation per block at run time: ACT == -2).  One specialised epilogue per kernel - five
// inlined copies behind a switch cost registers (256 + a spill) and minutes of compile time.
template <int TAG> struct G2TagAct { static constexpr int act = -2; };
template <> struct G2TagAct<GT_VIT_OUT> { static constexpr int act = ACT_NONE; };
template <> struct G2TagAct<GT_VIT_FC2> { static constexpr int act = ACT_NONE; };
template <> struct G2TagAct<GT_LLM_O> { static constexpr int act = ACT_NONE; };
template <> struct G2TagAct<GT_LLM_DOWN> { static constexpr int act = ACT_NONE; };
template <> struct G2TagAct<GT_LLM_GATEUP> { static constexpr int act = ACT_SILU_MUL; };
template <> struct G2TagAct<GT_VIT_FC1> { static constexpr int act = ACT_QUICK_GELU; };
// (an erf-GELU tower - config.hidden_act - runs GT_OTHER)
static int g2_tag_act(int tag) {
    switch (tag) {
        case GT_VIT_OUT: case GT_VIT_FC2: case GT_LLM_O: case GT_LLM_DOWN: return ACT_NONE;
        case GT_LLM_GATEUP: return ACT_SILU_MUL;
        case GT_VIT_FC1: return ACT_QUICK_GELU;
        default: return -2;
    }
}
template <int TAG>
__device__ __forceinline__ void g2_epilogue_row(const GemmArgs& a, f4 (&acc)[4][8], int mb, int nb, int lane, int w, char* smem, int wfree, int afree) {
    if constexpr (TAG == GT_OTHER) {
        // The generic kernel (projector, patch embedding, aur_linear) used to run the ladder variant for every activation: 78 KB of
        // epilogue code of which a tile executes a scattered tenth - 20 k cycles per tile on a quiet chip, 36 k once the operands have pushed
        // the lines out of L2 (round 4, profiles/r04_gemm_epilogue_probe.log), against 8-9 k for the 7.5 KB of a compile-time activation.  Plain
        // bias / residual launches now take their own lean copy; only a launch with an activation pays for the ladder.
        if (a.act == ACT_NONE) g2_epilogue_row_act<ACT_NONE>(a, acc, mb, nb, lane, w, smem, wfree, afree);
        else g2_epilogue_row_act<-2>(a, acc, mb, nb, lane, w, smem, wfree, afree);
    } else {
        g2_epilogue_row_act<G2TagAct<TAG>::act>(a, acc, mb, nb, lane, w, smem, wfree, afree);
    }
}

template <int EPI, int TAG>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbn = a.Npad >> 8;
    const int nbm = (a.M + 255) >> 8;
    const int nwg = nbn * nbm;
    // tile of a block id: rounds of gridDim.x tiles are compact blocks of the tile space, one 4 x sn sub-block per XCD (tile_order.h).
    // gridDim.x is a multiple of 8 whenever a workgroup walks more than one tile, so its tiles keep the XCD (bid % 8) the order assumes.
    // (Rounds 1-2 walked per-XCD tile ranges - the guide's T1 remap - where nothing is shared between XCDs in time: measured twice, removed.)
    TileOrder ord;
    tile_order_init(ord, nbm, nbn, (int)gridDim.x);
    int bm, bn;
    G2Src src;
    tile_of_bid(ord, blockIdx.x, bm, bn);
    g2_sources(a, bm, bn, w, lane, src);
    g2_prologue(a, src, smem, w);
    bool first = true;
    for (int bid = blockIdx.x; bid < nwg; bid += gridDim.x) {
        f4 acc[4][8];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[t][u] = f4{0.f, 0.f, 0.f, 0.f};
        const int nb = bn * 256 + (w & 3) * 64;
        const int mb = bm * 256 + (w >> 2) * 128;
        const bool vmode = (EPI == EPI_QKV) && (nb >= a.q_cols + a.k_cols);
        const int nxt = bid + gridDim.x;
        if (vmode) g2_mainloop<EPI, true>(a, src, smem, acc, w, lane, first);
        else g2_mainloop<EPI, false>(a, src, smem, acc, w, lane, first);
        first = false;
        if (nxt < nwg) {
            // every wave is past the K loop's last barrier, all LDS reads of this tile are done -> start the next tile's loads now, they fly
            // while this tile's epilogue converts and stores
            tile_of_bid(ord, nxt, bm, bn);
            g2_sources(a, bm, bn, w, lane, src);
            g2_prologue(a, src, smem, w);
        }
        // 16-byte row accesses need 8-column alignment of every row (ldc, ldr multiples of 8 halves; SiLU*up writes n / 2: ldc % 4)
        const bool wide = EPI == EPI_ROW && a.out_rows == nullptr && (a.act == ACT_SILU_MUL ? (a.ldc & 3) == 0 : ((a.ldc | (a.resid ? a.ldr : 0)) & 7) == 0);
        // LDS the next tile's staged K-tiles do not occupy: W buffer 2 and the second half of A buffer 1
        if (wide) g2_epilogue_row<TAG>(a, acc, mb, nb, lane, w, smem, 2, 1);
        else gemm_epilogue<EPI, 4, 8>(a, acc, mb, nb, lane, vmode);
    }
}

static int g_gemm256_cus = 256;

template <int EPI, int TAG>
static hipError_t g2_attr() {
    return hipFuncSetAttribute((const void*)gemm256_kernel<EPI, TAG>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
}
hipError_t gemm256_init() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= 8)
        g_gemm256_cus = cus;
    hipError_t e;
    if ((e = g2_attr<EPI_ROW, GT_OTHER>()) != hipSuccess || (e = g2_attr<EPI_ROW, GT_VIT_OUT>()) != hipSuccess ||
        (e = g2_attr<EPI_ROW, GT_VIT_FC1>()) != hipSuccess || (e = g2_attr<EPI_ROW, GT_VIT_FC2>()) != hipSuccess ||
        (e = g2_attr<EPI_ROW, GT_LLM_O>()) != hipSuccess || (e = g2_attr<EPI_ROW, GT_LLM_GATEUP>()) != hipSuccess ||
        (e = g2_attr<EPI_ROW, GT_LLM_DOWN>()) != hipSuccess || (e = g2_attr<EPI_QKV, GT_OTHER>()) != hipSuccess ||
        (e = g2_attr<EPI_QKV, GT_VIT_QKV>()) != hipSuccess || (e = g2_attr<EPI_QKV, GT_LLM_QKV>()) != hipSuccess)
        return e;
    return hipSuccess;
}

int gemm256_grid_cap() { return g_gemm256_cus & ~7; }

bool gemm256_eligible(const GemmArgs& a) {
    // whole 256-column tiles, and more 256x256 tiles than HALF of the persistent grid G (one 160 KiB workgroup per CU; max_wgs under a
    // CU mask).  Cost model in rounds of the 128x128 kernel (2 workgroups per CU, measured 1.3-1.5x slower per flop): ceil(T / G) x ~1.4
    // against ceil(2 T / G) x 1 - the big tile wins as soon as the small one needs a second round.  Rounds 1-3 asked for T >= 224 and sent
    // the o / down projections of a single-clip prefill (144 tiles) to the 128x128 kernel: 116 / 296 us against 98 / 229 us (round 4,
    // tools/microbench.py --batch 8), 2.7 ms of a 50 ms time to first token.  The two kernels agree bit for bit, so this is speed only.
    if (a.Npad & 255) return false;
    const int64_t tiles = (int64_t)(a.Npad >> 8) * ((a.M + 255) >> 8);
    const int G = a.max_wgs > 0 ? ((a.max_wgs + 7) & ~7) : (g_gemm256_cus & ~7);
    return 2 * tiles > G;
}

hipError_t launch_gemm256(const GemmArgs& a, int epi, hipStream_t s) {
    int ntiles = (a.Npad >> 8) * ((a.M + 255) >> 8);
    const int cap = a.max_wgs > 0 ? ((a.max_wgs + 7) & ~7) : (g_gemm256_cus & ~7);        // one 160 KiB workgroup per CU
    if (ntiles > cap) ntiles = cap;
    dim3 grid(ntiles), block(512);
#define G2_LAUNCH(E, T) hipLaunchKernelGGL((gemm256_kernel<E, T>), grid, block, G2_LDS, s, a)
    if (epi == EPI_ROW) {
        // a tagged instantiation carries its projection's activation at compile time: anything else runs the generic one
        switch (g2_tag_act(a.tag) == a.act ? a.tag : GT_OTHER) {
            case GT_VIT_OUT: G2_LAUNCH(EPI_ROW, GT_VIT_OUT); break;
            case GT_VIT_FC1: G2_LAUNCH(EPI_ROW, GT_VIT_FC1); break;
            case GT_VIT_FC2: G2_LAUNCH(EPI_ROW, GT_VIT_FC2); break;
            case GT_LLM_O: G2_LAUNCH(EPI_ROW, GT_LLM_O); break;
            case GT_LLM_GATEUP: G2_LAUNCH(EPI_ROW, GT_LLM_GATEUP); break;
            case GT_LLM_DOWN: G2_LAUNCH(EPI_ROW, GT_LLM_DOWN); break;
            default: G2_LAUNCH(EPI_ROW, GT_OTHER); break;
        }
    } else {
        switch (a.tag) {
            case GT_VIT_QKV: G2_LAUNCH(EPI_QKV, GT_VIT_QKV); break;
            case GT_LLM_QKV: G2_LAUNCH(EPI_QKV, GT_LLM_QKV); break;
            default: G2_LAUNCH(EPI_QKV, GT_OTHER); break;
        }
    }
#undef G2_LAUNCH
    return hipGetLastError();
}
