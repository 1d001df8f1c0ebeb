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

// Which phase of K-tile t issues which half-tile of LDS-DMA prefetch (all three keep the same issue ORDER A1(t+1), W0(t+2), W1(t+2),
// A0(t+2), hence the same counted wait vmcnt(6) in phase 3 and the same arithmetic):
//   0: one half-tile per phase - A1 | W0 | W1 | A0     (rounds 1-3)
//   1: none | A1, W0 | W1, A0 | none                   the two read-heavy / waiting phases (0: 12 fragment reads, 3: the vmcnt wait) issue nothing
//   2: none | A1, W0 | W1 | A0
//   4: nothing in the load segments: the two instructions of a half-tile are issued INSIDE the phase's MFMA burst (after the 4th and the
//      12th MFMA), A1 | W0 | W1 | A0 by phase - the per-CU LDS-DMA path takes one 1 KiB instruction per 16 cycles, and a burst of 8-16
//      of them at the head of a load segment is what made the segment outlast the partner's burst; phase 3 then waits vmcnt(4)
// Round 4 measured the segments (tools/gemm_lab/ts_probe.py, profiles/r04_gemm_segments.log): with schedule 0 the load segments of phases 0
// and 3 outlast the partner group's 16-MFMA burst (329 / 222-287 cycles against 292) while phases 1 and 2 idle at the barrier.  An LDS-DMA
// instruction costs its wave 60-90 cycles wherever it sits (in a load segment or between MFMAs: schedule 4 lengthens the bursts by what
// it takes off the load segments); without any operand DMA the same kernel runs 1.28x faster - the price of feeding a 256 x 256 tile.
// Start-up stagger of the persistent workgroups (0 = none; S = S slots per XCD).  All tiles of a GEMM take the same time, so without it
// every CU reaches its epilogue at the same moment and 256 x 128 KiB of output hit the memory system in one burst: round 4 measured the
// epilogue at 38-46 k cycles per tile whatever the activation (tools/gemm_lab/ts_probe.py) = 3.4 B per cycle per CU = HBM's write rate,
// while the K loops around it leave HBM idle.  Workgroup b sleeps ((b / 8) % S) / S of a tile's duration before its first tile: epilogues
// of some CUs then overlap K loops of others.
#ifndef G2_CONT
#define G2_CONT 0      // 1 = continuous pipeline across tile boundaries (g2_mainloop): built and bit-exact in round 4, measured 1-2 % SLOWER than
#endif                 //     the per-tile prologue (profiles/r04_gemm_segments.log: the prologue burst was not what the epilogue waits for) - off
#ifndef G2_STAGGER
#define G2_STAGGER 0
#endif
#ifndef G2_STORE_AUX
#define G2_STORE_AUX 2  // cache policy of the wide epilogue's output stores: 0 default, 2 nt (g2_store_c)
#endif
#ifndef G2_SCHED
#define G2_SCHED 2      // round 4: +3..6 % on every shape of the path against schedule 0 (profiles/r04_gemm_segments.log), same bits
#endif
// ---- lab instantiations (round 3: tools/cumask/contention_lab.py, now in the history; never launched by the product path: GemmArgs.lab == 0 there).
// TAG >= GT_LAB_BASE selects a deliberately altered kernel that answers "what does the front-end GEMM take away from a concurrent
// decode stream?": cache-policy bits on the operand DMAs, no DMA at all (power / clock only), no MFMA (fabric traffic only), or the
// activation operand read from a K-tile-major image (DRAM-friendly 32 KiB blocks instead of 128-byte row pieces; results are garbage).
// Compiled only with -DAUR_LABS (python -m aurora_amd.build --labs -> libaurora_hip_labs.so); the product library carries G2Lab<..>::id == 0 only.
template <int TAG> struct G2Lab {
#ifdef AUR_LABS
    static constexpr int id = TAG >= GT_LAB_BASE ? TAG - GT_LAB_BASE : 0;
#else
    static_assert(TAG < GT_LAB_BASE, "lab instantiations need -DAUR_LABS");
    static constexpr int id = 0;
#endif
    static constexpr int aux_a = (id == 1 || id == 3) ? 2 : (id == 4 ? 16 : 0);     // 2 = nt, 16 = sc1
    static constexpr int aux_w = (id == 2 || id == 3) ? 2 : (id == 4 ? 16 : 0);
    static constexpr bool no_dma = id == 5 || id == 14;
    static constexpr bool no_mfma = id == 6;
    static constexpr bool a_tiled = id == 7 || id == 13;
    // 8-10, 13, 14: s_memtime stamps around every segment of every phase -> g2_ts (aur_lab_gemm_ts); 9 / 11: DMA schedule 1, 10 / 12: 2
    // 26: stamps on a kernel with a compile-time activation (none), i.e. the PRODUCT's epilogue (7.5 KB of code; the other labs carry the generic
    // run-time activation ladder in each of their 32 blocks: 78 KB)
    static constexpr bool ts = (id >= 8 && id <= 10) || id == 13 || id == 14 || id == 16 || id == 19 || id == 20 || id == 21 || id == 23 || id == 24 || id == 26;
    static constexpr bool cont = (G2_CONT != 0 || id == 22 || id == 23);   // 22 / 23 (= with stamps): the continuous pipeline
    static constexpr bool wide_only = id == 24;        // stamps + the direct epilogue compiled OUT (a 30 KB kernel instead of 127 KB: instruction-cache probe)
    static constexpr bool epi_nostore = id == 20;      // stamps + the wide epilogue without its global stores (where do its cycles go?)
    static constexpr bool epi_halfstore = id == 21;    // stamps + NO LDS transposition in the wide epilogue (garbage output; both stores stay)
    static constexpr int sched = (id == 9 || id == 11) ? 1 : (id == 10 || id == 12 || id >= 17) ? 2 : (id == 15 || id == 16) ? 4 : G2_SCHED;     // 15 / 16: schedule 4 without / with stamps
    // 17 / 18: workgroups of one XCD start 0 .. 3/4 (17, 19 = with stamps) or 0 .. 7/8 (18) of a tile's duration apart
    static constexpr int stagger = (id == 17 || id == 19) ? 4 : id == 18 ? 8 : G2_STAGGER;
    // 25: the wide epilogue's output stores with the default cache policy (the product's are non-temporal: tools/gemm_lab/store_probe.py)
    static constexpr int store_aux = id == 25 ? 0 : G2_STORE_AUX;
};
#ifdef AUR_LABS
// lab 8: per workgroup and wave, cycles summed over the K loops of all its tiles, [phase 0..3][load segment, barrier 1, lgkmcnt wait,
// MFMA burst, barrier 2], then [20] = phases timed, [21] = total cycles of the K loops.  Stamps are s_memtime (SMEM: counted by lgkmcnt)
// issued WITHOUT a wait of their own: they are collected behind the schedule's existing s_waitcnt lgkmcnt(0).
__device__ unsigned g2_ts2[256 * 8 * 4];
__device__ unsigned g2_ts[256 * 8 * 32];     // + [22] tile head + K loop, [23] next tile's prologue issue, [24] epilogue, [25] tiles (cycles per wave)
#endif
template <int AUX>
__device__ __forceinline__ void glds16x(const void* gsrc_lane, void* lds_wave_base) {
    if constexpr (AUX == 0) __builtin_amdgcn_global_load_lds(GPTR(gsrc_lane), LPTR(lds_wave_base), 16, 0, 0);
    else if constexpr (AUX == 2) __builtin_amdgcn_global_load_lds(GPTR(gsrc_lane), LPTR(lds_wave_base), 16, 0, 2);
    else __builtin_amdgcn_global_load_lds(GPTR(gsrc_lane), LPTR(lds_wave_base), 16, 0, 16);
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
template <int TAG>
__device__ __forceinline__ void g2_sources(const GemmArgs& a, int bm, int bn, int w, int lane, G2Src& src, bool do_a0 = true, bool do_a1 = true, bool do_w = true) {
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
            if (h == 0 ? do_a0 : do_a1) {                  // block-uniform
                if constexpr (G2Lab<TAG>::a_tiled) src.a[h][i] = a.A + ((int64_t)bm * (a.K >> 6) * 256 + h * 128 + row) * 64 + c * 8;
                else src.a[h][i] = a.A + (int64_t)m * a.lda + c * 8;
            }
            const int f = grp;                      // fragment of the half-tile: n16 = f >> 1, kk = f & 1
            if (do_w) src.w[h][i] = a.W + ((int64_t)(bn * 16 + h * 8 + (f >> 1)) * K32 + (f & 1)) * AUR_FRAG_HALVES + lane * 8;
        }
}
// kt = K-tile of the SOURCE tile, ab / wb = the LDS buffer it lands in.  Inside a tile buffer = (kt + phase) mod 2 / mod 3; the phases
// carry over from tile to tile (continuous pipeline, gemm256_kernel), so source index and buffer are separate arguments.
template <int TAG>
__device__ __forceinline__ void g2_stage_a(const G2Src& src, char* smem, int w, int h, int kt, int ab) {
    if constexpr (G2Lab<TAG>::no_dma) return;
    char* dst = smem + ab * G2_ABUF + h * G2_SLOT;
    const int64_t step = G2Lab<TAG>::a_tiled ? (int64_t)kt * 256 * 64 : (int64_t)kt * 64;
    glds16x<G2Lab<TAG>::aux_a>(src.a[h][0] + step, dst + (w * 2 + 0) * 1024);
    glds16x<G2Lab<TAG>::aux_a>(src.a[h][1] + step, dst + (w * 2 + 1) * 1024);
}
template <int TAG>
__device__ __forceinline__ void g2_stage_w(const G2Src& src, char* smem, int w, int h, int kt, int wb) {
    if constexpr (G2Lab<TAG>::no_dma) return;
    char* dst = smem + G2_WBASE + wb * G2_WBUF + h * G2_SLOT;
    glds16x<G2Lab<TAG>::aux_w>(src.w[h][0] + (int64_t)kt * 2 * AUR_FRAG_HALVES, dst + (w * 2 + 0) * 1024);
    glds16x<G2Lab<TAG>::aux_w>(src.w[h][1] + (int64_t)kt * 2 * AUR_FRAG_HALVES, dst + (w * 2 + 1) * 1024);
}
// one of the two instructions of a half-tile (schedule 4 issues them inside the MFMA bursts)
template <int TAG>
__device__ __forceinline__ void g2_stage_a1(const G2Src& src, char* smem, int w, int h, int kt, int ab, int i) {
    if constexpr (G2Lab<TAG>::no_dma) return;
    char* dst = smem + ab * G2_ABUF + h * G2_SLOT;
    const int64_t step = G2Lab<TAG>::a_tiled ? (int64_t)kt * 256 * 64 : (int64_t)kt * 64;
    glds16x<G2Lab<TAG>::aux_a>(src.a[h][i] + step, dst + (w * 2 + i) * 1024);
}
template <int TAG>
__device__ __forceinline__ void g2_stage_w1(const G2Src& src, char* smem, int w, int h, int kt, int wb, int i) {
    if constexpr (G2Lab<TAG>::no_dma) return;
    char* dst = smem + G2_WBASE + wb * G2_WBUF + h * G2_SLOT;
    glds16x<G2Lab<TAG>::aux_w>(src.w[h][i] + (int64_t)kt * 2 * AUR_FRAG_HALVES, dst + (w * 2 + i) * 1024);
}
// prologue of a workgroup's FIRST tile (and of every tile when the K loop is too short for the continuous pipeline): K-tile 0 completely
// (8 instructions) into A buffer 0 / W buffer 0, plus W0, W1, A0 of K-tile 1 (6 instructions that may stay in flight) into W buffer 1 /
// A buffer 1 - buffer phases (0, 0)
template <int TAG>
__device__ __forceinline__ void g2_prologue(const GemmArgs& a, const G2Src& src, char* smem, int w) {
    g2_stage_a<TAG>(src, smem, w, 0, 0, 0);
    g2_stage_a<TAG>(src, smem, w, 1, 0, 0);
    g2_stage_w<TAG>(src, smem, w, 0, 0, 0);
    g2_stage_w<TAG>(src, smem, w, 1, 0, 0);
    if ((a.K >> 6) > 1) {
        g2_stage_w<TAG>(src, smem, w, 0, 1, 1);
        g2_stage_w<TAG>(src, smem, w, 1, 1, 1);
        g2_stage_a<TAG>(src, smem, w, 0, 1, 1);
    }
}

// K loop of one tile.
// CONTINUOUS PIPELINE (round 4, compile-time option G2_CONT, OFF: see the define).  The default flow issues the next tile's prologue (3.5
// K-tiles' worth of LDS-DMA, 112 KiB per CU) in one burst after the K loop; the hypothesis was that the epilogue's first load - or a
// spilled scalar's reload - waits behind that burst with vmcnt(0).  With G2_CONT the LAST TWO K-tiles of a tile stage the NEXT tile's
// K-tiles 0 and 1 exactly where a middle K-tile stages t + 1 / t + 2:
// same issue order, same counted waits, the LDS buffer rotation (ap: A buffer of K-tile 0, wp: W buffer of K-tile 0) simply carries over
// the tile boundary, and phase 3 of the last K-tile retires the next tile's K-tile 0 like any other.  The source pointers are switched
// to the next tile IN PLACE (W and A0 at K-tile nkt - 2, A1 at nkt - 1: the current tile no longer needs them), so no registers are
// added.  Result: bit-exact (the whole kernel suite passes on it), the prologue burst is gone from the profile - and the epilogue still
// takes its 30 k cycles (3.5 k per 16-row block with or without stores, LDS transposition or a small code footprint: labs 20 / 21 / 24),
// while the K loop pays ~4 % for the extra branches.  Kept as a measured option, not the default.
// `wait` = what to wait for on entry: 0 the first tile's prologue (counted), 1 drain everything (short K loops keep the old
// per-tile prologue), 2 nothing (the previous tile's loop already retired this tile's K-tile 0).
template <int EPI, bool VMODE, int TAG>
__device__ __forceinline__ void g2_mainloop(const GemmArgs& a, G2Src& src, char* smem, f4 (&acc)[4][8], int w, int lane, int wait, int ap, int wp,
                                            bool has_next, int bm2, int bn2) {
    const int wr = w >> 2, wc = w & 3;
    const int r = lane & 15, g = lane >> 4;
    const int nkt = a.K >> 6;
    auto stage_a = [&](int h, int kt, int ab) { g2_stage_a<TAG>(src, smem, w, h, kt, ab); };
    auto stage_w = [&](int h, int kt, int wb) { g2_stage_w<TAG>(src, smem, w, h, kt, wb); };

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
    // ---- lab 8 (AUR_LABS): segment stamps.  tA: phase start, tB: before barrier 1, tC: after it, tD: after the lgkmcnt wait, tE: after
    // the MFMA burst.  Everything compiles away when G2Lab<TAG>::ts is false.
    unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, tE = 0, tCp = 0, tDp = 0, tEp = 0, tK0 = 0;
    [[maybe_unused]] unsigned tacc[4][5] = {{0}}, tphases = 0;
    bool thave = false;
#define G2_TS(v)                                                      \
    do {                                                              \
        if constexpr (G2Lab<TAG>::ts) asm volatile("s_memtime %0" : "=s"(v)); \
    } while (0)
#define G2_TS_COLLECT(P)                                                                                     \
    do {                                                                                                     \
        if constexpr (G2Lab<TAG>::ts) {                                                                      \
            asm volatile("" : "+s"(tA), "+s"(tB), "+s"(tC), "+s"(tDp), "+s"(tEp));   /* behind the lgkmcnt(0) above */ \
            tacc[P][0] += (unsigned)(tB - tA);                                                               \
            tacc[P][1] += (unsigned)(tC - tB);                                                               \
            if (thave) {                                                                                     \
                tacc[((P) + 3) & 3][2] += (unsigned)(tDp - tCp);                                             \
                tacc[((P) + 3) & 3][3] += (unsigned)(tEp - tDp);                                             \
                tacc[((P) + 3) & 3][4] += (unsigned)(tA - tEp);                                              \
                ++tphases;                                                                                   \
            }                                                                                                \
            thave = true;                                                                                    \
            tCp = tC;                                                                                        \
        }                                                                                                    \
    } while (0)
#define G2_COMPUTE(NH, MH, wf, D0, D1)                                                                           \
    do {                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        G2_TS_COLLECT(((NH) == 0 && (MH) == 0) ? 0 : ((NH) == 1 && (MH) == 0) ? 1 : ((NH) == 1 && (MH) == 1) ? 2 : 3); \
        G2_TS(tD);                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        if (G2Lab<TAG>::no_mfma) __builtin_amdgcn_s_sleep(4);      /* lab: the burst's duration without its power */ \
        else                                                                                                 \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                      \
            _Pragma("unroll") for (int tt = 0; tt < 2; ++tt)                                                  \
                _Pragma("unroll") for (int uu = 0; uu < 4; ++uu) {                                            \
                    if (VMODE)                                                                               \
                        acc[(NH) * 2 + tt][(MH) * 4 + uu] = mfma16(af[uu * 2 + kk], wf[tt * 2 + kk], acc[(NH) * 2 + tt][(MH) * 4 + uu]); \
                    else                                                                                     \
                        acc[(NH) * 2 + tt][(MH) * 4 + uu] = mfma16(wf[tt * 2 + kk], af[uu * 2 + kk], acc[(NH) * 2 + tt][(MH) * 4 + uu]); \
                    if (G2Lab<TAG>::sched == 4 && uu == 3 && tt == 0) {                                      \
                        __builtin_amdgcn_sched_barrier(0);                                                   \
                        if (kk == 0) { D0; } else { D1; }                                                    \
                        __builtin_amdgcn_sched_barrier(0);                                                   \
                    }                                                                                        \
                }                                                                                            \
        __builtin_amdgcn_s_setprio(0);                                                                       \
        G2_TS(tE);                                                                                           \
        tDp = tD;                                                                                            \
        tEp = tE;                                                                                            \
    } while (0)

    // ---- the prologue was issued by the caller: K-tile 0 must have landed, W0, W1, A0 of K-tile 1 may stay in flight
    if (wait == 0 && nkt > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (wait != 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G2_BARRIER();
    if (wr == 1) G2_BARRIER();                    // stagger: waves 4-7 run one barrier behind

    if constexpr (G2Lab<TAG>::ts) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tK0));
    int wb = wp;                                           // weight buffer of K-tile t: (t + wp) % 3
    for (int t = 0; t < nkt; ++t) {
        const int ab0 = (t + ap) & 1, ab1 = ab0 ^ 1;       // A buffer of K-tiles t (and t + 2) / t + 1
        const char* abuf = smem + ab0 * G2_ABUF;
        const char* wbuf = smem + G2_WBASE + wb * G2_WBUF;
        const int wb2 = wb == 0 ? 2 : wb - 1;              // W buffer of K-tile t + 2
        // K-tiles t + 1 / t + 2 of THIS tile, or - in its last two iterations - K-tiles 0 / 1 of the next one (block-uniform)
        const bool nx1 = t + 1 >= nkt, nx2 = t + 2 >= nkt;
        const bool n1 = !nx1 || has_next, n2 = !nx2 || has_next;
        const int k1 = nx1 ? t + 1 - nkt : t + 1, k2 = nx2 ? t + 2 - nkt : t + 2;
        if (has_next && t >= nkt - 2) g2_sources<TAG>(a, bm2, bn2, w, lane, src, t == nkt - 2, t == nkt - 1, t == nkt - 2);
        constexpr int SCHED = G2Lab<TAG>::sched;
        // phase 0: quadrant (n half 0, m half 0)
        G2_TS(tA);
        read_a(abuf, 0);
        read_w(wbuf, 0, wf0);
        if (SCHED == 0 && n1) stage_a(1, k1, ab1);
        G2_TS(tB);
        G2_BARRIER();
        G2_TS(tC);
        G2_COMPUTE(0, 0, wf0, if (n1) g2_stage_a1<TAG>(src, smem, w, 1, k1, ab1, 0), if (n1) g2_stage_a1<TAG>(src, smem, w, 1, k1, ab1, 1));
        G2_BARRIER();
        // phase 1: (n half 1, m half 0)
        G2_TS(tA);
        read_w(wbuf, 1, wf1);
        if (SCHED != 0 && SCHED != 4 && n1) stage_a(1, k1, ab1);
        if (SCHED != 4 && n2) stage_w(0, k2, wb2);
        G2_TS(tB);
        G2_BARRIER();
        G2_TS(tC);
        G2_COMPUTE(1, 0, wf1, if (n2) g2_stage_w1<TAG>(src, smem, w, 0, k2, wb2, 0), if (n2) g2_stage_w1<TAG>(src, smem, w, 0, k2, wb2, 1));
        G2_BARRIER();
        // phase 2: (n half 1, m half 1)
        G2_TS(tA);
        read_a(abuf, 1);
        if (SCHED != 4 && n2) stage_w(1, k2, wb2);
        if (SCHED == 1 && n2) stage_a(0, k2, ab0);
        G2_TS(tB);
        G2_BARRIER();
        G2_TS(tC);
        G2_COMPUTE(1, 1, wf1, if (n2) g2_stage_w1<TAG>(src, smem, w, 1, k2, wb2, 0), if (n2) g2_stage_w1<TAG>(src, smem, w, 1, k2, wb2, 1));
        G2_BARRIER();
        // phase 3: (n half 0, m half 1); retire K-tile t+1's half-tiles, keep W0, W1, A0 of t+2 in flight
        G2_TS(tA);
        if (n2) {
            if (SCHED != 1 && SCHED != 4) stage_a(0, k2, ab0);
            if (SCHED == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // W0, W1 of t+2 stay in flight; A0(t+2) follows in this phase's burst
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        G2_TS(tB);
        G2_BARRIER();
        G2_TS(tC);
        G2_COMPUTE(0, 1, wf0, if (n2) g2_stage_a1<TAG>(src, smem, w, 0, k2, ab0, 0), if (n2) g2_stage_a1<TAG>(src, smem, w, 0, k2, ab0, 1));
        G2_BARRIER();
        wb = wb == 2 ? 0 : wb + 1;
    }
    if (wr == 0) G2_BARRIER();                    // match the stagger barrier of waves 4-7
#ifdef AUR_LABS
    if constexpr (G2Lab<TAG>::ts) {
        unsigned long long tK1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tK1));
        if (lane == 0 && blockIdx.x < 256) {
            unsigned* o = g2_ts + (blockIdx.x * 8 + w) * 32;
            for (int p = 0; p < 4; ++p)
                for (int q = 0; q < 5; ++q) o[p * 5 + q] += tacc[p][q];
            o[20] += tphases;
            o[21] += (unsigned)(tK1 - tK0);
        }
    }
#endif
#undef G2_TS
#undef G2_TS_COLLECT
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
// (tools/gemm_lab/ts_probe.py) - so the activation ladder is resolved outside the 32 (u, t) blocks instead of inside each.
// ACT == -2 (GT_OTHER: projector, patch embedding, microbenchmarks): the activation is resolved at run time inside the blocks.
#define G2_EST(i)                                                                                     \
    do {                                                                                              \
        if (U == 0 && fine) asm volatile("s_memtime %0" : "=s"(fine[i])::"memory");                   \
    } while (0)
// One output vector of the wide epilogue.  POL 2 (product): a non-temporal store when the launch asks for it (GemmArgs.nt_out: outputs larger
// than the L2s together; engine.hip ctx_gemm).  Every round of tiles writes 256 x 128 KiB - the capacity of
// the eight L2s - in one burst, because all CUs reach their epilogue together; written with the default policy the output evicts the operand
// panels the next tiles share through L2 and the K loops AFTER the epilogue pay for it (round 4, tools/gemm_lab/store_probe.py and
// conc_probe.py: the per-tile fixed cost is 9.7 us with up to 128 CUs running and 16.4 us with 256; a K = 128 sweep 500 -> 310 us, K = 1280
// 1180 -> 1085 us, prefill gate/up -3 %, with `nt`; sc1 / sc0 sc1 write-through stores gain less).  Nothing on this path re-reads C from L2:
// the consumer is the next launch, and C is 40-800 MB.
template <int POL, typename V>
__device__ __forceinline__ void g2_store_c(bool nt, half_t* dst, const V& v) {
    if (POL == 2 && nt) __builtin_nontemporal_store(v, (V*)dst);
    else *(V*)dst = v;
}
template <int U, int ACT, int STORES = 2, int POL = 0>
__device__ __forceinline__ void g2_epilogue_row_u(const GemmArgs& a, f4 (&acc)[4][8], const f4 (&bias)[4], const h8 (&res)[8][2], int mb, int nb, int lane, char* slab,
                                                  unsigned long long* fine = nullptr) {        // lab: 5 stamps inside row block 0
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
        if (STORES != -1) *(f4*)(slab + r * G2_SLAB_STRIDE + (t * 16 + 4 * g) * 4) = v;
        else asm volatile("" ::"v"(v));                  // lab ablation (STORES == -1): no LDS transposition at all (the output is garbage)
    }
    asm volatile("" ::: "memory");                       // LDS writes above, reads below: same wave, program order
    G2_EST(0);                                           // the 4 ds_write_b128 are issued
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int row = hh * 8 + row0;
        const int m = mb + U * 16 + row;
        const f4 x0 = STORES != -1 ? *(const f4*)(slab + row * G2_SLAB_STRIDE + chunk * 32) : acc[hh][U];
        const f4 x1 = STORES != -1 ? *(const f4*)(slab + row * G2_SLAB_STRIDE + chunk * 32 + 16) : acc[hh + 2][U];
        if (U == 0 && fine) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (hh == 0) G2_EST(1); else G2_EST(3);      // this half's two ds_read_b128 have returned
        }
        if (m >= a.M || n >= a.n_real) continue;
        if (STORES >= 0 && hh >= STORES) {                                  // lab ablation only (STORES < 2): keep the values alive, skip the store
            asm volatile("" ::"v"(x0), "v"(x1));
            continue;
        }
        if (ACT == ACT_SILU_MUL || (ACT == -2 && a.act == ACT_SILU_MUL)) {      // (gate, up) interleaved: 4 outputs at columns n / 2 .. n / 2 + 3
            h4 o;
            o[0] = (half_t)(silu_f(x0[0]) * x0[1]);
            o[1] = (half_t)(silu_f(x0[2]) * x0[3]);
            o[2] = (half_t)(silu_f(x1[0]) * x1[1]);
            o[3] = (half_t)(silu_f(x1[2]) * x1[3]);
            half_t* dst = a.C + (int64_t)m * a.ldc + (n >> 1);
            if (n + 8 <= a.n_real) g2_store_c<POL>(a.nt_out != 0, dst, o);
            else g2_store_c<POL>(a.nt_out != 0, dst, h2{o[0], o[1]});             // n_real % 4 == 0: the chunk holds 4 real columns
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
            if (full) g2_store_c<POL>(a.nt_out != 0, dst, o);
            else g2_store_c<POL>(a.nt_out != 0, dst, h4{o[0], o[1], o[2], o[3]});
        }
        if (hh == 0) G2_EST(2); else G2_EST(4);          // this half's store is issued
    }
    asm volatile("" ::: "memory");
}
#undef G2_EST
// The wide epilogue issues NO load after its first store.  Rounds 2-3 loaded the residual row (and looked up `out_rows[m]`) inside each of
// the 16 row blocks: hipcc answers a load next to in-flight LDS-DMA with `s_waitcnt vmcnt(0)` (guide, "three .s-level traps" (b)), and
// because stores and loads share vmcnt each block then waited for the PREVIOUS block's stores to be acknowledged by memory - 16 dependent
// store round trips per tile = 38-46 k cycles, whatever the activation (round 4, tools/gemm_lab/ts_probe.py: 40 % of a K = 1280 tile).
// Now every residual vector of the tile is fetched up front into the registers the dead operand fragments leave free (16 x 16 bytes per
// lane), one wait, and the 16 stores go out back to back.  (`out_rows` launches - the projector's last GEMM - take the direct epilogue.)
template <int ACT, int STORES = 2, bool TS = false, int POL = 0>
__device__ __forceinline__ void g2_epilogue_row_act(const GemmArgs& a, f4 (&acc)[4][8], int mb, int nb, int lane, int w, char* smem, int wfree, int afree,
                                                    unsigned long long* st = nullptr) {
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
    if constexpr (TS) asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st[0])::"memory");     // bias + residual have landed
    g2_epilogue_row_u<0, ACT, STORES, POL>(a, acc, bias, res, mb, nb, lane, slab, TS ? st + 3 : nullptr);
    if constexpr (TS) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st[1])::"memory");
    g2_epilogue_row_u<1, ACT, STORES, POL>(a, acc, bias, res, mb, nb, lane, slab);
    g2_epilogue_row_u<2, ACT, STORES, POL>(a, acc, bias, res, mb, nb, lane, slab);
    g2_epilogue_row_u<3, ACT, STORES, POL>(a, acc, bias, res, mb, nb, lane, slab);
    g2_epilogue_row_u<4, ACT, STORES, POL>(a, acc, bias, res, mb, nb, lane, slab);
    g2_epilogue_row_u<5, ACT, STORES, POL>(a, acc, bias, res, mb, nb, lane, slab);
    g2_epilogue_row_u<6, ACT, STORES, POL>(a, acc, bias, res, mb, nb, lane, slab);
    g2_epilogue_row_u<7, ACT, STORES, POL>(a, acc, bias, res, mb, nb, lane, slab);
    if constexpr (TS) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st[2])::"memory");
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
#ifdef AUR_LABS
template <> struct G2TagAct<GT_LAB_BASE + 26> { static constexpr int act = ACT_NONE; };
#endif      // an erf-GELU tower (config.hidden_act) runs GT_OTHER
static int g2_tag_act(int tag) {
    switch (tag) {
        case GT_VIT_OUT: case GT_VIT_FC2: case GT_LLM_O: case GT_LLM_DOWN: return ACT_NONE;
        case GT_LLM_GATEUP: return ACT_SILU_MUL;
        case GT_VIT_FC1: return ACT_QUICK_GELU;
        default: return -2;
    }
}
template <int TAG>
__device__ __forceinline__ void g2_epilogue_row(const GemmArgs& a, f4 (&acc)[4][8], int mb, int nb, int lane, int w, char* smem, int wfree, int afree,
                                                unsigned long long* st = nullptr) {
    constexpr int STORES = G2Lab<TAG>::epi_nostore ? 0 : G2Lab<TAG>::epi_halfstore ? -1 : 2;
    if constexpr (TAG == GT_OTHER) {
        // The generic kernel (projector, patch embedding, aur_linear) used to run the ladder variant for every activation: 78 KB of
        // epilogue code of which a tile executes a scattered tenth - 20 k cycles per tile on a quiet chip, 36 k once the operands have pushed
        // the lines out of L2 (tools/gemm_lab/epi_probe.py), against 8-9 k for the 7.5 KB of a compile-time activation.  Plain bias /
        // residual launches now take their own lean copy; only a launch with an activation pays for the ladder.
        if (a.act == ACT_NONE) g2_epilogue_row_act<ACT_NONE, STORES, G2Lab<TAG>::ts, G2Lab<TAG>::store_aux>(a, acc, mb, nb, lane, w, smem, wfree, afree, st);
        else g2_epilogue_row_act<-2, STORES, G2Lab<TAG>::ts, G2Lab<TAG>::store_aux>(a, acc, mb, nb, lane, w, smem, wfree, afree, st);
    } else {
        g2_epilogue_row_act<G2TagAct<TAG>::act, STORES, G2Lab<TAG>::ts, G2Lab<TAG>::store_aux>(a, acc, mb, nb, lane, w, smem, wfree, afree, st);
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
    // tile of a block id.  tile_order 1 (default): rounds of gridDim.x tiles are compact blocks of the tile space, one 4 x sn
    // sub-block per XCD (tile_order.h).  tile_order 0 (round 1 / early round 2): XCD-aware bijective remap (guide T1), then super-rows
    // of 4 M-tiles, column-major inside - every XCD walks its own contiguous range of the tile space, so an XCD's 32 consecutive tiles
    // are 4 (M) x 8 (N) and nothing is shared between XCDs in time.  gridDim.x is a multiple of 8 whenever a workgroup walks more than
    // one tile, so its tiles keep the XCD (bid % 8) both orders assume.
    TileOrder ord;
    tile_order_init(ord, nbm, nbn, a.tile_order >= 1 ? (int)gridDim.x : 0);
    auto tile_of = [&](int bid, int& bm, int& bn) {
        if (ord.full > 0 || a.tile_order >= 1) {
            tile_of_bid(ord, bid, bm, bn);
            return;
        }
        const int xcd = bid & 7, q = nwg >> 3, rem8 = nwg & 7;
        const int lid = (xcd < rem8 ? xcd * (q + 1) : rem8 * (q + 1) + (xcd - rem8) * q) + (bid >> 3);
        const int sr = lid / (4 * nbn), rem = lid - sr * 4 * nbn;
        const int rows = (nbm - 4 * sr) < 4 ? (nbm - 4 * sr) : 4;
        bn = rem / rows;
        bm = 4 * sr + rem % rows;
    };
    if constexpr (G2Lab<TAG>::stagger > 0) {
        const int slot = ((int)blockIdx.x >> 3) % G2Lab<TAG>::stagger;          // block b runs on XCD b % 8: CUs of one XCD take different slots
        const int tile_cycles = (a.K >> 6) * 3000 + 40000;                   // K loop + epilogue at ~2 GHz (order of magnitude is enough)
        const int naps = slot * tile_cycles / G2Lab<TAG>::stagger / (64 * 100);
        if (nwg > (int)gridDim.x)                                             // only when a workgroup walks more than one tile
            for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(100);
    }
    int bm, bn;
    G2Src src;
    tile_of(blockIdx.x, bm, bn);
    g2_sources<TAG>(a, bm, bn, w, lane, src);
    g2_prologue<TAG>(a, src, smem, w);
    const int nkt = a.K >> 6;
    const bool cont = G2Lab<TAG>::cont && nkt >= 3;          // the continuous pipeline needs K-tiles nkt - 2 and nkt - 1 to be distinct from K-tile 0
    int ap = 0, wp = 0;                                      // LDS buffers of this tile's K-tile 0 (g2_mainloop)
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
        const bool has_next = nxt < nwg;
        int bm2 = bm, bn2 = bn;
        if (has_next) tile_of(nxt, bm2, bn2);
        [[maybe_unused]] unsigned long long tT0 = 0, tT1 = 0, tT2 = 0, tT3 = 0;
        if constexpr (G2Lab<TAG>::ts) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tT0));
        const int wait = first ? 0 : (cont ? 2 : 1);
        if (vmode) g2_mainloop<EPI, true, TAG>(a, src, smem, acc, w, lane, wait, ap, wp, cont && has_next, bm2, bn2);
        else g2_mainloop<EPI, false, TAG>(a, src, smem, acc, w, lane, wait, ap, wp, cont && has_next, bm2, bn2);
        if constexpr (G2Lab<TAG>::ts) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tT1));
        first = false;
        if (cont) {                                          // the next tile's K-tiles 0 and 1 are staged already: only the buffer phases move on
            ap = (ap + nkt) & 1;
            wp = (wp + nkt) % 3;
        } else if (has_next) {
            // short K loops: every wave is past the K loop's last barrier, all LDS reads of this tile are done -> start the next tile's
            // loads now, they fly while this tile's epilogue converts and stores
            g2_sources<TAG>(a, bm2, bn2, w, lane, src);
            g2_prologue<TAG>(a, src, smem, w);
        }
        bm = bm2;
        bn = bn2;
        if constexpr (G2Lab<TAG>::ts) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tT2));
        // 16-byte row accesses need 8-column alignment of every row (ldc, ldr multiples of 8 halves; SiLU*up writes n / 2: ldc % 4)
        const bool wide = EPI == EPI_ROW && a.wide_epilogue && a.out_rows == nullptr && (a.act == ACT_SILU_MUL ? (a.ldc & 3) == 0 : ((a.ldc | (a.resid ? a.ldr : 0)) & 7) == 0);
        // LDS the next tile's staged K-tiles do not occupy: the W buffer of its K-tile 2 and the second half of the A buffer of its K-tile 1
        unsigned long long est[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if constexpr (G2Lab<TAG>::wide_only) g2_epilogue_row<TAG>(a, acc, mb, nb, lane, w, smem, (wp + 2) % 3, (ap + 1) & 1, est);
        else if (wide) g2_epilogue_row<TAG>(a, acc, mb, nb, lane, w, smem, (wp + 2) % 3, (ap + 1) & 1, est);
        else gemm_epilogue<EPI, 4, 8>(a, acc, mb, nb, lane, vmode);
#ifdef AUR_LABS
        if constexpr (G2Lab<TAG>::ts) {
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tT3));
            if (lane == 0 && blockIdx.x < 256) {
                unsigned* o = g2_ts + (blockIdx.x * 8 + w) * 32;
                o[22] += (unsigned)(tT1 - tT0);
                o[23] += (unsigned)(tT2 - tT1);
                o[24] += (unsigned)(tT3 - tT2);
                o[25] += 1;
                o[26] += (unsigned)(est[0] - tT2);       // epilogue entry -> bias / residual landed
                o[27] += (unsigned)(est[1] - est[0]);    // row block 0
                o[28] += (unsigned)(est[2] - est[1]);    // row blocks 1 .. 7
                o[29] += (unsigned)(tT3 - est[2]);
                o[30] += (unsigned)(est[3] - est[0]);    // row block 0: bias adds + 4 ds_write_b128 issued
                o[31] += (unsigned)(est[4] - est[3]);    //              first half's reads back
                g2_ts2[(blockIdx.x * 8 + w) * 4 + 0] += (unsigned)(est[5] - est[4]);    // first half converted + stored
                g2_ts2[(blockIdx.x * 8 + w) * 4 + 1] += (unsigned)(est[6] - est[5]);    // second half's reads back
                g2_ts2[(blockIdx.x * 8 + w) * 4 + 2] += (unsigned)(est[7] - est[6]);    // second half stored
            }
        }
#endif
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
#ifdef AUR_LABS
    if ((e = g2_attr<EPI_ROW, GT_LAB_BASE + 1>()) != hipSuccess || (e = g2_attr<EPI_ROW, GT_LAB_BASE + 2>()) != hipSuccess ||
        (e = g2_attr<EPI_ROW, GT_LAB_BASE + 3>()) != hipSuccess || (e = g2_attr<EPI_ROW, GT_LAB_BASE + 4>()) != hipSuccess ||
        (e = g2_attr<EPI_ROW, GT_LAB_BASE + 5>()) != hipSuccess || (e = g2_attr<EPI_ROW, GT_LAB_BASE + 6>()) != hipSuccess ||
        (e = g2_attr<EPI_ROW, GT_LAB_BASE + 7>()) != hipSuccess || (e = g2_attr<EPI_ROW, GT_LAB_BASE + 8>()) != hipSuccess ||
        (e = g2_attr<EPI_ROW, GT_LAB_BASE + 9>()) != hipSuccess || (e = g2_attr<EPI_ROW, GT_LAB_BASE + 10>()) != hipSuccess ||
        (e = g2_attr<EPI_ROW, GT_LAB_BASE + 11>()) != hipSuccess || (e = g2_attr<EPI_ROW, GT_LAB_BASE + 12>()) != hipSuccess ||
        (e = g2_attr<EPI_ROW, GT_LAB_BASE + 13>()) != hipSuccess || (e = g2_attr<EPI_ROW, GT_LAB_BASE + 14>()) != hipSuccess ||
        (e = g2_attr<EPI_ROW, GT_LAB_BASE + 15>()) != hipSuccess || (e = g2_attr<EPI_ROW, GT_LAB_BASE + 16>()) != hipSuccess ||
        (e = g2_attr<EPI_ROW, GT_LAB_BASE + 17>()) != hipSuccess || (e = g2_attr<EPI_ROW, GT_LAB_BASE + 18>()) != hipSuccess ||
        (e = g2_attr<EPI_ROW, GT_LAB_BASE + 19>()) != hipSuccess || (e = g2_attr<EPI_ROW, GT_LAB_BASE + 20>()) != hipSuccess ||
        (e = g2_attr<EPI_ROW, GT_LAB_BASE + 21>()) != hipSuccess || (e = g2_attr<EPI_ROW, GT_LAB_BASE + 22>()) != hipSuccess ||
        (e = g2_attr<EPI_ROW, GT_LAB_BASE + 23>()) != hipSuccess || (e = g2_attr<EPI_ROW, GT_LAB_BASE + 24>()) != hipSuccess ||
        (e = g2_attr<EPI_ROW, GT_LAB_BASE + 25>()) != hipSuccess || (e = g2_attr<EPI_ROW, GT_LAB_BASE + 26>()) != hipSuccess)
        return e;
#endif
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
    if (epi == EPI_ROW && a.lab > 0) {            // lab instantiations only
#ifndef AUR_LABS
        return hipErrorInvalidValue;
#else
        switch (a.lab) {
            case 1: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 1); break;
            case 2: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 2); break;
            case 3: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 3); break;
            case 4: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 4); break;
            case 5: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 5); break;
            case 6: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 6); break;
            case 7: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 7); break;
            case 8: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 8); break;
            case 9: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 9); break;
            case 10: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 10); break;
            case 11: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 11); break;
            case 12: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 12); break;
            case 13: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 13); break;
            case 14: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 14); break;
            case 15: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 15); break;
            case 16: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 16); break;
            case 17: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 17); break;
            case 18: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 18); break;
            case 19: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 19); break;
            case 20: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 20); break;
            case 21: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 21); break;
            case 22: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 22); break;
            case 23: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 23); break;
            case 24: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 24); break;
            case 25: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 25); break;
            case 26: G2_LAUNCH(EPI_ROW, GT_LAB_BASE + 26); break;
            default: return hipErrorInvalidValue;
        }
#endif
    } else if (epi == EPI_ROW) {
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

#ifdef AUR_LABS
// lab 8 read-out (libaurora_hip_labs.so only; not part of include/aurora_hip.h): copies g2_ts to the host and clears it
extern "C" int aur_lab_gemm_ts2(unsigned* dst_host, int clear) {
    if (hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g2_ts2), sizeof(unsigned) * 256 * 8 * 4) != hipSuccess) return -1;
    if (clear) {
        static unsigned zeros[256 * 8 * 4];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g2_ts2), zeros, sizeof zeros) != hipSuccess) return -1;
    }
    return 0;
}
extern "C" int aur_lab_gemm_ts(unsigned* dst_host, int clear) {
    if (hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g2_ts), sizeof(unsigned) * 256 * 8 * 32) != hipSuccess) return -1;
    if (clear) {
        static unsigned zeros[256 * 8 * 32];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g2_ts), zeros, sizeof zeros) != hipSuccess) return -1;
    }
    return 0;
}
#endif
