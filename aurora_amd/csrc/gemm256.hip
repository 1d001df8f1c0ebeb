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
#include "gemm_epilogue.h"

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

template <int EPI, bool VMODE>
__device__ __forceinline__ void g2_mainloop(const GemmArgs& a, char* smem, f4 (&acc)[4][8], int bm, int bn, int w, int lane) {
    const int wr = w >> 2, wc = w & 3;
    const int r = lane & 15, g = lane >> 4;
    const int K32 = a.K >> 5, nkt = a.K >> 6;
    const int m0 = bm * 256;

    // per-thread LDS-DMA sources: 2 instructions per half-tile
    const half_t* a_src[2][2];
    const half_t* w_src[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int grp = w * 2 + i;
            const int row = grp * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            int m = m0 + h * 128 + row;
            m = m < a.M ? m : a.M - 1;
            a_src[h][i] = a.A + (int64_t)m * a.lda + c * 8;
            const int f = grp;                      // fragment of the half-tile: n16 = f >> 1, kk = f & 1
            w_src[h][i] = a.W + ((int64_t)(bn * 16 + h * 8 + (f >> 1)) * K32 + (f & 1)) * AUR_FRAG_HALVES + lane * 8;
        }
    auto stage_a = [&](int h, int kt) {
        char* dst = smem + (kt & 1) * G2_ABUF + h * G2_SLOT;
        glds16(a_src[h][0] + kt * 64, dst + (w * 2 + 0) * 1024);
        glds16(a_src[h][1] + kt * 64, dst + (w * 2 + 1) * 1024);
    };
    auto stage_w = [&](int h, int kt, int wb) {        // wb = kt % 3
        char* dst = smem + G2_WBASE + wb * G2_WBUF + h * G2_SLOT;
        glds16(w_src[h][0] + (int64_t)kt * 2 * AUR_FRAG_HALVES, dst + (w * 2 + 0) * 1024);
        glds16(w_src[h][1] + (int64_t)kt * 2 * AUR_FRAG_HALVES, dst + (w * 2 + 1) * 1024);
    };

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
#define G2_COMPUTE(NH, MH, wf)                                                                                   \
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

    // ---- prologue: K-tile 0 completely, plus W0, W1, A0 of K-tile 1 (stay in flight)
    stage_a(0, 0);
    stage_a(1, 0);
    stage_w(0, 0, 0);
    stage_w(1, 0, 0);
    if (nkt > 1) {
        stage_w(0, 1, 1);
        stage_w(1, 1, 1);
        stage_a(0, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    G2_BARRIER();
    if (wr == 1) G2_BARRIER();                    // stagger: waves 4-7 run one barrier behind

    int wb = 0;                                            // weight buffer of K-tile t (t % 3)
    for (int t = 0; t < nkt; ++t) {
        const char* abuf = smem + (t & 1) * G2_ABUF;
        const char* wbuf = smem + G2_WBASE + wb * G2_WBUF;
        const int wb2 = wb == 0 ? 2 : wb - 1;              // (t + 2) % 3
        const bool n1 = t + 1 < nkt, n2 = t + 2 < nkt;     // block-uniform
        // phase 0: quadrant (n half 0, m half 0)
        read_a(abuf, 0);
        read_w(wbuf, 0, wf0);
        if (n1) stage_a(1, t + 1);
        G2_BARRIER();
        G2_COMPUTE(0, 0, wf0);
        G2_BARRIER();
        // phase 1: (n half 1, m half 0)
        read_w(wbuf, 1, wf1);
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

template <int EPI>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbn = a.Npad >> 8;
    const int nbm = (a.M + 255) >> 8;
    const int nwg = nbn * nbm;
    // Persistent mode (gemm256_set_max_wgs): a fixed number of workgroups walks the tiles, so the GEMM occupies only that many
    // CUs (one 160 KiB workgroup per CU) and leaves the others to a concurrently running stream.  gridDim.x is a multiple of 8
    // there, so a workgroup's tiles keep the XCD (bid % 8) the remap assumes.  Default: one workgroup per tile.
    for (int bid = blockIdx.x; bid < nwg; bid += gridDim.x) {
    int lid;
    {   // XCD-aware bijective remap (guide T1)
        const int xcd = bid & 7, q = nwg >> 3, rem = nwg & 7;
        lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    }
    // super-rows of 4 M-tiles, column-major inside: an XCD's 32 consecutive tiles are 4 (M) x 8 (N), so every weight
    // half-tile is fetched from HBM once and re-used from the XCD's L2 by 4 workgroups, every activation tile by 8
    const int sr = lid / (4 * nbn), rem = lid - sr * 4 * nbn;
    const int rows = (nbm - 4 * sr) < 4 ? (nbm - 4 * sr) : 4;
    const int bn = rem / rows, bm = 4 * sr + rem % rows;
    f4 acc[4][8];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[t][u] = f4{0.f, 0.f, 0.f, 0.f};
    const int nb = bn * 256 + (w & 3) * 64;
    const int mb = bm * 256 + (w >> 2) * 128;
    const bool vmode = (EPI == EPI_QKV) && (nb >= a.q_cols + a.k_cols);
    if (vmode) g2_mainloop<EPI, true>(a, smem, acc, bm, bn, w, lane);
    else g2_mainloop<EPI, false>(a, smem, acc, bm, bn, w, lane);
    gemm_epilogue<EPI, 4, 8>(a, acc, mb, nb, lane, vmode);
    __syncthreads();                              // the next tile's prologue refills LDS
    }
}

hipError_t gemm256_init() {
    hipError_t e = hipFuncSetAttribute((const void*)gemm256_kernel<EPI_ROW>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)gemm256_kernel<EPI_QKV>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
}

bool gemm256_eligible(const GemmArgs& a) {
    // whole 256-column tiles, and enough 256x256 tiles to fill the 256 CUs at one workgroup each
    if (a.Npad & 255) return false;
    const int64_t tiles = (int64_t)(a.Npad >> 8) * ((a.M + 255) >> 8);
    return tiles >= 224;
}

static int g_gemm256_max_wgs = 0;
void gemm256_set_max_wgs(int n) { g_gemm256_max_wgs = n > 0 ? (n + 7) & ~7 : 0; }

hipError_t launch_gemm256(const GemmArgs& a, int epi, hipStream_t s) {
    int ntiles = (a.Npad >> 8) * ((a.M + 255) >> 8);
    if (g_gemm256_max_wgs > 0 && ntiles > g_gemm256_max_wgs) ntiles = g_gemm256_max_wgs;
    dim3 grid(ntiles), block(512);
    if (epi == EPI_ROW) hipLaunchKernelGGL(gemm256_kernel<EPI_ROW>, grid, block, G2_LDS, s, a);
    else hipLaunchKernelGGL(gemm256_kernel<EPI_QKV>, grid, block, G2_LDS, s, a);
    return hipGetLastError();
}
