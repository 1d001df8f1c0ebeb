// MFMA GEMM for gfx950:  C[M, N] = A[M, K] (fp16 row-major) x W[N, K]^T (fp16, fragment-packed)
//
// Replaces every nn.Linear / conv-as-GEMM on the AuroraCap path (reference: aurora.py:613-616,634-636,
// 699 q/k/v/out_proj; HF CLIPMLP fc1/fc2 via aurora.py:710; modeling_projector.py:20-33; HF Llama
// q/k/v/o/gate/up/down_proj) - fp16 operands, fp32 accumulation, fused epilogues.
//
// Design (MI355X-first):
//  * W is pre-packed once into FRAG tiles [N/16][K/32][64 lanes][8] (common.h): a wave stages one
//    1 KiB fragment with a single global_load_lds_dwordx4 (LDS-DMA, lane-linear) and reads it back
//    with one conflict-free ds_read_b128 - no swizzle, no VGPR round trip.
//  * A stays row-major in HBM; its 128-byte tile rows are LDS-DMA'd with the XOR swizzle applied on
//    the SOURCE chunk index (guide rule 21), chunk' = chunk ^ ((row >> 1) & 7), which makes the
//    16-lane ds_read_b128 groups conflict-free.
//  * 128x128x64 block tile, 256 threads = 2x2 waves of 64x64, 16x16x32 f16 MFMA, double-buffered LDS,
//    one barrier per K-step, next tile's DMA in flight during the MFMAs.
//  * The MFMA is issued "swapped" (W fragment as the A operand): a lane ends up with 4 consecutive
//    output columns of one row -> 8-byte row-major stores.  For the V third of a fused QKV projection
//    the operands are exchanged so a lane holds 4 consecutive TOKENS of one feature instead, which is
//    exactly a V^T operand fragment.  The QKV epilogue therefore writes Q, K (PAIRED-d fragments, with
//    RoPE applied in-register for Llama) and V^T (PAIRED-token fragments) straight into the attention
//    operand layout / paged KV cache with 16-byte stores: attention never transposes anything.
#include "kernels.h"
#include "gemm_epilogue.h"

#define BM 128
#define BN 128
#define BK 64
#define STAGE_BYTES 32768
#define GEMM_LDS (2 * STAGE_BYTES)

template <int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w & 1, wm = w >> 1;
    const int r = lane & 15, g = lane >> 4;

    // XCD-aware bijective remap (guide T1): consecutive logical tiles share an XCD's L2.
    const int nbn = a.Npad / BN;
    const int nbm = (a.M + BM - 1) / BM;
    const int nwg = nbn * nbm;
    int lid;
    {
        const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, rem = nwg & 7;
        lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    }
    const int bm = lid / nbn, bn = lid % nbn;
    const int m0 = bm * BM, n0 = bn * BN;
    const int K32 = a.K >> 5, nkt = a.K >> 6;

    // per-lane source pointers for the A tile (row clamp keeps tail tiles in bounds)
    const half_t* a_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (w * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int m = m0 + row;
        m = m < a.M ? m : a.M - 1;
        a_src[i] = a.A + (int64_t)m * a.lda + c * 8;
    }
    const half_t* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = w * 4 + i;
        w_src[i] = a.W + ((int64_t)(bn * 8 + (f >> 1)) * K32 + (f & 1)) * AUR_FRAG_HALVES + lane * 8;
    }

    auto stage = [&](int kt, int buf) {
        char* Wt = smem + buf * STAGE_BYTES;
        char* At = Wt + 16384;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(w_src[i] + (int64_t)kt * 2 * AUR_FRAG_HALVES, Wt + (w * 4 + i) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(a_src[i] + kt * BK, At + (w * 4 + i) * 1024);
    };

    f4 acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = f4{0.f, 0.f, 0.f, 0.f};

    const bool vmode = (EPI == EPI_QKV) && (n0 + wn * 64 >= a.q_cols + a.k_cols);

    // LDS byte offsets of this lane's A fragments (swizzled), constant over the K loop
    int a_off[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int row = wm * 64 + u * 16 + r;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) a_off[u][kk] = 16384 + row * 128 + (((kk * 4 + g) ^ ((row >> 1) & 7)) << 4);
    }

    stage(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nkt) stage(kt + 1, (kt + 1) & 1);
        const char* base = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            h8 wf[4], af[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) wf[t] = *(const h8*)(base + ((wn * 4 + t) * 2 + kk) * 1024 + lane * 16);
#pragma unroll
            for (int u = 0; u < 4; ++u) af[u] = *(const h8*)(base + a_off[u][kk]);
            if (vmode) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[t][u] = mfma16(af[u], wf[t], acc[t][u]);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[t][u] = mfma16(wf[t], af[u], acc[t][u]);
            }
        }
    }

    // ------------------------------------------------------------------ epilogues (shared with the 256x256 kernel: the two agree bit for bit)
    if (EPI == EPI_ROW && a.act == ACT_NONE) gemm_epilogue<EPI, 4, 4, ACT_NONE>(a, acc, m0 + wm * 64, n0 + wn * 64, lane, vmode);
    else if (EPI == EPI_ROW && a.act == ACT_QUICK_GELU) gemm_epilogue<EPI, 4, 4, ACT_QUICK_GELU>(a, acc, m0 + wm * 64, n0 + wn * 64, lane, vmode);
    else if (EPI == EPI_ROW && a.act == ACT_SILU_MUL) gemm_epilogue<EPI, 4, 4, ACT_SILU_MUL>(a, acc, m0 + wm * 64, n0 + wn * 64, lane, vmode);
    else gemm_epilogue<EPI, 4, 4>(a, acc, m0 + wm * 64, n0 + wn * 64, lane, vmode);
}

hipError_t gemm_init() {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<EPI_ROW>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)gemm_kernel<EPI_QKV>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
}


static hipError_t launch_gemm128(const GemmArgs& a, int epi, hipStream_t s) {
    const int nbn = a.Npad / BN, nbm = (a.M + BM - 1) / BM;
    dim3 grid(nbn * nbm), block(256);
    if (epi == EPI_ROW)
        hipLaunchKernelGGL(gemm_kernel<EPI_ROW>, grid, block, GEMM_LDS, s, a);
    else
        hipLaunchKernelGGL(gemm_kernel<EPI_QKV>, grid, block, GEMM_LDS, s, a);
    return hipGetLastError();
}

hipError_t launch_gemm(const GemmArgs& a, int epi, hipStream_t s) {
    // the QKV epilogue treats a 16-row block as 16 consecutive tokens of one sequence, one page and one fragment row (gemm_epilogue.h)
    // EPI_QKV: every 16-row block is 16 consecutive tokens of one sequence, page and fragment row; the V third pairs two of them
    if (epi == EPI_QKV && (a.rows_per_seq <= 0 || ((a.M | a.rows_per_seq | a.pos0 | a.kv.page_tokens) & 15) != 0 || (a.rows_per_seq & 31) != 0 ||
                           a.M % a.rows_per_seq != 0)) return hipErrorInvalidValue;
    if (!((a.gemm_mode == 1 && gemm256_eligible(a)) || (a.gemm_mode == 2 && (a.Npad & 255) == 0))) return launch_gemm128(a, epi, s);
    // Tile quantisation: the persistent 256x256 kernel runs ceil(tiles / G) rounds of G workgroups, and a GEMM of few rounds can
    // leave most of the last one idle (o / down projection of a 4-clip prefill pass: 544 tiles = 4.25 rounds of 128, 2.1 of 256).
    // When the last round would be less than ~60 % full, the bottom rows go to the 128x128 kernel instead (a quarter of the work
    // per tile at ~0.75 of the rate): 4.25 -> ~4.4 rounds' time instead of 5.  Row-wise split, EPI_ROW only (rows carry no position
    // there); both kernels accumulate every output element over k in the same order, so the result is bit-identical (tested).
    if (a.tail_split && epi == EPI_ROW) {
        const int nbn = a.Npad >> 8, nbm = (a.M + 255) >> 8, tiles = nbn * nbm;
        const int G = a.max_wgs > 0 ? ((a.max_wgs + 7) & ~7) : gemm256_grid_cap();
        const int full = tiles / G;
        if (full >= 2 && tiles - full * G > 0) {
            const int mfull = full * G / nbn;                           // whole M-tile rows that fit `full` rounds
            const int m0 = mfull * 256;
            if (m0 > 0 && m0 < a.M) {
                const int small = ((a.M - m0 + BM - 1) / BM) * (a.Npad / BN);
                const float t_new = (float)((mfull * nbn + G - 1) / G) + 0.34f * (float)((small + G - 1) / G) + 0.03f;
                if (t_new + 0.1f < (float)(full + 1)) {
                    GemmArgs hi = a, lo = a;
                    hi.M = m0;
                    lo.M = a.M - m0;
                    lo.A = a.A + (int64_t)m0 * a.lda;
                    if (a.out_rows) lo.out_rows = a.out_rows + m0;
                    else lo.C = a.C + (int64_t)m0 * a.ldc;
                    if (a.resid) lo.resid = a.resid + (int64_t)m0 * a.ldr;
                    hipError_t e = launch_gemm256(hi, epi, s);
                    if (e != hipSuccess) return e;
                    return launch_gemm128(lo, epi, s);
                }
            }
        }
    }
    return launch_gemm256(a, epi, s);
}

// ---------------------------------------------------------------------------------------------
// Weight packing: row-major W[N_src, K_src] fp16 -> FRAG tiles [Npad/16][Kpad/32][64][8] (LINEAR).
// row_map[n] = source row of packed row n (or -1 -> zero row): expresses QKV concatenation, ViT head
// padding 80->96, the Llama RoPE pairing permutation and gate/up interleaving without copying weights
// on the host.
// ---------------------------------------------------------------------------------------------
__global__ void pack_weight_kernel(const half_t* __restrict__ w, int n_src, int k_src, int ld_src,
                                   const int32_t* __restrict__ row_map, int npad, int kpad, half_t* __restrict__ out) {
    const int64_t piece = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one 16-byte piece per thread
    const int K32 = kpad >> 5;
    const int64_t total = (int64_t)(npad >> 4) * K32 * 64;
    if (piece >= total) return;
    const int lane = piece & 63;
    const int64_t frag = piece >> 6;
    const int k32 = frag % K32;
    const int n16 = frag / K32;
    const int r = lane & 15, g = lane >> 4;
    const int n = n16 * 16 + r;
    const int src_row = row_map ? row_map[n] : (n < n_src ? n : -1);
    h8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k32 * 32 + g * 8 + j;
        o[j] = (src_row >= 0 && k < k_src) ? w[(int64_t)src_row * ld_src + k] : (half_t)0.f;
    }
    *(h8*)(out + piece * 8) = o;
}

hipError_t launch_pack_weight(const half_t* w, int n_src, int k_src, int ld_src, const int32_t* row_map, int npad,
                              int kpad, half_t* out, hipStream_t s) {
    const int64_t total = (int64_t)(npad >> 4) * (kpad >> 5) * 64;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, n_src, k_src, ld_src,
                       row_map, npad, kpad, out);
    return hipGetLastError();
}
