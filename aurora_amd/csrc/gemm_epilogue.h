// Shared GEMM epilogue for a wave-owned region of TN x TM accumulator tiles (16x16 each).
// acc[t][u] holds, in the MFMA 16x16 C/D map (lane: row = 4*(lane>>4)+i, col = lane&15):
//   normal  : D[row = n (tile t)][col = m (tile u)]  - 4 consecutive output columns of one row per lane
//   V region: D[row = m (tile u)][col = n (tile t)]  - 4 consecutive tokens of one feature per lane
// See gemm.hip for the layout rationale (row-major stores, PAIRED Q/K fragments, V^T fragments).
#pragma once
#include "kernels.h"

// Q fragments and K / V page fragments: written in one burst per round of tiles, read by the NEXT launch (attention).  nt: GemmArgs.nt_out
template <typename V>
__device__ __forceinline__ void qkv_store(bool nt, half_t* dst, const V& v) {
    if (nt) __builtin_nontemporal_store(v, (V*)dst);
    else *(V*)dst = v;
}

// "The loaded value is needed HERE": an empty asm that reads the register makes hipcc place its `s_waitcnt` for the load at this point, in
// straight-line code.  Left to the first real use - inside the `m < M` branch around a store - the wait does not dominate the later uses, is
// repeated in front of every store and, stores and loads sharing vmcnt, then waits for the previous store's round trip.
template <typename T>
__device__ __forceinline__ void landed(const T& v) {
    asm volatile("" ::"v"(v));
}

// ---- EPI_QKV: Q / K columns.  Per (column pair block p, row block u) the lane holds 4 + 4 PAIRED features of one token: bias, RoPE, one
// 16-byte store into the Q fragment buffer or the token's K page.
// Loads are batched in front of stores.  Rounds 1-3 looked up the page, fetched the four (cos, sin) pairs and stored inside each of the 16
// (p, u) blocks: a load behind a store waits for `vmcnt(0)`, i.e. for the store to be acknowledged by memory (stores and loads share the
// counter; hipcc cannot count across the `m < M` branch around a store), so a tile paid ~30 dependent memory round trips - the QKV GEMMs
// ran 20-25 % below gate/up at the same K (round 4: 1.05 vs 1.31 PF/s).  Now: bias and the K pages of ALL row blocks up front (no branch
// around a load: row blocks past M read the last real block's entries and store nothing), then per PAIR of row blocks the RoPE entries
// (32 registers), convert, store - TM / 2 - 1 store round trips per tile instead of one per block.  Same expressions per element as before.
template <int TN, int TM>
__device__ __forceinline__ void qkv_epilogue_qk(const GemmArgs& a, f4 (&acc)[TN][TM], int mb, int nb, int lane) {
    constexpr int NP = TN / 2, UH = 2, NPH = TM / UH;
    const int r = lane & 15, g = lane >> 4;
    const KvLayout& kv = a.kv;
    const bool is_q = nb < a.q_cols;
    const int nreg = is_q ? nb : nb - a.q_cols;
    const int T16 = a.rows_per_seq >> 4;
    int head[NP], blk[NP];
    f4 b1[NP], b2[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int blkg = (nreg >> 5) + p;
        head[p] = blkg / kv.kblk;                      // >= kv.heads: region padding (to a multiple of 64 columns), nothing is stored
        blk[p] = blkg % kv.kblk;
        const int n1 = nb + p * 32 + 4 * g;
        b1[p] = a.bias ? *(const f4*)(a.bias + n1) : f4{0.f, 0.f, 0.f, 0.f};
        b2[p] = a.bias ? *(const f4*)(a.bias + n1 + 16) : f4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        landed(b1[p]);
        landed(b2[p]);
    }
    // Per row block u the 16 rows mb + 16 u + r are 16 consecutive tokens of ONE sequence, one page and one 16-token fragment row
    // (rows_per_seq, pos0, the page size and mb are multiples of 16; M is a multiple of rows_per_seq): sequence, page and fragment row
    // are wave-uniform - scalar divisions, a scalar page-table load - and a lane adds only its own 16 bytes (lane * 8 halves).
    int tokb[TM];                                       // first token of the row block in its sequence (uniform)
    int64_t row_off[TM];                                // halves from the Q buffer / the layer's page pool, without the (head, block) term (uniform)
#pragma unroll
    for (int u = 0; u < TM; ++u) {
        const int mrow = min(mb + u * 16, a.M - 16);   // row blocks past M: the last real block's entries are read, nothing is stored
        const int seq = mrow / a.rows_per_seq;
        tokb[u] = mrow - seq * a.rows_per_seq;
        if (is_q) row_off[u] = (((int64_t)seq * kv.heads) * T16 + (tokb[u] >> 4)) * kv.kblk * AUR_FRAG_HALVES;
        else {
            const int pos = a.pos0 + tokb[u];
            const int64_t pid = kv.page_table ? (int64_t)kv.page_table[(int64_t)(a.seq0 + seq) * kv.max_pages + pos / kv.page_tokens] : (int64_t)(a.seq0 + seq);
            row_off[u] = pid * kv.page_halves + kfrag_off(kv, 0, (pos % kv.page_tokens) >> 4, 0);
        }
    }
    half_t* const dst_base = (is_q ? a.Qf : kv.base) + lane * 8;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
        f4 cs[NP][UH][2];                              // four (cos, sin) pairs = 32 bytes per (p, row block)
        if (a.rope) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int uu = 0; uu < UH; ++uu) {
                    const f4* src = (const f4*)(a.rope + (int64_t)(a.pos0 + tokb[ph * UH + uu] + r) * (a.hd >> 1) + blk[p] * 16 + 4 * g);
                    cs[p][uu][0] = src[0];
                    cs[p][uu][1] = src[1];
                }
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int uu = 0; uu < UH; ++uu) {
                    landed(cs[p][uu][0]);
                    landed(cs[p][uu][1]);
                }
        }
        h8 o[NP][UH];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int uu = 0; uu < UH; ++uu) {
                const int u = ph * UH + uu;
                float x1[4], x2[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    x1[i] = acc[2 * p][u][i] + b1[p][i];
                    x2[i] = acc[2 * p + 1][u][i] + b2[p][i];
                }
                if (a.rope) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float cx = cs[p][uu][i >> 1][2 * (i & 1)], cy = cs[p][uu][i >> 1][2 * (i & 1) + 1];
                        const float y1 = x1[i] * cx - x2[i] * cy;
                        const float y2 = x2[i] * cx + x1[i] * cy;
                        x1[i] = y1;
                        x2[i] = y2;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o[p][uu][i] = (half_t)x1[i];
                    o[p][uu][4 + i] = (half_t)x2[i];
                }
            }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (head[p] >= kv.heads) continue;
            // (head, block) term: Q fragments [seq][head][tok16][blk], K fragments of a page [head][tok16][blk]
            const int64_t hb = is_q ? ((int64_t)head[p] * T16 * kv.kblk + blk[p]) * AUR_FRAG_HALVES : kfrag_off(kv, head[p], 0, blk[p]);
#pragma unroll
            for (int uu = 0; uu < UH; ++uu) {
                const int u = ph * UH + uu;
                if (mb + u * 16 < a.M) qkv_store(a.nt_out != 0, dst_base + row_off[u] + hb, o[p][uu]);
            }
        }
    }
}

// ---- EPI_QKV: V columns (MFMA operands exchanged: the lane holds 4 consecutive tokens of one feature, two row blocks make a 16-byte V^T
// piece).  Bias and the pages of all row-block pairs are fetched before the first store (same reason as above).
template <int TN, int TM>
__device__ __forceinline__ void qkv_epilogue_v(const GemmArgs& a, f4 (&acc)[TN][TM], int mb, int nb, int lane) {
    const int r = lane & 15, g = lane >> 4;
    const KvLayout& kv = a.kv;
    const int nreg = nb - a.q_cols - a.k_cols;
    float b[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) b[t] = a.bias ? a.bias[nb + t * 16 + r] : 0.f;
#pragma unroll
    for (int t = 0; t < TN; ++t) landed(b[t]);
    half_t* page_dst[TM / 2];
    int64_t pid[TM / 2];
    int tok_of[TM / 2];
#pragma unroll
    for (int pu = 0; pu < TM / 2; ++pu) {
        const int mrow = min(mb + pu * 32, (a.M - 1) & ~31);          // row blocks past M: the last real block's page (nothing is stored)
        tok_of[pu] = mrow % a.rows_per_seq;
        pid[pu] = a.seq0 + mrow / a.rows_per_seq;
    }
    if (kv.page_table) {
#pragma unroll
        for (int pu = 0; pu < TM / 2; ++pu) pid[pu] = kv.page_table[pid[pu] * kv.max_pages + (a.pos0 + tok_of[pu]) / kv.page_tokens];
    }
#pragma unroll
    for (int pu = 0; pu < TM / 2; ++pu) {
        const int pos = a.pos0 + tok_of[pu];
        page_dst[pu] = mb + pu * 32 < a.M ? kv.base + pid[pu] * kv.page_halves + vfrag_off(kv, 0, 0, (pos % kv.page_tokens) >> 5) + (g * 16 + r) * 8 : nullptr;
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int idx = (nreg >> 4) + t;
        const int head = idx / kv.vd16, d16 = idx % kv.vd16;
        if (head >= kv.heads) continue;            // Npad padding beyond the V region
        const int64_t hd_off = vfrag_off(kv, head, d16, 0) - kv.v_off;
#pragma unroll
        for (int pu = 0; pu < TM / 2; ++pu) {
            if (!page_dst[pu]) continue;
            h8 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[i] = (half_t)(acc[t][2 * pu][i] + b[t]);
                o[4 + i] = (half_t)(acc[t][2 * pu + 1][i] + b[t]);
            }
            qkv_store(a.nt_out != 0, page_dst[pu] + hd_off, o);
        }
    }
}

// ACTC: the activation as a compile-time constant, or -2 = read a.act at run time.  The run-time ladder is inlined into each of the TN x TM
// blocks (erf-GELU alone is ~40 instructions per element): 56 KB of epilogue in the 128x128 kernel, of which a launch executes a scattered
// fraction - instruction fetch, not arithmetic, then sets its time (gemm256.hip g2_epilogue_row, tools/gemm_lab/epi_probe.py).  The callers
// dispatch the common activations to their own lean copy once per tile.
template <int EPI, int TN, int TM, int ACTC = -2>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f4 (&acc)[TN][TM], int mb, int nb, int lane, bool vmode) {
    const int r = lane & 15, g = lane >> 4;
    const int act = ACTC == -2 ? a.act : ACTC;
    if (EPI == EPI_ROW) {
        // Every load of the region - output row map, bias, residual pieces - is issued before its first store and pinned there
        // (`landed`): rounds 1-3 fetched bias and residual inside each of the TN x TM blocks, behind the previous block's store, and each such
        // load waited for that store's round trip (see qkv_epilogue_qk).  Same arithmetic per element.
        int orow[TM];
#pragma unroll
        for (int u = 0; u < TM; ++u) {
            const int m = mb + u * 16 + r;
            orow[u] = m >= a.M ? -1 : (a.out_rows ? a.out_rows[m] : m);
        }
        f4 bias[TN];
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int n = nb + t * 16 + 4 * g;
            bias[t] = f4{0.f, 0.f, 0.f, 0.f};
            if (a.bias && n < a.n_real) {              // bias holds Npad >= n + 4 entries; columns past n_real are never stored
#pragma unroll
                for (int i = 0; i < 4; ++i) bias[t][i] = a.bias[n + i];
            }
        }
        const bool with_res = a.resid && act != ACT_SILU_MUL;
#pragma unroll
        for (int u = 0; u < TM; ++u) landed(orow[u]);
#pragma unroll
        for (int t = 0; t < TN; ++t) landed(bias[t]);
        // residual pieces: all row blocks of the 128x128 kernel at once, two at a time in the 256x256 kernel (its direct path is the fallback of the wide epilogue - the projector's row-mapped last GEMM - and has 128 accumulators live: more would spill)
        constexpr int UH = TM > 4 ? 2 : TM;
#pragma unroll
        for (int u0 = 0; u0 < TM; u0 += UH) {
            h4 res[UH][TN];
            if (with_res) {
#pragma unroll
                for (int uu = 0; uu < UH; ++uu)
#pragma unroll
                    for (int t = 0; t < TN; ++t) {
                        const int m = min(mb + (u0 + uu) * 16 + r, a.M - 1), n = min(nb + t * 16 + 4 * g, a.n_real - 4);      // clamped: loads need no branch
                        res[uu][t] = *(const h4*)(a.resid + (int64_t)m * a.ldr + n);
                    }
#pragma unroll
                for (int uu = 0; uu < UH; ++uu)
#pragma unroll
                    for (int t = 0; t < TN; ++t) landed(res[uu][t]);
            }
#pragma unroll
            for (int uu = 0; uu < UH; ++uu) {
                const int u = u0 + uu;
                if (orow[u] < 0) continue;
#pragma unroll
                for (int t = 0; t < TN; ++t) {
                    const int n = nb + t * 16 + 4 * g;
                    if (n >= a.n_real) continue;
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = acc[t][u][i] + bias[t][i];
                    if (act == ACT_SILU_MUL) {
                        h2 o;
                        o[0] = (half_t)(silu_f(v[0]) * v[1]);
                        o[1] = (half_t)(silu_f(v[2]) * v[3]);
                        *(h2*)(a.C + (int64_t)orow[u] * a.ldc + (n >> 1)) = o;
                    } else {
                        if (act == ACT_QUICK_GELU) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = quick_gelu_f(v[i]);
                        } else if (act == ACT_GELU) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = gelu_erf_f(v[i]);
                        } else if (act >= ACT_SILU) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = act_other_f(v[i], act);
                        }
                        if (with_res) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] += (float)res[uu][t][i];
                        }
                        h4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] = (half_t)v[i];
                        *(h4*)(a.C + (int64_t)orow[u] * a.ldc + n) = o;
                    }
                }
            }
        }
    } else if (!vmode) {
        qkv_epilogue_qk<TN, TM>(a, acc, mb, nb, lane);
    } else {
        qkv_epilogue_v<TN, TM>(a, acc, mb, nb, lane);
    }
}
