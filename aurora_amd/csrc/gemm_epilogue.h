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

template <int EPI, int TN, int TM>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f4 (&acc)[TN][TM], int mb, int nb, int lane, bool vmode) {
    const int r = lane & 15, g = lane >> 4;
    if (EPI == EPI_ROW) {
#pragma unroll
        for (int u = 0; u < TM; ++u) {
            const int m = mb + u * 16 + r;
            if (m >= a.M) continue;
            const int orow = a.out_rows ? a.out_rows[m] : m;
            if (orow < 0) continue;
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const int n = nb + t * 16 + 4 * g;
                if (n >= a.n_real) continue;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[t][u][i] + (a.bias ? a.bias[n + i] : 0.f);
                if (a.act == ACT_SILU_MUL) {
                    h2 o;
                    o[0] = (half_t)(silu_f(v[0]) * v[1]);
                    o[1] = (half_t)(silu_f(v[2]) * v[3]);
                    *(h2*)(a.C + (int64_t)orow * a.ldc + (n >> 1)) = o;
                } else {
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
                    if (a.resid) {
                        const h4 rr = *(const h4*)(a.resid + (int64_t)m * a.ldr + n);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] += (float)rr[i];
                    }
                    h4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (half_t)v[i];
                    *(h4*)(a.C + (int64_t)orow * a.ldc + n) = o;
                }
            }
        }
    } else {
        const KvLayout& kv = a.kv;
        if (!vmode) {
            const bool is_q = nb < a.q_cols;
            const int nreg = is_q ? nb : nb - a.q_cols;
            const int T16 = a.rows_per_seq >> 4;
#pragma unroll
            for (int p = 0; p < TN / 2; ++p) {
                const int blkg = (nreg >> 5) + p;
                const int head = blkg / kv.kblk, blk = blkg % kv.kblk;
                if (head >= kv.heads) continue;            // region padding (to a multiple of 64 columns)
                const int n1 = nb + p * 32 + 4 * g, n2 = n1 + 16;
#pragma unroll
                for (int u = 0; u < TM; ++u) {
                    const int m = mb + u * 16 + r;
                    if (m >= a.M) continue;
                    const int seq = m / a.rows_per_seq, tok = m % a.rows_per_seq;
                    float x1[4], x2[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        x1[i] = acc[2 * p][u][i] + (a.bias ? a.bias[n1 + i] : 0.f);
                        x2[i] = acc[2 * p + 1][u][i] + (a.bias ? a.bias[n2 + i] : 0.f);
                    }
                    if (a.rope) {
                        const float2* cs = a.rope + (int64_t)(a.pos0 + tok) * (a.hd >> 1) + blk * 16 + 4 * g;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float2 c = cs[i];
                            const float y1 = x1[i] * c.x - x2[i] * c.y;
                            const float y2 = x2[i] * c.x + x1[i] * c.y;
                            x1[i] = y1;
                            x2[i] = y2;
                        }
                    }
                    h8 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        o[i] = (half_t)x1[i];
                        o[4 + i] = (half_t)x2[i];
                    }
                    if (is_q) {
                        half_t* dst = a.Qf + ((((int64_t)seq * kv.heads + head) * T16 + (tok >> 4)) * kv.kblk + blk) * AUR_FRAG_HALVES +
                                      (g * 16 + (tok & 15)) * 8;
                        qkv_store(a.nt_out != 0, dst, o);
                    } else {
                        const int pos = a.pos0 + tok;
                        half_t* dst = kv_page(kv, a.seq0 + seq, pos) + kfrag_off(kv, head, (pos % kv.page_tokens) >> 4, blk) +
                                      (g * 16 + (pos & 15)) * 8;
                        qkv_store(a.nt_out != 0, dst, o);
                    }
                }
            }
        } else {
            const int nreg = nb - a.q_cols - a.k_cols;
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const int idx = (nreg >> 4) + t;
                const int head = idx / kv.vd16, d16 = idx % kv.vd16;
                if (head >= kv.heads) continue;            // Npad padding beyond the V region
                const float b = a.bias ? a.bias[nb + t * 16 + r] : 0.f;
#pragma unroll
                for (int pu = 0; pu < TM / 2; ++pu) {
                    const int mrow = mb + pu * 32;
                    if (mrow >= a.M) continue;
                    const int seq = mrow / a.rows_per_seq, tok = mrow % a.rows_per_seq;
                    const int pos = a.pos0 + tok;
                    h8 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        o[i] = (half_t)(acc[t][2 * pu][i] + b);
                        o[4 + i] = (half_t)(acc[t][2 * pu + 1][i] + b);
                    }
                    half_t* dst = kv_page(kv, a.seq0 + seq, pos) + vfrag_off(kv, head, d16, (pos % kv.page_tokens) >> 5) +
                                  (g * 16 + r) * 8;
                    qkv_store(a.nt_out != 0, dst, o);
                }
            }
        }
    }
}
