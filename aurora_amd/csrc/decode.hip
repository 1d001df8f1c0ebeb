// Autoregressive decode kernels for gfx950 (HBM-bound): skinny weight-streaming GEMM, paged decode
// attention over fragment-packed KV, argmax + next-token bookkeeping.
//
// Reference: third-party transformers LlamaForCausalLM.generate greedy loop as called at
// inference.py:89-96 (one position per step with a KV cache, argmax, EOS / max_new_tokens stop).
//
// skinny GEMM   y[b, n] = sum_k x[b, k] W[n, k]   for b <= 16 rows:
//   Weights are the same FRAG tiles the prefill GEMM uses ([N/16][K/32][64][8]); a wave streams one
//   1 KiB fragment per global_load_dwordx4 straight into VGPRs (no LDS round trip for a once-read
//   operand), x (<= 16 x K halves) is staged once per workgroup into LDS and read back as the MFMA B
//   operand; one 16x16x32 MFMA per KiB of weights keeps the VALU idle and the loop purely HBM-bound.
//   A workgroup owns NT n16 tiles over the full K; its 4 waves split K interleaved (wave w takes
//   k32 = 4i + w, so the 4 waves walk 4 consecutive KiB), reduced through LDS in fixed order.
//
// decode attention: one wave per (sequence, head, split of 64-token pages).  K fragments (rows = tokens)
//   and V^T fragments (rows = d) stream from the paged cache as contiguous 1 KiB pieces; q is replicated
//   across the 16 MFMA columns.  S^T accumulators become the PAIRED-token P operand in-lane.
#include "kernels.h"

#define SK_LDS_BUDGET (72 * 1024)

template <int NT, int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void skinny_kernel(SkinnyArgs a, int KC) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int K32 = a.K >> 5;
    const int tile0 = blockIdx.x * NT;
    half_t* xs = (half_t*)smem;
    const int xrow = (c < a.B) ? c : c % a.B;

    f4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};

    // The weight stream is ONE continuous software pipeline over all of this wave's k32 tiles (wave w owns
    // tiles w, w+4, ...): the first batch is in flight before x is staged, and it keeps running across the
    // x-chunk boundaries (chunks exist only because B x K halves may exceed the LDS budget).
    constexpr int U = 4;
    const int nit = a.K / (32 * NW);             // tiles per wave (wave w owns tiles w, w+NW, ...)
    const int tpc = KC / (32 * NW);              // tiles per wave per x-chunk (multiple of U when there are several chunks)
    const int ldxs = KC + 8;
    const half_t* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wp[t] = a.W + ((int64_t)(tile0 + t) * K32 + w) * AUR_FRAG_HALVES + lane * 8;
    const half_t* xp = xs + xrow * ldxs + w * 32 + g * 8;
    const int nfull = nit / U * U;
    h8 cur[U][NT], nxt[U][NT];
    if (nfull > 0) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                cur[u][t] = __builtin_nontemporal_load((const h8*)(wp[t] + (int64_t)u * NW * AUR_FRAG_HALVES));
    }
    // Fused RMSNorm (HF LlamaRMSNorm) at zero extra passes: y = rstd[b] * sum_k W[n,k] * (w_norm[k] * x[b,k]).
    // The staging pass multiplies by w_norm and accumulates sum(x^2) per row on the fly; rstd[b] scales the
    // accumulators in the epilogue.  Row b is staged and summed by wave (b mod NW) alone: lane partials over pieces
    // lane, lane+64, ... in chunk order, then a fixed butterfly - the same arithmetic for any batch size.
    const size_t x_bytes = (size_t)a.B * ldxs * 2, red_bytes = (size_t)NW * NT * 64 * 16;
    float* rs = (float*)(smem + (x_bytes > red_bytes ? x_bytes : red_bytes));   // [16] sum(x^2) per row
    auto stage = [&](int chunk) {               // block-uniform: every wave calls it at the same tile index
        __syncthreads();
        const int kc0 = chunk * KC;
        const int kc = (a.K - kc0) < KC ? (a.K - kc0) : KC;
        const int ppr = kc >> 3;
        if (a.norm_w) {
#pragma unroll
            for (int rr = 0; rr < 16 / NW; ++rr) {
                const int b = w + rr * NW;               // wave-uniform
                if (b < a.B) {
                    float ss = 0.f;
                    for (int p = lane; p < ppr; p += 64) {
                        h8 v = *(const h8*)(a.x + (int64_t)b * a.ldx + kc0 + p * 8);
                        const float* nw = a.norm_w + kc0 + p * 8;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float xv = (float)v[e];
                            ss += xv * xv;
                            v[e] = (half_t)(nw[e] * xv);
                        }
                        *(h8*)(xs + b * ldxs + p * 8) = v;
                    }
                    ss = wave_sum(ss);
                    if (lane == 0) rs[b] = (chunk == 0 ? 0.f : rs[b]) + ss;
                }
            }
        } else {
            for (int idx = tid; idx < a.B * ppr; idx += 64 * NW) {
                const int b = idx / ppr, p = idx % ppr;
                *(h8*)(xs + b * ldxs + p * 8) = *(const h8*)(a.x + (int64_t)b * a.ldx + kc0 + p * 8);
            }
        }
        __syncthreads();
    };
    int staged = -1;
    int i0 = 0;
    for (; i0 < nfull; i0 += U) {
        const int chunk = i0 / tpc;
        if (chunk != staged) {
            stage(chunk);
            staged = chunk;
        }
        const bool more = (i0 + 2 * U <= nfull);            // wave-uniform: one branch per batch
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    nxt[u][t] = __builtin_nontemporal_load((const h8*)(wp[t] + (int64_t)(i0 + U + u) * NW * AUR_FRAG_HALVES));
        }
        const int il = i0 - chunk * tpc;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const h8 xf = *(const h8*)(xp + (il + u) * (32 * NW));
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma16(cur[u][t], xf, acc[t]);
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int t = 0; t < NT; ++t) cur[u][t] = nxt[u][t];
        }
    }
    for (; i0 < nit; ++i0) {                     // tail (< U tiles)
        const int chunk = i0 / tpc;
        if (chunk != staged) {
            stage(chunk);
            staged = chunk;
        }
        const h8 xf = *(const h8*)(xp + (i0 - chunk * tpc) * (32 * NW));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const h8 wf = __builtin_nontemporal_load((const h8*)(wp[t] + (int64_t)i0 * NW * AUR_FRAG_HALVES));
            acc[t] = mfma16(wf, xf, acc[t]);
        }
    }
    // cross-wave reduction in fixed order
    __syncthreads();
    float* red = (float*)smem;       // [NW waves][NT][64 lanes][4]
#pragma unroll
    for (int t = 0; t < NT; ++t) *(f4*)(red + ((w * NT + t) * 64 + lane) * 4) = acc[t];
    __syncthreads();
    if (w != 0) return;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        f4 s = *(const f4*)(red + ((0 * NT + t) * 64 + lane) * 4);
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) {
            const f4 p = *(const f4*)(red + ((ww * NT + t) * 64 + lane) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] += p[i];
        }
        acc[t] = s;
    }
    const int b = c;
    if (b >= a.B) return;
    if (a.norm_w) {                                   // fused RMSNorm: per-row 1/rms applied to the contraction
        const float rstd = rsqrtf(rs[b] / (float)a.K + a.norm_eps);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[t][i] *= rstd;
    }

    if (MODE == SK_ROW || MODE == SK_LOGITS || MODE == SK_SILU_MUL) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = (tile0 + t) * 16 + 4 * g;
            if (n >= a.n_real) continue;
            if (MODE == SK_LOGITS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a.out32[(int64_t)b * a.n_real + n + i] = acc[t][i];
            } else if (MODE == SK_SILU_MUL) {
                h2 o;
                o[0] = (half_t)(silu_f(acc[t][0]) * acc[t][1]);
                o[1] = (half_t)(silu_f(acc[t][2]) * acc[t][3]);
                *(h2*)(a.out + (int64_t)b * a.ldo + (n >> 1)) = o;
            } else {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[t][i];
                if (a.resid) {
                    const h4 rr = *(const h4*)(a.resid + (int64_t)b * a.ldr + n);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += (float)rr[i];
                }
                h4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (half_t)v[i];
                *(h4*)(a.out + (int64_t)b * a.ldo + n) = o;
            }
        }
    } else if (NT == 2) {     // SK_QKV: the workgroup owns one PAIRED 32-column block
        const KvLayout& kv = a.kv;
        const int nb = tile0 * 16;
        const int pos = a.pos[b];
        const int seq = a.seq_ids ? a.seq_ids[b] : b;
        half_t* page = kv_page(kv, seq, pos);
        if (nb < a.q_cols + a.k_cols) {
            const bool is_q = nb < a.q_cols;
            const int nreg = is_q ? nb : nb - a.q_cols;
            const int blkg = nreg >> 5;
            const int head = blkg / kv.kblk, blk = blkg % kv.kblk;
            if (head >= kv.heads) return;
            const float2* cs = a.rope + (int64_t)pos * (a.hd >> 1) + blk * 16 + 4 * g;
            h8 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float2 cc = cs[i];
                const float x1 = acc[0][i], x2 = acc[NT - 1][i];
                o[i] = (half_t)(x1 * cc.x - x2 * cc.y);
                o[4 + i] = (half_t)(x2 * cc.x + x1 * cc.y);
            }
            if (is_q) *(h8*)(a.qbuf + ((((int64_t)b * kv.heads + head) * kv.kblk + blk) * 4 + g) * 8) = o;
            else *(h8*)(page + kfrag_off(kv, head, (pos % kv.page_tokens) >> 4, blk) + (g * 16 + (pos & 15)) * 8) = o;
        } else {
            const int nreg = nb - a.q_cols - a.k_cols;
            const int tp = pos & 31;
            const int gt = (tp & 15) >> 2, jt = (tp & 3) + ((tp >> 4) << 2);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int idx = (nreg >> 4) + t;
                const int head = idx / kv.vd16, d16 = idx % kv.vd16;
                if (head >= kv.heads) continue;
                half_t* fr = page + vfrag_off(kv, head, d16, (pos % kv.page_tokens) >> 5);
#pragma unroll
                for (int i = 0; i < 4; ++i) fr[(gt * 16 + 4 * g + i) * 8 + jt] = (half_t)acc[t][i];
            }
        }
    }
}

template <int NT, int MODE, int NW>
static hipError_t launch_skinny_t(const SkinnyArgs& a, hipStream_t s) {
    // chunk K so that B x (KC + 8) halves fit the LDS budget; chunk boundaries fall on whole U-tile batches
    constexpr int G = 32 * NW * 4;
    int kc = a.K;
    const int maxk = (SK_LDS_BUDGET / (2 * a.B) - 8) / G * G;
    if (kc > maxk) {
        const int nch = (a.K + maxk - 1) / maxk;
        kc = (((a.K + nch - 1) / nch) + G - 1) / G * G;
    }
    size_t lds = (size_t)a.B * (kc + 8) * 2;                // x image
    const size_t red = (size_t)NW * NT * 64 * 16;
    if (lds < red) lds = red;
    lds += 64;                                              // 16 row sums (fused RMSNorm), after both regions
    hipLaunchKernelGGL((skinny_kernel<NT, MODE, NW>), dim3(a.Npad / (16 * NT)), dim3(64 * NW), lds, s, a, kc);
    return hipGetLastError();
}

template <int NT, int MODE, int NW>
static hipError_t skinny_attr() {
    return hipFuncSetAttribute((const void*)skinny_kernel<NT, MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
}
hipError_t skinny_init() {
    hipError_t e;
    if ((e = skinny_attr<1, SK_ROW, 4>()) != hipSuccess) return e;
    if ((e = skinny_attr<1, SK_ROW, 8>()) != hipSuccess) return e;
    if ((e = skinny_attr<1, SK_LOGITS, 4>()) != hipSuccess) return e;
    if ((e = skinny_attr<2, SK_SILU_MUL, 4>()) != hipSuccess) return e;
    return skinny_attr<2, SK_QKV, 4>();
}

hipError_t launch_skinny(const SkinnyArgs& a, hipStream_t s) {
    if (a.B < 1 || a.B > 16 || (a.K & 127) || (a.Npad & 31)) return hipErrorInvalidValue;
    switch (a.mode) {
        case SK_ROW:
            // few output tiles (N = hidden): 8 waves per workgroup double the loads in flight per CU
            if (a.waves == 8 && (a.K & 255) == 0) return launch_skinny_t<1, SK_ROW, 8>(a, s);
            return launch_skinny_t<1, SK_ROW, 4>(a, s);
        case SK_LOGITS: return launch_skinny_t<1, SK_LOGITS, 4>(a, s);
        case SK_SILU_MUL: return launch_skinny_t<2, SK_SILU_MUL, 4>(a, s);
        case SK_QKV: return launch_skinny_t<2, SK_QKV, 4>(a, s);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------ decode attention
template <int KBLK, int VD16>
__global__ __launch_bounds__(64) void decode_attn_kernel(DecAttnArgs a) {
    const int lane = threadIdx.x;
    const int g = lane >> 4;
    const int sp = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
    const KvLayout& kv = a.kv;
    const int npos = a.pos[b] + 1;
    const int seq = a.seq_ids ? a.seq_ids[b] : b;
    const int npages = (npos + kv.page_tokens - 1) / kv.page_tokens;
    const int p_first = sp * a.pages_per_split;
    int p_last = p_first + a.pages_per_split;
    p_last = p_last < npages ? p_last : npages;
    const int64_t pidx = ((int64_t)b * a.heads + head) * a.nsplit + sp;

    h8 qf[KBLK];
#pragma unroll
    for (int blk = 0; blk < KBLK; ++blk)
        qf[blk] = *(const h8*)(a.qbuf + ((((int64_t)b * a.heads + head) * KBLK + blk) * 4 + g) * 8);

    f4 acc_o[VD16];
#pragma unroll
    for (int d = 0; d < VD16; ++d) acc_o[d] = f4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const float sc = a.scale * 1.4426950408889634f;

    for (int p = p_first; p < p_last; ++p) {
        const half_t* page = kv_page(kv, seq, p * kv.page_tokens);
        for (int kb = 0; kb < (kv.page_tokens >> 6); ++kb) {
            const int key0 = p * kv.page_tokens + kb * 64;
            if (key0 >= npos) break;
            h8 kf[4][KBLK];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int blk = 0; blk < KBLK; ++blk)
                    kf[kt][blk] = __builtin_nontemporal_load((const h8*)(page + kfrag_off(kv, head, kb * 4 + kt, blk) + lane * 8));
            h8 vf[VD16][2];
#pragma unroll
            for (int d = 0; d < VD16; ++d)
#pragma unroll
                for (int b32 = 0; b32 < 2; ++b32)
                    vf[d][b32] = __builtin_nontemporal_load((const h8*)(page + vfrag_off(kv, head, d, kb * 2 + b32) + lane * 8));
            f4 s[4];
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                s[kt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int blk = 0; blk < KBLK; ++blk) s[kt] = mfma16(kf[kt][blk], qf[blk], s[kt]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int key = key0 + kt * 16 + 4 * g + i;
                    const float v = key < npos ? s[kt][i] * sc : -INFINITY;
                    s[kt][i] = v;
                    mx = fmaxf(mx, v);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);     // key0 < npos => m_new finite
            m_run = m_new;
            h8 pf[2];
            float ps = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float pv = __builtin_amdgcn_exp2f(s[kt][i] - m_new);
                    ps += pv;
                    pf[kt >> 1][(kt & 1) * 4 + i] = (half_t)pv;
                }
            l_run = l_run * alpha + ps;
#pragma unroll
            for (int d = 0; d < VD16; ++d) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc_o[d][i] *= alpha;
                acc_o[d] = mfma16(vf[d][0], pf[0], acc_o[d]);
                acc_o[d] = mfma16(vf[d][1], pf[1], acc_o[d]);
            }
        }
    }
    float l = l_run;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if ((lane & 15) == 0) {
#pragma unroll
        for (int d = 0; d < VD16; ++d)
            *(f4*)(a.part_o + pidx * a.hd + d * 16 + 4 * g) = acc_o[d];
    }
    if (lane == 0) {
        a.part_ml[pidx * 2 + 0] = m_run;
        a.part_ml[pidx * 2 + 1] = l;
    }
}

// Variant 1: software-pipelined over 64-token pages.  While page p is being reduced (QK^T -> softmax -> PV),
// its V fragments and page p+1's K fragments are already in flight (32 KiB per wave), so a wave never
// idles for a full HBM round trip between its MFMA bursts.  Requires page_tokens == 64.
template <int KBLK, int VD16>
__global__ __launch_bounds__(64) void decode_attn_pipe_kernel(DecAttnArgs a) {
    const int lane = threadIdx.x;
    const int g = lane >> 4;
    const int sp = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
    const KvLayout& kv = a.kv;
    const int npos = a.pos[b] + 1;
    const int seq = a.seq_ids ? a.seq_ids[b] : b;
    const int npages = (npos + 63) >> 6;
    const int p_first = sp * a.pages_per_split;
    int p_last = p_first + a.pages_per_split;
    p_last = p_last < npages ? p_last : npages;
    const int64_t pidx = ((int64_t)b * a.heads + head) * a.nsplit + sp;

    h8 qf[KBLK];
#pragma unroll
    for (int blk = 0; blk < KBLK; ++blk)
        qf[blk] = *(const h8*)(a.qbuf + ((((int64_t)b * a.heads + head) * KBLK + blk) * 4 + g) * 8);
    f4 acc_o[VD16];
#pragma unroll
    for (int d = 0; d < VD16; ++d) acc_o[d] = f4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const float sc = a.scale * 1.4426950408889634f;
    const int64_t koff = kfrag_off(kv, head, 0, 0) + lane * 8;        // K frags of a (page, head): 4*KBLK contiguous KiB
    const int64_t voff = vfrag_off(kv, head, 0, 0) + lane * 8;        // V frags: VD16*2 contiguous KiB

    h8 kf[4 * KBLK], kn[4 * KBLK];
    if (p_first < p_last) {
        const half_t* page = kv_page(kv, seq, p_first * 64);
#pragma unroll
        for (int i = 0; i < 4 * KBLK; ++i) kf[i] = __builtin_nontemporal_load((const h8*)(page + koff + i * AUR_FRAG_HALVES));
    }
    for (int p = p_first; p < p_last; ++p) {
        const half_t* page = kv_page(kv, seq, p * 64);
        h8 vf[VD16 * 2];
#pragma unroll
        for (int i = 0; i < VD16 * 2; ++i) vf[i] = __builtin_nontemporal_load((const h8*)(page + voff + i * AUR_FRAG_HALVES));
        const bool more = p + 1 < p_last;                 // wave-uniform
        if (more) {
            const half_t* pn = kv_page(kv, seq, (p + 1) * 64);
#pragma unroll
            for (int i = 0; i < 4 * KBLK; ++i) kn[i] = __builtin_nontemporal_load((const h8*)(pn + koff + i * AUR_FRAG_HALVES));
        }
        const int key0 = p * 64;
        f4 s[4];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int blk = 0; blk < KBLK; ++blk) s[kt] = mfma16(kf[kt * KBLK + blk], qf[blk], s[kt]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = key0 + kt * 16 + 4 * g + i;
                const float v = key < npos ? s[kt][i] * sc : -INFINITY;
                s[kt][i] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        h8 pf[2];
        float ps = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float pv = __builtin_amdgcn_exp2f(s[kt][i] - m_new);
                ps += pv;
                pf[kt >> 1][(kt & 1) * 4 + i] = (half_t)pv;
            }
        l_run = l_run * alpha + ps;
#pragma unroll
        for (int d = 0; d < VD16; ++d) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc_o[d][i] *= alpha;
            acc_o[d] = mfma16(vf[d * 2 + 0], pf[0], acc_o[d]);
            acc_o[d] = mfma16(vf[d * 2 + 1], pf[1], acc_o[d]);
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < 4 * KBLK; ++i) kf[i] = kn[i];
        }
    }
    float l = l_run;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if ((lane & 15) == 0) {
#pragma unroll
        for (int d = 0; d < VD16; ++d) *(f4*)(a.part_o + pidx * a.hd + d * 16 + 4 * g) = acc_o[d];
    }
    if (lane == 0) {
        a.part_ml[pidx * 2 + 0] = m_run;
        a.part_ml[pidx * 2 + 1] = l;
    }
}

__global__ void decode_attn_combine_kernel(DecAttnArgs a) {
    const int head = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    if (d >= a.hd) return;
    const int64_t p0 = ((int64_t)b * a.heads + head) * a.nsplit;
    float M = -INFINITY;
    for (int s = 0; s < a.nsplit; ++s) M = fmaxf(M, a.part_ml[(p0 + s) * 2]);
    float num = 0.f, den = 0.f;
    for (int s = 0; s < a.nsplit; ++s) {
        const float m = a.part_ml[(p0 + s) * 2];
        if (m == -INFINITY) continue;
        const float wgt = __builtin_amdgcn_exp2f(m - M);
        num += wgt * a.part_o[(p0 + s) * a.hd + d];
        den += wgt * a.part_ml[(p0 + s) * 2 + 1];
    }
    a.out[(int64_t)b * a.ldo + head * a.hd + d] = (half_t)(num / den);
}

hipError_t launch_decode_attention(const DecAttnArgs& a, hipStream_t s) {
    if (a.kv.page_tokens & 63) return hipErrorInvalidValue;
    dim3 grid(a.nsplit, a.heads, a.B);
    const bool pipe = a.variant == 1 && a.kv.page_tokens == 64;
    if (pipe && a.kv.kblk == 4 && a.kv.vd16 == 8) hipLaunchKernelGGL((decode_attn_pipe_kernel<4, 8>), grid, dim3(64), 0, s, a);
    else if (pipe && a.kv.kblk == 2 && a.kv.vd16 == 4) hipLaunchKernelGGL((decode_attn_pipe_kernel<2, 4>), grid, dim3(64), 0, s, a);
    else if (pipe && a.kv.kblk == 1 && a.kv.vd16 == 2) hipLaunchKernelGGL((decode_attn_pipe_kernel<1, 2>), grid, dim3(64), 0, s, a);
    else if (a.kv.kblk == 4 && a.kv.vd16 == 8) hipLaunchKernelGGL((decode_attn_kernel<4, 8>), grid, dim3(64), 0, s, a);
    else if (a.kv.kblk == 2 && a.kv.vd16 == 4) hipLaunchKernelGGL((decode_attn_kernel<2, 4>), grid, dim3(64), 0, s, a);
    else if (a.kv.kblk == 1 && a.kv.vd16 == 2) hipLaunchKernelGGL((decode_attn_kernel<1, 2>), grid, dim3(64), 0, s, a);
    else return hipErrorInvalidValue;
    hipLaunchKernelGGL(decode_attn_combine_kernel, dim3(a.heads, a.B), dim3(a.hd <= 64 ? 64 : 128), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------ argmax + advance
// Greedy step bookkeeping: token = first argmax of the fp32 logits; append to out_ids unless the sequence
// already finished; EOS marks it finished; x_next[b] = embed[token]; pos[b] += advance_pos.
__global__ __launch_bounds__(256) void argmax_advance_kernel(const float* __restrict__ logits, int vocab,
                                                             const half_t* __restrict__ embed, int d, int eos_id, int max_new,
                                                             int32_t* __restrict__ out_ids, int32_t* __restrict__ out_len,
                                                             int32_t* __restrict__ finished, int32_t* __restrict__ pos,
                                                             half_t* __restrict__ x_next, int ldx, int advance_pos, int set_pos) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* lg = logits + (int64_t)b * vocab;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < vocab; i += 256) {
        const float v = lg[i];
        if (v > best || (v == best && i < bi)) {
            best = v;
            bi = i;
        }
    }
    sv[tid] = best;
    si[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            const float v2 = sv[tid + o];
            const int i2 = si[tid + o];
            if (v2 > sv[tid] || (v2 == sv[tid] && i2 < si[tid])) {
                sv[tid] = v2;
                si[tid] = i2;
            }
        }
        __syncthreads();
    }
    int tok = si[0];
    if (tok == 0x7fffffff) tok = 0;
    if (tid == 0) {
        if (!finished[b]) {
            const int n = out_len[b];
            if (n < max_new) {
                out_ids[(int64_t)b * max_new + n] = tok;
                out_len[b] = n + 1;
            }
            if (tok == eos_id) finished[b] = 1;
        }
        if (set_pos >= 0) pos[b] = set_pos;
        if (advance_pos) pos[b] += 1;
    }
    for (int cidx = tid; cidx < (d >> 3); cidx += 256)
        *(h8*)(x_next + (int64_t)b * ldx + cidx * 8) = *(const h8*)(embed + (int64_t)tok * d + cidx * 8);
}

hipError_t launch_argmax_advance(const float* logits, int B, int vocab, const half_t* embed, int d, int eos_id,
                                 int max_new, int32_t* out_ids, int32_t* out_len, int32_t* finished, int32_t* pos,
                                 half_t* x_next, int ldx, int advance_pos, int set_pos, hipStream_t s) {
    hipLaunchKernelGGL(argmax_advance_kernel, dim3(B), dim3(256), 0, s, logits, vocab, embed, d, eos_id, max_new, out_ids,
                       out_len, finished, pos, x_next, ldx, advance_pos, set_pos);
    return hipGetLastError();
}
