// Flash attention over operand fragments (ViT encoder attention and Llama causal prefill), gfx950.
//
// Reference: AuroraAttention.forward softmax(q k^T) v (aurora.py:647-688; the `+ size.log()` term at
// :671-674 is a per-query-row constant, i.e. a softmax no-op - SURVEY fact 6 - and is not computed);
// HF Llama causal SDPA for the prefill.
//
// Everything arrives in MFMA operand form, written by the QKV GEMM epilogue (gemm.hip):
//   Q, K : [tok16][d-block][FRAG], rows = tokens, k-slots = d (PAIRED)  -> S^T = K Q^T
//   V^T  : [d16][tok32][FRAG],     rows = d,      k-slots = tokens (PAIRED) -> O^T = V^T P^T
// S^T accumulators (lane: 4 keys of one query) ARE the PAIRED-token B operand of the PV MFMA after a
// cvt to fp16: the softmax and P never leave the lane's registers, there is no transpose and no LDS
// traffic besides the lane-linear K/V fragment reads (ds_read_b128 at base + lane*16, conflict-free).
// K/V tiles are staged by LDS-DMA (global_load_lds_dwordx4), double-buffered, one barrier per 64 keys.
// A workgroup = 4 waves x 32 queries; each wave keeps 2 query tiles so every K/V fragment read feeds
// two MFMAs.  (Round 6 built the 4-tile form - 64 queries per wave, half the LDS reads per flop, 321 registers = one wave per SIMD at
// head_dim 128, 235 = two at head_dim 80 - as a template parameter and measured it against this one in one process, bitwise the same output
// (profiles/r06_attn_qtiles_ab.log): Llama prefill attention 4 x 2142 245 -> 352 us, 1 x 2142 82 -> 103 us, a ViT pass of 32 frames
// 26.03 -> 26.2 ms.  As hipcc schedules it, one wave per SIMD runs its QK MFMAs, its softmax and its PV MFMAs one after the other; with
// 2-3 waves per SIMD they overlap across waves.  The 4-tile form needs the hand-placed MFMA / VALU interleave of the guide's T15 / T19;
// not kept.)
#include "kernels.h"

template <int KBLK, int VD16>
__global__ __launch_bounds__(256) void attn_kernel(AttnArgs a) {
    constexpr int NF = 4 * KBLK + 2 * VD16;       // fragments (KiB) per 64-key stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    // XCD-aware block -> (query block, head, sequence) map (guide T1).  Workgroup b runs on XCD b % 8, and the K / V fragments of one
    // (sequence, head) - 1.1 MB for a 2142-token Llama prefix, 0.26 MB for a ViT frame - are re-read by every query block of that pair:
    // with the plain (qb, head, seq) grid the 17 query blocks of a pair were dealt over all 8 XCDs and each XCD's L2 fetched the pair's
    // K / V for itself (PMC round 3: 1.06 GB per 4-clip prefill launch against 0.28 GB of Q + K + V + O; 3.9 TB/s of fabric traffic -
    // and 885 us beside a decode stream against 449 us alone).  Now all query blocks of a pair run on ONE XCD, back to back (heavy
    // causal blocks first), so its K / V cross the fabric once and live in that XCD's L2 while they are needed (~2 pairs at a time).
    const int nqb = a.nqb, npair = a.heads * a.nseq;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int pair = xcd + 8 * (j / nqb);
    if (pair >= npair) return;                     // the grid is padded to 8 x ceil(pairs / 8) x nqb
    const int qb = nqb - 1 - (j % nqb);            // heavy (late, causal) blocks first
    const int head = pair % a.heads, seq = pair / a.heads;
    const KvLayout& kv = a.kv;
    const int T16 = a.rows_per_seq >> 4, T32 = a.rows_per_seq >> 5;
    const int q0 = qb * 128 + w * 32;

    // Q fragments -> registers
    h8 qf[2][KBLK];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        int tq = (q0 >> 4) + qt;
        tq = tq < T16 ? tq : T16 - 1;
#pragma unroll
        for (int b = 0; b < KBLK; ++b)
            qf[qt][b] = *(const h8*)(a.Qf + ((((int64_t)seq * a.heads + head) * T16 + tq) * KBLK + b) * AUR_FRAG_HALVES + lane * 8);
    }

    int nkb = (a.t + 63) >> 6;
    if (a.causal) {
        const int lastq = a.q_pos0 + qb * 128 + 127;
        const int lim = (lastq >> 6) + 1;
        nkb = nkb < lim ? nkb : lim;
    }

    // 64-token pages (the paged KV cache): key block kb IS page kb of the sequence, whose K fragments of this head are
    // 4*KBLK contiguous KiB and whose V^T fragments are 2*VD16 contiguous KiB -> ONE page-table lookup per block (fetched a
    // block ahead) and constant offsets, instead of an integer division + dependent table load per fragment.
    const bool p64 = kv.page_tokens == 64 && kv.page_table != nullptr;
    const int32_t* prow = p64 ? kv.page_table + (int64_t)(a.seq0 + seq) * kv.max_pages : nullptr;
    const int64_t koff = kfrag_off(kv, head, 0, 0) + lane * 8, voff = vfrag_off(kv, head, 0, 0) + lane * 8;
    int pid_next = p64 ? prow[0] : 0;
    auto stage = [&](int kb, int buf) {
        char* dst = smem + buf * (NF * 1024);
        if (p64) {
            const half_t* page = kv.base + (int64_t)pid_next * kv.page_halves;
            if (kb + 1 < nkb) pid_next = prow[kb + 1];
#pragma unroll
            for (int i = 0; i < NF / 4; ++i) {
                const int f = w + 4 * i;
                glds16(f < 4 * KBLK ? page + koff + f * AUR_FRAG_HALVES : page + voff + (f - 4 * KBLK) * AUR_FRAG_HALVES, dst + f * 1024);
            }
            for (int f = w + 4 * (NF / 4); f < NF; f += 4)
                glds16(f < 4 * KBLK ? page + koff + f * AUR_FRAG_HALVES : page + voff + (f - 4 * KBLK) * AUR_FRAG_HALVES, dst + f * 1024);
            return;
        }
        if (kv.page_table == nullptr && kv.page_tokens >= a.rows_per_seq) {     // one "page" per sequence (ViT frames): no division
            const half_t* page = kv.base + (int64_t)(a.seq0 + seq) * kv.page_halves;
            const int t16pp = kv.page_tokens >> 4, t32pp = kv.page_tokens >> 5;
            for (int f = w; f < NF; f += 4) {
                const half_t* src;
                if (f < 4 * KBLK) {
                    int t16 = kb * 4 + f / KBLK;
                    t16 = t16 < T16 ? t16 : T16 - 1;
                    src = page + (((int64_t)head * t16pp + t16) * KBLK + f % KBLK) * AUR_FRAG_HALVES;
                } else {
                    const int fv = f - 4 * KBLK;
                    int t32 = kb * 2 + (fv & 1);
                    t32 = t32 < T32 ? t32 : T32 - 1;
                    src = page + kv.v_off + (((int64_t)head * VD16 + (fv >> 1)) * t32pp + t32) * AUR_FRAG_HALVES;
                }
                glds16(src + lane * 8, dst + f * 1024);
            }
            return;
        }
        for (int f = w; f < NF; f += 4) {
            const half_t* src;
            if (f < 4 * KBLK) {
                int t16 = kb * 4 + f / KBLK;
                t16 = t16 < T16 ? t16 : T16 - 1;
                const int tok = t16 << 4;
                src = kv_page(kv, a.seq0 + seq, tok) + kfrag_off(kv, head, (tok % kv.page_tokens) >> 4, f % KBLK);
            } else {
                const int fv = f - 4 * KBLK;
                int t32 = kb * 2 + (fv & 1);
                t32 = t32 < T32 ? t32 : T32 - 1;
                const int tok = t32 << 5;
                src = kv_page(kv, a.seq0 + seq, tok) + vfrag_off(kv, head, fv >> 1, (tok % kv.page_tokens) >> 5);
            }
            glds16(src + lane * 8, dst + f * 1024);
        }
    };

    f4 acc_o[VD16][2];
#pragma unroll
    for (int d = 0; d < VD16; ++d)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) acc_o[d][qt] = f4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    const float sc = a.scale * 1.4426950408889634f;

    stage(0, 0);
    for (int kb = 0; kb < nkb; ++kb) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kb + 1 < nkb) stage(kb + 1, (kb + 1) & 1);
        const char* base = smem + (kb & 1) * (NF * 1024);

        f4 s[4][2];
        __builtin_amdgcn_s_setprio(1);                 // MFMA groups at raised priority: the SIMD's other wave issues its softmax in the gaps
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt][0] = f4{0.f, 0.f, 0.f, 0.f};
            s[kt][1] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < KBLK; ++b) {
                const h8 kf = *(const h8*)(base + (kt * KBLK + b) * 1024 + lane * 16);
                s[kt][0] = mfma16(kf, qf[0][b], s[kt][0]);
                s[kt][1] = mfma16(kf, qf[1][b], s[kt][1]);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        h8 pf[2][2];
        // wave-uniform: unless every key of this block is valid for every query of this wave, mask first (diagonal / last block)
        if (!((kb * 64 + 63 < a.t) && (!a.causal || kb * 64 + 63 <= a.q_pos0 + q0))) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int query = a.q_pos0 + q0 + qt * 16 + c;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int key = kb * 64 + kt * 16 + 4 * g + i;
                        const bool ok = key < a.t && (!a.causal || key <= query);
                        s[kt][qt][i] = ok ? s[kt][qt][i] : -INFINITY;
                    }
            }
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float mx = -INFINITY;                          // running maxima are kept on the RAW scores (sc > 0)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int i = 0; i < 4; ++i) mx = fmaxf(mx, s[kt][qt][i]);
            mx = xor16_max(mx);
            mx = xor32_max(mx);
            const float m_new = fmaxf(m_run[qt], mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            // the softmax is VALU-bound (27-31 % MFMA busy, PMC): after the first few key blocks a query's running maximum rarely
            // moves, and then alpha == 1 exactly - the rescale of l and of the VD16 x 4 output accumulators is skipped for the whole
            // wave (wave-uniform branch; multiplying by 1.0f is the identity, so the result is bitwise the same)
            const bool moved = m_new != m_run[qt];
            const float ms = m_use * sc;
            float ps = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][qt][i], sc, -ms));     // exp2(-inf) = 0 for masked keys
                    ps += p;
                    pf[qt][kt >> 1][(kt & 1) * 4 + i] = (half_t)p;
                }
            if (__builtin_amdgcn_ballot_w64(moved) != 0ull) {
                const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_use) * sc);
                m_run[qt] = m_new;
                l_run[qt] = l_run[qt] * alpha + ps;
#pragma unroll
                for (int d = 0; d < VD16; ++d)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc_o[d][qt][i] *= alpha;
            } else {
                l_run[qt] = l_run[qt] + ps;
            }
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int d = 0; d < VD16; ++d)
#pragma unroll
            for (int b32 = 0; b32 < 2; ++b32) {
                const h8 vf = *(const h8*)(base + (4 * KBLK + d * 2 + b32) * 1024 + lane * 16);
                acc_o[d][0] = mfma16(vf, pf[0][b32], acc_o[d][0]);
                acc_o[d][1] = mfma16(vf, pf[1][b32], acc_o[d][1]);
            }
        __builtin_amdgcn_s_setprio(0);
    }

#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float l = l_run[qt];
        l = xor16_sum(l);
        l = xor32_sum(l);
        const float inv = 1.0f / l;
        const int query = q0 + qt * 16 + c;
        if (query < a.rows_per_seq) {
            half_t* orow = a.O + ((int64_t)seq * a.rows_per_seq + query) * a.ldo + head * a.hd + 4 * g;
#pragma unroll
            for (int d = 0; d < VD16; ++d) {
                h4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (half_t)(acc_o[d][qt][i] * inv);
                *(h4*)(orow + d * 16) = o;
            }
        }
    }
}

template <int KBLK, int VD16>
static hipError_t launch_attn_t(const AttnArgs& a, hipStream_t s) {
    constexpr int lds = 2 * (4 * KBLK + 2 * VD16) * 1024;
    AttnArgs b = a;
    b.nqb = (a.rows_per_seq + 127) / 128;
    const int npair = a.heads * a.nseq;
    dim3 grid(8 * ((npair + 7) / 8) * b.nqb);
    hipLaunchKernelGGL((attn_kernel<KBLK, VD16>), grid, dim3(256), lds, s, b);
    return hipGetLastError();
}

template <int KBLK, int VD16>
static hipError_t attn_init_t() {
    return hipFuncSetAttribute((const void*)attn_kernel<KBLK, VD16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               2 * (4 * KBLK + 2 * VD16) * 1024);
}
hipError_t attn_init() {
    hipError_t e;
    if ((e = attn_init_t<3, 5>()) != hipSuccess) return e;
    if ((e = attn_init_t<4, 8>()) != hipSuccess) return e;
    if ((e = attn_init_t<1, 1>()) != hipSuccess) return e;
    if ((e = attn_init_t<1, 2>()) != hipSuccess) return e;
    return attn_init_t<2, 4>();
}

hipError_t launch_attention(const AttnArgs& a, hipStream_t s) {
    if (a.kv.kblk == 3 && a.kv.vd16 == 5) return launch_attn_t<3, 5>(a, s);    // ViT-H: head_dim 80 (padded 96)
    if (a.kv.kblk == 4 && a.kv.vd16 == 8) return launch_attn_t<4, 8>(a, s);    // Llama-7B: head_dim 128
    if (a.kv.kblk == 1 && a.kv.vd16 == 1) return launch_attn_t<1, 1>(a, s);    // test configs: head_dim 16
    if (a.kv.kblk == 1 && a.kv.vd16 == 2) return launch_attn_t<1, 2>(a, s);    // test configs: head_dim 32
    if (a.kv.kblk == 2 && a.kv.vd16 == 4) return launch_attn_t<2, 4>(a, s);    // test configs: head_dim 64
    return hipErrorInvalidValue;
}
