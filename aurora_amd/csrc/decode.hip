// Autoregressive decode kernels for gfx950 (HBM-bound): skinny weight-streaming GEMM, paged decode
// attention over fragment-packed KV, argmax + next-token bookkeeping.
//
// Reference: third-party transformers LlamaForCausalLM.generate greedy loop as called at
// inference.py:89-96 (one position per step with a KV cache, argmax, EOS / max_new_tokens stop).
//
// skinny GEMM   y[b, n] = sum_k x[b, k] W[n, k]   for b <= 64 rows (1-4 MFMA column groups of 16):
//   * Weights are the same FRAG tiles the prefill GEMM uses ([N/16][K/32][64][8]); a wave streams one
//     1 KiB fragment per global_load_dwordx4 (non-temporal) straight into VGPRs - no LDS round trip for a
//     once-read operand.
//   * Activations travel BETWEEN decode kernels in "x-fragment" form, [ceil(B/16)][K/32][64 lanes][8] with
//     lane = g*16 + (b % 16): exactly the MFMA B operand.  A wave reads the fragment it needs with one 16-byte
//     load per lane (L2-resident: B x K halves <= 700 KiB), so there is no per-workgroup LDS staging, no K
//     chunking and no barrier inside the stream loop; producers (RMSNorm, attention combine, the SiLU*up
//     epilogue) write this layout directly.
//   * One 16x16x32 MFMA per KiB of weights (x 2 column groups when B > 16) keeps the VALU idle: the loop is
//     purely HBM-bound.  A workgroup owns NT n16 tiles over the full K; its NW waves split K interleaved
//     (wave w takes k32 = w + NW*i, so the waves walk consecutive KiB), reduced through LDS in fixed order
//     (deterministic, batch-invariant).
//
//   * No RMSNorm kernels in the decode step: the norm WEIGHTS are folded into the consuming projection at pack time
//     (W' = W diag(w_norm)), the residual stream lives in x-fragment form, the residual-producing epilogues (o / down
//     projection, next-token embedding) accumulate sum(x^2) per row as 2^-28 fixed point with 64-bit integer atomics
//     (integer addition commutes: bitwise deterministic and batch-invariant), and the consumer scales its accumulators
//     by rstd[b] = rsqrt(sum/K + eps).
//
// decode attention: one wave per (sequence, head, split of 64-token pages).  K fragments (rows = tokens)
//   and V^T fragments (rows = d) stream from the paged cache as contiguous 1 KiB pieces; dot products on the VALU (v_dot2c_f32_f16),
//   one query per (sequence, head): see decode_attn_dot_kernel.
#include "kernels.h"

// address (halves) of the 16-byte piece holding x[b][k .. k+8) (k % 8 == 0) in x-fragment form
__device__ __forceinline__ int64_t xfrag_piece(int b, int k, int K32) {
    return ((((int64_t)(b >> 4) * K32 + (k >> 5)) * 64) + ((k >> 3) & 3) * 16 + (b & 15)) * 8;
}

// Lane whose 16-byte piece of an x fragment this lane loads for column group nb.  A lane whose batch row lies outside the live range
// [b_lo, b_hi) reads the piece of a LIVE row of its group instead of its own: the same address as that lane, so the wave's request
// touches only the live rows' 64-byte sectors - at one live row 4 x 16 bytes of a 1 KiB fragment instead of all of it.  (Measured at one
// slot, round 5: the o and down projections read as many x bytes as weight bytes, 128 KiB resp. 344 KiB of mostly-zero rows per
// workgroup through the same vector-memory path as the HBM stream.)  The MFMA columns of a dead row then hold a copy of a live row's
// products; columns are independent and skinny_store never stores a dead one, so every live output keeps its bits whatever the batch.
__device__ __forceinline__ int xsrc_lane(const SkinnyArgs& a, int lane, int nb) {
    const int c = lane & 15;
    int lo = a.b_lo - nb * 16, hi = a.b_hi - nb * 16;            // live columns of this group: [lo, hi) clipped to [0, 16)
    lo = lo < 0 ? 0 : lo;
    hi = hi > 16 ? 16 : hi;
    const int cc = lo >= hi ? 0 : (c < lo ? lo : (c >= hi ? hi - 1 : c));
    return (lane & ~15) | cc;
}

#define SSQ_SCALE 268435456.0f      /* 2^28 */
// sum(x^2) accumulators are striped over AUR_SSQ_SLOTS copies, [slot][AUR_MAX_BATCH]: the residual-producing epilogues add into
// slot (n16 tile % SLOTS), so the 256 integer atomics that hit one row's accumulator per projection spread over 16 L2 lines
// instead of queueing on one (they cost the o / down projections ~5 us each at 64 rows: 18.4 us in the engine vs 13.7 us with a
// plain store epilogue, tools/gemv_lab); the consumer adds the slots - integers: any order gives the same bits.
__device__ __forceinline__ float ssq_to_rstd(const unsigned long long* ssq, int b, int k, float eps) {
    unsigned long long s = 0ull;
#pragma unroll
    for (int sl = 0; sl < AUR_SSQ_SLOTS; ++sl) s += ssq[sl * AUR_MAX_BATCH + b];
    return rsqrtf((float)((double)s * (1.0 / 268435456.0)) / (float)k + eps);
}
__device__ __forceinline__ void ssq_set(unsigned long long* ssq, int b, unsigned long long v) {        // one writer per row
    ssq[b] = v;
#pragma unroll
    for (int sl = 1; sl < AUR_SSQ_SLOTS; ++sl) ssq[sl * AUR_MAX_BATCH + b] = 0ull;
}
__device__ __forceinline__ void ssq_clear(unsigned long long* ssq, int tid) {                          // tid < AUR_MAX_BATCH
#pragma unroll
    for (int sl = 0; sl < AUR_SSQ_SLOTS; ++sl) ssq[sl * AUR_MAX_BATCH + tid] = 0ull;
}

// Epilogue shared by every skinny-GEMM structure: the lane holds acc[t][nb][i] = y[n = (tile0 + t) * 16 + 4g + i][b = nb * 16 + c]
// (MFMA 16x16x32 C/D map), already reduced over K.  SK_QKV: a Q / K workgroup-unit is one PAIRED 32-column block (NT == 2:
// tiles 2p and 2p + 1 hold d and its RoPE partner d + hd/2); a V unit is NT consecutive n16 tiles.
// Per-lane inputs of the epilogue that do not depend on the projection itself (this lane's batch rows b = nb * 16 + c): loaded
// BEFORE the weight stream starts, so that their L2 round trips (16 accumulator stripes per row, then position -> page table)
// hide under it instead of serialising behind the last MFMA (measured at B = 64: 2-3 us per launch on the QKV / gate-up kernels).
template <int NB>
struct SkPre {
    float rstd[NB];
    int pos[NB];
    half_t* page[NB];
};
template <int MODE, int NB, bool WITH_RSTD = true>
__device__ __forceinline__ void skinny_prefetch(const SkinnyArgs& a, const int lane, SkPre<NB>& pre, const int tile0 = 0, const int nb_base = 0) {
    const int c = lane & 15;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int b = (nb_base + nb) * 16 + c;
        const bool live = b >= a.b_lo && b < a.b_hi;
        pre.rstd[nb] = (WITH_RSTD && a.ssq_in && live) ? ssq_to_rstd(a.ssq_in, b, a.K, a.norm_eps) : 1.f;
        pre.pos[nb] = 0;
        pre.page[nb] = nullptr;
        if (MODE == SK_QKV && live) {
            pre.pos[nb] = a.pos[b];
            pre.page[nb] = kv_page(a.kv, a.seq_ids ? a.seq_ids[b] : b, pre.pos[nb]);
        }
    }
}

template <int NT, int MODE, int NB>
__device__ __forceinline__ void skinny_store(const SkinnyArgs& a, const int tile0, f4 (&acc)[NT][NB], const int lane, const SkPre<NB>& pre,
                                             const int nb_base = 0) {
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int b = (nb_base + nb) * 16 + c;
        if (b < a.b_lo || b >= a.b_hi) continue;
        if (a.ssq_in) {                                   // folded RMSNorm: per-row 1/rms of the (un-normalised) input
            const float rstd = pre.rstd[nb];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][nb][i] *= rstd;
        }
        if (MODE == SK_ROW || MODE == SK_LOGITS || MODE == SK_SILU_MUL) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int tile = tile0 + t;
                const int n = tile * 16 + 4 * g;
                if (n >= a.n_real) continue;
                if (MODE == SK_LOGITS) {
                    float* dst = a.out32 + (int64_t)b * a.n_real + n;
                    if ((a.n_real & 3) == 0) *(f4*)dst = acc[t][nb];      // n % 4 == 0: one 16-byte store
                    else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) dst[i] = acc[t][nb][i];
                    }
                } else if (MODE == SK_SILU_MUL) {
                    // h[b][k], k = n/2 + {0,1} = tile*8 + 2g + {0,1}, written in x-fragment form for the down projection
                    h2 o;
                    o[0] = (half_t)(silu_f(acc[t][nb][0]) * acc[t][nb][1]);
                    o[1] = (half_t)(silu_f(acc[t][nb][2]) * acc[t][nb][3]);
                    half_t* dst = a.out_f + ((((int64_t)(b >> 4) * a.out_k32 + (tile >> 2)) * 64) + (tile & 3) * 16 + (b & 15)) * 8 + 2 * g;
                    *(h2*)dst = o;
                } else {
                    // residual update in place, in x-fragment form: x[b][n..n+3] += y; and sum(x_new^2) for the next norm
                    half_t* px = a.xres + xfrag_piece(b, n & ~7, a.n_real >> 5) + (n & 7);
                    const h4 rr = *(const h4*)px;
                    h4 o;
                    float p = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        o[i] = (half_t)(acc[t][nb][i] + (float)rr[i]);
                        p += (float)o[i] * (float)o[i];
                    }
                    *(h4*)px = o;
                    p = xor16_sum(p);
                    p = xor32_sum(p);
                    if (g == 0 && a.ssq_out)
                        atomicAdd(a.ssq_out + (tile & (AUR_SSQ_SLOTS - 1)) * AUR_MAX_BATCH + b, (unsigned long long)__float2ll_rn(p * SSQ_SCALE));
                }
            }
        } else {                  // SK_QKV: one PAIRED 32-column block (Q / K, NT == 2) or NT n16 tiles of V
            const KvLayout& kv = a.kv;
            const int nbc = tile0 * 16;
            const int pos = pre.pos[nb];
            half_t* page = pre.page[nb];
            if (nbc < a.q_cols + a.k_cols) {
                const bool is_q = nbc < a.q_cols;
                const int nreg = is_q ? nbc : nbc - a.q_cols;
                const int blkg = nreg >> 5;
                const int head = blkg / kv.kblk, blk = blkg % kv.kblk;
                if (head >= kv.heads) continue;
                // (cos, sin) stay in the epilogue: prefetching them is a THIRD dependent round trip (pos -> table) queued ahead of the
                // weight stream's in-order vmcnt - measured +10 us on this kernel
                const float2* cs = a.rope + (int64_t)pos * (a.hd >> 1) + blk * 16 + 4 * g;
                h8 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float2 cc = cs[i];
                    const float x1 = acc[0][nb][i], x2 = acc[NT - 1][nb][i];
                    o[i] = (half_t)(x1 * cc.x - x2 * cc.y);
                    o[4 + i] = (half_t)(x2 * cc.x + x1 * cc.y);
                }
                if (is_q) *(h8*)(a.qbuf + ((((int64_t)b * kv.heads + head) * kv.kblk + blk) * 4 + g) * 8) = o;
                else *(h8*)(page + kfrag_off(kv, head, (pos % kv.page_tokens) >> 4, blk) + (g * 16 + (pos & 15)) * 8) = o;
            } else {
                const int nreg = nbc - a.q_cols - a.k_cols;
                const int tp = pos & 31;
                const int gt = (tp & 15) >> 2, jt = (tp & 3) + ((tp >> 4) << 2);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int idx = (nreg >> 4) + t;
                    const int head = idx / kv.vd16, d16 = idx % kv.vd16;
                    if (head >= kv.heads) continue;
                    half_t* fr = page + vfrag_off(kv, head, d16, (pos % kv.page_tokens) >> 5);
#pragma unroll
                    for (int i = 0; i < 4; ++i) fr[(gt * 16 + 4 * g + i) * 8 + jt] = (half_t)acc[t][nb][i];
                }
            }
        }
    }
}

template <int NT, int MODE, int NW, int NB>
__global__ __launch_bounds__(64 * NW) void skinny_kernel(SkinnyArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.ssq_zero && blockIdx.x == 0 && tid < AUR_MAX_BATCH) ssq_clear(a.ssq_zero, tid);  // reset the accumulator a LATER kernel fills
    const int K32 = a.K >> 5;
    const int tile0 = blockIdx.x * NT;
    SkPre<NB> pre;
    if (w == 0) skinny_prefetch<MODE, NB>(a, lane, pre, tile0);                            // wave 0 runs the epilogue

    f4 acc[NT][NB];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = f4{0.f, 0.f, 0.f, 0.f};

    // ONE continuous software pipeline over this wave's k32 tiles: U tiles of W (and x) fragments in registers,
    // the next U in flight.
    constexpr int U = 4;                         // (round 5: 8 is slower at 1-2 column groups: o 7.8 -> 8.0, down 17.3 -> 17.7, gate/up 31.7 -> 35.2 us at one slot)
    const int nit = a.K / (32 * NW);
    const half_t* wp[NT];
    const half_t* xp[NB];
#pragma unroll
    for (int t = 0; t < NT; ++t) wp[t] = a.W + ((int64_t)(tile0 + t) * K32 + w) * AUR_FRAG_HALVES + lane * 8;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) xp[nb] = a.xf + ((int64_t)nb * K32 + w) * AUR_FRAG_HALVES + xsrc_lane(a, lane, nb) * 8;
    constexpr int64_t STEP = (int64_t)NW * AUR_FRAG_HALVES;
    const int nfull = nit / U * U;
    int i0 = 0;
    if constexpr (NB > 2) {
        // 3-4 column groups: x fragments (L2 hits) outnumber the weight fragments, so they get their own shallow
        // pipeline (a ring of U register buffers, U-1 k32 tiles ahead) and the HBM stream keeps its full depth of U tiles.
        h8 cw[U][NT], nw[U][NT], xr[U][NB];
        if (nfull > 0) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int t = 0; t < NT; ++t) cw[u][t] = __builtin_nontemporal_load((const h8*)(wp[t] + u * STEP));
#pragma unroll
            for (int u = 0; u < U - 1; ++u)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) xr[u][nb] = *(const h8*)(xp[nb] + u * STEP);
        }
        for (; i0 < nfull; i0 += U) {
            const bool more = (i0 + 2 * U <= nfull);
            if (more) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int t = 0; t < NT; ++t) nw[u][t] = __builtin_nontemporal_load((const h8*)(wp[t] + (i0 + U + u) * STEP));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (u + U - 1 < U || more) {                  // tile i0 + u + U-1 lies in a full batch: buffer (u + U-1) % U
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) xr[(u + U - 1) % U][nb] = *(const h8*)(xp[nb] + (i0 + u + U - 1) * STEP);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[t][nb] = mfma16(cw[u][t], xr[u][nb], acc[t][nb]);
            }
            if (more) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int t = 0; t < NT; ++t) cw[u][t] = nw[u][t];
            }
        }
    } else {
    h8 cw[U][NT], cx[U][NB], nw[U][NT], nx[U][NB];
    if (nfull > 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int t = 0; t < NT; ++t) cw[u][t] = __builtin_nontemporal_load((const h8*)(wp[t] + u * STEP));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) cx[u][nb] = *(const h8*)(xp[nb] + u * STEP);
        }
    }
    for (; i0 < nfull; i0 += U) {
        const bool more = (i0 + 2 * U <= nfull);            // wave-uniform: one branch per batch
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int t = 0; t < NT; ++t) nw[u][t] = __builtin_nontemporal_load((const h8*)(wp[t] + (i0 + U + u) * STEP));
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) nx[u][nb] = *(const h8*)(xp[nb] + (i0 + U + u) * STEP);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[t][nb] = mfma16(cw[u][t], cx[u][nb], acc[t][nb]);
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int t = 0; t < NT; ++t) cw[u][t] = nw[u][t];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) cx[u][nb] = nx[u][nb];
            }
        }
    }
    }
    for (; i0 < nit; ++i0) {                     // tail (< U tiles)
        h8 xf[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) xf[nb] = *(const h8*)(xp[nb] + i0 * STEP);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const h8 wf = __builtin_nontemporal_load((const h8*)(wp[t] + i0 * STEP));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[t][nb] = mfma16(wf, xf[nb], acc[t][nb]);
        }
    }
    // cross-wave reduction in fixed order
    float* red = (float*)smem;       // [NW waves][NT*NB][64 lanes][4]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) *(f4*)(red + (((w * NT + t) * NB + nb) * 64 + lane) * 4) = acc[t][nb];
    __syncthreads();
    if (w != 0) return;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f4 s = *(const f4*)(red + (((0 * NT + t) * NB + nb) * 64 + lane) * 4);
#pragma unroll
            for (int ww = 1; ww < NW; ++ww) {
                const f4 p = *(const f4*)(red + (((ww * NT + t) * NB + nb) * 64 + lane) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i] += p[i];
            }
            acc[t][nb] = s;
        }

    skinny_store<NT, MODE, NB>(a, tile0, acc, lane, pre);
}

// ------------------------------------------------------------------------------------ skinny GEMM, x through LDS
// Structure for engines that decode MORE than 32 sequences at once (3-4 MFMA column groups).  There the x fragments a wave needs
// outweigh its weight fragments (4 KiB of x per KiB of W for a one-tile wave), and every byte of it crosses the same per-CU vector
// memory path as the HBM stream: measured at B = 64 on MI355X the per-wave-x kernel above spends 40-50 % of its time on x
// (gate/up 49 us vs 32 us with the x loads stubbed out; tools/gemv_lab).  Here ONE workgroup per CU owns T consecutive n16 tiles
// (T x KS consumer waves: wave -> tile w / KS, k phase w % KS) and NL LOADER waves bring the x fragments in once per workgroup by
// LDS-DMA (global_load_lds, x-fragment form is already lane-linear), chunk by chunk (KC k32 tiles x NB column groups) into a ring
// of NBUF LDS buffers behind a COUNTED vmcnt; one s_barrier per chunk hands chunk ci to the consumers and chunk ci-1's buffer back
// to the loaders.  Consumers stream their weight fragments straight into VGPRs (U-deep register pipeline that runs across chunk
// boundaries; their vmcnt never sees a DMA) and read x with conflict-free ds_read_b128.  x traffic per CU drops from
// (tiles per CU) x |x| to |x|; barriers are rare (KC = 8: 16 per K = 4096).
//
// Accumulation order per output is fixed by (KS, KC, S) - shape constants - never by the batch: bitwise deterministic and
// batch-invariant like the kernel above.
//
// SK_ROW (N = hidden: one tile per CU) cannot share x across tiles without splitting K across workgroups: blockIdx.y = k split s
// of S, each workgroup writes its fp32 partial tile lane-linearly, and skinny_row_reduce_kernel sums the S partials in fixed order
// and applies the residual / sum(x^2) epilogue.
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#define SKX_BAR()                              \
    do {                                       \
        asm volatile("" ::: "memory");         \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
        asm volatile("" ::: "memory");         \
    } while (0)

// tiles of workgroup c: SK_QKV: unit c = PAIRED block c (tiles 2c, 2c + 1) + V tile v0 + c; otherwise tiles_lo (+1 for c < n_hi) consecutive
struct SkxGeom {
    int tiles_lo, n_hi;     // generic split of N16 tiles over gridDim.x workgroups
    int v_tile0;            // SK_QKV: first V tile (= (q_cols + k_cols) / 16)
    int S;                  // k splits (gridDim.y); > 1 only for SK_ROW
    float* part;            // SK_ROW split-K partials, lane-linear [S][N16][NB][64][4]
    int* cnt;               // SK_ROW: arrival counter per tile (fused reduce), nullptr = skinny_row_reduce_kernel follows
};

// 16-byte agent-scope accesses of the split-K hand-over (a tile's S workgroups may sit on different XCDs, whose L2s are not coherent for
// plain accesses): write-through `sc1` stores, L1-bypassing `sc1` loads - the guide's "sc1 payload -> drained vmcnt -> relaxed agent
// fetch_add, sc1 loads on the reducer" form of the in-launch split-K reduction.  Both go through the raw-buffer BUILTINS (aux 16 = sc1),
// i.e. they are instructions hipcc knows: it counts them in vmcnt and pads their hazards.
//
// Round 3 wrote the partials with inline asm (`global_store_dwordx4 .. sc1`) and was caught producing different captions in one serving
// run of three.  Root cause (profiles/r04_fused_reduce_rootcause.txt): hipcc treats an asm statement as one opaque
// instruction and does not pad its hazards, and gfx950 requires wait states between a VMEM store of more than 64 bits and a VALU write of
// the store's DATA registers.  The NB >= 4 instantiations came out as
//     global_store_dwordx4 v[18:19], v[10:13], off sc1 ; s_mov_b64 s[0:1], 0x400 ; v_lshl_add_u64 v[10:11], v[18:19], 0, s[0:1]
// - the next store's address computed INTO the previous store's data registers one state later.  Alone the store unit has read its data by
// then; with the memory pipeline backed up beside a front end it sometimes has not, and a partial goes out with two dwords of an
// address in it.  The protocol (counter, sc1, drain) was sound (the asm form and its soak script are in the history up to commit 282d3cc; profiles/r04_fused_reduce_soak.log).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "the split-K hand-over relies on gfx950 sc1 semantics"
#endif
typedef unsigned u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t part_rsrc(float* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000);       // raw buffer: byte offsets, bounds-checked against `bytes`
}
__device__ __forceinline__ void st_agent16(__amdgpu_buffer_rsrc_t r, unsigned byte_off, const f4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), r, byte_off, 0, 16);
}
__device__ __forceinline__ f4 ld_agent16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}

// TPW (tiles per consumer wave; 2 only for the split-K residual projections on the half grid): a wave accumulates TWO consecutive tiles
// against every x fragment it reads from LDS, so a workgroup owns 2 * T_MAX tiles per ring pass - on a stream that owns half of the CUs the
// residual projections then run one round of 128 workgroups instead of two of 256, and x crosses a CU's LDS once instead of twice.  Every
// tile keeps its k phases and its k order: bitwise the TPW == 1 result.
template <int T_MAX, int KS, int MODE, int NB, int KC, int NBUF, int U, int NL, int TPW = 1>
__global__ __launch_bounds__(64 * (T_MAX * KS + NL)) void skinny_lds_kernel(SkinnyArgs a, SkxGeom gm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];         // NBUF x KC x NB KiB ring, then 1/rms of every batch row
    constexpr int NCW = T_MAX * KS, Q = KC / KS;                        // consumer waves; steps per wave per chunk
    constexpr int PIECES = KC * NB / NL;                                // DMA instructions per chunk per loader wave
    constexpr int CHUNK_BYTES = KC * NB * 1024;
    static_assert(KC % KS == 0 && (Q % U == 0 || U % Q == 0), "chunk steps per wave vs pipeline depth");
    static_assert((KC * NB) % NL == 0, "pieces per chunk must split evenly over the loader waves");
    static_assert((NBUF - 1) * PIECES <= 63, "vmcnt is a 6-bit counter");
    static_assert(NBUF >= 2 && NBUF <= 8, "ring depth");
    static_assert(MODE != SK_QKV || T_MAX % 3 == 0, "SK_QKV: a workgroup owns T_MAX / 3 units of one PAIRED block + one V tile");
    static_assert(TPW == 1 || (TPW == 2 && MODE == SK_ROW && KS > 1), "two tiles per wave: split-K residual projections only");
    float* rstd_lds = (float*)(smem + NBUF * CHUNK_BYTES);              // [AUR_MAX_BATCH], written by loader wave 0 before the first barrier
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = blockIdx.x, s = blockIdx.y;
    if (a.ssq_zero && c == 0 && s == 0 && tid < AUR_MAX_BATCH) ssq_clear(a.ssq_zero, tid); // reset the accumulator a LATER kernel fills
    const int K32 = a.K >> 5, KE = K32 / gm.S, kb = s * KE;
    const int nchunk = (KE + KC - 1) / KC;

    if (w >= NCW) {                                            // ---------------- loader waves (piece q of a chunk -> loader q % NL)
        // few instructions, all on the critical path of every consumer wave.  Measured per mode at 128 rows (tools/gpu/run_ab.sh): the
        // split-K residual projections gain on a stream that owns half of the CUs (down 45.0 -> 43.0 us, o 27.0 -> 25.9 us; the same
        // on all CUs), QKV loses (60.7 -> 64.6 us), gate/up does not care
        if constexpr (MODE == SK_ROW) __builtin_amdgcn_s_setprio(3);
        const int lw = w - NCW;
        const int xl = NB == 1 ? xsrc_lane(a, lane, 0) : lane;   // one column group: dead rows re-read a live row's piece (see xsrc_lane)
        auto issue = [&](int ci) {
            char* buf = smem + (ci % NBUF) * CHUNK_BYTES;
#pragma unroll
            for (int q = 0; q < PIECES; ++q) {
                const int piece = q * NL + lw, kk = piece / NB, nb = piece % NB;
                int kr = ci * KC + kk;
                kr = kr < KE ? kr : KE - 1;                    // always PIECES instructions per chunk: the vmcnt below counts them
                glds16(a.xf + ((int64_t)nb * K32 + kb + kr) * AUR_FRAG_HALVES + xl * 8, buf + piece * 1024);
            }
        };
        int next = 0;
        for (; next < NBUF - 1 && next < nchunk; ++next) issue(next);
        // folded RMSNorm: 1/rms of every live batch row ONCE per workgroup (16 accumulator stripes per row), while the first chunks
        // fly; the first chunk barrier publishes it to the epilogue waves.  (Per epilogue wave it was 16 x NB loads ahead of the
        // weight stream - what made a spread epilogue lose on the gate/up kernel.)
        if (lw == 0) {
            for (int b = lane; b < AUR_MAX_BATCH; b += 64)
                rstd_lds[b] = (a.ssq_in && b >= a.b_lo && b < a.b_hi) ? ssq_to_rstd(a.ssq_in, b, a.K, a.norm_eps) : 1.f;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        for (int ci = 0; ci < nchunk; ++ci) {
            switch (next - ci - 1) {                           // chunks that may stay in flight once chunk ci has landed (<= NBUF - 2)
                case 0: wait_vmcnt<0>(); break;
                case 1: wait_vmcnt<PIECES>(); break;
                case 2: wait_vmcnt<(NBUF > 3 ? 2 : 0) * PIECES>(); break;
                case 3: wait_vmcnt<(NBUF > 4 ? 3 : 0) * PIECES>(); break;
                case 4: wait_vmcnt<(NBUF > 5 ? 4 : 0) * PIECES>(); break;
                case 5: wait_vmcnt<(NBUF > 6 ? 5 : 0) * PIECES>(); break;
                case 6: wait_vmcnt<(NBUF > 7 ? 6 : 0) * PIECES>(); break;
                default: wait_vmcnt<0>(); break;
            }
            SKX_BAR();                                         // chunk ci is in LDS; chunk ci-1's buffer is free again
            if (next < nchunk) {
                issue(next);
                ++next;
            }
        }
        return;
    }
    const int t = w / KS, p = w % KS;
    constexpr int UN = MODE == SK_QKV ? T_MAX / 3 : 1;         // SK_QKV units per workgroup: unit u = c + u * gridDim.x
    int ntile, tile;
    if (MODE == SK_QKV) {
        ntile = T_MAX;
        tile = t < 2 * UN ? 2 * (c + (t >> 1) * (int)gridDim.x) + (t & 1) : gm.v_tile0 + c + (t - 2 * UN) * (int)gridDim.x;
    } else {
        ntile = gm.tiles_lo + (c < gm.n_hi ? 1 : 0);
        tile = c * gm.tiles_lo + (c < gm.n_hi ? c : gm.n_hi) + TPW * t;
    }
    if (TPW * t >= ntile) return;                              // idle consumer (ended waves do not count at s_barrier)
    // Epilogue waves.  SK_QKV spreads the RoPE / page-table / KV-store tail: the 2 * KS waves of a PAIRED block share its column
    // groups (NBPP each), the KS waves of a V tile theirs (NBPV each) - with one epilogue wave per tile 8 column groups serialised
    // behind the last MFMA (128 rows: 52.7 -> 36 us).  The other modes keep ONE epilogue wave per tile (k phase 0).
    constexpr int NBPP = MODE == SK_QKV ? (NB + 2 * KS - 1) / (2 * KS) : NB;      // PAIRED block: groups per wave
    constexpr int NBPV = MODE == SK_QKV ? (NB + KS - 1) / KS : NB;                // V tile / every other mode
    const bool split = MODE == SK_ROW && gm.S > 1;
    const bool pair_tile = MODE == SK_QKV && t < 2 * UN;
    const int eg0 = MODE != SK_QKV ? 0 : pair_tile ? ((t & 1) * KS + p) * NBPP : p * NBPV;       // first column group of this wave's epilogue
    const bool epi_wave = !split && (MODE == SK_QKV ? eg0 < NB : p == 0);
    SkPre<NBPV> pre;
    if (epi_wave) skinny_prefetch<MODE, NBPV, false>(a, lane, pre, tile, eg0);   // positions / page pointers (1/rms comes from LDS)
    f4 acc[TPW][NB];
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[tp][nb] = f4{0.f, 0.f, 0.f, 0.f};
    const half_t* wbase = a.W + ((int64_t)tile * K32 + kb) * AUR_FRAG_HALVES + lane * 8;
    const int64_t wtile = (int64_t)K32 * AUR_FRAG_HALVES;       // tile + 1 is K32 fragments further on
    const int NS = nchunk * Q;                                 // steps of this wave (the last chunk may hold invalid steps)
    // step j -> chunk j / Q, kk = (j % Q) * KS + p; k index clamped: an invalid step re-reads the last fragment against x = 0
    auto kof = [&](int j, bool& valid) {
        const int kr = (j / Q) * KC + (j % Q) * KS + p;
        valid = j < NS && kr < KE;
        return valid ? kr : KE - 1;
    };
    h8 cw[U][TPW], nw[U][TPW];
    bool vv;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp) cw[u][tp] = __builtin_nontemporal_load((const h8*)(wbase + tp * wtile + (int64_t)kof(u, vv) * AUR_FRAG_HALVES));
    for (int j0 = 0; j0 < NS; j0 += U) {
        const bool more = j0 + U < NS;
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int tp = 0; tp < TPW; ++tp) nw[u][tp] = __builtin_nontemporal_load((const h8*)(wbase + tp * wtile + (int64_t)kof(j0 + U + u, vv) * AUR_FRAG_HALVES));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u;
            if (j < NS && j % Q == 0) SKX_BAR();               // chunk j / Q landed (and chunk j / Q - 1 is released)
            const int ci = j / Q, kk = (j % Q) * KS + p;
            const char* buf = smem + (ci % NBUF) * CHUNK_BYTES + lane * 16;
            bool valid;
            (void)kof(j, valid);
            h8 xr[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) xr[nb] = *(const h8*)(buf + (kk * NB + nb) * 1024);
            if (!valid) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) xr[nb] = h8{0, 0, 0, 0, 0, 0, 0, 0};
            }
#pragma unroll
            for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[tp][nb] = mfma16(cw[u][tp], xr[nb], acc[tp][nb]);
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int tp = 0; tp < TPW; ++tp) cw[u][tp] = nw[u][tp];
        }
    }
    const int N16 = a.Npad >> 4;
    auto load_rstd = [&](auto& pr, int g0) {                   // 1/rms of this lane's batch rows (published by loader wave 0)
        constexpr int cnt = sizeof(pr.rstd) / sizeof(float);
#pragma unroll
        for (int nb = 0; nb < cnt; ++nb) {
            const int b = (g0 + nb) * 16 + (lane & 15);
            pr.rstd[nb] = b < AUR_MAX_BATCH ? rstd_lds[b] : 1.f;
        }
    };
    if constexpr (TPW == 2) {
        // two tiles per wave: the k phases of tile `tile + tp` meet in LDS one tile at a time (the scratch of one round fills the ring), phase
        // order 0, 1, .. as ever; the split's partial goes to the reduce launch.  Every consumer wave passes every barrier.
        float* red = (float*)smem;
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp) {
            SKX_BAR();                                         // x is read (round 0) / the previous round's scratch is read (round 1)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) *(f4*)(red + ((w * NB + nb) * 64 + lane) * 4) = acc[tp][nb];
            SKX_BAR();
            if (p == 0 && TPW * t + tp < ntile) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    f4 sum = *(const f4*)(red + ((w * NB + nb) * 64 + lane) * 4);
#pragma unroll
                    for (int pp = 1; pp < KS; ++pp) {
                        const f4 q = *(const f4*)(red + (((w + pp) * NB + nb) * 64 + lane) * 4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) sum[i] += q[i];
                    }
                    *(f4*)(gm.part + ((((int64_t)s * (a.Npad >> 4) + tile + tp) * NB + nb) * 64 + lane) * 4) = sum;
                }
            }
        }
        return;
    }
    // ---- K phases of a tile (and, for SK_QKV, the two tiles of a PAIRED block) meet in LDS; fixed summation order (phase 0, 1, ...)
    if (KS > 1 || MODE == SK_QKV) {
        SKX_BAR();                                             // every consumer is done reading x (the loaders have exited)
        float* red = (float*)smem;                             // [consumer wave][NB][64][4]  (<= 12 x 4 KiB: inside the ring)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) *(f4*)(red + ((w * NB + nb) * 64 + lane) * 4) = acc[0][nb];
        SKX_BAR();
        if (!epi_wave && !split) return;
        if (split && gm.cnt == nullptr && p != 0) return;
        // column groups [g0, g0 + cnt) of the tile whose first wave is w0
        auto gather = [&](int w0, int g0, auto& dst) {
            constexpr int cnt = sizeof(dst) / sizeof(f4);
#pragma unroll
            for (int nb = 0; nb < cnt; ++nb) {
                const int nbg = g0 + nb;
                f4 sum = f4{0.f, 0.f, 0.f, 0.f};
                if (nbg < NB) {
                    sum = *(const f4*)(red + ((w0 * NB + nbg) * 64 + lane) * 4);
#pragma unroll
                    for (int pp = 1; pp < KS; ++pp) {
                        const f4 q = *(const f4*)(red + (((w0 + pp) * NB + nbg) * 64 + lane) * 4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) sum[i] += q[i];
                    }
                }
                dst[nb] = sum;
            }
        };
        if (pair_tile) {
            const int w0 = (t & ~1) * KS;                      // first wave of the PAIRED block's first tile
            f4 pr[2][NBPP];
            gather(w0, eg0, pr[0]);
            gather(w0 + KS, eg0, pr[1]);
            SkPre<NBPP> pp2;
#pragma unroll
            for (int nb = 0; nb < NBPP; ++nb) {
                pp2.pos[nb] = pre.pos[nb];
                pp2.page[nb] = pre.page[nb];
            }
            load_rstd(pp2, eg0);
            skinny_store<2, MODE, NBPP>(a, tile & ~1, pr, lane, pp2, eg0);
            return;
        }
        f4 one[1][NBPV];
        if constexpr (MODE == SK_ROW) {
            if (split && gm.cnt != nullptr) {
                // ---- split-K with the reduce IN the kernel: k split s publishes its partial tile (agent scope), counts itself in, and
                // the split that finds the tile complete sums the S = 4 partials in the fixed order s = 0, 1, 2, 3 and runs the residual
                // epilogue - exactly the arithmetic of skinny_row_reduce_kernel (bitwise), without its launch.  The tail is shared by the
                // tile's KS waves (column groups p, p + KS, ..).  A tile's 4 splits are workgroups c, c + gridDim.x, ..: the same XCD when
                // gridDim.x % 8 == 0, so in practice the partials are L2 hits; correctness does not rely on it.
                int* last_lds = (int*)(rstd_lds + AUR_MAX_BATCH);               // [T_MAX]
                const __amdgpu_buffer_rsrc_t prs = part_rsrc(gm.part, gm.S * N16 * NB * 1024);
                if (p == 0) {
                    gather(t * KS, 0, one[0]);
                    const unsigned off0 = (unsigned)((((s * N16 + tile) * NB) * 64 + lane) * 16);
#pragma unroll
                    for (int nb = 0; nb < NBPV; ++nb) st_agent16(prs, off0 + nb * 1024, one[0][nb]);
                    // the partial has reached the coherence point before the arrival is counted.  An asm wait on purpose: hipcc may
                    // drop its own vmcnt(0) ahead of an atomic when its scoreboard is empty (guide, Guideline 16 pitfall 12)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) {
                        const int arrived = __hip_atomic_fetch_add(gm.cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const int last = arrived == gm.S - 1;
                        if (last) __hip_atomic_store(gm.cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
                        last_lds[t] = last;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                SKX_BAR();
                if (!last_lds[t]) return;
                constexpr int GPW = (NB + KS - 1) / KS;                          // column groups per wave of the tail
                f4 q[GPW][4];
                const unsigned sstride = (unsigned)(N16 * NB * 1024);
#pragma unroll
                for (int i = 0; i < GPW; ++i) {
                    int nb = p + i * KS;
                    nb = nb < NB ? nb : NB - 1;
                    const unsigned off = (unsigned)(((tile * NB + nb) * 64 + lane) * 16);
#pragma unroll
                    for (int ss = 0; ss < 4; ++ss) q[i][ss] = ld_agent16(prs, off + ss * sstride);
                }
#pragma unroll
                for (int i = 0; i < GPW; ++i) {
                    const int nb = p + i * KS;
                    if (nb >= NB) continue;
                    SkPre<1> pre1;
                    skinny_prefetch<SK_ROW, 1>(a, lane, pre1, tile, nb);
                    f4 acc1[1][1];
                    acc1[0][0] = q[i][0];
#pragma unroll
                    for (int ss = 1; ss < 4; ++ss)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc1[0][0][e] += q[i][ss][e];
                    skinny_store<1, SK_ROW, 1>(a, tile, acc1, lane, pre1, nb);
                }
                return;
            }
        }
        gather(t * KS, eg0, one[0]);
        if (split) {                                           // split-K partial, lane-linear: one 16-byte store per lane and column group
#pragma unroll
            for (int nb = 0; nb < NBPV; ++nb) *(f4*)(gm.part + ((((int64_t)s * N16 + tile) * NB + nb) * 64 + lane) * 4) = one[0][nb];
            return;
        }
        load_rstd(pre, eg0);
        skinny_store<1, MODE, NBPV>(a, tile, one, lane, pre, eg0);
        return;
    }
    // KS == 1 (not SK_QKV): the wave holds the whole K sum of its tile and finishes it alone (NBPV == NB, eg0 == 0)
    if (split) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) *(f4*)(gm.part + ((((int64_t)s * N16 + tile) * NB + nb) * 64 + lane) * 4) = acc[0][nb];
        return;
    }
    f4 one[1][NBPV];
#pragma unroll
    for (int nb = 0; nb < NBPV; ++nb) one[0][nb] = acc[0][nb];
    load_rstd(pre, 0);
    skinny_store<1, MODE, NBPV>(a, tile, one, lane, pre);
}

// second stage of the split-K SK_ROW projection: y = sum_s part[s] (s ascending: fixed order), then the residual epilogue.
// One wave per (n16 tile, column group): N16 x NB waves (2048 at 128 rows) - the first version walked the NB groups inside a
// wave (64 workgroups in all: 9.4 us per launch at 128 rows, twice per layer).
__global__ __launch_bounds__(256) void skinny_row_reduce_kernel(SkinnyArgs a, SkxGeom gm, int NB) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int N16 = a.Npad >> 4;
    if (idx >= N16 * NB) return;
    const int tile = idx / NB, nb = idx - tile * NB;
    SkPre<1> pre;
    skinny_prefetch<SK_ROW, 1>(a, lane, pre, tile, nb);
    f4 acc[1][1];
    acc[0][0] = *(const f4*)(gm.part + (((int64_t)tile * NB + nb) * 64 + lane) * 4);
    for (int s = 1; s < gm.S; ++s) {
        const f4 q = *(const f4*)(gm.part + ((((int64_t)s * N16 + tile) * NB + nb) * 64 + lane) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[0][0][i] += q[i];
    }
    skinny_store<1, SK_ROW, 1>(a, tile, acc, lane, pre, nb);
}

static int g_skx_cus = 256;

template <int T_MAX, int KS, int MODE, int NB, int KC, int NBUF, int U, int NL, int TPW = 1>
static hipError_t launch_skx_t(const SkinnyArgs& a, const SkxGeom& gm, dim3 grid, hipStream_t s) {
    constexpr int ring = NBUF * KC * NB * 1024;
    constexpr int lds = ring + AUR_MAX_BATCH * 4 + 64;         // + 1/rms per batch row + the fused split-K reduce's "last split" flags
    static_assert(ring >= T_MAX * KS * NB * 1024, "the reduction scratch lives inside the x ring");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)skinny_lds_kernel<T_MAX, KS, MODE, NB, KC, NBUF, U, NL, TPW>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((skinny_lds_kernel<T_MAX, KS, MODE, NB, KC, NBUF, U, NL, TPW>), grid, dim3(64 * (T_MAX * KS + NL)), lds, s, a, gm);
    return hipGetLastError();
}

template <int NB>
static hipError_t launch_skx_nb(const SkinnyArgs& a, float* part, hipStream_t s) {
    // chunk depth by column groups: the x ring is NBUF x KC x NB KiB of LDS.  Up to 4 groups (64 rows) KC = 8 (fewest barriers:
    // 21.6 vs 25.4 us on the QKV shape against KC = 4); 5-8 groups (128 rows): 2 x 8 ("ring", default) or 4 x 4 k32 tiles - 128 KiB.
    constexpr int KC = NB <= 4 ? 8 : 4;
    constexpr int KCR = NB <= 4 ? 6 : 3, NBUFR = NB <= 4 ? 5 : 4, NLR = NB <= 4 ? 2 : 3;       // split-K residual projection
    const int N16 = a.Npad >> 4;
    SkxGeom gm{};
    gm.S = 1;
    gm.part = part;
    // half grid (a.half_grid, 5-8 column groups): the launch is meant for a stream that owns HALF of the CUs (the decode stream while a
    // front end runs on the other half): half as many workgroups with twice the tiles each, so that a CU streams x through its LDS
    // once instead of once per round of workgroups (QKV 62 -> 39 us, gate/up 77 -> 58 us on 16 CUs per XCD, tools/gemv_lab).  Same
    // k phases per tile, same summation order: bitwise the full-grid result.
    const bool half = a.half_grid && NB > 4;
    auto split = [&](int tmax, int cus) {                      // N16 tiles over min(N16, max(cus, ceil(N16 / tmax))) workgroups
        int G = (N16 + tmax - 1) / tmax;
        if (G < cus) G = N16 < cus ? N16 : cus;
        gm.tiles_lo = N16 / G;
        gm.n_hi = N16 % G;
        return G;
    };
    switch (a.mode) {
        case SK_QKV: {
            // unit c = PAIRED block c of the Q / K region + V tile c (equal counts by construction: q_cols == k_cols == V cols);
            // a workgroup owns unit c (full grid) or units c and c + pairs / 2 (half grid).  2 k phases per tile: 31.0 us against
            // 35.5 us with 4 at 128 rows, 22.7 vs 23.1 at 64 (tools/gemv_lab) - and 6 tiles x 2 phases still fit one workgroup.
            const int pairs = (a.q_cols + a.k_cols) >> 5;
            gm.v_tile0 = (a.q_cols + a.k_cols) >> 4;
            if (N16 - gm.v_tile0 < pairs) return hipErrorInvalidValue;
            if constexpr (NB > 4) {
                if (half && (pairs & 1) == 0) return launch_skx_t<6, 2, SK_QKV, NB, 8, 2, 2, 4>(a, gm, dim3(pairs / 2, 1), s);
                return launch_skx_t<3, 2, SK_QKV, NB, 8, 2, 2, 4>(a, gm, dim3(pairs, 1), s);      // 5-8 column groups: 2 x 8 k32 ring (4 x 4 was 52.4 vs 55.8 us)
            }
            // (round 5: U = 8 at one column group - twice the weight bytes in flight per wave - is SLOWER: 20.1 -> 22.5 us at one slot)
            // (also tried at one column group: 4 k phases per tile = 12 consumer waves instead of 6: 20.7 us against 20.1 - the QKV kernel at
            // few slots is not short of bytes in flight)
            return launch_skx_t<3, 2, SK_QKV, NB, KC, 4, 4, 4>(a, gm, dim3(pairs, 1), s);
        }
        case SK_SILU_MUL:
            // engines of more than 64 slots (a.gu_ks == 1): ONE k phase per tile, so that the half grid's 11 tiles per workgroup fit
            // (128 rows: full grid 51.0 us against 49.5 with 2 phases, half grid on 16 CUs per XCD 58 against 77; 64 rows 35.5 vs 33.4).
            // A constant of the engine, never of the live batch: results stay batch-invariant.
            if (a.gu_ks == 1) {
                if constexpr (NB > 4) {
                    if (half) return launch_skx_t<11, 1, SK_SILU_MUL, NB, 4, 4, 4, 4>(a, gm, dim3(split(11, g_skx_cus / 2), 1), s);
                    return launch_skx_t<6, 1, SK_SILU_MUL, NB, 8, 2, 8, 4>(a, gm, dim3(split(6, g_skx_cus), 1), s);
                }
                return launch_skx_t<6, 1, SK_SILU_MUL, NB, KC, 4, 8, 4>(a, gm, dim3(split(6, g_skx_cus), 1), s);
            }
            if constexpr (NB > 4) return launch_skx_t<6, 2, SK_SILU_MUL, NB, 8, 2, 4, 4>(a, gm, dim3(split(6, g_skx_cus), 1), s);
            return launch_skx_t<6, 2, SK_SILU_MUL, NB, KC, 4, 4, 4>(a, gm, dim3(split(6, g_skx_cus), 1), s);
        case SK_LOGITS: return launch_skx_t<8, 1, SK_LOGITS, NB, KC, 4, 4, 4>(a, gm, dim3(split(8, g_skx_cus), 1), s);
        case SK_ROW: {
            // split-K 4: 4 tiles x 3 k phases per workgroup, N16 / 4 n-blocks x 4 k splits (= 256 workgroups at N = 4096)
            gm.S = 4;
            gm.tiles_lo = 4;
            gm.n_hi = 0;
            // (round 3 re-swept this geometry in tools/gemv_lab, profiles/r03_gemv_lab_row.log: 2 k phases over chunks of 8 k32 is 1-2 us faster
            // there, o 18.9 -> 18.0 us, down 31.5 -> 29.4 us incl. the reduce, but neutral in the engine - 19.3 / 30.1 us either way - and
            // rocprofv3's counter mode crashed in this launch with that instantiation: kept as it was)
            gm.cnt = a.row_cnt;
            if constexpr (NB > 4) {
                // half grid (a stream that owns half of the CUs): 8 tiles per workgroup, two per wave - one round of N16 / 8 x 4 workgroups;
                // same k phases and k order per tile: bitwise the full grid's partials.  (The in-kernel reduce keeps the full grid.)
                if (half && gm.cnt == nullptr && (N16 & 7) == 0) {
                    gm.tiles_lo = 8;
                    hipError_t e2 = launch_skx_t<4, 3, SK_ROW, NB, KCR, NBUFR, 1, NLR, 2>(a, gm, dim3(N16 / 8, 4), s);      // U = 1: two tiles x (current + next) fragments = the full-grid form's bytes in flight, and 128 registers without a spill
                    if (e2 != hipSuccess) return e2;
                    hipLaunchKernelGGL(skinny_row_reduce_kernel, dim3((N16 * NB + 3) / 4), dim3(256), 0, s, a, gm, NB);
                    return hipGetLastError();
                }
            }
            hipError_t e = launch_skx_t<4, 3, SK_ROW, NB, KCR, NBUFR, 2, NLR>(a, gm, dim3(N16 / 4, 4), s);
            if (e != hipSuccess || gm.cnt != nullptr) return e;
            hipLaunchKernelGGL(skinny_row_reduce_kernel, dim3((N16 * NB + 3) / 4), dim3(256), 0, s, a, gm, NB);
            return hipGetLastError();
        }
    }
    return hipErrorInvalidValue;
}

// SK_ROW takes the split-K LDS structure only where it pays (K >= 8192: the down projection; at K = 4096 the extra kernel
// boundary of the reduce eats the gain: o projection 13.7 us either way, tools/gemv_lab)
bool skinny_lds_applies(const SkinnyArgs& a) {
    if (a.variant != 1) return false;
    if (a.mode == SK_ROW) return a.part != nullptr && (a.K & 127) == 0 && (a.Npad & 63) == 0;      // the engine passes part for K >= 8192
    return true;
}

template <int NT, int MODE, int NW, int NB>
static hipError_t launch_skinny_t(const SkinnyArgs& a, hipStream_t s) {
    const size_t lds = (size_t)NW * NT * NB * 64 * 16;
    hipLaunchKernelGGL((skinny_kernel<NT, MODE, NW, NB>), dim3(a.Npad / (16 * NT)), dim3(64 * NW), lds, s, a);
    return hipGetLastError();
}

hipError_t skinny_init() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
        g_skx_cus = cus;
    return hipSuccess;
}

template <int NB>
static hipError_t launch_skinny_nb(const SkinnyArgs& a, hipStream_t s) {
    switch (a.mode) {
        case SK_ROW:
            // few output tiles (N = hidden): 8 waves per workgroup double the loads in flight per CU
            // (round 5: 16 waves on the o projection at 1 and 8 slots: 7.67 us against 7.55 with 8)
            // (4 waves measured twice, rounds 2 and 4: slower on every shape; K % 256 != 0 keeps them)
            if ((a.K & 255) == 0) return launch_skinny_t<1, SK_ROW, 8, NB>(a, s);
            return launch_skinny_t<1, SK_ROW, 4, NB>(a, s);
        case SK_LOGITS: return launch_skinny_t<1, SK_LOGITS, 4, NB>(a, s);
        case SK_SILU_MUL: return launch_skinny_t<2, SK_SILU_MUL, 4, NB>(a, s);
        case SK_QKV: return launch_skinny_t<2, SK_QKV, 4, NB>(a, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_skinny(const SkinnyArgs& a, hipStream_t s) {
    if (a.B < 1 || a.B > AUR_MAX_BATCH || (a.K & 127) || (a.Npad & 31)) return hipErrorInvalidValue;
    const int nb = (a.B + 15) >> 4;                         // MFMA column groups of 16 batch rows
    if (skinny_lds_applies(a)) {
        switch (nb) {
            case 1: return launch_skx_nb<1>(a, a.part, s);
            case 2: return launch_skx_nb<2>(a, a.part, s);
            case 3: return launch_skx_nb<3>(a, a.part, s);
            case 4: return launch_skx_nb<4>(a, a.part, s);
            case 5: return launch_skx_nb<5>(a, a.part, s);
            case 6: return launch_skx_nb<6>(a, a.part, s);
            case 7: return launch_skx_nb<7>(a, a.part, s);
            default: return launch_skx_nb<8>(a, a.part, s);
        }
    }
    switch (nb) {                                           // x fragments per wave: engines of <= 32 slots (and K < 8192 residual projections up to 64 rows)
        case 1: return launch_skinny_nb<1>(a, s);
        case 2: return launch_skinny_nb<2>(a, s);
        case 3: return launch_skinny_nb<3>(a, s);
        case 4: return launch_skinny_nb<4>(a, s);
        default: return hipErrorInvalidValue;               // > 64 rows exist only on the LDS structure (the engine routes every projection there)
    }
}

// ------------------------------------------------------------------------------------ x-fragment producers
// xf[b0 + row] = (RMSNorm(x[row]) if w else x[row]) in x-fragment form.  One wave per row.
// One 1024-thread workgroup per group of 16 batch rows (rows b0 .. b0+rows-1 all lie in groups starting at b0 & ~15):
//   phase 1: wave r computes sum(x^2) of row r (one wave per row, fixed butterfly order)           -> rstd[r] in LDS
//   phase 2: wave w builds k32 tiles w, w+16, ...: lane (g, r) reads x[r][k32*32 + g*8 .. +8], normalises, and the
//            64 lanes store one complete 1 KiB fragment (fully coalesced; rows outside [b0, b0+rows) keep zeros).
template <int NWV>
__global__ __launch_bounds__(64 * NWV) void xfrag_norm_kernel(const half_t* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                          float eps, int rows, int d, int b0, half_t* __restrict__ xf,
                                                          unsigned long long* __restrict__ ssq_out) {
    __shared__ float rs[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int grp = (b0 >> 4) + blockIdx.x;                 // 16-row group handled by this workgroup
    const int K32 = d >> 5;
    for (int rr = wv; rr < 16; rr += NWV) {                 // phase 1: wave wv <-> rows wv, wv + NWV, ... of the group
        const int b = grp * 16 + rr;
        const int row = b - b0;
        float ss = 0.f;
        if (row >= 0 && row < rows && (w || ssq_out)) {
            const half_t* xr = x + row * ldx;
            const int nchunk = d >> 3;
            for (int c0 = 0; c0 < nchunk; c0 += 512) {          // batches of 8 independent 16-byte loads per lane
                h8 t[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int cch = c0 + lane + i * 64;
                    if (cch < nchunk) t[i] = *(const h8*)(xr + cch * 8);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int cch = c0 + lane + i * 64;
                    if (cch < nchunk) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) ss += (float)t[i][j] * (float)t[i][j];
                    }
                }
            }
        }
        ss = wave_sum(ss);
        if (lane == 0) {
            rs[rr] = w ? rsqrtf(ss / (float)d + eps) : 1.f;
            if (ssq_out && row >= 0 && row < rows) ssq_set(ssq_out, b, (unsigned long long)__float2ll_rn(ss * SSQ_SCALE));
        }
    }
    __syncthreads();
    const int r = lane & 15, g = lane >> 4;
    const int b = grp * 16 + r, row = b - b0;
    const bool live = row >= 0 && row < rows;
    const float rstd = rs[r];
    const half_t* xr = x + (live ? row : 0) * ldx + g * 8;
    half_t* dst = xf + ((int64_t)grp * K32 * 64 + lane) * 8;
    for (int k0 = 0; k0 < K32; k0 += 8 * NWV) {                // batches of 8 tiles per wave: loads first, then math + stores
        h8 t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k32 = k0 + wv + i * NWV;
            if (live && k32 < K32) t[i] = *(const h8*)(xr + k32 * 32);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k32 = k0 + wv + i * NWV;
            if (live && k32 < K32) {
                h8 o = t[i];
                if (w) {
                    const float* wk = w + k32 * 32 + g * 8;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (half_t)(wk[j] * (float)(half_t)((float)t[i][j] * rstd));
                }
                *(h8*)(dst + (int64_t)k32 * AUR_FRAG_HALVES) = o;
            }
        }
    }
}

hipError_t launch_xfrag_norm(const half_t* x, int64_t ldx, const float* w, float eps, int rows, int d, int b0, half_t* xf,
                             unsigned long long* ssq_out, hipStream_t s, int waves) {
    if ((d & 31) || rows < 1) return hipErrorInvalidValue;
    const int g0 = b0 >> 4, g1 = (b0 + rows - 1) >> 4;
    if (waves == 16) hipLaunchKernelGGL(xfrag_norm_kernel<16>, dim3(g1 - g0 + 1), dim3(1024), 0, s, x, ldx, w, eps, rows, d, b0, xf, ssq_out);
    else if (waves == 8) hipLaunchKernelGGL(xfrag_norm_kernel<8>, dim3(g1 - g0 + 1), dim3(512), 0, s, x, ldx, w, eps, rows, d, b0, xf, ssq_out);
    else hipLaunchKernelGGL(xfrag_norm_kernel<4>, dim3(g1 - g0 + 1), dim3(256), 0, s, x, ldx, w, eps, rows, d, b0, xf, ssq_out);
    return hipGetLastError();
}

// generic row-major [rows, d] -> x-fragment form for any d % 32 == 0 (test entry point / wide rows)
__global__ void xfrag_pack_kernel(const half_t* __restrict__ x, int64_t ldx, int rows, int d, half_t* __restrict__ xf) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nchunk = d >> 3;
    if (id >= (int64_t)rows * nchunk) return;
    const int b = id / nchunk, cch = id % nchunk;
    *(h8*)(xf + xfrag_piece(b, cch * 8, d >> 5)) = *(const h8*)(x + b * ldx + cch * 8);
}
hipError_t launch_xfrag_pack(const half_t* x, int64_t ldx, int rows, int d, half_t* xf, hipStream_t s) {
    if (d & 31) return hipErrorInvalidValue;
    const int64_t n = (int64_t)rows * (d >> 3);
    hipLaunchKernelGGL(xfrag_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, ldx, rows, d, xf);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------ decode attention
// One wave per (sequence, head, split of 64-token pages), on the VALU (v_dot2c_f32_f16).  One query per (sequence, head) makes the decode
// attention a GEMV: an MFMA form (rounds 1-3: q replicated over the 16 columns of a 16x16x32 tile; docs/rounds/r01-r04.md section 4)
// throws 15 / 16 of every product away - 32 MFMAs = 524 k MACs per 32 KiB page for 16 k useful ones, and 12-20 W of a socket that runs
// at its power cap.  A lane keeps the layout the fragments give it - lane (g, r) holds token kt * 16 + r x the 8 PAIRED dims of group g
// of a K fragment, feature d16 * 16 + r x the 8 PAIRED tokens of group g of a V^T fragment:
//   Q K^T : 4 v_dot2c per fragment into ONE fp32 per (lane, kt); the 4 dim groups of a token meet by two row exchanges (xor 16, 32)
//   softmax: 4 scores per lane
//   P     : 64 probabilities -> LDS as fp16 in token order (lane l writes token l), read back as the PAIRED token groups of the V^T
//           fragments (two 8-byte reads per 32 tokens): 128 B of LDS per wave, same wave writes and reads - no barrier
//   P V   : 4 v_dot2c per fragment into one fp32 per (lane, d16); the 4 token groups meet once, after the last page
// Software-pipelined over pages: K of page p + 1 is re-loaded into the registers the Q K^T products of page p have just read, V of page p
// is in flight with it (<= 256 registers: two waves per SIMD, up to 256 KiB of loads in flight per CU).  Deterministic, batch-invariant.
// One output feature d of (sequence b, head): the splits' partials combined in the fixed order s = 0, 1, .. (deterministic whatever order
// they were produced in).  (Rounds 4-5 also had the last-arriving split combine inside the attention kernel, behind an agent-scope
// hand-over: bitwise the same, 0.3 % slower at 8 slots and 7 % slower at one - removed.)
__device__ __forceinline__ float attn_part_ld(const float* p) { return *p; }
__device__ __forceinline__ void attn_combine_feature(const DecAttnArgs& a, int b, int head, int d) {
    const int64_t p0 = ((int64_t)b * a.heads + head) * a.nsplit;
    float M = -INFINITY;
    float num = 0.f, den = 0.f;
    if (a.nsplit <= 16) {
        // the engine's split counts (<= 16; a 32-wide version of this path cost 1.7 us per launch in registers and scalar spills): every partial is fetched before the first is used - 3 x nsplit independent loads, one round
        // trip - then the same operations in the same order as the loop below (rounds 1-4 ran that loop for every count: 2 x nsplit
        // DEPENDENT round trips, 6.4 us per launch at one slot where the attention itself takes 8.9 us)
        float m[16], l[16], o[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int sc = s < a.nsplit ? s : a.nsplit - 1;
            m[s] = attn_part_ld(a.part_ml + (p0 + sc) * 2);
            l[s] = attn_part_ld(a.part_ml + (p0 + sc) * 2 + 1);
            o[s] = attn_part_ld(a.part_o + (p0 + sc) * a.hd + d);
        }
#pragma unroll
        for (int s = 0; s < 16; ++s)
            if (s < a.nsplit) M = fmaxf(M, m[s]);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (s >= a.nsplit || m[s] == -INFINITY) continue;
            const float wgt = __builtin_amdgcn_exp2f(m[s] - M);
            num = __builtin_fmaf(wgt, o[s], num);
            den = __builtin_fmaf(wgt, l[s], den);
        }
    } else {
        for (int s = 0; s < a.nsplit; ++s) M = fmaxf(M, attn_part_ld(a.part_ml + (p0 + s) * 2));
        for (int s = 0; s < a.nsplit; ++s) {
            const float m = attn_part_ld(a.part_ml + (p0 + s) * 2);
            if (m == -INFINITY) continue;
            const float wgt = __builtin_amdgcn_exp2f(m - M);
            num = __builtin_fmaf(wgt, attn_part_ld(a.part_o + (p0 + s) * a.hd + d), num);
            den = __builtin_fmaf(wgt, attn_part_ld(a.part_ml + (p0 + s) * 2 + 1), den);
        }
    }
    const int k = head * a.hd + d;                 // attention output in x-fragment form (input of the o projection)
    a.out_f[xfrag_piece(b, k & ~7, a.out_k32) + (k & 7)] = (half_t)(num / den);
}
__device__ __forceinline__ float dot8(const h8 a, const h8 b, float c) {
    c = __builtin_amdgcn_fdot2(h2{a[0], a[1]}, h2{b[0], b[1]}, c, false);
    c = __builtin_amdgcn_fdot2(h2{a[2], a[3]}, h2{b[2], b[3]}, c, false);
    c = __builtin_amdgcn_fdot2(h2{a[4], a[5]}, h2{b[4], b[5]}, c, false);
    c = __builtin_amdgcn_fdot2(h2{a[6], a[7]}, h2{b[6], b[7]}, c, false);
    return c;
}
// S = splits of a (sequence, head) that live in ONE workgroup (one wave each).  S == 1: a wave per workgroup; the splits of a pair are
// separate workgroups and decode_attn_combine_kernel joins them.  S > 1 (engines whose batch x heads alone gives one workgroup per
// CU, 8-15 slots of 32 heads: round 5): the workgroup holds ALL splits of its pair and joins them through LDS behind one barrier -
// the same operations in the same order as the combine kernel (same bits), without its launch or the partials' round trip.
template <int KBLK, int VD16, int S = 1>
__global__ __launch_bounds__(64 * S, S == 1 ? 2 : 1) void decode_attn_dot_kernel(DecAttnArgs a) {
    __shared__ __attribute__((aligned(16))) half_t p16s[S][64];
    const int lane = threadIdx.x & 63;
    const int wv = S == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    half_t* p16 = p16s[wv];
    const int g = lane >> 4, r = lane & 15;
    const int sp = blockIdx.x * S + wv, head = blockIdx.y, b = blockIdx.z;
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
    float acc_o[VD16];                                   // feature d16 * 16 + r, summed over this lane's token group so far
#pragma unroll
    for (int d = 0; d < VD16; ++d) acc_o[d] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;                // l_run: this lane's tokens (r, 16 + r, 32 + r, 48 + r of every page)
    const float sc = a.scale * 1.4426950408889634f;
    const int64_t koff = kfrag_off(kv, head, 0, 0) + lane * 8;
    const int64_t voff = vfrag_off(kv, head, 0, 0) + lane * 8;

    // ---- the page pipeline: K of page p + 1 is requested into the registers the Q K^T products of page p have just read, V^T of page p at the
    // top of its step.  hipcc sizes the waits of the V products for the path on which the conditional K requests were NOT issued
    // (`s_waitcnt vmcnt(15 .. 0)` where `vmcnt(31 .. 16)` would do), i.e. every step drains K(p + 1) before its last V product.  Round 6 built
    // the exact-count forms - the last page as a peeled step, and on top of it V^T requested a whole step ahead into a second register set
    // (316 registers, one wave per SIMD) - and measured them against this loop in one process (tools/gpu/attn_ab.py at 128 x 32 x 2143,
    // profiles/r06_attn_pipeline_ab.log, r06_attn_pipeline_ab2.log): whole chip 656 -> 652 / 654 us, 16 CUs per XCD (the masked steps of the
    // serving loop: seven of eight) 750 -> 753 / 743 us.  Nothing: the kernel is bound by what a CU keeps in flight (~47 GB/s per CU on half of
    // the chip, 19.6 B per clock - a plain streaming kernel reaches the same 5.5-5.8 TB/s there, profiles/r03_mall_lab.log) and by HBM on
    // the whole chip (6.85 TB/s), not by its own waits.  The loop stays as it was.
    h8 kf[4 * KBLK];
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
        const int key0 = p * 64;
        float s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            float part = 0.f;
#pragma unroll
            for (int blk = 0; blk < KBLK; ++blk) part = dot8(kf[kt * KBLK + blk], qf[blk], part);
            s[kt] = part;
        }
        if (more) {                                       // K of the next page into the registers the products above have just read
            const half_t* pn = kv_page(kv, seq, (p + 1) * 64);
#pragma unroll
            for (int i = 0; i < 4 * KBLK; ++i) kf[i] = __builtin_nontemporal_load((const h8*)(pn + koff + i * AUR_FRAG_HALVES));
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            float v = s[kt];
            v = xor16_sum(v);                   // the 4 dim groups of token kt * 16 + r: every one of its 4 lanes gets the same bits
            v = xor32_sum(v);
            v = (key0 + kt * 16 + r) < npos ? v * sc : -INFINITY;
            s[kt] = v;
            mx = fmaxf(mx, v);
        }
        mx = row16_max(mx);                               // over the 16 token lanes (the 4 groups already agree); DPP, order-free
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float ps = 0.f, mine = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const float pv = __builtin_amdgcn_exp2f(s[kt] - m_new);
            ps += pv;
            mine = kt == g ? pv : mine;                   // lane l = g * 16 + r publishes token l of the page
        }
        l_run = l_run * alpha + ps;
        p16[lane] = (half_t)mine;
        h8 pf[2];
#pragma unroll
        for (int b32 = 0; b32 < 2; ++b32) {               // PAIRED token order of a V^T fragment: tokens 4g .. 4g + 3 and 16 + 4g .. of the 32
            const h4 lo = *(const h4*)(p16 + b32 * 32 + 4 * g);
            const h4 hi = *(const h4*)(p16 + b32 * 32 + 16 + 4 * g);
            pf[b32] = h8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int d = 0; d < VD16; ++d) {
            float o = acc_o[d] * alpha;
            o = dot8(vf[d * 2 + 0], pf[0], o);
            o = dot8(vf[d * 2 + 1], pf[1], o);
            acc_o[d] = o;
        }
    }
    float l = l_run;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) l += __shfl_xor(l, o, 64);       // over the token lanes; the 4 groups hold the same sum
#pragma unroll
    for (int d = 0; d < VD16; ++d) {                      // the 4 token groups of a feature
        acc_o[d] = xor16_sum(acc_o[d]);
        acc_o[d] = xor32_sum(acc_o[d]);
    }
    if constexpr (S > 1) {
        __shared__ float lo[S][VD16 * 16], lm[S], ll[S];
#pragma unroll
        for (int d = 0; d < VD16; ++d)
            if ((d & 3) == g) lo[wv][d * 16 + r] = acc_o[d];
        if (lane == 0) {
            lm[wv] = m_run;
            ll[wv] = l;
        }
        __syncthreads();
        for (int d = threadIdx.x; d < a.hd; d += 64 * S) {       // attn_combine_feature on the workgroup's own partials
            float M = -INFINITY;
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2) M = fmaxf(M, lm[s2]);
            float num = 0.f, den = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2) {
                if (lm[s2] == -INFINITY) continue;
                const float wgt = __builtin_amdgcn_exp2f(lm[s2] - M);
                num = __builtin_fmaf(wgt, lo[s2][d], num);
                den = __builtin_fmaf(wgt, ll[s2], den);
            }
            const int k = head * a.hd + d;
            a.out_f[xfrag_piece(b, k & ~7, a.out_k32) + (k & 7)] = (half_t)(num / den);
        }
        return;
    }
    if (a.nsplit == 1) {
#pragma unroll
        for (int d = 0; d < VD16; ++d) {
            if ((d & 3) != g) continue;                   // every lane stores its share: features d16 = g, g + 4
            const int k = head * a.hd + d * 16 + r;
            a.out_f[xfrag_piece(b, k & ~7, a.out_k32) + (k & 7)] = (half_t)(acc_o[d] / l);
        }
        return;
    }
    // decode_attn_combine_kernel follows
#pragma unroll
    for (int d = 0; d < VD16; ++d)
        if ((d & 3) == g) a.part_o[pidx * a.hd + d * 16 + r] = acc_o[d];
    if (lane == 0) {
        a.part_ml[pidx * 2 + 0] = m_run;
        a.part_ml[pidx * 2 + 1] = l;
    }
}

__global__ void decode_attn_combine_kernel(DecAttnArgs a) {
    const int head = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    if (d >= a.hd) return;
    attn_combine_feature(a, b, head, d);
}

hipError_t launch_decode_attention_main(const DecAttnArgs& a, hipStream_t s) {
    if (a.kv.page_tokens != 64) return hipErrorInvalidValue;        // the page pipeline is written for 64-token pages (aur_create enforces it)
    dim3 grid(a.nsplit, a.heads, a.B);
    if (a.local_splits > 1) {                           // all splits of a (sequence, head) in one workgroup: no combine launch
        if (a.local_splits != a.nsplit || a.kv.kblk != 4 || a.kv.vd16 != 8) return hipErrorInvalidValue;
        grid.x = 1;
        if (a.nsplit == 2) hipLaunchKernelGGL((decode_attn_dot_kernel<4, 8, 2>), grid, dim3(128), 0, s, a);
        else if (a.nsplit == 4) hipLaunchKernelGGL((decode_attn_dot_kernel<4, 8, 4>), grid, dim3(256), 0, s, a);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (a.kv.kblk == 4 && a.kv.vd16 == 8) hipLaunchKernelGGL((decode_attn_dot_kernel<4, 8>), grid, dim3(64), 0, s, a);
    else if (a.kv.kblk == 2 && a.kv.vd16 == 4) hipLaunchKernelGGL((decode_attn_dot_kernel<2, 4>), grid, dim3(64), 0, s, a);
    else if (a.kv.kblk == 1 && a.kv.vd16 == 2) hipLaunchKernelGGL((decode_attn_dot_kernel<1, 2>), grid, dim3(64), 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
bool decode_attention_needs_combine(const DecAttnArgs& a) { return a.nsplit > 1 && a.local_splits <= 1; }     // a single split normalises and stores its output itself
hipError_t launch_decode_attention_combine(const DecAttnArgs& a, hipStream_t s) {
    if (!decode_attention_needs_combine(a)) return hipSuccess;
    hipLaunchKernelGGL(decode_attn_combine_kernel, dim3(a.heads, a.B), dim3(a.hd <= 64 ? 64 : 128), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_decode_attention(const DecAttnArgs& a, hipStream_t s) {
    hipError_t e = launch_decode_attention_main(a, s);
    return e != hipSuccess ? e : launch_decode_attention_combine(a, s);
}

// ------------------------------------------------------------------------------------ argmax + advance
// Greedy step bookkeeping for slot b = b0 + blockIdx.x: token = first argmax of the fp32 logits; append to out_ids unless
// the sequence already finished; EOS marks it finished; the next input x[b] = embed[token] is written in x-fragment
// form together with sum(x^2) (2^-28 fixed point) for the first folded RMSNorm of the next step; pos[b] += advance_pos.
// 1024 threads scan the row with 16-byte loads (one 256-thread block took 125 dependent 4-byte loads per thread: 40 us per step whatever
// the batch - 1 % of an 8-slot step); the FIRST maximum wins whatever the scan order (value, then index).  The embedding copy and its
// sum(x^2) keep the 256-thread loop and tree of rounds 1-3: same additions in the same order, same bits.
#define ARGMAX_THREADS 1024
__device__ __forceinline__ void argmax_take(float& best, int& bi, float v, int i) {
    if (v > best || (v == best && i < bi)) {
        best = v;
        bi = i;
    }
}
__global__ __launch_bounds__(ARGMAX_THREADS) void argmax_advance_kernel(const float* __restrict__ logits, int b0, int vocab,
                                                             const half_t* __restrict__ embed, int d, int eos_id, int max_new,
                                                             int32_t* __restrict__ out_ids, int32_t* __restrict__ out_len,
                                                             int32_t* __restrict__ finished, int32_t* __restrict__ pos,
                                                             half_t* __restrict__ xf, unsigned long long* __restrict__ ssq,
                                                             int advance_pos, int set_pos) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const int b = b0 + blockIdx.x, tid = threadIdx.x;
    const float* lg = logits + (int64_t)b * vocab;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if ((vocab & 3) == 0) {                        // rows stay 16-byte aligned
        const int n4 = vocab >> 2;
#pragma unroll 4
        for (int i = tid; i < n4; i += ARGMAX_THREADS) {
            const f4 v = *(const f4*)(lg + 4 * i);
#pragma unroll
            for (int j = 0; j < 4; ++j) argmax_take(best, bi, v[j], 4 * i + j);
        }
    } else {
        for (int i = tid; i < vocab; i += ARGMAX_THREADS) argmax_take(best, bi, lg[i], i);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_take(best, bi, __shfl_xor(best, o), __shfl_xor(bi, o));
    if ((tid & 63) == 0) {
        sv[tid >> 6] = best;
        si[tid >> 6] = bi;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < ARGMAX_THREADS / 64; ++w) argmax_take(best, bi, sv[w], si[w]);
        si[0] = bi;
    }
    __syncthreads();
    int tok = si[0];
    if (tok == 0x7fffffff) tok = 0;
    __syncthreads();
    if (tid == 0) {
        const bool was_finished = finished[b] != 0;
        if (!was_finished) {
            const int n = out_len[b];
            if (n < max_new) {
                out_ids[(int64_t)b * max_new + n] = tok;
                out_len[b] = n + 1;
            }
            if (tok == eos_id || n + 1 >= max_new) finished[b] = 1;     // EOS, or the output row is full
        }
        if (set_pos >= 0) pos[b] = set_pos;
        // a finished (or idle) slot stops advancing: it keeps recomputing one position inside its own pages, so extra
        // decode steps (other slots still running, continuous batching) can never walk it past its KV allocation
        if (advance_pos && !was_finished) pos[b] += 1;
    }
    float ss = 0.f;
    if (tid < 256) {
        for (int cidx = tid; cidx < (d >> 3); cidx += 256) {
            const h8 v = *(const h8*)(embed + (int64_t)tok * d + cidx * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += (float)v[j] * (float)v[j];
            *(h8*)(xf + xfrag_piece(b, cidx * 8, d >> 5)) = v;
        }
        sv[tid] = ss;
    }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {            // fixed-order tree: deterministic
        if (tid < o) sv[tid] += sv[tid + o];
        __syncthreads();
    }
    if (tid == 0) ssq_set(ssq, b, (unsigned long long)__float2ll_rn(sv[0] * SSQ_SCALE));
}

hipError_t launch_argmax_advance(const float* logits, int b0, int nb, int vocab, const half_t* embed, int d, int eos_id,
                                 int max_new, int32_t* out_ids, int32_t* out_len, int32_t* finished, int32_t* pos,
                                 half_t* xf, unsigned long long* ssq, int advance_pos, int set_pos, hipStream_t s) {
    hipLaunchKernelGGL(argmax_advance_kernel, dim3(nb), dim3(ARGMAX_THREADS), 0, s, logits, b0, vocab, embed, d, eos_id, max_new, out_ids,
                       out_len, finished, pos, xf, ssq, advance_pos, set_pos);
    return hipGetLastError();
}
