// Token merging (ToMe) step for gfx950 - bipartite soft matching + size-weighted merge.
//
// Reference: src/xtuner/xtuner/model/tome.py:36-81 (bipartite_soft_matching, class_token=True),
// tome.py:207-219 (merge_wavg), called per ViT layer at aurora.py:746-747.
//
// BIT-EXACT CONTRACT: on identical metric bytes every index (node_idx, src, unm, dst) equals the CPU
// oracle (oracle/tome_ref.c).  This is achieved by using one fully specified fp32 operation order on
// both sides - this file is compiled with -ffp-contract=off and spells every fused op as fmaf():
//   n2 = fmaf-chain_k m*m;  mhat = m / sqrtf(n2)  (IEEE sqrt, div);  S_ij = fmaf-chain_k A_ik*B_jk;
//   first-max over j;  rank by (node_max desc, i asc);  merge: acc = x_B*s_B, += x_A*s_A in rank
//   order, / s_tot.
// Each score chain is evaluated by ONE thread in k order (no split-k, no tree), so parallelism never
// changes rounding.  Cross-thread reductions compare (value, index) pairs lexicographically, which is
// order-independent.
//
// Four short launches per step, no cross-workgroup hand-over inside any of them (rounds 2-4 ranked a frame in the LAST workgroup of
// the match launch to arrive, behind relaxed agent-scope atomics; round 5 gives the ranking its own launch, so every dependency of
// the step is a kernel boundary):
//   1. prep   : metric = mean over heads of the layer's K fragments (or the caller's metric), the k-ordered norm chain by ONE thread
//               per row, IEEE division, written as fp32 MFMA operand blocks ("mfrag": 16 token rows x 16 k per 1 KiB group,
//               element (lane, q) = mhat[row lane & 15][k = 16 g + 4 q + (lane >> 4)]) - the A (even) and B (odd) tokens of a
//               frame in separate block lists, zero-padded;
//   2. match  : scores on the fp32 matrix cores.  v_mfma_f32_16x16x4_f32 accumulates its four k in order, one rounding per
//               product - bitwise the fmaf chain of the contract (MI355X_MICROARCH.md, "exact f32 (== fmaf chain, bitwise)") -
//               so a workgroup owns 16 A rows (the MFMA's N side, stationary in 20 VGPRs at c = 80) and its four waves
//               stream the B blocks of the frame through the M side: lane (g, n) then holds S[i = n][j = 16 jb + 4 g + 0..3], the running first-max
//               is per lane, and the lane groups and waves of an A row meet once at the end.  All waves of a frame run on ONE XCD
//               (blockIdx -> frame map below): the frame's B half is fetched from HBM once and re-read from that L2;
//   3. select : one workgroup per frame ranks node_max (integer keys in LDS: (value desc, index asc) is a total order, so the count
//               is exact whatever the thread order) and writes src / dst / unm;
//   4. merge (+ LayerNorm 2): one wave per output row, 16-byte loads / stores; with LN the normalised row (aurora.py:750) is
//               written beside the merged one from the registers that hold it - the arithmetic of norm_kernel<false> on the
//               rounded fp16 row, bitwise.
#include "kernels.h"

// ------------------------------------------------------------------ metric from K fragments
// metric[f][tok][d] = (sum over heads h ascending of K[f][h][tok][d]) * (1/H)      (aurora.py:639)
__global__ void tome_metric_kernel(KvLayout kv, int frames, int t, int hd, float* __restrict__ metric) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int pieces = kv.kblk * 4;
    const int64_t total = (int64_t)frames * t * pieces;
    if (id >= total) return;
    const int pc = id % pieces;
    const int tok = (id / pieces) % t;
    const int f = id / ((int64_t)pieces * t);
    const int blk = pc >> 2, g = pc & 3;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const half_t* page = kv.base + (int64_t)f * kv.page_halves;
    for (int h = 0; h < kv.heads; ++h) {
        const h8 v = *(const h8*)(page + kfrag_off(kv, h, tok >> 4, blk) + (g * 16 + (tok & 15)) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
    }
    const float inv = 1.0f / (float)kv.heads;
    float* mrow = metric + ((int64_t)f * t + tok) * hd;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int d = blk * 32 + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
        if (d < hd) mrow[d] = acc[j] * inv;
    }
}

hipError_t launch_tome_metric(const KvLayout& kv, int frames, int t, int hd, float* metric, hipStream_t s) {
    const int64_t total = (int64_t)frames * t * kv.kblk * 4;
    hipLaunchKernelGGL(tome_metric_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, kv, frames, t, hd, metric);
    return hipGetLastError();
}

// ------------------------------------------------------------------ prep: metric (+ mean over heads) -> normalised MFMA operand blocks
// One workgroup per (frame, 32 consecutive tokens) = 16 A rows (even tokens) + 16 B rows (odd tokens) = one A block and one B block.
// SRC_KV: the 32 rows are the mean over heads (ascending, fp32) of the layer's K fragments - one thread per 16-byte piece (2 x kblk x 64
// threads: consecutive threads read consecutive pieces of a fragment, coalesced 1 KiB per 16 tokens), four heads' loads in flight -
// else rows of the caller's metric.  ONE thread per row runs the k-ordered fmaf chain of the norm (the contract of this file), every
// element is divided (IEEE) as it is stored.
// mfrag frame layout: [nbA A blocks][nbB B blocks] x [G = ceil(c / 16) groups][64 lanes][4 floats].
__host__ __device__ inline int tome_nblk(int rows) { return (rows + 15) >> 4; }

template <bool SRC_KV>
__global__ __launch_bounds__(1024) void tome_prep_kernel(KvLayout kv, const float* __restrict__ metric, int t, int c, int ldw,
                                                         float* __restrict__ metric_out, float* __restrict__ mfrag) {
    extern __shared__ __attribute__((aligned(16))) float msm[];       // [32][ldw] rows + [32] norms
    float* nrm = msm + 32 * ldw;
    const int tid = threadIdx.x, nthr = blockDim.x, bx = blockIdx.x, f = blockIdx.y;
    const int tok0 = bx * 32;
    if (SRC_KV) {
        const int items = 2 * kv.kblk * 64;                           // (tok16 half, d-block, fragment lane)
        const half_t* page = kv.base + (int64_t)f * kv.page_halves;
        const float inv = 1.0f / (float)kv.heads;
        for (int it = tid; it < items; it += nthr) {
            const int fl = it & 63, blk = (it >> 6) % kv.kblk, th = (it >> 6) / kv.kblk;
            const int g = fl >> 4, tl = th * 16 + (fl & 15), tok = tok0 + tl;
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
            if (tok < t) {
                const half_t* p0 = page + kfrag_off(kv, 0, tok >> 4, blk) + fl * 8;
                const int64_t hstride = kfrag_off(kv, 1, 0, 0) - kfrag_off(kv, 0, 0, 0);
                int h = 0;
                // sixteen, then four independent 16-byte loads in flight per thread; summed in head order whatever the batching (round 6; four
                // in flight before: 37.4 -> 36.8 us at 128 frames per launch in the kernel trace - the launch is not short of loads in flight)
                for (; h + 16 <= kv.heads; h += 16) {
                    h8 v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = *(const h8*)(p0 + (h + u) * hstride);
#pragma unroll
                    for (int u = 0; u < 16; ++u)
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] += (float)v[u][j];
                }
                for (; h + 4 <= kv.heads; h += 4) {
                    h8 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = *(const h8*)(p0 + (h + u) * hstride);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] += (float)v[u][j];
                }
                for (; h < kv.heads; ++h) {
                    const h8 v = *(const h8*)(p0 + h * hstride);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = blk * 32 + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
                msm[tl * ldw + d] = acc[j] * inv;
            }
        }
    } else {
        for (int idx = tid; idx < 32 * c; idx += nthr) {
            const int tl = idx / c, k = idx - tl * c;
            const int tok = tok0 + tl;
            msm[tl * ldw + k] = tok < t ? metric[((int64_t)f * t + tok) * c + k] : 0.f;
        }
    }
    __syncthreads();
    if (tid < 32) {
        const float* m = msm + tid * ldw;
        float n2 = 0.0f;
        for (int k = 0; k < c; ++k) n2 = fmaf(m[k], m[k], n2);
        nrm[tid] = sqrtf(n2);
    }
    if (SRC_KV && metric_out) {
        for (int idx = tid; idx < 32 * c; idx += nthr) {
            const int tl = idx / c, k = idx - tl * c;
            if (tok0 + tl < t) metric_out[((int64_t)f * t + tok0 + tl) * c + k] = msm[tl * ldw + k];
        }
    }
    __syncthreads();
    const int ta = (t + 1) >> 1, tb = t >> 1, nbA = tome_nblk(ta), nbB = tome_nblk(tb), G = (c + 15) >> 4;
    float* fb = mfrag + (int64_t)f * (nbA + nbB) * G * 256;
    for (int it = tid; it < 2 * G * 64; it += nthr) {
        const int half = it / (G * 64), rem = it - half * (G * 64);
        const int g = rem >> 6, lane = rem & 63;
        if (half == 1 && bx >= nbB) break;                              // t odd: the last workgroup holds an A row only
        const int tl = 2 * (lane & 15) + half, kq = lane >> 4;
        const float n = nrm[tl];
        f4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 16 * g + 4 * q + kq;
            v[q] = (tok0 + tl < t && k < c) ? msm[tl * ldw + k] / n : 0.f;
        }
        *(f4*)(fb + ((int64_t)(half ? nbA + bx : bx) * G + g) * 256 + lane * 4) = v;
    }
}

// ------------------------------------------------------------------ match (tome.py:52-60) on v_mfma_f32_16x16x4_f32
// MFMA roles: M side = 16 B tokens of block jb (operand a), N side = the workgroup's 16 A tokens (operand b), K = 4 metric channels per
// instruction, 4 G instructions per block in ascending k: D[m][n] = fmaf chain over k of Bhat[j][k] * Ahat[i][k], start 0.0f.
// One workgroup = one A block; its four waves take the B blocks jb = w, w + 4, w + 8, .. (ascending inside a wave), two blocks in
// flight per iteration (two independent accumulator chains cover the instruction's 40-cycle dependent latency; the next pair's
// operands are loaded before the current pair's MFMAs issue).  The per-lane first maxima meet across the 4 lane groups of a row, then
// across the 4 waves through LDS: (value desc, j asc) picks the first maximum whatever the order of the meeting.
__device__ __forceinline__ f4 mfma_f32x4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void first_max_take(float& best, int& bj, float ov, int oj) {
    if (ov > best || (ov == best && oj < bj)) {
        best = ov;
        bj = oj;
    }
}

template <int G>
__global__ __launch_bounds__(256) void tome_match_kernel(const float* __restrict__ mfrag, int frames, int t,
                                                         float* __restrict__ node_max, int32_t* __restrict__ node_idx) {
    __shared__ float sv[4][16];
    __shared__ int sj[4][16];
    const int ta = (t + 1) >> 1, tb = t >> 1, nbA = tome_nblk(ta), nbB = tome_nblk(tb);
    // blockIdx -> (frame, A block): workgroups b, b + 8, b + 16, .. share an XCD (observed placement: block b runs on XCD b % 8; a speed
    // assumption only), so a frame's workgroups are dealt to ONE residue class and re-read its B half from one L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int f = (slot / nbA) * 8 + xcd, ib = slot % nbA;
    if (f >= frames) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const f4* fb = (const f4*)(mfrag + (int64_t)f * (nbA + nbB) * G * 256);
    const f4* ap = fb + (int64_t)ib * G * 64 + lane;
    const f4* bp = fb + (int64_t)nbA * G * 64 + lane;
    float best = -INFINITY;
    int bj = 0x7fffffff;
    if (wave < nbB) {
        f4 a[G], b0[G], b1[G];
        const int jn = wave + 4 < nbB ? wave + 4 : wave;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            a[g] = ap[g * 64];
            b0[g] = bp[((int64_t)wave * G + g) * 64];
            b1[g] = bp[((int64_t)jn * G + g) * 64];
        }
        const int jrow = 4 * (lane >> 4);
        for (int jb = wave; jb < nbB; jb += 8) {
            f4 n0[G], n1[G];
            const int j2 = jb + 8 < nbB ? jb + 8 : jb, j3 = jb + 12 < nbB ? jb + 12 : jb;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                n0[g] = bp[((int64_t)j2 * G + g) * 64];
                n1[g] = bp[((int64_t)j3 * G + g) * 64];
            }
            f4 acc0 = f4{0.f, 0.f, 0.f, 0.f}, acc1 = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc0 = mfma_f32x4(b0[g][q], a[g][q], acc0);
                    acc1 = mfma_f32x4(b1[g][q], a[g][q], acc1);
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {                               // first maximum, j ascending inside the lane (tome.py:60)
                const int j = jb * 16 + jrow + i;
                const float sv0 = acc0[i];
                if (j == 0 || (j < tb && sv0 > best)) {
                    best = sv0;
                    bj = j;
                }
            }
            if (jb + 4 < nbB) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = (jb + 4) * 16 + jrow + i;
                    const float sv1 = acc1[i];
                    if (j < tb && sv1 > best) {
                        best = sv1;
                        bj = j;
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                b0[g] = n0[g];
                b1[g] = n1[g];
            }
        }
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) first_max_take(best, bj, __shfl_xor(best, o, 64), __shfl_xor(bj, o, 64));
    }
    if (lane < 16) {
        sv[wave][lane] = best;
        sj[wave][lane] = bj;
    }
    __syncthreads();
    const int i = ib * 16 + lane;
    if (wave == 0 && lane < 16 && i < ta) {
        // wave 0 holds j = 0 (the unconditional first candidate of the oracle's scan); the others join by (value desc, j asc)
#pragma unroll
        for (int w = 1; w < 4; ++w) first_max_take(best, bj, sv[w][lane], sj[w][lane]);
        if (i == 0) {                         // class token row: scores = -inf (tome.py:55-56)
            best = -INFINITY;
            bj = 0;
        }
        if (bj == 0x7fffffff) bj = 0;
        node_max[(int64_t)f * ta + i] = best;
        node_idx[(int64_t)f * ta + i] = bj;
    }
}

// ------------------------------------------------------------------ select (tome.py:61-69)
// rank by (node_max desc, i asc); src = the r best in rank order, unm = the rest in ascending i, dst = node_idx[src].  ONE workgroup of
// 1024 threads per frame (a first version of round 5 ranked the frame in every workgroup of the merge launch: 184 copies of the same
// 133 k comparisons per frame made that launch VALU-bound, 32 us at 8 frames and 72 us at 32).  Keys: the floats mapped to
// order-preserving unsigned integers (-0.0 first canonicalised to +0.0, as float comparison treats them as equal) above the
// complemented index: ONE 64-bit compare per pair decides (value desc, index asc), a total order, so the count is exact whatever the
// thread order.  The count of a row is split over 8 thread groups (group h scans an eighth of the keys) and joined by integer atomics
// in LDS.  Outputs are plain stores: the merge launch is a kernel boundary later.
__global__ __launch_bounds__(1024) void tome_select_kernel(const float* __restrict__ node_max, const int32_t* __restrict__ node_idx, int t, int r,
                                                           int32_t* __restrict__ unm, int32_t* __restrict__ src, int32_t* __restrict__ dst) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long ssm64[];
    const int tid = threadIdx.x, f = blockIdx.x;
    const int ta = (t + 1) >> 1, nu = ta - r, tap = (ta + 15) & ~15;
    unsigned long long* keys = ssm64;        // [tap]
    int* rk = (int*)(keys + tap);            // [ta]
    int* srcl = rk + ta;                     // [r]
    for (int i = tid; i < tap; i += 1024) {
        unsigned long long key = 0ull;
        if (i < ta) {
            const uint32_t u = __float_as_uint(node_max[(int64_t)f * ta + i] + 0.0f);
            key = ((unsigned long long)((u & 0x80000000u) ? ~u : (u | 0x80000000u)) << 32) | (uint32_t)(0xffffffffu - (uint32_t)i);
            rk[i] = 0;
        }
        keys[i] = key;
    }
    __syncthreads();
    {
        const int h = tid >> 7, il = tid & 127;
        const int seg = tap >> 3, j0 = h * seg, j1 = j0 + seg;              // tap % 16 == 0: eight even-length segments
        for (int i = il; i < ta; i += 128) {
            const unsigned long long k = keys[i];
            int cnt = 0;
            for (int j = j0; j < j1; j += 2) {
                const ulonglong2 k2 = *(const ulonglong2*)(keys + j);
                cnt += k2.x > k ? 1 : 0;
                cnt += k2.y > k ? 1 : 0;
            }
            if (cnt) atomicAdd(rk + i, cnt);
        }
    }
    __syncthreads();
    for (int i = tid; i < ta; i += 1024) {
        const int rank = rk[i];
        if (rank < r) {
            srcl[rank] = i;
            src[(int64_t)f * r + rank] = i;
            dst[(int64_t)f * r + rank] = node_idx[(int64_t)f * ta + i];
        }
    }
    __syncthreads();
    for (int i = tid; i < ta; i += 1024) {
        if (rk[i] >= r) {
            int before = 0;
            for (int q = 0; q < r; ++q) before += srcl[q] < i ? 1 : 0;
            unm[(int64_t)f * nu + (i - before)] = i;
        }
    }
}

// ------------------------------------------------------------------ merge (tome.py:71-81, 207-219) (+ LayerNorm 2, aurora.py:750)
// One wave per output row; a lane owns chunks lane, lane + 64, ... of V halves (V = 8: 16-byte loads / stores when d % 8 == 0).
// Sum order per element: the B (or unmerged A) row first, then the merged A rows in rank order - as the oracle does.  LN: the normalised row is written beside the merged one from the registers
// that hold it - norm.hip's norm_kernel<false> on the ROUNDED fp16 row, operation for operation (its two FMAs are explicit there).
#define TM_ROWS 4       // output rows per workgroup: one per wave
template <int V, int MAXC, bool LN>
__global__ __launch_bounds__(256) void tome_merge_kernel(TomeArgs a) {
    typedef half_t hv __attribute__((ext_vector_type(V)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f = blockIdx.y;
    const int ta = (a.t + 1) >> 1, nu = ta - a.r, t_out = a.t - a.r;
    const int o = blockIdx.x * TM_ROWS + wave;
    if (o >= a.t_out_pad) return;
    const int nchunk = a.d / V;
    const half_t* xf = a.x + (int64_t)f * a.t_pad * a.d;
    const float* sf = a.size ? a.size + (int64_t)f * a.t_pad : nullptr;
    half_t* orow = a.x_out + ((int64_t)f * a.t_out_pad + o) * a.d;
    half_t* yrow = LN ? a.ln_out + ((int64_t)f * a.t_out_pad + o) * a.d : nullptr;
    float* so = a.size_out + (int64_t)f * a.t_out_pad + o;
    float acc[MAXC][V];
    float st = 1.0f;
    if (o >= t_out) {       // padding rows: zeros, size 1
#pragma unroll
        for (int i = 0; i < MAXC; ++i)
#pragma unroll
            for (int j = 0; j < V; ++j) acc[i][j] = 0.f;
    } else {
        // (B rows: the first 64 (dst, src) pairs are requested BEFORE the row itself, so that all of the wave's loads share one round trip)
        const int32_t* dstf = a.dst + (int64_t)f * a.r;
        const int32_t* srcf = a.src + (int64_t)f * a.r;
        int my_dst = -1, my_src = 0;
        if (o >= nu && lane < a.r) {
            my_dst = dstf[lane];
            my_src = srcf[lane];
        }
        const int tok = o < nu ? 2 * a.unm[(int64_t)f * nu + o] : 2 * (o - nu) + 1;
        const float s0 = sf ? sf[tok] : 1.0f;
        st = s0;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + i * 64;
            if (c < nchunk) {
                const hv v = *(const hv*)(xf + (int64_t)tok * a.d + c * V);
#pragma unroll
                for (int j = 0; j < V; ++j) acc[i][j] = (float)v[j] * s0;
            }
        }
        if (o >= nu) {
            // the frame's (dst, src) pairs, 64 per pass: lane q holds pair q (one load each, in flight beside the row's own loads above -
            // a first version staged the lists in LDS behind a barrier: one more dependent round trip per workgroup); the pairs that
            // merge into this row are the set bits of a ballot, visited in ascending q = rank order
            const int jrow = o - nu;
            for (int q0 = 0; q0 < a.r; q0 += 64) {
                if (q0 > 0) {
                    const int q = q0 + lane;
                    my_dst = q < a.r ? dstf[q] : -1;
                    my_src = q < a.r ? srcf[q] : 0;
                }
                unsigned long long hits = __ballot(my_dst == jrow);
                while (hits) {
                    const int bq = __builtin_ctzll(hits);
                    hits &= hits - 1;
                    const int tsq = 2 * __builtin_amdgcn_readlane(my_src, bq);
                    const float ss = sf ? sf[tsq] : 1.0f;
#pragma unroll
                    for (int i = 0; i < MAXC; ++i) {
                        const int c = lane + i * 64;
                        if (c < nchunk) {
                            const hv v = *(const hv*)(xf + (int64_t)tsq * a.d + c * V);
#pragma unroll
                            for (int j = 0; j < V; ++j) {
                                const float p = (float)v[j] * ss;
                                acc[i][j] = acc[i][j] + p;
                            }
                        }
                    }
                    st = st + ss;
                }
            }
        }
    }
    float ls = 0.f;                      // LN: sum of the ROUNDED row, in norm_kernel<false>'s order
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nchunk) {
            hv ov;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                ov[j] = (half_t)(acc[i][j] / st);
                if (LN) {
                    acc[i][j] = (float)ov[j];
                    ls = ls + acc[i][j];
                }
            }
            *(hv*)(orow + c * V) = ov;
        }
    }
    if (lane == 0) *so = st;
    if (LN) {
        ls = wave_sum(ls);
        const float mean = ls / (float)a.d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + i * 64;
            if (c < nchunk) {
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float dlt = acc[i][j] - mean;
                    q = __builtin_fmaf(dlt, dlt, q);
                }
            }
        }
        q = wave_sum(q);
        const float rstd = rsqrtf(q / (float)a.d + a.ln_eps);
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + i * 64;
            if (c < nchunk) {
                hv ov;
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const int k = c * V + j;
                    const float tv = (acc[i][j] - mean) * rstd;
                    ov[j] = (half_t)__builtin_fmaf(tv, a.ln_w[k], a.ln_b[k]);
                }
                *(hv*)(yrow + c * V) = ov;
            }
        }
    }
}

hipError_t tome_init() {
    return hipFuncSetAttribute((const void*)tome_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
}

// floats of the mfrag scratch one frame of t tokens x c channels needs (engine.hip sizes its workspace with it)
int64_t tome_mfrag_floats(int t, int c) { return (int64_t)(tome_nblk((t + 1) >> 1) + tome_nblk(t >> 1)) * ((c + 15) >> 4) * 256; }

template <int G>
static void launch_match_g(const TomeArgs& a, hipStream_t s) {
    const int nbA = tome_nblk((a.t + 1) >> 1);
    const unsigned grid = (unsigned)(((a.frames + 7) / 8) * 8 * nbA);
    hipLaunchKernelGGL((tome_match_kernel<G>), dim3(grid), dim3(256), 0, s, a.mhat, a.frames, a.t, a.node_max, a.node_idx);
}

hipError_t launch_tome_step(const TomeArgs& a, hipStream_t s) {
    const int ta = (a.t + 1) >> 1;
    if (a.r <= 0 || a.r > (a.t - 1) / 2 || a.d > 2048 || (a.d & 3) || ta > 4096 || a.c < 1 || a.c > 128) return hipErrorInvalidValue;
    if (a.ln_out && (a.d & 7)) return hipErrorInvalidValue;
    const dim3 pgrid((a.t + 31) / 32, a.frames);
    if (a.kv) {                                                // metric = mean over heads of this layer's K fragments (aurora.py:639)
        const int hdp = a.kv->kblk * 32;
        if (a.c > hdp || a.kv->kblk > 8) return hipErrorInvalidValue;
        const int ldw = hdp + 1;
        hipLaunchKernelGGL((tome_prep_kernel<true>), pgrid, dim3(2 * a.kv->kblk * 64), (size_t)(32 * ldw + 32) * 4, s, *a.kv, (const float*)nullptr,
                           a.t, a.c, ldw, a.metric_out, a.mhat);
    } else {
        const int ldw = ((a.c + 15) & ~15) + 1;
        hipLaunchKernelGGL((tome_prep_kernel<false>), pgrid, dim3(256), (size_t)(32 * ldw + 32) * 4, s, KvLayout{}, a.metric, a.t, a.c, ldw,
                           (float*)nullptr, a.mhat);
    }
    switch ((a.c + 15) >> 4) {
        case 1: launch_match_g<1>(a, s); break;
        case 2: launch_match_g<2>(a, s); break;
        case 3: launch_match_g<3>(a, s); break;
        case 4: launch_match_g<4>(a, s); break;
        case 5: launch_match_g<5>(a, s); break;
        case 6: launch_match_g<6>(a, s); break;
        case 7: launch_match_g<7>(a, s); break;
        default: launch_match_g<8>(a, s); break;
    }
    const size_t lds_s = (size_t)((ta + 15) & ~15) * 8 + (size_t)(ta + a.r + 4) * 4;      // keys (8 bytes each) + ranks + src list
    hipLaunchKernelGGL(tome_select_kernel, dim3(a.frames), dim3(1024), lds_s, s, a.node_max, a.node_idx, a.t, a.r, a.unm, a.src, a.dst);
    const dim3 grid((a.t_out_pad + TM_ROWS - 1) / TM_ROWS, a.frames);
    const size_t lds = 0;
    if ((a.d & 7) == 0) {
        if (a.ln_out) hipLaunchKernelGGL((tome_merge_kernel<8, 4, true>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((tome_merge_kernel<8, 4, false>), grid, dim3(256), lds, s, a);
    } else {
        hipLaunchKernelGGL((tome_merge_kernel<4, 8, false>), grid, dim3(256), lds, s, a);
    }
    return hipGetLastError();
}
