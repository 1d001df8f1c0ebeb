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
// Three launches per step (round 1: five):
//   1. metric + normalise: metric = mean over heads of the layer's K fragments, read once, normalised in LDS (the C-ABI entry
//      aur_tome_step, whose metric is an input, runs the normalise half alone);
//   2. match + select: grid over 32-row A blocks x frames, 4x4 register tiles from k-major LDS images, first-max per A row; the
//      LAST workgroup of a frame to finish (one atomic counter per frame, release / acquire fences around it) ranks the frame's
//      A rows and writes src / dst / unm - which workgroup that is does not matter, the inputs are complete and the rule is fixed;
//   3. merge: one wave per output row, 16-byte loads, HBM-bound (reads t rows, writes t-r rows of the [frames, t, D] state).
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

// ------------------------------------------------------------------ normalise (tome.py:51)
__global__ void tome_normalize_kernel(const float* __restrict__ metric, int64_t rows, int c, float* __restrict__ mhat) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const float* m = metric + row * c;
    float n2 = 0.0f;
    for (int k = 0; k < c; ++k) n2 = fmaf(m[k], m[k], n2);
    const float nrm = sqrtf(n2);
    float* o = mhat + row * c;
    for (int k = 0; k < c; ++k) o[k] = m[k] / nrm;
}

// ------------------------------------------------------------------ metric from K fragments + normalise, one launch
// A workgroup owns R = 256 / pieces token rows (pieces = 8-wide slices of a padded head): every thread sums its slice over the heads
// (ascending, fp32: the bytes tome_metric_kernel produces) into LDS, ONE thread per row runs the k-ordered fmaf chain of the norm
// (the contract of this file), then the row is divided and stored.  metric_out (optional) receives the un-normalised rows.
__global__ __launch_bounds__(256) void tome_metric_norm_kernel(KvLayout kv, int frames, int t, int hd, float* __restrict__ metric_out,
                                                               float* __restrict__ mhat) {
    extern __shared__ __attribute__((aligned(16))) float msm[];       // [R][hdp] rows + [R] norms
    const int pieces = kv.kblk * 4, hdp = pieces * 8;
    const int R = 256 / pieces;
    const int tid = threadIdx.x;
    const int rl = tid / pieces, pc = tid - rl * pieces;
    const int64_t row = (int64_t)blockIdx.x * R + rl;                 // over frames * t
    const int64_t rows = (int64_t)frames * t;
    float* nrm = msm + R * hdp;
    const bool live = rl < R && row < rows;
    if (live) {
        const int f = (int)(row / t), tok = (int)(row - (int64_t)f * t);
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
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = blk * 32 + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
            msm[rl * hdp + d] = acc[j] * inv;
        }
    }
    __syncthreads();
    if (tid < R && (int64_t)blockIdx.x * R + tid < rows) {
        const float* m = msm + tid * hdp;
        float n2 = 0.0f;
        for (int k = 0; k < hd; ++k) n2 = fmaf(m[k], m[k], n2);
        nrm[tid] = sqrtf(n2);
    }
    __syncthreads();
    for (int idx = tid; idx < R * hd; idx += 256) {
        const int r2 = idx / hd, k = idx - r2 * hd;
        const int64_t row2 = (int64_t)blockIdx.x * R + r2;
        if (row2 >= rows) break;
        const float v = msm[r2 * hdp + k];
        if (metric_out) metric_out[row2 * hd + k] = v;
        mhat[row2 * hd + k] = v / nrm[r2];
    }
}

// ------------------------------------------------------------------ match (tome.py:52-60)
#define TM_RA 32      // A rows per workgroup
#define TM_JB 128     // B rows per pass
__global__ __launch_bounds__(256) void tome_match_kernel(const float* __restrict__ mhat, int t, int c, int r,
                                                         float* __restrict__ node_max, int32_t* __restrict__ node_idx,
                                                         int32_t* __restrict__ counters, int32_t* __restrict__ unm,
                                                         int32_t* __restrict__ src, int32_t* __restrict__ dst) {
    extern __shared__ __attribute__((aligned(16))) float tsm[];
    float* As = tsm;                         // [c][TM_RA]
    float* Bs = tsm + c * TM_RA;             // [c][TM_JB]
    float* rv = Bs + c * TM_JB;              // [TM_RA][32] reduction values
    int* ri = (int*)(rv + TM_RA * 32);       // [TM_RA][32]
    const int tid = threadIdx.x;
    const int f = blockIdx.y;
    const int ta = (t + 1) >> 1, tb = t >> 1;
    const int i0 = blockIdx.x * TM_RA;
    const float* mf = mhat + (int64_t)f * t * c;
    // A rows (even tokens) -> k-major LDS
    for (int idx = tid; idx < TM_RA * c; idx += 256) {
        const int il = idx % TM_RA, k = idx / TM_RA;
        const int i = i0 + il;
        As[k * TM_RA + il] = (i < ta) ? mf[(int64_t)(2 * i) * c + k] : 0.f;
    }
    const int ig = tid >> 5;          // 0..7  -> rows ig*4 .. +3
    const int jg = tid & 31;          // 0..31 -> cols jg*4 .. +3 of each pass
    float best[4];
    int bj[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        best[a] = -INFINITY;
        bj[a] = 0x7fffffff;
    }
    for (int j0 = 0; j0 < tb; j0 += TM_JB) {
        __syncthreads();
        for (int idx = tid; idx < TM_JB * c; idx += 256) {
            const int jl = idx % TM_JB, k = idx / TM_JB;
            const int j = j0 + jl;
            Bs[k * TM_JB + jl] = (j < tb) ? mf[(int64_t)(2 * j + 1) * c + k] : 0.f;
        }
        __syncthreads();
        float acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = 0.0f;
        for (int k = 0; k < c; ++k) {
            const f4 av = *(const f4*)(As + k * TM_RA + ig * 4);
            const f4 bv = *(const f4*)(Bs + k * TM_JB + jg * 4);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(av[a], bv[b], acc[a][b]);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int j = j0 + jg * 4 + b;
                const float sv = acc[a][b];
                if (j < tb && (sv > best[a] || (sv == best[a] && j < bj[a]))) {
                    best[a] = sv;
                    bj[a] = j;
                }
            }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        rv[(ig * 4 + a) * 32 + jg] = best[a];
        ri[(ig * 4 + a) * 32 + jg] = bj[a];
    }
    __syncthreads();
    if (tid < TM_RA) {
        const int i = i0 + tid;
        if (i < ta) {
            float bv = -INFINITY;
            int bi = 0x7fffffff;
            for (int q = 0; q < 32; ++q) {
                const float v = rv[tid * 32 + q];
                const int j = ri[tid * 32 + q];
                if (j != 0x7fffffff && (v > bv || (v == bv && j < bi) || bi == 0x7fffffff)) {
                    bv = v;
                    bi = j;
                }
            }
            if (i == 0) {                     // class token row: scores = -inf (tome.py:55-56)
                bv = -INFINITY;
                bi = 0;
            }
            // agent-scope stores / loads for the hand-over below: the frame's workgroups sit on different XCDs, whose L2s are not
            // coherent for plain accesses.  (Full release / acquire fences - an L2 write-back per workgroup - cost ~100 us per launch.)
            // This is the guide's "sc1 payload -> drained vmcnt -> sc1 flag, sc1 loads on the consumer" form (MI355X_MICROARCH.md,
            // valid forms / handoff-flag row): relaxed agent-scope atomics lower to write-through `sc1` stores and L1-bypassing
            // `sc1` loads on gfx950, and the hand-written s_waitcnt below orders payload before arrival.  It is NOT a HIP-memory-model
            // release / acquire pair - hence the architecture guard; tests/test_gpu_kernels.py stresses it under uneven load.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "tome_match_kernel's hand-over relies on gfx950 sc1 write-through semantics; use release/acquire fences on other targets"
#endif
            __hip_atomic_store(node_max + (int64_t)f * ta + i, bv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(node_idx + (int64_t)f * ta + i, bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- select (tome.py:61-69) by the last workgroup of the frame: rank by (node_max desc, i asc); src = the r best in rank
    //      order, unm = the rest in ascending i.  A workgroup's rows have reached the coherence point (vmcnt 0) before its arrival
    //      is counted; the workgroup that finds the frame complete reads them back at agent scope.
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const int arrived = __hip_atomic_fetch_add(counters + f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = arrived == (int)gridDim.x - 1;
        if (s_last) __hip_atomic_store(counters + f, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next layer
    }
    __syncthreads();
    if (!s_last) return;
    float* nm = tsm;                   // [ta]   (the match images are dead)
    int* is_src = (int*)(tsm + ta);    // [ta]
    for (int i = tid; i < ta; i += 256) nm[i] = __hip_atomic_load(node_max + (int64_t)f * ta + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    for (int i = tid; i < ta; i += 256) {
        const float v = nm[i];
        int rank = 0;
        for (int i2 = 0; i2 < ta; ++i2) {
            const float v2 = nm[i2];
            rank += (v2 > v || (v2 == v && i2 < i)) ? 1 : 0;
        }
        is_src[i] = rank < r ? 1 : 0;
        if (rank < r) {
            src[(int64_t)f * r + rank] = i;
            dst[(int64_t)f * r + rank] = __hip_atomic_load(node_idx + (int64_t)f * ta + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    for (int i = tid; i < ta; i += 256) {
        if (!is_src[i]) {
            int before = 0;
            for (int i2 = 0; i2 < i; ++i2) before += is_src[i2];
            unm[(int64_t)f * (ta - r) + (i - before)] = i;
        }
    }
}

// ------------------------------------------------------------------ merge (tome.py:71-81, 207-219)
// One wave per output row; a lane owns chunks lane, lane + 64, ... of V halves (V = 8: 16-byte loads / stores when d % 8 == 0).
// Sum order per element: the B (or unmerged A) row first, then the merged A rows in rank order - as the oracle does.
template <int V, int MAXC>
__global__ __launch_bounds__(256) void tome_merge_kernel(TomeArgs a) {
    typedef half_t hv __attribute__((ext_vector_type(V)));
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int f = blockIdx.y;
    if (o >= a.t_out_pad) return;
    const int ta = (a.t + 1) >> 1, nu = ta - a.r, t_out = a.t - a.r;
    const int nchunk = a.d / V;
    half_t* orow = a.x_out + ((int64_t)f * a.t_out_pad + o) * a.d;
    float* so = a.size_out + (int64_t)f * a.t_out_pad + o;
    if (o >= t_out) {       // padding rows: zeros, size 1
        hv z;
#pragma unroll
        for (int j = 0; j < V; ++j) z[j] = (half_t)0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + i * 64;
            if (c < nchunk) *(hv*)(orow + c * V) = z;
        }
        if (lane == 0) *so = 1.0f;
        return;
    }
    const half_t* xf = a.x + (int64_t)f * a.t_pad * a.d;
    const float* sf = a.size ? a.size + (int64_t)f * a.t_pad : nullptr;
    float acc[MAXC][V];
    float st;
    int tok;
    if (o < nu) tok = 2 * a.unm[(int64_t)f * nu + o];
    else tok = 2 * (o - nu) + 1;
    {
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
    }
    if (o >= nu) {
        const int jrow = o - nu;
        const int32_t* dstf = a.dst + (int64_t)f * a.r;
        const int32_t* srcf = a.src + (int64_t)f * a.r;
        for (int q = 0; q < a.r; ++q) {
            if (dstf[q] != jrow) continue;
            const int tsq = 2 * srcf[q];
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
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nchunk) {
            hv ov;
#pragma unroll
            for (int j = 0; j < V; ++j) ov[j] = (half_t)(acc[i][j] / st);
            *(hv*)(orow + c * V) = ov;
        }
    }
    if (lane == 0) *so = st;
}

hipError_t tome_init() {
    return hipFuncSetAttribute((const void*)tome_match_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);   // + 4 bytes of static LDS (the select flag)
}

hipError_t launch_tome_step(const TomeArgs& a, hipStream_t s) {
    const int ta = (a.t + 1) >> 1;
    if (a.r <= 0 || a.r > (a.t - 1) / 2 || a.d > 2048 || (a.d & 3) || ta > 4096 || !a.counters) return hipErrorInvalidValue;
    const int64_t rows = (int64_t)a.frames * a.t;
    if (a.kv) {                                                // metric from this layer's K fragments, normalised in the same launch
        const int pieces = a.kv->kblk * 4, R = 256 / pieces;
        if (pieces < 1 || pieces > 256 || a.c > pieces * 8) return hipErrorInvalidValue;
        hipLaunchKernelGGL(tome_metric_norm_kernel, dim3((unsigned)((rows + R - 1) / R)), dim3(256), (size_t)(R * pieces * 8 + R) * 4, s,
                           *a.kv, a.frames, a.t, a.c, a.metric_out, a.mhat);
    } else {
        hipLaunchKernelGGL(tome_normalize_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, a.metric, rows, a.c, a.mhat);
    }
    size_t lds_m = (size_t)(a.c * (TM_RA + TM_JB) + TM_RA * 64) * sizeof(float);
    if (lds_m < (size_t)ta * 8) lds_m = (size_t)ta * 8;        // the select tail re-uses the images: [ta] values + [ta] flags
    hipLaunchKernelGGL(tome_match_kernel, dim3((ta + TM_RA - 1) / TM_RA, a.frames), dim3(256), lds_m, s, a.mhat, a.t, a.c, a.r,
                       a.node_max, a.node_idx, a.counters, a.unm, a.src, a.dst);
    const dim3 grid((a.t_out_pad + 3) / 4, a.frames);
    if ((a.d & 7) == 0) hipLaunchKernelGGL((tome_merge_kernel<8, 4>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((tome_merge_kernel<4, 8>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}
