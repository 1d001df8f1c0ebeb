/* Bit-exact CPU oracle for one ToMe step.  TEST INFRASTRUCTURE ONLY - never linked into the
 * product library (aurora_amd/csrc); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg load it.
 *
 * Restates rese1f/aurora src/xtuner/xtuner/model/tome.py:36-81 (bipartite_soft_matching with
 * class_token=True) and tome.py:207-219 (merge_wavg) with a FULLY SPECIFIED fp32 operation
 * order, so that the HIP kernels (aurora_amd/csrc/tome.hip), which use the identical order,
 * reproduce every index bit-for-bit on identical metric bytes:
 *
 *   n2[i]   = fma-chain over k ascending of m[i][k]*m[i][k], start 0.0f         (tome.py:51 norm)
 *   mh[i][k]= m[i][k] / sqrtf(n2[i])           (IEEE correctly rounded sqrt, div)  (tome.py:51)
 *   S[i][j] = fma-chain over k ascending of A[i][k]*B[j][k], start 0.0f         (tome.py:53)
 *   S[0][*] = -inf                                                               (tome.py:55-56)
 *   node_max/idx = first maximum over j (strict >)                               (tome.py:60)
 *   order   = node_max descending, ties -> lower i first (stable argsort)        (tome.py:61)
 *   src = first r of order, unm = rest sorted ascending, dst = node_idx[src]     (tome.py:63-69)
 *   merge:  acc = x_B*s_B; for k ascending with dst[k]==j: acc += x_A[src k]*s_A  (tome.py:71-81,
 *           out = acc / s_tot                                                     207-219)
 *
 * Pinned against the reference's own tome.py via tests/golden (tests/test_oracle_golden.py).
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; fmaf() is exact in libm either way).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* metric [t][c] -> indices.  ta = ceil(t/2) A rows (even tokens), tb = floor(t/2) B rows.
 * r must already be clamped: 0 < r <= (t-1)/2.  Outputs: node_max[ta], node_idx[ta],
 * unm[ta-r], src[r], dst[r].  Returns 0, or -1 on bad arguments. */
int tome_match_ref(const float* metric, int t, int c, int r,
                   float* node_max, int32_t* node_idx,
                   int32_t* unm, int32_t* src, int32_t* dst) {
    if (t < 3 || c < 1 || r < 1 || r > (t - 1) / 2) return -1;
    const int ta = (t + 1) / 2, tb = t / 2;
    float* mh = (float*)malloc((size_t)t * c * sizeof(float));
    if (!mh) return -1;
    for (int i = 0; i < t; ++i) {
        const float* m = metric + (size_t)i * c;
        float n2 = 0.0f;
        for (int k = 0; k < c; ++k) n2 = fmaf(m[k], m[k], n2);
        const float nrm = sqrtf(n2);
        for (int k = 0; k < c; ++k) mh[(size_t)i * c + k] = m[k] / nrm;
    }
    for (int i = 0; i < ta; ++i) {
        const float* a = mh + (size_t)(2 * i) * c;
        float best = -INFINITY;
        int bi = 0;
        if (i > 0) {
            for (int j = 0; j < tb; ++j) {
                const float* b = mh + (size_t)(2 * j + 1) * c;
                float s = 0.0f;
                for (int k = 0; k < c; ++k) s = fmaf(a[k], b[k], s);
                if (j == 0 || s > best) { best = s; bi = j; }
            }
        }
        node_max[i] = best;
        node_idx[i] = bi;
    }
    free(mh);
    /* rank by (node_max desc, i asc) */
    int nu = 0;
    for (int i = 0; i < ta; ++i) {
        int rank = 0;
        for (int i2 = 0; i2 < ta; ++i2) {
            if (node_max[i2] > node_max[i] || (node_max[i2] == node_max[i] && i2 < i)) ++rank;
        }
        if (rank < r) { src[rank] = i; }
        else { unm[nu++] = i; }            /* i ascending => unm ascending */
    }
    for (int k = 0; k < r; ++k) dst[k] = node_idx[src[k]];
    return 0;
}

/* x [t][d] fp32 (values exactly representable in the kernel's fp16 input), size [t] ->
 * x_out [t-r][d], size_out [t-r].  Row order: A[unm] then all B rows (tome.py:81). */
int tome_merge_ref(const float* x, const float* size, int t, int d, int r,
                   const int32_t* unm, const int32_t* src, const int32_t* dst,
                   float* x_out, float* size_out) {
    const int ta = (t + 1) / 2, tb = t / 2, nu = ta - r;
    for (int o = 0; o < nu; ++o) {
        const int tok = 2 * unm[o];
        const float s = size[tok];
        for (int k = 0; k < d; ++k) {
            const float p = x[(size_t)tok * d + k] * s;
            x_out[(size_t)o * d + k] = p / s;
        }
        size_out[o] = s;
    }
    for (int j = 0; j < tb; ++j) {
        const int tok = 2 * j + 1;
        float st = size[tok];
        float* out = x_out + (size_t)(nu + j) * d;
        for (int k = 0; k < d; ++k) out[k] = x[(size_t)tok * d + k] * size[tok];
        for (int q = 0; q < r; ++q) {
            if (dst[q] != j) continue;
            const int ts = 2 * src[q];
            const float ss = size[ts];
            for (int k = 0; k < d; ++k) {
                const float p = x[(size_t)ts * d + k] * ss;
                out[k] = out[k] + p;
            }
            st = st + ss;
        }
        for (int k = 0; k < d; ++k) out[k] = out[k] / st;
        size_out[nu + j] = st;
    }
    return 0;
}
