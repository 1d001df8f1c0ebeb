// Shared device/host definitions for the MI355X (gfx950) AuroraCap kernels.
//
// Data-layout vocabulary used by every kernel in this directory ("everything is a fragment"):
//
//  * FRAG  = one MFMA 16x16x32 f16 operand tile: 16 rows x 32 k-slots, 1 KiB, stored as
//            [lane 0..63][8 halves] with lane = g*16 + r  (r = row 0..15, g = k-group 0..3).
//            A wave reads it with ONE 16-byte load per lane at base + lane*16: perfectly coalesced
//            from HBM, lane-linear for LDS-DMA (global_load_lds), conflict-free for ds_read_b128.
//  * k-slot maps.  Contraction only needs both operands to agree on which k sits in slot (g, j):
//      LINEAR : k = g*8 + j                       (weights packed from row-major, staged activations)
//      PAIRED : k = 4g + j (j<4), 16 + 4g + (j-4) (j>=4)   (what two adjacent 16-wide accumulator
//               tiles of one lane hold: no cross-lane traffic to turn GEMM output into an operand)
//  * MFMA 16x16x32 C/D map (guide section 3): lane holds D[row = 4*(lane>>4) + i][col = lane & 15], i = 0..3.
#pragma once
#define AUR_MAX_BATCH 128    /* decode slots per bank: up to 8 MFMA column groups of 16 batch rows (KV at 128 x 2.4k tokens: 156 of 288 GB) */
#define AUR_STAMP_RING 4096 /* decode steps whose in-loop attention interval is kept (option "decode_stamp_layer") */
#define AUR_SSQ_SLOTS 16     /* stripes of the sum(x^2) accumulators (power of two): decode.hip ssq_to_rstd */
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define AUR_FRAG_HALVES 512          // 1 KiB
#define AUR_WAVE 64

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// 16-byte async global -> LDS copy: LDS destination = wave-uniform base + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc_lane, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GPTR(gsrc_lane), LPTR(lds_wave_base), 16, 0, 0);
}

__device__ __forceinline__ f4 mfma16(h8 a, h8 b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// Lane exchanges over the 4 rows of 16 lanes on the VALU (gfx950: v_permlane16_swap / v_permlane32_swap) instead of ds_bpermute_b32
// through the LDS crossbar (~100+ cycles of latency behind the same lgkmcnt as the kernel's real LDS reads).  Both lanes of a pair
// compute `own op partner` from the same two values and + / max are commutative: results are BITWISE those of
// `v op __shfl_xor(v, 16)` / `(.., 32)`.  swap16(A, B): A.row1 <-> B.row0, A.row3 <-> B.row2; swap32(A, B): A.rows23 <-> B.rows01.
__device__ __forceinline__ float xor16_sum(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_max(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float rows_sum(float v) { return xor32_sum(xor16_sum(v)); }      // over the 4 lanes l, l ^ 16, l ^ 32, l ^ 48 (16 first)
__device__ __forceinline__ float rows_max(float v) { return xor32_max(xor16_max(v)); }
// max over the 16 lanes of a row, in every lane of it: DPP row rotations (order does not matter for max)
__device__ __forceinline__ float row16_max(float v) {
#define AUR_ROR(x, n) __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x120 + (n), 0xf, 0xf, false))
    v = fmaxf(v, AUR_ROR(v, 8));
    v = fmaxf(v, AUR_ROR(v, 4));
    v = fmaxf(v, AUR_ROR(v, 2));
    v = fmaxf(v, AUR_ROR(v, 1));
#undef AUR_ROR
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {      // same pairing order as before (32, 16, 8, ..): bitwise unchanged
    v = xor32_sum(v);
    v = xor16_sum(v);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = xor32_max(v);
    v = xor16_max(v);
    return row16_max(v);
}

// x * sigmoid(k x) with the hardware reciprocal and base-2 exponential: v_mul, v_exp_f32, v_add, v_rcp_f32 (1 ulp), v_mul - five
// instructions.  (Rounds 2-3 wrote `__fdividef(x, 1 + __expf(..))` believing it lowered to v_rcp_f32; round 4 read the ISA: without
// -ffast-math hipcc expands it to the full IEEE sequence - v_div_scale x 2, v_rcp, 4 fma, v_div_fmas, v_div_fixup - ~16 instructions per
// element, 128 elements per lane in the epilogue of every 256x256 MLP tile, which is VALU-issue-bound: tools/gemm_lab/ts_probe.py.)
// The outputs are rounded to fp16 right after, 13 bits above the reciprocal's last-ulp difference.
__device__ __forceinline__ float sigmoid_scaled_f(float x, float k_log2e) {      // 1 / (1 + exp(-k x)), k_log2e = k * log2(e)
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-k_log2e * x));
}
__device__ __forceinline__ float quick_gelu_f(float x) { return x * sigmoid_scaled_f(x, 2.4554669595930157f); }     // 1.702 * log2(e)
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_scaled_f(x, 1.4426950408889634f); }
// the activations a ProjectorConfig.hidden_act may name besides the two above (act = ACT_SILU / ACT_RELU / ACT_GELU_TANH, kernels.h); one
// uniform branch per element behind the two common cases, never on the ViT / Llama path
__device__ __forceinline__ float act_other_f(float x, int act) {
    if (act == 4) return silu_f(x);
    if (act == 5) return fmaxf(x, 0.f);
    return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}

// PAIRED k-slot helpers: position of element k (0..31) inside a fragment lane group.
__device__ __host__ __forceinline__ int paired_g(int k) { return (k & 15) >> 2; }
__device__ __host__ __forceinline__ int paired_j(int k) { return (k & 3) + ((k >> 4) << 2); }

// ---------------------------------------------------------------------------------------------
// Paged fragment stores (K fragments / V^T fragments), shared by the ViT (one "page" per frame)
// and the LLM KV cache (page table).  A page holds `page_tokens` tokens of ALL heads of one layer:
//   K part: [head][tok16 in page][blk = hd_pad/32][FRAG]           (rows = tokens, k = d PAIRED)
//   V part: [head][d16 = hd/16][tok32 in page][FRAG]               (rows = d, k = tokens PAIRED)
// ---------------------------------------------------------------------------------------------
struct KvLayout {
    half_t* base;            // layer base
    const int32_t* page_table;   // [seq][max_pages] or nullptr (page id = seq)
    int max_pages;           // row length of page_table
    int page_tokens;         // LLM cache: 64 (aur_create); ViT scratch: the padded token count, a multiple of 32
    int heads;
    int kblk;                // d-blocks of 32 per head in K fragments (hd_pad / 32)
    int vd16;                // d16 tiles per head in V fragments (hd / 16)
    int64_t page_halves;     // halves per page (K part + V part)
    int64_t v_off;           // offset of the V part inside a page (halves)
};

__device__ __forceinline__ half_t* kv_page(const KvLayout& L, int seq, int tok) {
    const int pi = tok / L.page_tokens;
    const int64_t pid = L.page_table ? (int64_t)L.page_table[(int64_t)seq * L.max_pages + pi] : (int64_t)seq;
    return L.base + pid * L.page_halves;
}
// address (halves) of K fragment (head, tok16-in-page, blk) inside a page
__device__ __forceinline__ int64_t kfrag_off(const KvLayout& L, int head, int t16, int blk) {
    return (((int64_t)head * (L.page_tokens >> 4) + t16) * L.kblk + blk) * AUR_FRAG_HALVES;
}
__device__ __forceinline__ int64_t vfrag_off(const KvLayout& L, int head, int d16, int t32) {
    return L.v_off + (((int64_t)head * L.vd16 + d16) * (L.page_tokens >> 5) + t32) * AUR_FRAG_HALVES;
}

#define HIP_CHECK_RET(expr)                                                     \
    do {                                                                        \
        hipError_t _e = (expr);                                                 \
        if (_e != hipSuccess) return aur_fail(ctx, AUR_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)
