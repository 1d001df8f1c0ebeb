// Internal kernel-launcher interface (C++ side, not part of the C ABI).
#pragma once
#include "common.h"

enum { EPI_ROW = 0, EPI_QKV = 1 };
enum { ACT_NONE = 0, ACT_QUICK_GELU = 1, ACT_GELU = 2, ACT_SILU_MUL = 3, ACT_SILU = 4, ACT_RELU = 5, ACT_GELU_TANH = 6 };   // 1, 2, 4, 5, 6 = AUR_ACT_*

struct GemmArgs {
    const half_t* A;        // [M, K] row-major activations
    int lda;
    const half_t* W;        // FRAG-packed weights [Npad/16][K/32][512]
    const float* bias;      // [Npad] fp32 or nullptr
    int M, Npad, K;         // Npad % 128 == 0, K % 64 == 0
    // ---- EPI_ROW
    half_t* C;              // row-major output
    int ldc;
    const half_t* resid;    // optional residual [M, ldr] (may alias C)
    int ldr;
    const int32_t* out_rows;   // optional output row map (m -> row of C, <0 = drop)
    int act;
    int n_real;             // columns actually stored (ACT_SILU_MUL stores n_real/2)
    // ---- EPI_QKV
    int rows_per_seq;       // rows of A per sequence / frame (multiple of 32)
    int q_cols, k_cols;     // padded column counts of the Q and K regions (multiples of 64)
    int hd;                 // true head dim (V region head width; rope table row = hd/2)
    half_t* Qf;             // Q fragments [seq][head][tok16][kblk][FRAG]
    KvLayout kv;            // K fragments / V^T fragments destination
    const float2* rope;     // [pos][hd/2] (cos, sin) or nullptr
    int pos0;               // position of row 0 of each sequence (multiple of 32)
    int seq0;               // first sequence's row in the page table
    // ---- per-ctx tuning knobs (aur_set_option; filled by the engine at every launch - nothing is process-global)
    int gemm_mode;          // 0: 128x128 kernel only, 1: auto, 2: force 256x256 when Npad % 256 == 0
    int max_wgs;            // 256x256 kernel: > 0 = at most this many persistent workgroups (= CUs); 0 = one per CU
    int tail_split;         // 1 = a mostly idle last round of the 256x256 kernel is replaced by a 128x128 launch over the bottom rows
    int nt_out;             // 1 = the output of this launch is larger than the L2s together: written with non-temporal stores (ctx_gemm)
    int tag;                // GT_*: which projection of the path this is.  No effect on the arithmetic: the 256x256 kernel is
                            // instantiated once per tag so that profiler traces (rocprofv3 groups by kernel NAME; the persistent
                            // grid is the same for every shape) separate the shapes - profiles/*_kernel_stats.txt, *_pmc.json
};
enum { GT_OTHER = 0, GT_VIT_QKV, GT_VIT_OUT, GT_VIT_FC1, GT_VIT_FC2, GT_LLM_QKV, GT_LLM_O, GT_LLM_GATEUP, GT_LLM_DOWN, GT_COUNT };

hipError_t gemm_init();
hipError_t attn_init();
hipError_t tome_init();
hipError_t skinny_init();
hipError_t launch_gemm(const GemmArgs& a, int epi, hipStream_t s);
// gemm256.hip: 256x256x64 staggered two-group kernel for large shapes (auto-selected by launch_gemm)
hipError_t gemm256_init();
bool gemm256_eligible(const GemmArgs& a);
int gemm256_grid_cap();           // persistent workgroups of the 256x256 kernel when max_wgs == 0
hipError_t launch_gemm256(const GemmArgs& a, int epi, hipStream_t s);
hipError_t launch_pack_weight(const half_t* w, int n_src, int k_src, int ld_src, const int32_t* row_map, int npad,
                              int kpad, half_t* out, hipStream_t s);

// ---- norm.hip
hipError_t launch_layernorm(const half_t* x, int ldx, const float* w, const float* b, float eps, int rows, int d,
                            half_t* y, int ldy, hipStream_t s);
hipError_t launch_rmsnorm(const half_t* x, int ldx, const float* w, float eps, int rows, int d, half_t* y, int ldy,
                          hipStream_t s);

// ---- attn.hip (prefill / ViT flash attention over fragments)
struct AttnArgs {
    const half_t* Qf;       // [seq][head][tok16][kblk][FRAG]
    KvLayout kv;
    int seq0;               // page-table row of sequence 0
    int nseq, heads;
    int rows_per_seq;       // padded query rows per sequence (multiple of 32)
    int t;                  // valid keys / queries per sequence
    int causal;
    float scale;            // softmax scale (hd^-0.5)
    half_t* O;              // [nseq*rows_per_seq, ldo] row-major, head h at columns h*hd
    int ldo;
    int hd;
    int nqb;                // filled by launch_attention: query blocks of 128 rows per sequence
    int q_pos0;             // position of query row 0 of every sequence among its keys (causal mask): 0 = the Q buffer starts at the
                            // sequence's first token; > 0 = it holds only a tail of the sequence (the pruned last prefill layer)
};
hipError_t launch_attention(const AttnArgs& a, hipStream_t s);

// ---- tome.hip
struct TomeArgs {
    int frames, t, t_pad, r, c;      // c = metric channels (head_dim); r already clamped, > 0
    int d;                           // hidden size
    const float* metric;             // [frames][t][c] fp32
    const half_t* x;                 // [frames][t_pad][d]
    const float* size;               // [frames][t_pad] (nullptr = ones)
    int t_out_pad;                   // padded rows per frame of the output
    half_t* x_out;                   // [frames][t_out_pad][d]
    float* size_out;                 // [frames][t_out_pad]
    // scratch / optional parity outputs (device): node_max [frames][ta], node_idx [frames][ta],
    // unm [frames][ta - r], src [frames][r], dst [frames][r]
    float* node_max;
    int32_t* node_idx;
    int32_t* unm;
    int32_t* src;
    int32_t* dst;
    float* mhat;                     // scratch: the normalised metric as fp32 MFMA operand blocks, tome_mfrag_floats(t, c) per frame
    const KvLayout* kv;              // non-null: metric = mean over heads of these K fragments (aurora.py:639), `metric` unused
    float* metric_out;               // with kv: optional copy of the un-normalised metric [frames][t][c]
    // optional LayerNorm of the merged rows (aurora.py:750), written beside x_out from the merge launch: y = LN(x_out) as
    // norm_kernel<false> computes it (bitwise), all frames * t_out_pad rows.  ln_out == nullptr: none.  Needs d % 8 == 0.
    const float* ln_w;
    const float* ln_b;
    float ln_eps;
    half_t* ln_out;                  // [frames][t_out_pad][d]
};
hipError_t launch_tome_step(const TomeArgs& a, hipStream_t s);
int64_t tome_mfrag_floats(int t, int c);          // floats of TomeArgs.mhat per frame
// metric[f][tok][d] = mean over heads of K (read back from PAIRED K fragments of the ViT "pages")
hipError_t launch_tome_metric(const KvLayout& kv, int frames, int t, int hd, float* metric, hipStream_t s);

// ---- vit.hip
hipError_t launch_im2col(const half_t* pixels, int frames, int chans, int height, int width, int patch, int kpad, half_t* out,
                         hipStream_t s);
hipError_t launch_vit_assemble(const half_t* patches, const half_t* cls, const half_t* pos, const float* ln_w,
                               const float* ln_b, float eps, int frames, int npatch, int d, int t_pad, half_t* x,
                               hipStream_t s);
hipError_t launch_strip_cls(const half_t* x, int frames, int t, int t_pad, int d, half_t* out, hipStream_t s);
hipError_t launch_pad_rows(const half_t* src, const float* size_src, int frames, int t, int t_pad, int d, half_t* dst,
                           float* size_dst, hipStream_t s);
hipError_t launch_pos_interp(const half_t* pos, int n, int gh, int gw, int d, half_t* out, hipStream_t s);
hipError_t launch_unpad_rows(const half_t* src, const float* size_src, int frames, int t, int t_pad, int d, half_t* dst,
                             float* size_dst, hipStream_t s);
hipError_t launch_gather_rows(const half_t* src, int ld_src, const int32_t* rows, int nrows, int d, half_t* dst,
                              int ld_dst, hipStream_t s);
hipError_t launch_embed_rows(const half_t* table, int d, const int32_t* ids, const int32_t* dst_rows, int n,
                             half_t* dst, int ld_dst, hipStream_t s);

// ---- decode.hip
struct SkinnyArgs {
    const half_t* xf;       // activations in x-fragment form [ceil(B/16)][K/32][64][8], lane = g*16 + (b % 16)
    const half_t* W;        // FRAG-packed [Npad/16][K/32][512]
    int B, Npad, K, n_real;
    int mode;               // SK_ROW, SK_LOGITS, SK_SILU_MUL, SK_QKV
    int b_lo, b_hi;         // only batch columns b_lo <= b < b_hi are stored
    half_t* xres;           // SK_ROW: residual stream in x-fragment form (K32 = n_real/32), updated in place: x += y
    unsigned long long* ssq_out;        // SK_ROW: sum(x_new^2) per row [AUR_SSQ_SLOTS][AUR_MAX_BATCH], 2^-28 fixed point, integer atomics
    const unsigned long long* ssq_in;   // folded RMSNorm: sum(x^2) of the input rows -> acc *= rsqrt(ssq/K + eps); nullptr = none
    unsigned long long* ssq_zero;       // accumulator (all slots) to reset for a later kernel; nullptr = none
    float norm_eps;
    half_t* out_f;          // SK_SILU_MUL: x-fragment form with K32 = out_k32 (input of the down projection)
    int out_k32;
    float* out32;           // SK_LOGITS: [B, n_real] fp32
    // SK_QKV
    int q_cols, k_cols, hd;
    half_t* qbuf;           // [B][heads][kblk][4][8]   (PAIRED-d pieces)
    KvLayout kv;
    const float2* rope;
    const int32_t* pos;     // [B] device: position of the new token (= tokens already cached)
    const int32_t* seq_ids; // [B] page-table rows (nullptr = identity)
    // structure: 0 = x fragments per wave from L2 (engines of <= 32 decode slots), 1 = x through LDS once per workgroup
    // (decode.hip "skinny GEMM, x through LDS").  Chosen per ENGINE (capacity), never per live batch: batch invariance.
    int variant;
    float* part;            // variant 1, SK_ROW: split-K partials [4][Npad/16][ceil(B/16)][64][4] fp32 (nullptr: keep structure 0)
    int* row_cnt;           // variant 1, SK_ROW: arrival counters [Npad/16], zero between launches: the workgroup that completes a tile's 4
                            // partials sums them (fixed order) and runs the residual epilogue IN the projection kernel.  nullptr = a second
                            // launch does it (skinny_row_reduce_kernel; bitwise the same result)
    int gu_ks;              // variant 1, SK_SILU_MUL: k phases per tile, 2 (default) or 1 (engines of > 64 slots: the half grid needs it).
                            // A constant of the ENGINE: it fixes the summation order
    int half_grid;          // variant 1, > 64 rows, SK_QKV / SK_SILU_MUL: 1 = half as many workgroups with twice the tiles each, for
                            // launches into a stream that owns half of the CUs (bitwise the full-grid result)
};
enum { SK_ROW = 0, SK_LOGITS = 1, SK_SILU_MUL = 2, SK_QKV = 3 };
hipError_t launch_skinny(const SkinnyArgs& a, hipStream_t s);
// xf[b0 + row] = RMSNorm(x[row]) (w != nullptr) or x[row], in x-fragment form; x rows are ldx halves apart
hipError_t launch_xfrag_norm(const half_t* x, int64_t ldx, const float* w, float eps, int rows, int d, int b0, half_t* xf,
                             unsigned long long* ssq_out, hipStream_t s, int waves = 16);
hipError_t launch_xfrag_pack(const half_t* x, int64_t ldx, int rows, int d, half_t* xf, hipStream_t s);

struct DecAttnArgs {
    const half_t* qbuf;     // [B][heads][kblk][4][8]
    KvLayout kv;
    const int32_t* pos;     // [B]: keys valid = pos[b] + 1 (the new token's K/V is already written)
    const int32_t* seq_ids;
    int B, heads, hd;
    int nsplit, pages_per_split;
    float scale;
    float* part_o;          // [B][heads][nsplit][hd]
    float* part_ml;         // [B][heads][nsplit][2]
    half_t* out_f;          // [B, heads*hd] in x-fragment form (K32 = out_k32 = heads*hd/32)
    int out_k32;
    int local_splits;       // > 1 (== nsplit, 2 or 4): the splits of a (sequence, head) are the waves of ONE workgroup, joined through LDS (no combine launch)
};
hipError_t launch_decode_attention(const DecAttnArgs& a, hipStream_t s);            // main + combine
hipError_t launch_decode_attention_main(const DecAttnArgs& a, hipStream_t s);
hipError_t launch_decode_attention_combine(const DecAttnArgs& a, hipStream_t s);

hipError_t launch_argmax_advance(const float* logits, int b0, int nb, int vocab, const half_t* embed, int d, int eos_id,
                                 int max_new, int32_t* out_ids, int32_t* out_len, int32_t* finished, int32_t* pos,
                                 half_t* xf, unsigned long long* ssq, int advance_pos, int set_pos, hipStream_t s);
