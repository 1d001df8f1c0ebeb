// libaurora_hip.so - context, workspace carving, pipelines and the extern "C" boundary (include/aurora_hip.h).
//
// Pipelines restate the reference's call sequence for the AuroraCap inference path with one enqueue per
// kernel on the caller's stream:
//   aur_vit_encode      <- AuroraEncoder.forward + AuroraCLIPEncoder loop   (aurora.py:883-904, 772-860, 713-759)
//   aur_project_splice  <- projector + prepare_inputs_labels_for_multimodal (aurora.py:254-258, utils.py:138-295)
//   aur_llm_prefill / aur_llm_decode <- LlamaForCausalLM.generate greedy    (inference.py:89-96)
#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/aurora_hip.h"
#include "kernels.h"

struct Tensor {
    const void* p;
    int64_t bytes;
};

struct VitLayerW {
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *qkv_b, *out_b, *fc1_b, *fc2_b;
    const half_t *qkv_w, *out_w, *fc1_w, *fc2_w;
};
struct LlmLayerW {
    const float *ln1_w, *ln2_w;
    const half_t *qkv_w, *o_w, *gateup_w, *down_w;
};

struct StageTimer {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    double ms = 0;
    int64_t launches = 0;
    bool open = false;
    uint64_t range_id = 0;       // roctx range of the stage (AURORA_ROCTX=1)
    bool range_open = false;
};

struct aur_ctx {
    aur_config cfg;
    std::string err;
    std::unordered_map<std::string, Tensor> tensors;
    char* ws = nullptr;
    int64_t ws_bytes = 0;
    char* kvpool = nullptr;
    int64_t kv_bytes = 0;
    bool finalized = false;
    // derived: vision
    int v_hd, v_hd_pad, v_kblk, v_vd16, v_qcols, v_qkv_npad, v_dpad, v_mlp_pad, v_npatch, v_t0, v_t0pad, v_kpad, v_native_npatch;
    // derived: llm
    int l_hd, l_kblk, l_vd16, l_qkv_npad, l_gu_npad, l_dpad, l_vocab_pad, l_max_pages, l_ctx_pad;
    int64_t l_page_halves, l_layer_halves;
    int kv_seqs = 0;                 // KV sequences per bank (max_batch decode slots + cfg.spare_slots)
    // weights
    std::vector<VitLayerW> vl;
    std::vector<LlmLayerW> ll;
    const half_t *v_patch_w, *v_cls, *v_pos, *l_embed, *l_head_w;
    const float *v_preln_w, *v_preln_b, *l_norm_w;
    std::vector<const half_t*> p_w;      // projector Linear layers (modeling_projector.py:20-33): [proj_depth] packed weights
    std::vector<const float*> p_b;       //                                                        and biases (zeros for bias = False)
    int p_depth = 2, p_act = ACT_GELU;
    // workspace regions (vision)
    half_t *w_col, *w_patch, *w_xa, *w_xb, *w_xn, *w_qf, *w_kv, *w_attn, *w_h;
    float *w_metric, *w_mhat, *w_nmax, *w_sza, *w_szb;
    int32_t *w_nidx, *w_unm, *w_src, *w_dst;
    // workspace regions (llm)
    half_t *l_xn, *l_qf, *l_attn, *l_h, *l_p1, *d_x, *d_q, *d_attn, *d_h, *d_scr;
    unsigned long long *s_ssq_mlp, *s_ssq_attn;
    float *d_logits, *d_part_o, *d_part_ml;
    float2* l_rope;
    int32_t *s_pos, *s_ids, *s_len, *s_fin, *s_ptab;
    int32_t* ptab_rw() { return const_cast<int32_t*>(ptab_cur); }
    // generation state
    int batch = 0, max_new = 0, eos = -1, nsplit = 1, pps = 1;
    int last_prefill_len = 0, mb_nseq = 1;                         // tuning knobs (aur_set_option)
    int decode_half = 0;             // 1: the next aur_llm_decode calls target a stream that owns half of the CUs (own hipGraph)
    int gemm_mode = 1, gemm_max_wgs = 0, gemm_tail_split = 1, prune_last = 1, gemm_nt_out = -1;                           // GEMM knobs: per ctx, copied into GemmArgs at every launch
    int skinny_variant = 0, row_split_min_k = 8192;                             // decode projections: x through LDS (engines of > 32 slots)
    int skinny_variant_wide = 0;                                                // the two WIDE projections (QKV, gate/up): x through LDS at every capacity
    float* d_part_row = nullptr;                                                // split-K partials of the SK_ROW projection
    int* d_row_cnt = nullptr;                                                   // its arrival counters (zero between launches)
    int attn_local = 1;                                                         // 1: workgroup-local join of the decode attention's splits where it applies (mk_dec_attn); 0 = always decode_attn_combine_kernel (A/B and test)
    int tome_fused_ln = 1;                                                      // 1: LayerNorm 2 of a merging ViT layer comes out of the ToMe merge launch (bitwise norm_kernel's result; 0 = its own launch, A/B and test only)
    int fused_reduce = 0;                                                       // 1: the split-K reduce runs inside the projection kernel
    hipGraphExec_t graph = nullptr, graph_h = nullptr;      // decode step: full grid / half grid (decode_half)
    int graph_batch = 0;
    // Generation banks: double-buffered per-batch state (KV slots, residual stream, sum(x^2), logits, outputs, graph) so
    // that one batch can decode on one stream while the next batch's ViT + prefill run on another.  The members above
    // (d_x, s_*, d_logits, batch, max_new, eos, graph, ...) always alias the CURRENT bank (aur_select_bank).
    struct Bank {
        half_t* d_x;
        unsigned long long *s_ssq_mlp, *s_ssq_attn;
        float* d_logits;
        int32_t *s_pos, *s_ids, *s_len, *s_fin;
        const int32_t* ptab;
        int batch = 0, max_new = 0, eos = -1;
        hipGraphExec_t graph = nullptr, graph_h = nullptr;
        int graph_batch = 0;
    } banks[2];
    int nbanks = 1, cur_bank = 0;
    const int32_t* ptab_cur = nullptr;
    // caller-captured hipGraphs (aur_graph_begin / aur_graph_end / aur_graph_launch): a front end per shape bucket
    struct UserGraph {
        hipGraphExec_t exec = nullptr;
        int64_t nodes = 0;
    };
    std::vector<UserGraph> ugraphs;                      // handle = index + 1; destroyed slots keep exec == nullptr
    bool capturing = false;
    hipStream_t cap_stream = nullptr;
    // in-loop timing of the dominant decode kernel: option "decode_stamp_layer" brackets that layer's attention launch with two
    // one-thread launches that store the device's constant-rate clock (part of the captured step, so the timed loop itself is measured)
    int stamp_layer = -1;
    unsigned long long* d_stamp = nullptr;               // [0] steps stamped so far, [8 + 2 * (step % AUR_STAMP_RING) + {0, 1}] begin / end
    // profiling
    bool prof = false;
    std::unordered_map<std::string, StageTimer> timers;
    struct KernelEvents {                                // HIP-event pairs around every launch of one decode kernel
        std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
        size_t used = 0;
        double ms = 0;
        int64_t n = 0;
    } kev[4];                                            // 0: decode attention (main kernel), 1: gate/up GEMV, 2: ToMe step (3 launches), 3: lm_head + argmax
};

static std::string g_create_err;

static int aur_fail(aur_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_create_err = buf;
    return code;
}
#define CK(expr)                                                                                       \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) return aur_fail(ctx, AUR_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// Entry points that synchronise (or destroy graphs) refuse to run inside aur_graph_begin / aur_graph_end BEFORE touching the runtime:
// a synchronising HIP call on a capturing stream invalidates the capture and (ROCm 7.2) leaves the stream refusing later launches
#define NO_CAPTURE(who)                                                                                                     \
    do {                                                                                                                    \
        if (ctx->capturing) return aur_fail(ctx, AUR_ERR_STATE, "%s inside aur_graph_begin / aur_graph_end: it synchronises", who); \
    } while (0)

static inline int rup(int x, int m) { return (x + m - 1) / m * m; }
// every GEMM of a ctx goes through here: the ctx's tuning knobs travel in the argument block (no process-global state)
static hipError_t ctx_gemm(const aur_ctx* ctx, GemmArgs& a, int epi, hipStream_t s);
static inline int64_t rup64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

static hipError_t ctx_gemm(const aur_ctx* ctx, GemmArgs& a, int epi, hipStream_t s) {
    a.gemm_mode = ctx->gemm_mode;
    a.max_wgs = ctx->gemm_max_wgs;
    a.tail_split = ctx->gemm_tail_split;
    // Output policy.  A round of 256x256 tiles writes 256 x 128 KiB in one burst (every CU reaches its epilogue together) - the capacity
    // of the eight L2s; with the default policy that burst evicts the operand panels the next tiles share through L2.  Outputs the L2s
    // could not hold for the consumer anyway go out non-temporal (round 4: ViT fc1 +8 %, prefill gate/up +3 %, QKV +2-3 %); a single
    // clip's prefill (17-50 MB per projection, read back at once by the next launch) keeps the default (+1 % on its layer stack).
    const int64_t out_cols = epi == EPI_QKV ? (int64_t)a.q_cols + 2 * a.k_cols : (a.act == ACT_SILU_MUL ? a.n_real / 2 : a.n_real);
    a.nt_out = ctx->gemm_nt_out >= 0 ? ctx->gemm_nt_out : ((int64_t)a.M * out_cols * 2 > ((int64_t)32 << 20));
    return launch_gemm(a, epi, s);
}

// ------------------------------------------------------------------------------------------ schedule
extern "C" int32_t aur_tome_r(int32_t height, int32_t width, int32_t patch, double ratio, int32_t layers) {
    // aurora.py:895 - same operation order in doubles: W*H / patch**2 * (1 - ratio) / L, then int()
    const double v = (double)width * (double)height / (double)(patch * patch) * (1.0 - ratio) / (double)layers;
    return (int32_t)v;
}
extern "C" int32_t aur_tokens_at_layer(int32_t t0, int32_t r, int32_t layer) {
    int t = t0;
    for (int l = 0; l < layer; ++l) {
        int rl = r < (t - 1) / 2 ? r : (t - 1) / 2;      // tome.py:45
        if (rl > 0) t -= rl;
    }
    return t;
}
extern "C" const char* aur_version(void) { return "aurora_hip 0.1 (gfx950)"; }
extern "C" const char* aur_last_error(const aur_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

// ------------------------------------------------------------------------------------------ layout
struct Carver {
    int64_t off = 0;
    char* base;
    explicit Carver(char* b) : base(b) {}
    template <typename T>
    T* take(int64_t n) {
        T* p = base ? (T*)(base + off) : nullptr;
        off = rup64(off + n * (int64_t)sizeof(T), 256);
        return p;
    }
};

static void derive(aur_ctx* c) {
    const aur_config& g = c->cfg;
    c->v_hd = g.vit_hidden / g.vit_heads;
    c->v_hd_pad = rup(c->v_hd, 32);
    c->v_kblk = c->v_hd_pad / 32;
    c->v_vd16 = c->v_hd / 16;
    c->v_qcols = rup(g.vit_heads * c->v_hd_pad, 64);
    c->v_qkv_npad = rup(2 * c->v_qcols + g.vit_heads * c->v_hd, 256);
    c->v_dpad = rup(g.vit_hidden, 256);
    c->v_mlp_pad = rup(g.vit_mlp, 256);
    const int gw = g.vit_image / g.vit_patch;
    c->v_npatch = gw * gw;
    c->v_t0 = c->v_npatch + 1;
    c->v_t0pad = rup(c->v_t0, 32);
    const int gn = (g.vit_native_image > 0 ? g.vit_native_image : g.vit_image) / g.vit_patch;    // grid of the checkpoint's position table
    c->v_native_npatch = gn * gn;
    c->v_kpad = rup(g.vit_channels * g.vit_patch * g.vit_patch, 64);
    c->l_hd = g.llm_hidden / g.llm_heads;
    c->l_kblk = c->l_hd / 32;
    c->l_vd16 = c->l_hd / 16;
    c->l_qkv_npad = rup(3 * g.llm_hidden, 256);
    c->l_gu_npad = rup(2 * g.llm_mlp, 256);
    c->l_dpad = rup(g.llm_hidden, 256);
    c->l_vocab_pad = rup(g.llm_vocab, 256);
    c->l_max_pages = (g.max_ctx + g.page_tokens - 1) / g.page_tokens;
    c->l_ctx_pad = c->l_max_pages * g.page_tokens;
    c->l_page_halves = (int64_t)2 * g.llm_heads * g.page_tokens * c->l_hd;
    c->nbanks = g.num_banks == 2 ? 2 : 1;
    c->skinny_variant = g.max_batch > 32 ? 1 : 0;        // a function of the engine's capacity, never of the live batch
    // QKV and gate/up (N = 12288 / 22016: several tiles per CU share x) gain from x-through-LDS at EVERY capacity - 22.3-22.9 -> 20.4-20.8 us
    // and 33.3-33.8 -> 31.6-31.9 us at 1, 4 and 8 rows (round 4, tools/microbench.py --batch 1 / 4 / 8) - while o / down do not (9.1 /
    // 19.0-19.5 us either way or worse).  configs[3]'s per-GPU engine (8 slots): 5.80 -> 6.01 captions/s with this and the GEMM changes.
    c->skinny_variant_wide = 1;
    c->kv_seqs = g.max_batch + (g.spare_slots > 0 ? g.spare_slots : 0);     // KV sequences per bank: decode slots + spare prefill targets
    c->l_layer_halves = c->l_page_halves * c->l_max_pages * c->kv_seqs * c->nbanks;
    // decode attention: one wave per (sequence, head, split).  Enough splits to put ~512 waves on the GPU for small batches,
    // as few as possible (1) once the batch supplies them - every extra split re-reads q, writes a partial and lengthens
    // the combine (measured at a 2.2k context: B=64 344 us with 2 splits vs 362 with 10; B=1 15.1 us with 13 vs 24.8 with 38).
    // A function of the engine's capacity only, never of the live batch: results stay batch-invariant per engine.
    int want = (512 + g.llm_heads * g.max_batch - 1) / (g.llm_heads * g.max_batch);
    want = want < 1 ? 1 : (want > 16 ? 16 : want);          // 1 split (>= 512 sequence-heads): 333.6 us vs 333.4 us with 2 at 64 x 32, and no combine
    c->pps = (c->l_max_pages + want - 1) / want;
    if (c->pps < 1) c->pps = 1;
    c->nsplit = (c->l_max_pages + c->pps - 1) / c->pps;
}

static int64_t vit_page_halves(const aur_ctx* c, int t_pad) { return (int64_t)c->cfg.vit_heads * t_pad * (c->v_hd_pad + c->v_hd); }

static int64_t carve(aur_ctx* c, char* base) {
    const aur_config& g = c->cfg;
    Carver k(base);
    const int64_t F = g.max_frames, TP = c->v_t0pad, D = g.vit_hidden;
    c->w_col = k.take<half_t>(F * c->v_npatch * c->v_kpad);
    c->w_patch = k.take<half_t>(F * c->v_npatch * D);
    c->w_xa = k.take<half_t>(F * TP * D);
    c->w_xb = k.take<half_t>(F * TP * D);
    c->w_xn = k.take<half_t>(F * TP * D);
    c->w_qf = k.take<half_t>(F * g.vit_heads * TP * c->v_hd_pad);
    c->w_kv = k.take<half_t>(F * vit_page_halves(c, (int)TP));
    c->w_attn = k.take<half_t>(F * TP * D);
    c->w_h = k.take<half_t>(F * TP * g.vit_mlp);
    c->w_metric = k.take<float>(F * c->v_t0 * c->v_hd);
    c->w_mhat = k.take<float>(F * tome_mfrag_floats(c->v_t0, c->v_hd));          // fp32 MFMA operand blocks (tome.hip)
    const int64_t ta = (c->v_t0 + 1) / 2;
    c->w_nmax = k.take<float>(F * ta);
    c->w_nidx = k.take<int32_t>(F * ta);
    c->w_unm = k.take<int32_t>(F * ta);
    c->w_src = k.take<int32_t>(F * ta);
    c->w_dst = k.take<int32_t>(F * ta);
    c->w_sza = k.take<float>(F * TP);
    c->w_szb = k.take<float>(F * TP);
    // llm
    const int64_t B = g.max_batch, d = g.llm_hidden;
    const int64_t LP = (int64_t)c->l_ctx_pad * B;          // batched prefill: up to max_batch sequences in one pass
    c->l_xn = k.take<half_t>(LP * d);
    c->l_qf = k.take<half_t>(LP * d);
    c->l_attn = k.take<half_t>(LP * d);
    c->l_h = k.take<half_t>(LP * g.llm_mlp);
    c->l_p1 = k.take<half_t>(LP * d);
    c->l_rope = k.take<float2>((int64_t)c->l_ctx_pad * (c->l_hd / 2));
    const int64_t Bp = rup((int)B, 16);                    // x-fragment buffers hold whole groups of 16 rows
    for (int bk = 0; bk < 2; ++bk) {
        aur_ctx::Bank& K = c->banks[bk];
        K.d_x = k.take<half_t>(Bp * d);                    // residual stream    (x-fragment form)
        K.s_ssq_mlp = k.take<unsigned long long>(AUR_SSQ_SLOTS * AUR_MAX_BATCH);      // sum(x^2) per row after the MLP / embedding (2^-28 fixed point)
        K.s_ssq_attn = k.take<unsigned long long>(AUR_SSQ_SLOTS * AUR_MAX_BATCH);     // sum(x^2) per row after the attention residual
        K.d_logits = k.take<float>(B * g.llm_vocab);
        K.s_pos = k.take<int32_t>(B);
        K.s_ids = k.take<int32_t>(B * g.max_new_tokens);
        K.s_len = k.take<int32_t>(B);
        K.s_fin = k.take<int32_t>(B);
    }
    c->d_q = k.take<half_t>(B * d);
    c->d_attn = k.take<half_t>(Bp * d);                    // attention output  (x-fragment form)
    c->d_h = k.take<half_t>(Bp * g.llm_mlp);               // SiLU(gate)*up     (x-fragment form)
    c->d_scr = k.take<half_t>(AUR_MAX_BATCH * 16384);                 // scratch x-fragments for aur_linear_skinny
    c->d_part_row = k.take<float>((int64_t)4 * (c->l_dpad / 16) * (Bp / 16) * 256);          // [4 k splits][tiles][column groups][64 lanes][4]
    c->d_row_cnt = k.take<int>(c->l_dpad / 16);
    c->d_stamp = k.take<unsigned long long>(8 + 2 * AUR_STAMP_RING);
    c->d_part_o = k.take<float>(B * g.llm_heads * c->l_max_pages * c->l_hd);     // room for pages_per_split = 1
    c->d_part_ml = k.take<float>(B * g.llm_heads * c->l_max_pages * 2);
    c->s_ptab = k.take<int32_t>(2 * (int64_t)c->kv_seqs * c->l_max_pages);
    for (int bk = 0; bk < 2; ++bk) c->banks[bk].ptab = c->s_ptab ? c->s_ptab + (int64_t)bk * c->kv_seqs * c->l_max_pages : nullptr;
    {   // alias the current bank
        const aur_ctx::Bank& K = c->banks[c->cur_bank];
        c->d_x = K.d_x; c->s_ssq_mlp = K.s_ssq_mlp; c->s_ssq_attn = K.s_ssq_attn; c->d_logits = K.d_logits;
        c->s_pos = K.s_pos; c->s_ids = K.s_ids; c->s_len = K.s_len; c->s_fin = K.s_fin; c->ptab_cur = K.ptab;
    }
    return k.off;
}

// ------------------------------------------------------------------------------------------ lifecycle
extern "C" int aur_create(const aur_config* cfg, aur_ctx** out) {
    if (!cfg || !out) return aur_fail(nullptr, AUR_ERR_ARG, "aur_create: null argument");
    const aur_config& g = *cfg;
    if (g.vit_hidden % 64 || g.vit_mlp % 64 || g.vit_hidden % g.vit_heads || (g.vit_hidden / g.vit_heads) % 16)
        return aur_fail(nullptr, AUR_ERR_ARG, "vit dims: hidden/mlp must be multiples of 64, head_dim a multiple of 16");
    if (g.llm_hidden % 128 || g.llm_mlp % 128 || g.llm_hidden % g.llm_heads || (g.llm_hidden / g.llm_heads) % 32)
        return aur_fail(nullptr, AUR_ERR_ARG, "llm dims: hidden/mlp must be multiples of 128, head_dim a multiple of 32");
    if (g.max_batch < 1 || g.max_batch > AUR_MAX_BATCH) return aur_fail(nullptr, AUR_ERR_ARG, "max_batch must be in [1, %d]", AUR_MAX_BATCH);
    if (g.page_tokens != 64) return aur_fail(nullptr, AUR_ERR_ARG, "page_tokens must be 64 (the decode attention's page pipeline is written for 64-token pages)");
    if (g.vit_image % g.vit_patch) return aur_fail(nullptr, AUR_ERR_ARG, "image size must be a multiple of the patch size");
    {   // the ToMe step's limits (tome.hip launch_tome_step), checked here so that an out-of-range ViT fails at create time, not at the first encode
        const int64_t t0 = (int64_t)(g.vit_image / g.vit_patch) * (g.vit_image / g.vit_patch) + 1;
        if (g.vit_hidden > 2048) return aur_fail(nullptr, AUR_ERR_ARG, "vit_hidden %d exceeds 2048 (token-merge kernel: one row of at most 2048 features per wave)", g.vit_hidden);
        if (g.vit_hidden / g.vit_heads > 128) return aur_fail(nullptr, AUR_ERR_ARG, "vit head_dim %d exceeds 128 (token-merge metric: at most 128 channels)", g.vit_hidden / g.vit_heads);
        if ((t0 + 1) / 2 > 4096) return aur_fail(nullptr, AUR_ERR_ARG, "vit_image %d / patch %d gives %lld tokens per frame; the token-merge select kernel ranks at most 4096 = ceil(t / 2) of them", g.vit_image, g.vit_patch, (long long)t0);
    }
    if (g.vit_native_image < 0 || g.vit_native_image > g.vit_image || g.vit_native_image % g.vit_patch)
        return aur_fail(nullptr, AUR_ERR_ARG, "vit_native_image must be 0 or a multiple of the patch size <= vit_image");
    {
        hipError_t e;
        if ((e = gemm_init()) != hipSuccess || (e = attn_init()) != hipSuccess || (e = tome_init()) != hipSuccess ||
            (e = skinny_init()) != hipSuccess || (e = gemm256_init()) != hipSuccess)
            return aur_fail(nullptr, AUR_ERR_HIP, "kernel attribute init failed (is a gfx950 GPU visible?): %s", hipGetErrorString(e));
    }
    if (g.proj_depth < 0 || g.proj_depth > 8) return aur_fail(nullptr, AUR_ERR_ARG, "proj_depth must be 0 (= 2) or 1 .. 8");
    if (g.proj_act != 0 && g.proj_act != AUR_ACT_QUICK_GELU && g.proj_act != AUR_ACT_GELU && g.proj_act != AUR_ACT_SILU && g.proj_act != AUR_ACT_RELU &&
        g.proj_act != AUR_ACT_GELU_TANH)
        return aur_fail(nullptr, AUR_ERR_UNSUPPORTED, "proj_act %d is not one of AUR_ACT_*", g.proj_act);
    aur_ctx* c = new aur_ctx();
    c->cfg = g;
    c->p_depth = g.proj_depth > 0 ? g.proj_depth : 2;
    c->p_act = g.proj_act > 0 ? g.proj_act : ACT_GELU;
    derive(c);
    *out = c;
    return AUR_OK;
}

extern "C" void aur_destroy(aur_ctx* ctx) {
    if (!ctx) return;
    ctx->banks[ctx->cur_bank].graph = ctx->graph;
    ctx->banks[ctx->cur_bank].graph_h = ctx->graph_h;
    for (auto& K : ctx->banks) {
        if (K.graph) (void)hipGraphExecDestroy(K.graph);
        if (K.graph_h) (void)hipGraphExecDestroy(K.graph_h);
    }
    for (auto& ug : ctx->ugraphs)
        if (ug.exec) (void)hipGraphExecDestroy(ug.exec);
    for (auto& kv : ctx->timers) {
        if (kv.second.e0) (void)hipEventDestroy(kv.second.e0);
        if (kv.second.e1) (void)hipEventDestroy(kv.second.e1);
    }
    for (auto& kv : ctx->kev)
        for (auto& p : kv.ev) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    delete ctx;
}

extern "C" int aur_set_tensor(aur_ctx* ctx, const char* name, const void* dev_ptr, int64_t nbytes) {
    if (!ctx || !name || !dev_ptr) return aur_fail(ctx, AUR_ERR_ARG, "aur_set_tensor: null argument");
    ctx->tensors[name] = Tensor{dev_ptr, nbytes};
    ctx->finalized = false;
    return AUR_OK;
}
extern "C" int64_t aur_workspace_bytes(const aur_ctx* ctx) {
    aur_ctx tmp = *ctx;
    return carve(&tmp, nullptr);
}
extern "C" int64_t aur_kv_pool_bytes(const aur_ctx* ctx) { return ctx->l_layer_halves * ctx->cfg.llm_layers * 2; }
extern "C" int aur_set_workspace(aur_ctx* ctx, void* p, int64_t n) {
    if (n < aur_workspace_bytes(ctx)) return aur_fail(ctx, AUR_ERR_ARG, "workspace too small: %lld < %lld", (long long)n, (long long)aur_workspace_bytes(ctx));
    ctx->ws = (char*)p;
    ctx->ws_bytes = n;
    carve(ctx, ctx->ws);
    ctx->finalized = false;
    if (ctx->d_row_cnt) CK(hipMemset(ctx->d_row_cnt, 0, (size_t)(ctx->l_dpad / 16) * 4));     // split-K arrival counters, likewise
    if (ctx->d_stamp) CK(hipMemset(ctx->d_stamp, 0, (size_t)(8 + 2 * AUR_STAMP_RING) * 8));
    return AUR_OK;
}
extern "C" int aur_set_kv_pool(aur_ctx* ctx, void* p, int64_t n) {
    if (n < aur_kv_pool_bytes(ctx)) return aur_fail(ctx, AUR_ERR_ARG, "kv pool too small");
    ctx->kvpool = (char*)p;
    ctx->kv_bytes = n;
    return AUR_OK;
}

template <typename T>
static bool get(aur_ctx* c, const std::string& name, const T** out, int64_t min_bytes, bool required = true) {
    auto it = c->tensors.find(name);
    if (it == c->tensors.end()) {
        if (required) aur_fail(c, AUR_ERR_STATE, "missing tensor '%s'", name.c_str());
        *out = nullptr;
        return !required;
    }
    if (it->second.bytes < min_bytes) {
        aur_fail(c, AUR_ERR_STATE, "tensor '%s' has %lld bytes, need %lld", name.c_str(), (long long)it->second.bytes, (long long)min_bytes);
        return false;
    }
    *out = (const T*)it->second.p;
    return true;
}

extern "C" int aur_finalize(aur_ctx* ctx, void* stream) {
    NO_CAPTURE("aur_finalize");
    if (!ctx->ws) return aur_fail(ctx, AUR_ERR_STATE, "aur_finalize: workspace not set");
    const aur_config& g = ctx->cfg;
    hipStream_t s = (hipStream_t)stream;
    const bool have_vit = ctx->tensors.count("vit.patch.w") > 0;
    const bool have_llm = ctx->tensors.count("llm.embed") > 0;
    if (have_vit) {
        const int64_t D = g.vit_hidden;
        bool ok = get(ctx, "vit.patch.w", &ctx->v_patch_w, (int64_t)ctx->v_dpad * ctx->v_kpad * 2) &&
                  get(ctx, "vit.cls", &ctx->v_cls, D * 2) && get(ctx, "vit.pos", &ctx->v_pos, (int64_t)(ctx->v_native_npatch + 1) * D * 2) &&
                  get(ctx, "vit.preln.w", &ctx->v_preln_w, D * 4) && get(ctx, "vit.preln.b", &ctx->v_preln_b, D * 4);
        if (!ok) return AUR_ERR_STATE;
        // hidden_states[-2] needs layers 0 .. L-2 only: the last layer's weights are optional
        const int nl = g.vit_layers - 1;
        ctx->vl.assign(nl > 0 ? nl : 0, VitLayerW{});
        for (int l = 0; l < nl; ++l) {
            const std::string p = "vit." + std::to_string(l) + ".";
            VitLayerW& w = ctx->vl[l];
            ok = get(ctx, p + "ln1.w", &w.ln1_w, D * 4) && get(ctx, p + "ln1.b", &w.ln1_b, D * 4) &&
                 get(ctx, p + "ln2.w", &w.ln2_w, D * 4) && get(ctx, p + "ln2.b", &w.ln2_b, D * 4) &&
                 get(ctx, p + "qkv.w", &w.qkv_w, (int64_t)ctx->v_qkv_npad * D * 2) && get(ctx, p + "qkv.b", &w.qkv_b, (int64_t)ctx->v_qkv_npad * 4) &&
                 get(ctx, p + "out.w", &w.out_w, (int64_t)ctx->v_dpad * D * 2) && get(ctx, p + "out.b", &w.out_b, (int64_t)ctx->v_dpad * 4) &&
                 get(ctx, p + "fc1.w", &w.fc1_w, (int64_t)ctx->v_mlp_pad * D * 2) && get(ctx, p + "fc1.b", &w.fc1_b, (int64_t)ctx->v_mlp_pad * 4) &&
                 get(ctx, p + "fc2.w", &w.fc2_w, (int64_t)ctx->v_dpad * g.vit_mlp * 2) && get(ctx, p + "fc2.b", &w.fc2_b, (int64_t)ctx->v_dpad * 4);
            if (!ok) return AUR_ERR_STATE;
        }
    }
    if (have_llm) {
        if (!ctx->kvpool) return aur_fail(ctx, AUR_ERR_STATE, "aur_finalize: kv pool not set");
        const int64_t d = g.llm_hidden;
        bool ok = get(ctx, "llm.embed", &ctx->l_embed, (int64_t)g.llm_vocab * d * 2) &&
                  get(ctx, "llm.lm_head.w", &ctx->l_head_w, (int64_t)ctx->l_vocab_pad * d * 2);
        if (!ok) return AUR_ERR_STATE;
        ctx->ll.assign(g.llm_layers, LlmLayerW{});
        for (int l = 0; l < g.llm_layers; ++l) {
            const std::string p = "llm." + std::to_string(l) + ".";
            LlmLayerW& w = ctx->ll[l];
            ok = get(ctx, p + "qkv.w", &w.qkv_w, (int64_t)ctx->l_qkv_npad * d * 2) && get(ctx, p + "o.w", &w.o_w, (int64_t)ctx->l_dpad * d * 2) &&
                 get(ctx, p + "gateup.w", &w.gateup_w, (int64_t)ctx->l_gu_npad * d * 2) &&
                 get(ctx, p + "down.w", &w.down_w, (int64_t)ctx->l_dpad * g.llm_mlp * 2);
            if (!ok) return AUR_ERR_STATE;
        }
        // projector is part of the language-side prefix path
        ctx->p_w.clear();
        ctx->p_b.clear();
        if (ctx->tensors.count("proj.0.w")) {
            for (int i = 0; i < ctx->p_depth; ++i) {
                const half_t* pw;
                const float* pb;
                const std::string p = "proj." + std::to_string(i) + ".";
                ok = get(ctx, p + "w", &pw, (int64_t)ctx->l_dpad * (i == 0 ? g.vit_hidden : d) * 2) && get(ctx, p + "b", &pb, (int64_t)ctx->l_dpad * 4);
                if (!ok) return AUR_ERR_STATE;
                ctx->p_w.push_back(pw);
                ctx->p_b.push_back(pb);
            }
        }
        // RoPE table (HF linear scaling: position / factor), cos/sin in fp32 like HF
        const int half = ctx->l_hd / 2;
        std::vector<float2> tab((size_t)ctx->l_ctx_pad * half);
        for (int p = 0; p < ctx->l_ctx_pad; ++p)
            for (int i = 0; i < half; ++i) {
                const float inv = 1.0f / powf(g.rope_theta, (float)(2 * i) / (float)ctx->l_hd);
                const float ang = ((float)p / g.rope_factor) * inv;
                tab[(size_t)p * half + i] = make_float2(cosf(ang), sinf(ang));
            }
        CK(hipMemcpyAsync(ctx->l_rope, tab.data(), tab.size() * sizeof(float2), hipMemcpyHostToDevice, s));
        std::vector<int32_t> pt((size_t)2 * ctx->kv_seqs * ctx->l_max_pages);
        for (size_t i = 0; i < pt.size(); ++i) pt[i] = (int32_t)i;      // initial allocation: sequence q owns pages [q*max_pages, ...)
        CK(hipMemcpyAsync(ctx->s_ptab, pt.data(), pt.size() * 4, hipMemcpyHostToDevice, s));
        // decode scratch in x-fragment form: lanes of unused batch rows must hold finite values forever
        CK(hipMemsetAsync(ctx->d_attn, 0, (size_t)rup(g.max_batch, 16) * g.llm_hidden * 2, s));
        CK(hipMemsetAsync(ctx->d_h, 0, (size_t)rup(g.max_batch, 16) * g.llm_mlp * 2, s));
        // KV pool: attention reads whole 64-token pages and weights the not-yet-written tail of the last page with p = 0;
        // 0 x (stale bits that decode as NaN / Inf) would poison the output, so the pool starts all-zero and from then on
        // only finite K / V values are ever stored into it.
        CK(hipMemsetAsync(ctx->kvpool, 0, (size_t)ctx->kv_bytes, s));
        CK(hipStreamSynchronize(s));
    }
    if (!have_vit && !have_llm) return aur_fail(ctx, AUR_ERR_STATE, "aur_finalize: no weights were provided");
    ctx->finalized = true;
    return AUR_OK;
}

extern "C" int aur_pack_linear(aur_ctx* ctx, const void* w, int32_t n_src, int32_t k_src, int32_t ld_src,
                               const int32_t* row_map, int32_t npad, int32_t kpad, void* out, void* stream) {
    if (npad % 16 || kpad % 32 || !w || !out) return aur_fail(ctx, AUR_ERR_ARG, "aur_pack_linear: npad %% 16, kpad %% 32 required");
    CK(launch_pack_weight((const half_t*)w, n_src, k_src, ld_src, row_map, npad, kpad, (half_t*)out, (hipStream_t)stream));
    return AUR_OK;
}

// ------------------------------------------------------------------------------------------ profiling
// roctx ranges per stage (tracing): `rocprofv3 --marker-trace --kernel-trace` then groups the launches of a stage under
// its name instead of by kernel name only.  The marker library is looked up at run time (librocprofiler-sdk-roctx.so, else libroctx64.so)
// when AURORA_ROCTX=1 is set - the product library has no link-time dependency on a profiler, and without the variable a range costs one
// predictable branch.  Ranges bracket the host-side ENQUEUE of a stage (what roctx measures); the device time is in the kernel trace.
// A ctx is driven by ONE host thread at a time (include/aurora_hip.h: "calls on one ctx are not re-entrant"): the open range of a stage name is state of the ctx,
// keyed by (name, stream), so the same stage enqueued on two streams by that thread (a front end beside the decode) keeps two ranges apart.
// Ranges are START / STOP ranges with ids (not the push / pop stack): an entry point that fails between stage_begin and stage_end
// (every CK() is an early return) leaves ONE unclosed range, which the next stage_begin of that name closes - the nesting of later
// stages cannot be skewed.
struct Roctx {
    uint64_t (*start)(const char*) = nullptr;
    void (*stop)(uint64_t) = nullptr;
    Roctx() {
        const char* on = getenv("AURORA_ROCTX");
        if (!on || on[0] != '1') return;
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        start = (uint64_t (*)(const char*))dlsym(h, "roctxRangeStartA");
        stop = (void (*)(uint64_t))dlsym(h, "roctxRangeStop");
        if (!start || !stop) start = nullptr, stop = nullptr;
    }
};
static const Roctx& roctx() {
    static Roctx r;          // read-only after its thread-safe initialisation
    return r;
}
static std::string range_key(const char* name, hipStream_t s) {
    char buf[96];
    snprintf(buf, sizeof buf, "%s@%p", name, (void*)s);
    return buf;
}
static void stage_begin(aur_ctx* c, const char* name, hipStream_t s) {
    if (roctx().start) {
        StageTimer& tr = c->timers[range_key(name, s)];
        if (tr.range_open) roctx().stop(tr.range_id);       // left open by a call that failed mid-stage
        tr.range_id = roctx().start(name);
        tr.range_open = true;
    }
    if (!c->prof || c->capturing) return;
    StageTimer& t = c->timers[name];
    if (!t.e0) {
        (void)hipEventCreate(&t.e0);
        (void)hipEventCreate(&t.e1);
    }
    if (t.open) {      // fold the previous interval
        (void)hipEventSynchronize(t.e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, t.e0, t.e1);
        t.ms += ms;
        t.open = false;
    }
    (void)hipEventRecord(t.e0, s);
}
static void stage_end(aur_ctx* c, const char* name, hipStream_t s) {
    if (roctx().stop) {
        StageTimer& tr = c->timers[range_key(name, s)];
        if (tr.range_open) roctx().stop(tr.range_id);
        tr.range_open = false;
    }
    if (!c->prof || c->capturing) return;
    StageTimer& t = c->timers[name];
    (void)hipEventRecord(t.e1, s);
    t.open = true;
    t.launches++;
}
extern "C" int aur_profile_enable(aur_ctx* ctx, int32_t on) {
    NO_CAPTURE("aur_profile_enable");
    ctx->prof = on != 0;
    for (auto& kv : ctx->timers) {
        kv.second.ms = 0;
        kv.second.launches = 0;
        kv.second.open = false;
    }
    for (auto& kv : ctx->kev) {
        kv.used = 0;
        kv.ms = 0;
        kv.n = 0;
    }
    return AUR_OK;
}
static void kev_fold(aur_ctx::KernelEvents& k) {
    for (size_t i = 0; i < k.used; ++i) {
        (void)hipEventSynchronize(k.ev[i].second);
        float ms = 0;
        if (hipEventElapsedTime(&ms, k.ev[i].first, k.ev[i].second) == hipSuccess) {
            k.ms += ms;
            k.n++;
        }
    }
    k.used = 0;
}
static void kev_begin(aur_ctx::KernelEvents& k, hipStream_t s) {
    if (k.used == k.ev.size()) {
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        k.ev.push_back({a, b});
    }
    (void)hipEventRecord(k.ev[k.used].first, s);
}
static void kev_end(aur_ctx::KernelEvents& k, hipStream_t s) {
    (void)hipEventRecord(k.ev[k.used].second, s);
    k.used++;
    if (k.used >= 4096) kev_fold(k);
}
extern "C" int aur_profile_read(aur_ctx* ctx, const char* stage, double* ms_out, int64_t* launches_out) {
    const int which = !strcmp(stage, "decode_attn") ? 0 : !strcmp(stage, "decode_gemm_gateup") ? 1 : !strcmp(stage, "vit_tome") ? 2 : !strcmp(stage, "first_token") ? 3 : -1;
    if (which >= 0) {
        kev_fold(ctx->kev[which]);
        if (ms_out) *ms_out = ctx->kev[which].ms;
        if (launches_out) *launches_out = ctx->kev[which].n;
        return AUR_OK;
    }
    auto it = ctx->timers.find(stage);
    if (it == ctx->timers.end()) {
        if (ms_out) *ms_out = 0;
        if (launches_out) *launches_out = 0;
        return AUR_OK;
    }
    StageTimer& t = it->second;
    if (t.open) {
        (void)hipEventSynchronize(t.e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, t.e0, t.e1);
        t.ms += ms;
        t.open = false;
    }
    if (ms_out) *ms_out = t.ms;
    if (launches_out) *launches_out = t.launches;
    return AUR_OK;
}

// ------------------------------------------------------------------------------------------ vision
static KvLayout vit_kv(const aur_ctx* c, int t_pad) {
    KvLayout L;
    L.base = c->w_kv;
    L.page_table = nullptr;
    L.max_pages = 0;
    L.page_tokens = t_pad;
    L.heads = c->cfg.vit_heads;
    L.kblk = c->v_kblk;
    L.vd16 = c->v_vd16;
    L.page_halves = vit_page_halves(c, t_pad);
    L.v_off = (int64_t)c->cfg.vit_heads * t_pad * c->v_hd_pad;
    return L;
}

// One encoder layer on padded state x [F][t_pad][D] (in place for r == 0; x_out otherwise).  Returns via
// *t_out the new token count; result lives in *x_res / *size_res (one of the two ping-pong buffers).
static int vit_layer_run(aur_ctx* ctx, int l, int F, int t, int r, half_t* x, const float* size, half_t* x_alt, float* size_alt,
                         half_t** x_res, const float** size_res, int* t_out, hipStream_t s, bool want_metric = false) {
    const aur_config& g = ctx->cfg;
    const VitLayerW& w = ctx->vl[l];
    const int D = g.vit_hidden, t_pad = rup(t, 32), M = F * t_pad;
    CK(launch_layernorm(x, D, w.ln1_w, w.ln1_b, 1e-5f, M, D, ctx->w_xn, D, s));          // aurora.py:735 (eps: :709)
    GemmArgs q{};
    q.A = ctx->w_xn; q.lda = D; q.W = w.qkv_w; q.bias = w.qkv_b; q.M = M; q.Npad = ctx->v_qkv_npad; q.K = D;
    q.rows_per_seq = t_pad; q.q_cols = ctx->v_qcols; q.k_cols = ctx->v_qcols; q.hd = ctx->v_hd; q.Qf = ctx->w_qf;
    q.kv = vit_kv(ctx, t_pad); q.rope = nullptr; q.pos0 = 0; q.seq0 = 0; q.tag = GT_VIT_QKV;
    CK(ctx_gemm(ctx, q, EPI_QKV, s));                                                        // aurora.py:634-636
    AttnArgs at{};
    at.Qf = ctx->w_qf; at.kv = q.kv; at.seq0 = 0; at.nseq = F; at.heads = g.vit_heads; at.rows_per_seq = t_pad; at.t = t;
    at.causal = 0; at.scale = 1.0f / sqrtf((float)ctx->v_hd); at.O = ctx->w_attn; at.ldo = D; at.hd = ctx->v_hd;
    CK(launch_attention(at, s));                                                           // aurora.py:647-697
    GemmArgs o{};
    o.A = ctx->w_attn; o.lda = D; o.W = w.out_w; o.bias = w.out_b; o.M = M; o.Npad = ctx->v_dpad; o.K = D;
    o.C = x; o.ldc = D; o.resid = x; o.ldr = D; o.n_real = D; o.act = ACT_NONE; o.tag = GT_VIT_OUT;
    CK(ctx_gemm(ctx, o, EPI_ROW, s));                                                        // aurora.py:699, 743
    int rl = r < (t - 1) / 2 ? r : (t - 1) / 2;                                            // tome.py:45
    half_t* xc = x;
    const float* sc = size;
    int t2 = t;
    bool ln2_done = false;
    if (rl > 0) {
        TomeArgs ta{};
        t2 = t - rl;
        ta.frames = F; ta.t = t; ta.t_pad = t_pad; ta.r = rl; ta.c = ctx->v_hd; ta.d = D;
        ta.kv = &q.kv; ta.metric_out = want_metric ? ctx->w_metric : nullptr;             // metric = mean_h K, aurora.py:639
        ta.x = x; ta.size = size; ta.t_out_pad = rup(t2, 32); ta.x_out = x_alt; ta.size_out = size_alt;
        ta.node_max = ctx->w_nmax; ta.node_idx = ctx->w_nidx; ta.unm = ctx->w_unm; ta.src = ctx->w_src; ta.dst = ctx->w_dst;
        ta.mhat = ctx->w_mhat;
        // LayerNorm 2 (aurora.py:750, eps :711) of the merged rows comes out of the merge launch (tome.hip: bitwise norm_kernel<false>)
        ln2_done = ctx->tome_fused_ln && (D & 7) == 0;
        if (ln2_done) { ta.ln_w = w.ln2_w; ta.ln_b = w.ln2_b; ta.ln_eps = 1e-5f; ta.ln_out = ctx->w_xn; }
        if (ctx->prof) kev_begin(ctx->kev[2], s);
        CK(launch_tome_step(ta, s));                                                       // aurora.py:746-747
        if (ctx->prof) kev_end(ctx->kev[2], s);
        xc = x_alt;
        sc = size_alt;
    }
    const int t2_pad = rup(t2, 32), M2 = F * t2_pad;
    if (!ln2_done) CK(launch_layernorm(xc, D, w.ln2_w, w.ln2_b, 1e-5f, M2, D, ctx->w_xn, D, s));         // aurora.py:750
    GemmArgs f1{};
    f1.A = ctx->w_xn; f1.lda = D; f1.W = w.fc1_w; f1.bias = w.fc1_b; f1.M = M2; f1.Npad = ctx->v_mlp_pad; f1.K = D;
    f1.C = ctx->w_h; f1.ldc = g.vit_mlp; f1.n_real = g.vit_mlp; f1.act = g.vit_act == AUR_ACT_GELU ? ACT_GELU : ACT_QUICK_GELU; f1.tag = GT_VIT_FC1;
    CK(ctx_gemm(ctx, f1, EPI_ROW, s));
    GemmArgs f2{};
    f2.A = ctx->w_h; f2.lda = g.vit_mlp; f2.W = w.fc2_w; f2.bias = w.fc2_b; f2.M = M2; f2.Npad = ctx->v_dpad; f2.K = g.vit_mlp;
    f2.C = xc; f2.ldc = D; f2.resid = xc; f2.ldr = D; f2.n_real = D; f2.act = ACT_NONE; f2.tag = GT_VIT_FC2;
    CK(ctx_gemm(ctx, f2, EPI_ROW, s));                                                       // aurora.py:751-752
    *x_res = xc;
    *size_res = sc;
    *t_out = t2;
    return AUR_OK;
}

extern "C" int aur_vit_pos_interp(aur_ctx* ctx, int32_t height, int32_t width, void* pos_out, void* stream) {
    if (!ctx->finalized || ctx->vl.empty()) return aur_fail(ctx, AUR_ERR_STATE, "aur_vit_pos_interp: vision weights not finalized");
    const aur_config& g = ctx->cfg;
    const int gh = height / g.vit_patch, gw = width / g.vit_patch;
    if (gh < 1 || gw < 1 || !pos_out) return aur_fail(ctx, AUR_ERR_ARG, "aur_vit_pos_interp: input %dx%d, patch %d", height, width, g.vit_patch);
    int n = 1;
    while ((n + 1) * (n + 1) <= ctx->v_native_npatch) ++n;          // aurora.py:929-931: the native grid is square
    if (n * n != ctx->v_native_npatch) return aur_fail(ctx, AUR_ERR_STATE, "aur_vit_pos_interp: the native table holds %d patch rows, not a square", ctx->v_native_npatch);
    CK(launch_pos_interp(ctx->v_pos, n, gh, gw, g.vit_hidden, (half_t*)pos_out, (hipStream_t)stream));
    return AUR_OK;
}

extern "C" int aur_vit_encode_hw(aur_ctx* ctx, const void* pixels, int32_t frames, int32_t height, int32_t width,
                                 const void* pos_emb, int32_t r, void* out_tokens, int32_t* n_kept_out, void* stream) {
    if (!ctx->finalized || ctx->vl.empty()) return aur_fail(ctx, AUR_ERR_STATE, "aur_vit_encode: vision weights not finalized");
    const aur_config& g = ctx->cfg;
    if (frames < 1 || frames > g.max_frames) return aur_fail(ctx, AUR_ERR_ARG, "frames %d outside [1, %d]", frames, g.max_frames);
    if (r < 0) return aur_fail(ctx, AUR_ERR_ARG, "r must be >= 0");
    const int gh = height / g.vit_patch, gw = width / g.vit_patch, npatch = gh * gw, t0 = npatch + 1;
    if (gh < 1 || gw < 1 || t0 > ctx->v_t0)
        return aur_fail(ctx, AUR_ERR_ARG, "input %dx%d gives %d tokens per frame; this ctx holds up to %d (vit_image %d)", height, width,
                        t0, ctx->v_t0, g.vit_image);
    const bool native = gh == gw && npatch == ctx->v_native_npatch;
    if (!pos_emb && !native) return aur_fail(ctx, AUR_ERR_ARG, "a %dx%d patch grid needs an interpolated position table (pos_emb)", gh, gw);
    hipStream_t s = (hipStream_t)stream;
    stage_begin(ctx, "vit", s);
    const int D = g.vit_hidden;
    CK(launch_im2col((const half_t*)pixels, frames, g.vit_channels, height, width, g.vit_patch, ctx->v_kpad, ctx->w_col, s));
    GemmArgs pe{};
    pe.A = ctx->w_col; pe.lda = ctx->v_kpad; pe.W = ctx->v_patch_w; pe.bias = nullptr; pe.M = frames * npatch;
    pe.Npad = ctx->v_dpad; pe.K = ctx->v_kpad; pe.C = ctx->w_patch; pe.ldc = D; pe.n_real = D; pe.act = ACT_NONE;
    CK(ctx_gemm(ctx, pe, EPI_ROW, s));
    CK(launch_vit_assemble(ctx->w_patch, ctx->v_cls, pos_emb ? (const half_t*)pos_emb : ctx->v_pos, ctx->v_preln_w, ctx->v_preln_b,
                           g.vit_ln_eps, frames, npatch, D, rup(t0, 32), ctx->w_xa, s));
    half_t* x = ctx->w_xa;
    half_t* x_alt = ctx->w_xb;
    const float* size = nullptr;          // aurora.py:811: size = None at layer 0
    float* size_alt = ctx->w_sza;
    int t = t0;
    for (int l = 0; l < g.vit_layers - 1; ++l) {
        half_t* xr;
        const float* sr;
        int t2;
        int rc = vit_layer_run(ctx, l, frames, t, r, x, size, x_alt, size_alt, &xr, &sr, &t2, s);
        if (rc) return rc;
        if (xr != x) {           // merged into the alternate buffer: swap ping-pong roles
            x_alt = x;
            x = xr;
            size_alt = (sr == ctx->w_sza) ? ctx->w_szb : ctx->w_sza;
            size = sr;
        }
        t = t2;
    }
    CK(launch_strip_cls(x, frames, t, rup(t, 32), D, (half_t*)out_tokens, s));           // aurora.py:253
    if (n_kept_out) *n_kept_out = t - 1;
    stage_end(ctx, "vit", s);
    return AUR_OK;
}

extern "C" int aur_vit_encode(aur_ctx* ctx, const void* pixels, int32_t frames, int32_t r, void* out_tokens,
                              int32_t* n_kept_out, void* stream) {
    const int nat = ctx->cfg.vit_native_image > 0 ? ctx->cfg.vit_native_image : ctx->cfg.vit_image;
    return aur_vit_encode_hw(ctx, pixels, frames, nat, nat, nullptr, r, out_tokens, n_kept_out, stream);
}

extern "C" int aur_vit_layer(aur_ctx* ctx, int32_t layer, const void* x, const float* size, int32_t frames, int32_t t,
                             int32_t r, void* x_out, float* size_out, float* metric_out, int32_t* node_idx,
                             int32_t* unm_idx, int32_t* src_idx, int32_t* dst_idx, void* stream) {
    if (!ctx->finalized || layer < 0 || layer >= (int)ctx->vl.size()) return aur_fail(ctx, AUR_ERR_ARG, "aur_vit_layer: bad layer / not finalized");
    if (frames < 1 || frames > ctx->cfg.max_frames || t < 2 || t > ctx->v_t0) return aur_fail(ctx, AUR_ERR_ARG, "aur_vit_layer: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const int D = ctx->cfg.vit_hidden, t_pad = rup(t, 32);
    CK(launch_pad_rows((const half_t*)x, size, frames, t, t_pad, D, ctx->w_xa, size ? ctx->w_sza : nullptr, s));
    half_t* xr;
    const float* sr;
    int t2;
    int rc = vit_layer_run(ctx, layer, frames, t, r, ctx->w_xa, size ? ctx->w_sza : nullptr, ctx->w_xb, ctx->w_szb, &xr, &sr, &t2, s, metric_out != nullptr);
    if (rc) return rc;
    CK(launch_unpad_rows(xr, sr, frames, t2, rup(t2, 32), D, (half_t*)x_out, size_out, s));
    const int rl = r < (t - 1) / 2 ? r : (t - 1) / 2;
    if (rl > 0) {
        const int ta = (t + 1) / 2;
        if (metric_out) CK(hipMemcpyAsync(metric_out, ctx->w_metric, (size_t)frames * t * ctx->v_hd * 4, hipMemcpyDeviceToDevice, s));
        if (node_idx) CK(hipMemcpyAsync(node_idx, ctx->w_nidx, (size_t)frames * ta * 4, hipMemcpyDeviceToDevice, s));
        if (unm_idx) CK(hipMemcpyAsync(unm_idx, ctx->w_unm, (size_t)frames * (ta - rl) * 4, hipMemcpyDeviceToDevice, s));
        if (src_idx) CK(hipMemcpyAsync(src_idx, ctx->w_src, (size_t)frames * rl * 4, hipMemcpyDeviceToDevice, s));
        if (dst_idx) CK(hipMemcpyAsync(dst_idx, ctx->w_dst, (size_t)frames * rl * 4, hipMemcpyDeviceToDevice, s));
    } else if (metric_out) {
        KvLayout kv = vit_kv(ctx, t_pad);
        CK(launch_tome_metric(kv, frames, t, ctx->v_hd, metric_out, s));
    }
    return AUR_OK;
}

extern "C" int aur_tome_step(aur_ctx* ctx, const float* metric, const void* x, const float* size, int32_t frames, int32_t t,
                             int32_t c, int32_t d, int32_t r, void* x_out, float* size_out, int32_t* node_idx,
                             int32_t* unm_idx, int32_t* src_idx, int32_t* dst_idx, void* stream) {
    if (!ctx->ws) return aur_fail(ctx, AUR_ERR_STATE, "aur_tome_step: workspace not set");
    if (frames < 1 || t < 1 || c < 1 || (d & 7)) return aur_fail(ctx, AUR_ERR_ARG, "aur_tome_step: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const int rl = r < (t - 1) / 2 ? r : (t - 1) / 2;
    if (rl <= 0) {      // tome.py:47-48 do_nothing; merge_wavg still materialises size (tome.py:212-213)
        CK(hipMemcpyAsync(x_out, x, (size_t)frames * t * d * 2, hipMemcpyDeviceToDevice, s));
        CK(launch_unpad_rows((const half_t*)x, size, frames, t, t, d, (half_t*)x_out, size_out, s));
        return AUR_OK;
    }
    const int64_t ta = (t + 1) / 2;
    // scratch must fit the ctx workspace regions sized for the configured ViT
    if (c > 128) return aur_fail(ctx, AUR_ERR_ARG, "aur_tome_step: at most 128 metric channels (got %d)", c);
    if (ta > 4096) return aur_fail(ctx, AUR_ERR_ARG, "aur_tome_step: ceil(t / 2) = %lld exceeds 4096 (the select kernel ranks one frame's A tokens in one workgroup)", (long long)ta);
    if (d > 2048) return aur_fail(ctx, AUR_ERR_ARG, "aur_tome_step: d = %d exceeds 2048 (a merge workgroup holds one row of at most 2048 features per wave)", d);
    if ((int64_t)frames * tome_mfrag_floats(t, c) > (int64_t)ctx->cfg.max_frames * tome_mfrag_floats(ctx->v_t0, ctx->v_hd) || (int64_t)frames * ta > (int64_t)ctx->cfg.max_frames * ((ctx->v_t0 + 1) / 2) ||
        (int64_t)frames * rup(t, 32) * d > (int64_t)ctx->cfg.max_frames * ctx->v_t0pad * ctx->cfg.vit_hidden)
        return aur_fail(ctx, AUR_ERR_ARG, "aur_tome_step: problem larger than the workspace configured at aur_create");
    const int t_pad = rup(t, 32), t2 = t - rl, t2_pad = rup(t2, 32);
    CK(launch_pad_rows((const half_t*)x, size, frames, t, t_pad, d, ctx->w_xa, ctx->w_sza, s));
    TomeArgs a{};
    a.frames = frames; a.t = t; a.t_pad = t_pad; a.r = rl; a.c = c; a.d = d; a.metric = metric; a.x = ctx->w_xa;
    a.size = size ? ctx->w_sza : nullptr; a.t_out_pad = t2_pad; a.x_out = ctx->w_xb; a.size_out = ctx->w_szb;
    a.node_max = ctx->w_nmax; a.node_idx = ctx->w_nidx; a.unm = ctx->w_unm; a.src = ctx->w_src; a.dst = ctx->w_dst; a.mhat = ctx->w_mhat;
    CK(launch_tome_step(a, s));
    CK(launch_unpad_rows(ctx->w_xb, ctx->w_szb, frames, t2, t2_pad, d, (half_t*)x_out, size_out, s));
    if (node_idx) CK(hipMemcpyAsync(node_idx, ctx->w_nidx, (size_t)frames * ta * 4, hipMemcpyDeviceToDevice, s));
    if (unm_idx) CK(hipMemcpyAsync(unm_idx, ctx->w_unm, (size_t)frames * (ta - rl) * 4, hipMemcpyDeviceToDevice, s));
    if (src_idx) CK(hipMemcpyAsync(src_idx, ctx->w_src, (size_t)frames * rl * 4, hipMemcpyDeviceToDevice, s));
    if (dst_idx) CK(hipMemcpyAsync(dst_idx, ctx->w_dst, (size_t)frames * rl * 4, hipMemcpyDeviceToDevice, s));
    return AUR_OK;
}

// ------------------------------------------------------------------------------------------ generic ops
extern "C" int aur_linear(aur_ctx* ctx, const void* a, int32_t m, int32_t k, const void* w_packed, int32_t npad, int32_t n,
                          const float* bias, int32_t act, const void* resid, void* c, void* stream) {
    if (m < 1 || (k & 63) || (npad & 127) || n > npad || (n & 3)) return aur_fail(ctx, AUR_ERR_ARG, "aur_linear: need k %% 64 == 0, npad %% 128 == 0, n %% 4 == 0");
    GemmArgs g{};
    g.A = (const half_t*)a; g.lda = k; g.W = (const half_t*)w_packed; g.bias = bias; g.M = m; g.Npad = npad; g.K = k;
    g.C = (half_t*)c; g.ldc = n; g.resid = (const half_t*)resid; g.ldr = n; g.n_real = n;
    g.act = act == AUR_ACT_GELU ? ACT_GELU : act == AUR_ACT_QUICK_GELU ? ACT_QUICK_GELU : ACT_NONE;
    CK(ctx_gemm(ctx, g, EPI_ROW, (hipStream_t)stream));
    return AUR_OK;
}
extern "C" int aur_linear_skinny(aur_ctx* ctx, const void* a, int32_t m, int32_t k, const void* w_packed, int32_t npad,
                                 int32_t n, float* out, void* stream) {
    if (m < 1 || m > AUR_MAX_BATCH || (k & 127) || k > 16384 || (npad & 31) || n > npad || (n & 3))
        return aur_fail(ctx, AUR_ERR_ARG, "aur_linear_skinny: need m <= %d, k %% 128 == 0, k <= 16384", AUR_MAX_BATCH);
    if (!ctx->ws) return aur_fail(ctx, AUR_ERR_STATE, "aur_linear_skinny: workspace not set");
    hipStream_t st = (hipStream_t)stream;
    CK(hipMemsetAsync(ctx->d_scr, 0, (size_t)AUR_MAX_BATCH * 16384 * 2, st));
    CK(launch_xfrag_pack((const half_t*)a, k, m, k, ctx->d_scr, st));
    SkinnyArgs s{};
    s.xf = ctx->d_scr; s.W = (const half_t*)w_packed; s.B = m; s.b_lo = 0; s.b_hi = m; s.Npad = npad; s.K = k; s.n_real = n;
    s.mode = SK_LOGITS; s.out32 = out; s.variant = ctx->skinny_variant;
    CK(launch_skinny(s, st));
    return AUR_OK;
}
extern "C" int aur_layernorm(aur_ctx* ctx, const void* x, int32_t rows, int32_t d, const float* w, const float* b, float eps,
                             void* y, void* stream) {
    CK(launch_layernorm((const half_t*)x, d, w, b, eps, rows, d, (half_t*)y, d, (hipStream_t)stream));
    return AUR_OK;
}
extern "C" int aur_rmsnorm(aur_ctx* ctx, const void* x, int32_t rows, int32_t d, const float* w, float eps, void* y, void* stream) {
    CK(launch_rmsnorm((const half_t*)x, d, w, eps, rows, d, (half_t*)y, d, (hipStream_t)stream));
    return AUR_OK;
}

// ------------------------------------------------------------------------------------------ projector + splice
extern "C" int aur_project_splice(aur_ctx* ctx, const void* vis, int32_t nvis, const int32_t* vis_rows, const int32_t* text_ids,
                                  const int32_t* text_rows, int32_t ntext, int32_t seq_len, void* embeds, void* stream) {
    if (!ctx->finalized || ctx->p_w.empty()) return aur_fail(ctx, AUR_ERR_STATE, "aur_project_splice: projector weights not finalized");
    const aur_config& g = ctx->cfg;
    if (seq_len < 1 || seq_len > g.max_ctx || nvis + ntext != seq_len) return aur_fail(ctx, AUR_ERR_ARG, "aur_project_splice: seq_len %d (nvis %d + ntext %d), max_ctx %d", seq_len, nvis, ntext, g.max_ctx);
    hipStream_t s = (hipStream_t)stream;
    stage_begin(ctx, "project", s);
    const int d = g.llm_hidden, lp = rup(seq_len, 32);
    if (lp > seq_len) CK(hipMemsetAsync((half_t*)embeds + (int64_t)seq_len * d, 0, (size_t)(lp - seq_len) * d * 2, s));
    if (nvis > 0) {
        // modeling_projector.py:20-33: Linear, then (act, Linear) x (depth - 1); the activation rides in the producing GEMM's epilogue and
        // the LAST Linear writes straight into the spliced rows.  Intermediates ping-pong between l_p1 and l_xn (front-end scratch, free
        // until the prefill that follows on this stream).
        const half_t* in = (const half_t*)vis;
        int K = g.vit_hidden;
        for (int i = 0; i < ctx->p_depth; ++i) {
            const bool last = i == ctx->p_depth - 1;
            half_t* out = last ? (half_t*)embeds : ((i & 1) ? ctx->l_xn : ctx->l_p1);
            GemmArgs a{};
            a.A = in; a.lda = K; a.W = ctx->p_w[i]; a.bias = ctx->p_b[i]; a.M = nvis; a.Npad = ctx->l_dpad; a.K = K;
            a.C = out; a.ldc = d; a.n_real = d; a.act = last ? ACT_NONE : ctx->p_act; a.out_rows = last ? vis_rows : nullptr;
            CK(ctx_gemm(ctx, a, EPI_ROW, s));
            in = out;
            K = d;
        }
    }
    CK(launch_embed_rows(ctx->l_embed, d, text_ids, text_rows, ntext, (half_t*)embeds, d, s));   // utils.py:214-216
    stage_end(ctx, "project", s);
    return AUR_OK;
}

// ------------------------------------------------------------------------------------------ language model
static KvLayout llm_kv(const aur_ctx* c, int layer) {
    KvLayout L;
    L.base = (half_t*)c->kvpool + (int64_t)layer * c->l_layer_halves;
    L.page_table = c->ptab_cur;
    L.max_pages = c->l_max_pages;
    L.page_tokens = c->cfg.page_tokens;
    L.heads = c->cfg.llm_heads;
    L.kblk = c->l_kblk;
    L.vd16 = c->l_vd16;
    L.page_halves = c->l_page_halves;
    L.v_off = c->l_page_halves / 2;
    return L;
}

static void drop_graphs(aur_ctx* ctx) {
    if (ctx->graph) (void)hipGraphExecDestroy(ctx->graph);
    if (ctx->graph_h) (void)hipGraphExecDestroy(ctx->graph_h);
    ctx->graph = ctx->graph_h = nullptr;
}

extern "C" int aur_begin_batch(aur_ctx* ctx, int32_t batch, int32_t max_new_tokens, int32_t eos_id, void* stream) {
    NO_CAPTURE("aur_begin_batch");
    if (!ctx->finalized || ctx->ll.empty()) return aur_fail(ctx, AUR_ERR_STATE, "aur_begin_batch: language weights not finalized");
    const aur_config& g = ctx->cfg;
    if (batch < 1 || batch > g.max_batch) return aur_fail(ctx, AUR_ERR_ARG, "batch %d outside [1, %d]", batch, g.max_batch);
    if (max_new_tokens < 1 || max_new_tokens > g.max_new_tokens) return aur_fail(ctx, AUR_ERR_ARG, "max_new_tokens %d outside [1, %d]", max_new_tokens, g.max_new_tokens);
    hipStream_t s = (hipStream_t)stream;
    if ((ctx->graph || ctx->graph_h) && (ctx->max_new != max_new_tokens || ctx->eos != eos_id || ctx->graph_batch != batch)) {
        // eos / max_new / batch are by-value kernel arguments frozen inside the captured graph
        CK(hipStreamSynchronize(s));
        drop_graphs(ctx);
    }
    ctx->batch = batch;
    ctx->max_new = max_new_tokens;
    ctx->eos = eos_id;
    CK(hipMemsetAsync(ctx->s_len, 0, (size_t)g.max_batch * 4, s));
    CK(hipMemsetAsync(ctx->s_fin, 0, (size_t)g.max_batch * 4, s));
    CK(hipMemsetAsync(ctx->s_pos, 0, (size_t)g.max_batch * 4, s));
    CK(hipMemsetAsync(ctx->s_ids, 0, (size_t)g.max_batch * g.max_new_tokens * 4, s));
    const size_t bp = (size_t)rup(g.max_batch, 16);       // unused fragment lanes must hold finite values
    CK(hipMemsetAsync(ctx->d_x, 0, bp * g.llm_hidden * 2, s));          // bank-owned buffers only: the other bank may be decoding
    CK(hipMemsetAsync(ctx->s_ssq_mlp, 0, AUR_SSQ_SLOTS * AUR_MAX_BATCH * 8, s));
    CK(hipMemsetAsync(ctx->s_ssq_attn, 0, AUR_SSQ_SLOTS * AUR_MAX_BATCH * 8, s));
    return AUR_OK;
}

// logits for batch columns [b0, b0 + nb) from the residual rows in d_x (x-fragment form; final RMSNorm folded into the
// lm_head weights, 1/rms from s_ssq_mlp), then greedy bookkeeping + next-token embedding
static int lm_head_and_advance(aur_ctx* ctx, int b0, int nb, int advance, int set_pos, hipStream_t s) {
    const aur_config& g = ctx->cfg;
    SkinnyArgs h{};
    h.xf = ctx->d_x; h.W = ctx->l_head_w; h.B = ctx->batch; h.Npad = ctx->l_vocab_pad; h.K = g.llm_hidden; h.n_real = g.llm_vocab;
    h.mode = SK_LOGITS; h.out32 = ctx->d_logits; h.b_lo = b0; h.b_hi = b0 + nb; h.ssq_in = ctx->s_ssq_mlp; h.norm_eps = g.llm_rms_eps;
    h.variant = ctx->skinny_variant;
    CK(launch_skinny(h, s));
    CK(launch_argmax_advance(ctx->d_logits, b0, nb, g.llm_vocab, ctx->l_embed, g.llm_hidden, ctx->eos, ctx->max_new, ctx->s_ids, ctx->s_len,
                             ctx->s_fin, ctx->s_pos, ctx->d_x, ctx->s_ssq_mlp, advance, set_pos, s));
    return AUR_OK;
}

// One prefill layer; `which` selects the kernels (bit 0 norm1, 1 qkv, 2 attention, 3 o_proj, 4 norm2, 5 gate/up, 6 down)
// so that aur_microbench can time each of them in isolation.
static int prefill_layer(aur_ctx* ctx, int l, int which, int slot, int nseq, half_t* x, int seq_len, hipStream_t s) {
    const aur_config& g = ctx->cfg;
    const LlmLayerW& w = ctx->ll[l];
    const int d = g.llm_hidden, Mseq = rup(seq_len, 32), M = nseq * Mseq;
    if (which & 1) CK(launch_rmsnorm(x, d, nullptr, g.llm_rms_eps, M, d, ctx->l_xn, d, s));      // weight folded into qkv.w
    if (which & 2) {
        GemmArgs q{};
        q.A = ctx->l_xn; q.lda = d; q.W = w.qkv_w; q.bias = nullptr; q.M = M; q.Npad = ctx->l_qkv_npad; q.K = d;
        q.rows_per_seq = Mseq; q.q_cols = d; q.k_cols = d; q.hd = ctx->l_hd; q.Qf = ctx->l_qf; q.kv = llm_kv(ctx, l);
        q.rope = ctx->l_rope; q.pos0 = 0; q.seq0 = slot; q.tag = GT_LLM_QKV;
        CK(ctx_gemm(ctx, q, EPI_QKV, s));
    }
    if (which & 4) {
        AttnArgs at{};
        at.Qf = ctx->l_qf; at.kv = llm_kv(ctx, l); at.seq0 = slot; at.nseq = nseq; at.heads = g.llm_heads; at.rows_per_seq = Mseq; at.t = seq_len;
        at.causal = 1; at.scale = 1.0f / sqrtf((float)ctx->l_hd); at.O = ctx->l_attn; at.ldo = d; at.hd = ctx->l_hd;
        CK(launch_attention(at, s));
    }
    if (which & 8) {
        GemmArgs o{};
        o.A = ctx->l_attn; o.lda = d; o.W = w.o_w; o.M = M; o.Npad = ctx->l_dpad; o.K = d; o.C = x; o.ldc = d; o.resid = x; o.ldr = d;
        o.n_real = d; o.act = ACT_NONE; o.tag = GT_LLM_O;
        CK(ctx_gemm(ctx, o, EPI_ROW, s));
    }
    if (which & 16) CK(launch_rmsnorm(x, d, nullptr, g.llm_rms_eps, M, d, ctx->l_xn, d, s));     // weight folded into gateup.w
    if (which & 32) {
        GemmArgs gu{};
        gu.A = ctx->l_xn; gu.lda = d; gu.W = w.gateup_w; gu.M = M; gu.Npad = ctx->l_gu_npad; gu.K = d; gu.C = ctx->l_h; gu.ldc = g.llm_mlp;
        gu.n_real = 2 * g.llm_mlp; gu.act = ACT_SILU_MUL; gu.tag = GT_LLM_GATEUP;
        CK(ctx_gemm(ctx, gu, EPI_ROW, s));
    }
    if (which & 64) {
        GemmArgs dn{};
        dn.A = ctx->l_h; dn.lda = g.llm_mlp; dn.W = w.down_w; dn.M = M; dn.Npad = ctx->l_dpad; dn.K = g.llm_mlp; dn.C = x; dn.ldc = d;
        dn.resid = x; dn.ldr = d; dn.n_real = d; dn.act = ACT_NONE; dn.tag = GT_LLM_DOWN;
        CK(ctx_gemm(ctx, dn, EPI_ROW, s));
    }
    return AUR_OK;
}

// The LAST layer of a prefill pass.  After it only each sequence's last hidden state is read (final norm + lm_head for the first token),
// while every later decode step needs the layer's K and V of ALL positions.  So: K / V projection (+ RoPE + page write) over all rows as
// in any layer, and the query projection, attention, o projection, residual and the whole MLP on the LAST 128 ROWS of every sequence only
// (the attention kernel's query block; it holds position seq_len - 1 because rows are padded to a multiple of 32).  HF computes all rows
// and throws them away; every value that is kept is bit-identical here - a GEMM's per-element k order does not depend on M, the tile
// kernels agree bitwise (tests), and the query block sees the same keys.  Saves 83 % of one layer of 32: ~2.5 % of a prefill.
// Scratch: the compact [nseq * 128] activations live in l_h (free until this layer's gate/up), the compact Q fragments in l_qf.
static int prefill_last_layer(aur_ctx* ctx, int l, int slot, int nseq, half_t* x, int seq_len, hipStream_t s) {
    const aur_config& g = ctx->cfg;
    const LlmLayerW& w = ctx->ll[l];
    const int d = g.llm_hidden, mlp = g.llm_mlp, Mseq = rup(seq_len, 32), M = nseq * Mseq, LB = 128, Mc = nseq * LB, row0 = Mseq - LB;
    half_t* xn_c = ctx->l_h;
    half_t* attn_c = xn_c + (int64_t)Mc * d;
    half_t* x_c = attn_c + (int64_t)Mc * d;
    half_t* h_c = x_c + (int64_t)Mc * d;
    const size_t rowb = (size_t)d * 2;
    CK(launch_rmsnorm(x, d, nullptr, g.llm_rms_eps, M, d, ctx->l_xn, d, s));
    {   // K and V of every position (the Q third of the packed weight is skipped: its tiles come first)
        GemmArgs q{};
        q.A = ctx->l_xn; q.lda = d; q.W = w.qkv_w + (int64_t)(d / 16) * (d / 32) * AUR_FRAG_HALVES; q.M = M; q.Npad = ctx->l_qkv_npad - d; q.K = d;
        q.rows_per_seq = Mseq; q.q_cols = 0; q.k_cols = d; q.hd = ctx->l_hd; q.Qf = ctx->l_qf; q.kv = llm_kv(ctx, l);
        q.rope = ctx->l_rope; q.pos0 = 0; q.seq0 = slot; q.tag = GT_LLM_QKV;
        CK(ctx_gemm(ctx, q, EPI_QKV, s));
    }
    CK(hipMemcpy2DAsync(xn_c, LB * rowb, ctx->l_xn + (int64_t)row0 * d, Mseq * rowb, LB * rowb, nseq, hipMemcpyDeviceToDevice, s));
    CK(hipMemcpy2DAsync(x_c, LB * rowb, x + (int64_t)row0 * d, Mseq * rowb, LB * rowb, nseq, hipMemcpyDeviceToDevice, s));
    {   // queries of the last block
        GemmArgs q{};
        q.A = xn_c; q.lda = d; q.W = w.qkv_w; q.M = Mc; q.Npad = d; q.K = d;
        q.rows_per_seq = LB; q.q_cols = d; q.k_cols = 0; q.hd = ctx->l_hd; q.Qf = ctx->l_qf; q.kv = llm_kv(ctx, l);
        q.rope = ctx->l_rope; q.pos0 = row0; q.seq0 = slot; q.tag = GT_OTHER;
        CK(ctx_gemm(ctx, q, EPI_QKV, s));
    }
    {
        AttnArgs at{};
        at.Qf = ctx->l_qf; at.kv = llm_kv(ctx, l); at.seq0 = slot; at.nseq = nseq; at.heads = g.llm_heads; at.rows_per_seq = LB; at.t = seq_len;
        at.causal = 1; at.scale = 1.0f / sqrtf((float)ctx->l_hd); at.O = attn_c; at.ldo = d; at.hd = ctx->l_hd; at.q_pos0 = row0;
        CK(launch_attention(at, s));
    }
    {
        GemmArgs o{};
        o.A = attn_c; o.lda = d; o.W = w.o_w; o.M = Mc; o.Npad = ctx->l_dpad; o.K = d; o.C = x_c; o.ldc = d; o.resid = x_c; o.ldr = d;
        o.n_real = d; o.act = ACT_NONE; o.tag = GT_OTHER;
        CK(ctx_gemm(ctx, o, EPI_ROW, s));
    }
    CK(launch_rmsnorm(x_c, d, nullptr, g.llm_rms_eps, Mc, d, xn_c, d, s));
    {
        GemmArgs gu{};
        gu.A = xn_c; gu.lda = d; gu.W = w.gateup_w; gu.M = Mc; gu.Npad = ctx->l_gu_npad; gu.K = d; gu.C = h_c; gu.ldc = mlp;
        gu.n_real = 2 * mlp; gu.act = ACT_SILU_MUL; gu.tag = GT_OTHER;
        CK(ctx_gemm(ctx, gu, EPI_ROW, s));
    }
    {
        GemmArgs dn{};
        dn.A = h_c; dn.lda = mlp; dn.W = w.down_w; dn.M = Mc; dn.Npad = ctx->l_dpad; dn.K = mlp; dn.C = x_c; dn.ldc = d;
        dn.resid = x_c; dn.ldr = d; dn.n_real = d; dn.act = ACT_NONE; dn.tag = GT_OTHER;
        CK(ctx_gemm(ctx, dn, EPI_ROW, s));
    }
    CK(hipMemcpy2DAsync(x + (int64_t)row0 * d, Mseq * rowb, x_c, LB * rowb, LB * rowb, nseq, hipMemcpyDeviceToDevice, s));
    return AUR_OK;
}

// The layer stack of a prefill pass over KV sequences [seq0, seq0 + nseq): KV pages and front-end scratch only - no decode state.
static int prefill_layers(aur_ctx* ctx, int seq0, int nseq, void* embeds, int seq_len, hipStream_t s, const char* who) {
    if (!ctx->finalized || ctx->ll.empty()) return aur_fail(ctx, AUR_ERR_STATE, "%s: language weights not finalized", who);
    const aur_config& g = ctx->cfg;
    if (ctx->batch < 1) return aur_fail(ctx, AUR_ERR_STATE, "%s: no active batch (aur_begin_batch)", who);
    if (seq_len < 1 || seq_len + ctx->max_new > g.max_ctx) return aur_fail(ctx, AUR_ERR_ARG, "seq_len %d + max_new %d exceeds max_ctx %d", seq_len, ctx->max_new, g.max_ctx);
    stage_begin(ctx, "prefill", s);
    ctx->last_prefill_len = seq_len;
    const int Mseq = rup(seq_len, 32);
    for (int l = 0; l < g.llm_layers; ++l) {
        // the compact scratch of the pruned last layer must fit l_h: nseq * 128 * (3 d + mlp) <= nseq * Mseq * mlp (and LB < Mseq)
        const bool prune = ctx->prune_last && l == g.llm_layers - 1 && (int64_t)128 * (3 * g.llm_hidden + g.llm_mlp) <= (int64_t)Mseq * g.llm_mlp && Mseq > 128;
        int rc = prune ? prefill_last_layer(ctx, l, seq0, nseq, (half_t*)embeds, seq_len, s)
                       : prefill_layer(ctx, l, 0x7f, seq0, nseq, (half_t*)embeds, seq_len, s);
        if (rc) return rc;
    }
    return AUR_OK;
}
// First tokens of slots [slot0, slot0 + nseq) from the last prompt rows of `embeds` (the final hidden states `prefill_layers` left
// there): residual fragments + sum(x^2) - the lm_head applies the (folded) final RMSNorm itself -, logits, argmax, bookkeeping.
static int prefill_first_tokens(aur_ctx* ctx, int slot0, int nseq, void* embeds, int seq_len, hipStream_t s, const char* stage = "prefill") {
    const aur_config& g = ctx->cfg;
    const int d = g.llm_hidden, Mseq = rup(seq_len, 32);
    if (ctx->prof) kev_begin(ctx->kev[3], s);
    CK(launch_xfrag_norm((half_t*)embeds + (int64_t)(seq_len - 1) * d, (int64_t)Mseq * d, nullptr, g.llm_rms_eps, nseq, d, slot0, ctx->d_x, ctx->s_ssq_mlp, s));
    int rc = lm_head_and_advance(ctx, slot0, nseq, 0, seq_len, s);
    if (ctx->prof) kev_end(ctx->kev[3], s);       // closed on the error path too: the event ring and the roctx nesting stay balanced
    stage_end(ctx, stage, s);
    return rc;
}

extern "C" int aur_llm_prefill_batch(aur_ctx* ctx, int32_t slot0, int32_t nseq, void* embeds, int32_t seq_len, void* stream) {
    if (nseq < 1 || slot0 < 0 || slot0 + nseq > ctx->batch) return aur_fail(ctx, AUR_ERR_ARG, "slots [%d, %d) outside the batch of %d (aur_begin_batch)", slot0, slot0 + nseq, ctx->batch);
    hipStream_t s = (hipStream_t)stream;
    if (int rc = prefill_layers(ctx, slot0, nseq, embeds, seq_len, s, "aur_llm_prefill")) return rc;
    return prefill_first_tokens(ctx, slot0, nseq, embeds, seq_len, s);
}

extern "C" int aur_llm_prefill_stage(aur_ctx* ctx, int32_t seq0, int32_t nseq, void* embeds, int32_t seq_len, void* stream) {
    if (nseq < 1 || seq0 < 0 || seq0 + nseq > ctx->kv_seqs)
        return aur_fail(ctx, AUR_ERR_ARG, "aur_llm_prefill_stage: KV sequences [%d, %d) outside [0, %d) (max_batch + spare_slots)", seq0, seq0 + nseq, ctx->kv_seqs);
    // its own timer interval, opened and closed on THIS stream; the commit (decode stream, possibly several stages later) is timed
    // as "prefill_commit"
    if (int rc = prefill_layers(ctx, seq0, nseq, embeds, seq_len, (hipStream_t)stream, "aur_llm_prefill_stage")) return rc;
    stage_end(ctx, "prefill", (hipStream_t)stream);
    return AUR_OK;
}

// page-table rows of sequences [a0, a0 + n) <-> [b0, b0 + n)
__global__ void ptab_swap_kernel(int32_t* __restrict__ pt, int a0, int b0, int n, int max_pages) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * max_pages) return;
    const int j = i / max_pages, p = i - j * max_pages;
    int32_t* pa = pt + (int64_t)(a0 + j) * max_pages + p;
    int32_t* pb = pt + (int64_t)(b0 + j) * max_pages + p;
    const int32_t t = *pa;
    *pa = *pb;
    *pb = t;
}

extern "C" int aur_llm_prefill_commit(aur_ctx* ctx, int32_t slot0, int32_t nseq, int32_t seq0, void* embeds, int32_t seq_len, void* stream) {
    if (!ctx->finalized || ctx->ll.empty() || ctx->batch < 1) return aur_fail(ctx, AUR_ERR_STATE, "aur_llm_prefill_commit: no active batch");
    if (nseq < 1 || slot0 < 0 || slot0 + nseq > ctx->batch) return aur_fail(ctx, AUR_ERR_ARG, "aur_llm_prefill_commit: slots [%d, %d) outside the batch of %d", slot0, slot0 + nseq, ctx->batch);
    if (seq0 < 0 || seq0 + nseq > ctx->kv_seqs) return aur_fail(ctx, AUR_ERR_ARG, "aur_llm_prefill_commit: KV sequences [%d, %d) outside [0, %d)", seq0, seq0 + nseq, ctx->kv_seqs);
    if (seq0 != slot0 && seq0 < slot0 + nseq && slot0 < seq0 + nseq) return aur_fail(ctx, AUR_ERR_ARG, "aur_llm_prefill_commit: slot and sequence ranges overlap");
    if (seq_len < 1 || seq_len + ctx->max_new > ctx->cfg.max_ctx) return aur_fail(ctx, AUR_ERR_ARG, "seq_len %d + max_new %d exceeds max_ctx %d", seq_len, ctx->max_new, ctx->cfg.max_ctx);
    hipStream_t s = (hipStream_t)stream;
    stage_begin(ctx, "prefill_commit", s);
    if (seq0 != slot0) {
        const int n = nseq * ctx->l_max_pages;
        hipLaunchKernelGGL(ptab_swap_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ctx->ptab_rw(), slot0, seq0, nseq, ctx->l_max_pages);
        CK(hipGetLastError());
    }
    CK(hipMemsetAsync(ctx->s_len + slot0, 0, (size_t)nseq * 4, s));
    CK(hipMemsetAsync(ctx->s_fin + slot0, 0, (size_t)nseq * 4, s));
    CK(hipMemsetAsync(ctx->s_pos + slot0, 0, (size_t)nseq * 4, s));
    CK(hipMemsetAsync(ctx->s_ids + (int64_t)slot0 * ctx->max_new, 0, (size_t)nseq * ctx->max_new * 4, s));
    return prefill_first_tokens(ctx, slot0, nseq, embeds, seq_len, s, "prefill_commit");
}
extern "C" int aur_llm_prefill(aur_ctx* ctx, int32_t slot, void* embeds, int32_t seq_len, void* stream) {
    return aur_llm_prefill_batch(ctx, slot, 1, embeds, seq_len, stream);
}

// ---- decode-step kernel argument builders (shared by the step and by aur_microbench)
static SkinnyArgs mk_dec_qkv(aur_ctx* ctx, int l) {
    const aur_config& g = ctx->cfg;
    const int d = g.llm_hidden;
    SkinnyArgs q{};
    q.ssq_in = ctx->s_ssq_mlp; q.norm_eps = g.llm_rms_eps; q.ssq_zero = ctx->s_ssq_attn;
    q.xf = ctx->d_x; q.W = ctx->ll[l].qkv_w; q.B = ctx->batch; q.b_lo = 0; q.b_hi = ctx->batch; q.Npad = ctx->l_qkv_npad; q.K = d;
    q.n_real = 3 * d; q.mode = SK_QKV; q.q_cols = d; q.k_cols = d; q.hd = ctx->l_hd; q.qbuf = ctx->d_q; q.kv = llm_kv(ctx, l);
    q.rope = ctx->l_rope; q.pos = ctx->s_pos; q.seq_ids = nullptr; q.variant = ctx->skinny_variant_wide;
    q.half_grid = ctx->decode_half;
    return q;
}
static DecAttnArgs mk_dec_attn(aur_ctx* ctx, int l) {
    const aur_config& g = ctx->cfg;
    DecAttnArgs at{};
    at.qbuf = ctx->d_q; at.kv = llm_kv(ctx, l); at.pos = ctx->s_pos; at.seq_ids = nullptr; at.B = ctx->batch; at.heads = g.llm_heads;
    at.hd = ctx->l_hd; at.nsplit = ctx->nsplit; at.pages_per_split = ctx->pps; at.scale = 1.0f / sqrtf((float)ctx->l_hd);
    at.part_o = ctx->d_part_o; at.part_ml = ctx->d_part_ml; at.out_f = ctx->d_attn; at.out_k32 = g.llm_hidden / 32;
    // engines whose capacity alone gives one attention workgroup per CU and still split the context (8-15 slots of 32 heads): the splits of a
    // (sequence, head) are the waves of one workgroup and meet in LDS - bitwise the two-launch result, one launch per layer less
    at.local_splits = (g.max_batch * g.llm_heads >= gemm256_grid_cap() /* one attention workgroup per CU */ && (ctx->nsplit == 2 || ctx->nsplit == 4) && ctx->l_hd == 128 && ctx->attn_local) ? ctx->nsplit : 0;
    return at;
}
static SkinnyArgs mk_dec_o(aur_ctx* ctx, int l) {
    const int d = ctx->cfg.llm_hidden;
    SkinnyArgs o{};
    o.xf = ctx->d_attn; o.W = ctx->ll[l].o_w; o.B = ctx->batch; o.b_lo = 0; o.b_hi = ctx->batch; o.Npad = ctx->l_dpad; o.K = d; o.n_real = d;
    o.mode = SK_ROW; o.xres = ctx->d_x; o.ssq_out = ctx->s_ssq_attn;
    o.variant = ctx->skinny_variant; o.part = (d >= ctx->row_split_min_k || ctx->cfg.max_batch > 64) ? ctx->d_part_row : nullptr;
    o.row_cnt = ctx->fused_reduce ? ctx->d_row_cnt : nullptr; o.half_grid = ctx->decode_half;
    return o;
}
static SkinnyArgs mk_dec_gateup(aur_ctx* ctx, int l) {
    const aur_config& g = ctx->cfg;
    const int d = g.llm_hidden;
    SkinnyArgs gu{};
    gu.ssq_in = ctx->s_ssq_attn; gu.norm_eps = g.llm_rms_eps; gu.ssq_zero = ctx->s_ssq_mlp;
    gu.xf = ctx->d_x; gu.W = ctx->ll[l].gateup_w; gu.B = ctx->batch; gu.b_lo = 0; gu.b_hi = ctx->batch; gu.Npad = ctx->l_gu_npad; gu.K = d;
    gu.n_real = 2 * g.llm_mlp; gu.mode = SK_SILU_MUL; gu.out_f = ctx->d_h; gu.out_k32 = g.llm_mlp / 32; gu.variant = ctx->skinny_variant_wide;
    gu.gu_ks = g.max_batch > 64 ? 1 : 2; gu.half_grid = ctx->decode_half;
    return gu;
}
static SkinnyArgs mk_dec_down(aur_ctx* ctx, int l) {
    const aur_config& g = ctx->cfg;
    const int d = g.llm_hidden;
    SkinnyArgs dn{};
    dn.xf = ctx->d_h; dn.W = ctx->ll[l].down_w; dn.B = ctx->batch; dn.b_lo = 0; dn.b_hi = ctx->batch; dn.Npad = ctx->l_dpad; dn.K = g.llm_mlp;
    dn.n_real = d; dn.mode = SK_ROW; dn.xres = ctx->d_x; dn.ssq_out = ctx->s_ssq_mlp;
    dn.variant = ctx->skinny_variant; dn.part = (g.llm_mlp >= ctx->row_split_min_k || g.max_batch > 64) ? ctx->d_part_row : nullptr;
    dn.row_cnt = ctx->fused_reduce ? ctx->d_row_cnt : nullptr; dn.half_grid = ctx->decode_half;
    return dn;
}

// One thread stores the device's constant-rate clock (wall_clock64: hipDeviceAttributeWallClockRate kHz) for step `n` = ring[0];
// the closing stamp advances the step count.  Two of these bracket ONE layer's attention launch inside the captured decode step when
// option "decode_stamp_layer" is set: the interval they give is that kernel's duration in the loop the caller actually runs (beside a
// front end, on a CU-masked stream, replayed from the graph) plus two kernel boundaries (~3 us).
__global__ void stamp_kernel(unsigned long long* __restrict__ ring, int which) {
    const unsigned long long n = ring[0];
    ring[8 + 2 * (n % AUR_STAMP_RING) + which] = (unsigned long long)wall_clock64();
    if (which) ring[0] = n + 1;
}

static int enqueue_decode_step(aur_ctx* ctx, hipStream_t s, bool instrument) {
    const aur_config& g = ctx->cfg;
    const int B = ctx->batch;
    for (int l = 0; l < g.llm_layers; ++l) {
        CK(launch_skinny(mk_dec_qkv(ctx, l), s));
        {
            const DecAttnArgs at = mk_dec_attn(ctx, l);
            const bool stamp = l == ctx->stamp_layer && ctx->d_stamp && !instrument;
            if (instrument) kev_begin(ctx->kev[0], s);
            if (stamp) hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, s, ctx->d_stamp, 0);
            CK(launch_decode_attention_main(at, s));
            if (stamp) hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, s, ctx->d_stamp, 1);
            if (instrument) kev_end(ctx->kev[0], s);
            CK(launch_decode_attention_combine(at, s));
        }
        CK(launch_skinny(mk_dec_o(ctx, l), s));
        SkinnyArgs gu = mk_dec_gateup(ctx, l);
        if (instrument) kev_begin(ctx->kev[1], s);
        CK(launch_skinny(gu, s));
        if (instrument) kev_end(ctx->kev[1], s);
        CK(launch_skinny(mk_dec_down(ctx, l), s));
    }
    return lm_head_and_advance(ctx, 0, B, 1, -1, s);
}

extern "C" int aur_llm_decode(aur_ctx* ctx, int32_t steps, void* stream) {
    if (!ctx->finalized || ctx->ll.empty() || ctx->batch < 1) return aur_fail(ctx, AUR_ERR_STATE, "aur_llm_decode: call aur_begin_batch / aur_llm_prefill first");
    hipStream_t s = (hipStream_t)stream;
    if (steps <= 0) return AUR_OK;
    if (ctx->capturing) return aur_fail(ctx, AUR_ERR_STATE, "aur_llm_decode inside aur_graph_begin / aur_graph_end: the decode step replays its own graph");
    stage_begin(ctx, "decode", s);
    const bool use_graph = ctx->cfg.use_graph && !ctx->prof;
    if (use_graph) {
        if (ctx->graph_batch != ctx->batch) drop_graphs(ctx);
        hipGraphExec_t& gx = ctx->decode_half ? ctx->graph_h : ctx->graph;
        if (!gx) {
            hipGraph_t gr;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            int rc = enqueue_decode_step(ctx, s, false);
            hipError_t e = hipStreamEndCapture(s, &gr);
            if (rc) return rc;
            if (e != hipSuccess) return aur_fail(ctx, AUR_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
            CK(hipGraphInstantiate(&gx, gr, nullptr, nullptr, 0));
            (void)hipGraphDestroy(gr);
            ctx->graph_batch = ctx->batch;
        }
        for (int i = 0; i < steps; ++i) CK(hipGraphLaunch(gx, s));
    } else {
        for (int i = 0; i < steps; ++i) {
            int rc = enqueue_decode_step(ctx, s, ctx->prof);
            if (rc) return rc;
        }
    }
    stage_end(ctx, "decode", s);
    return AUR_OK;
}

// In-loop intervals of the stamped attention launch (option "decode_stamp_layer"): the last min(cap, steps stamped, AUR_STAMP_RING)
// steps, oldest first, in microseconds.  Synchronises the stream.
extern "C" int aur_decode_stamps_read(aur_ctx* ctx, double* us_out, int32_t cap, int64_t* steps_total_out, int32_t* n_out, void* stream) {
    NO_CAPTURE("aur_decode_stamps_read");
    if (!ctx->d_stamp || !us_out || cap < 1 || !n_out) return aur_fail(ctx, AUR_ERR_ARG, "aur_decode_stamps_read: workspace not set or null argument");
    hipStream_t s = (hipStream_t)stream;
    std::vector<unsigned long long> h(8 + 2 * AUR_STAMP_RING);
    CK(hipMemcpyAsync(h.data(), ctx->d_stamp, h.size() * 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    int dev = 0, khz = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
    if (khz <= 0) return aur_fail(ctx, AUR_ERR_HIP, "hipDeviceAttributeWallClockRate = %d", khz);
    const unsigned long long total = h[0];
    int64_t n = (int64_t)(total < (unsigned long long)AUR_STAMP_RING ? total : AUR_STAMP_RING);
    if (n > cap) n = cap;
    for (int64_t i = 0; i < n; ++i) {
        const unsigned long long step = total - (unsigned long long)n + (unsigned long long)i;
        const unsigned long long t0 = h[8 + 2 * (step % AUR_STAMP_RING)], t1 = h[8 + 2 * (step % AUR_STAMP_RING) + 1];
        us_out[i] = (double)(t1 - t0) * 1e3 / (double)khz;
    }
    if (steps_total_out) *steps_total_out = (int64_t)total;
    *n_out = (int32_t)n;
    return AUR_OK;
}

// ------------------------------------------------------------------------------------------ caller-captured graphs
// The front end of a clip (or of a group of equal-shape clips) is ~600 launches whose arguments depend only on the shape bucket
// (frames, input size, r, prompt structure, target KV sequences) once the inputs sit in fixed staging buffers.  A serving loop captures
// that call sequence once per bucket and replays it with one call - the design the reference tree's serving engine uses for its decode
// step (src/sglang/python/sglang/srt/model_executor/cuda_graph_runner.py:163-279: one captured graph per shape bucket, inputs copied
// into the graph's static buffers before each replay).  Everything between begin and end must be enqueue-only calls of THIS ctx on
// THIS stream from THIS thread (aur_vit_encode*, aur_project_splice, aur_llm_prefill_batch / _stage / _commit, aur_slot_reset /
// _retire / _collect, the kernel-level entry points); calls that synchronise (aur_get_outputs, aur_unfinished, aur_slot_state,
// aur_microbench, aur_decode_stamps_read, aur_set_option on a dec_* knob) and aur_llm_decode (it replays its own graph) are rejected by
// the runtime and fail the capture.  Kernel arguments, including the ctx's gemm_* knobs, are frozen at capture time.
extern "C" int aur_graph_begin(aur_ctx* ctx, void* stream) {
    if (!ctx->finalized) return aur_fail(ctx, AUR_ERR_STATE, "aur_graph_begin: weights not finalized");
    if (ctx->capturing) return aur_fail(ctx, AUR_ERR_STATE, "aur_graph_begin: a capture is already open on this ctx");
    if (ctx->prof) return aur_fail(ctx, AUR_ERR_STATE, "aur_graph_begin: per-stage profiling is on (aur_profile_enable): its events cannot be captured");
    CK(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    ctx->capturing = true;
    ctx->cap_stream = (hipStream_t)stream;
    return AUR_OK;
}
// Ends the capture opened by aur_graph_begin (always - also after a failed call in between) and instantiates the graph.
// *graph_out: handle >= 1 for aur_graph_launch; *nodes_out (may be NULL): number of graph nodes.
extern "C" int aur_graph_end(aur_ctx* ctx, void* stream, int32_t* graph_out, int64_t* nodes_out) {
    if (!ctx->capturing) return aur_fail(ctx, AUR_ERR_STATE, "aur_graph_end: no capture is open");
    if ((hipStream_t)stream != ctx->cap_stream) return aur_fail(ctx, AUR_ERR_ARG, "aur_graph_end: not the stream aur_graph_begin was given");
    ctx->capturing = false;
    hipGraph_t gr = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &gr);
    if (e != hipSuccess || !gr) {
        // a call of the CALLER's inside the capture synchronised: make sure the stream has really left capture mode and drop the sticky error
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
            hipGraph_t g2 = nullptr;
            (void)hipStreamEndCapture((hipStream_t)stream, &g2);
            if (g2) (void)hipGraphDestroy(g2);
        }
        (void)hipGetLastError();
        return aur_fail(ctx, AUR_ERR_HIP, "hipStreamEndCapture: %s (a call inside the capture synchronised or failed)", hipGetErrorString(e));
    }
    if (!graph_out) {
        (void)hipGraphDestroy(gr);
        return aur_fail(ctx, AUR_ERR_ARG, "aur_graph_end: graph_out is NULL");
    }
    size_t n = 0;
    (void)hipGraphGetNodes(gr, nullptr, &n);
    aur_ctx::UserGraph ug;
    e = hipGraphInstantiate(&ug.exec, gr, nullptr, nullptr, 0);
    (void)hipGraphDestroy(gr);
    if (e != hipSuccess) return aur_fail(ctx, AUR_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
    ug.nodes = (int64_t)n;
    size_t slot = 0;
    while (slot < ctx->ugraphs.size() && ctx->ugraphs[slot].exec) ++slot;
    if (slot == ctx->ugraphs.size()) ctx->ugraphs.push_back(ug);
    else ctx->ugraphs[slot] = ug;
    *graph_out = (int32_t)slot + 1;
    if (nodes_out) *nodes_out = ug.nodes;
    return AUR_OK;
}
extern "C" int aur_graph_launch(aur_ctx* ctx, int32_t graph, void* stream) {
    if (graph < 1 || graph > (int32_t)ctx->ugraphs.size() || !ctx->ugraphs[graph - 1].exec) return aur_fail(ctx, AUR_ERR_ARG, "aur_graph_launch: unknown graph %d", graph);
    if (ctx->capturing) return aur_fail(ctx, AUR_ERR_STATE, "aur_graph_launch: a capture is open on this ctx");
    CK(hipGraphLaunch(ctx->ugraphs[graph - 1].exec, (hipStream_t)stream));
    return AUR_OK;
}
// The caller guarantees that no launch of the graph is still queued or running (its own event / stream synchronisation).
extern "C" int aur_graph_destroy(aur_ctx* ctx, int32_t graph) {
    if (graph < 1 || graph > (int32_t)ctx->ugraphs.size() || !ctx->ugraphs[graph - 1].exec) return aur_fail(ctx, AUR_ERR_ARG, "aur_graph_destroy: unknown graph %d", graph);
    (void)hipGraphExecDestroy(ctx->ugraphs[graph - 1].exec);
    ctx->ugraphs[graph - 1] = aur_ctx::UserGraph{};
    return AUR_OK;
}

extern "C" int aur_get_outputs(aur_ctx* ctx, int32_t* ids_host, int32_t* lens_host, void* stream) {
    NO_CAPTURE("aur_get_outputs");
    hipStream_t s = (hipStream_t)stream;
    if (ids_host) CK(hipMemcpyAsync(ids_host, ctx->s_ids, (size_t)ctx->batch * ctx->max_new * 4, hipMemcpyDeviceToHost, s));
    if (lens_host) CK(hipMemcpyAsync(lens_host, ctx->s_len, (size_t)ctx->batch * 4, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    return AUR_OK;
}
extern "C" int aur_unfinished(aur_ctx* ctx, int32_t* count_host, void* stream) {
    NO_CAPTURE("aur_unfinished");
    hipStream_t s = (hipStream_t)stream;
    std::vector<int32_t> fin(ctx->batch), len(ctx->batch);
    CK(hipMemcpyAsync(fin.data(), ctx->s_fin, (size_t)ctx->batch * 4, hipMemcpyDeviceToHost, s));
    CK(hipMemcpyAsync(len.data(), ctx->s_len, (size_t)ctx->batch * 4, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    int n = 0;
    for (int b = 0; b < ctx->batch; ++b) n += (!fin[b] && len[b] < ctx->max_new) ? 1 : 0;
    *count_host = n;
    return AUR_OK;
}
// ---- per-slot control for continuous batching: slots of the active batch are independent sequences
static int slot_check(aur_ctx* ctx, int32_t slot, const char* who) {
    if (ctx->batch < 1) return aur_fail(ctx, AUR_ERR_STATE, "%s: no active batch (aur_begin_batch)", who);
    if (slot < 0 || slot >= ctx->batch) return aur_fail(ctx, AUR_ERR_ARG, "%s: slot %d outside the batch of %d", who, slot, ctx->batch);
    return AUR_OK;
}
extern "C" int aur_slot_reset(aur_ctx* ctx, int32_t slot, void* stream) {
    if (int rc = slot_check(ctx, slot, "aur_slot_reset")) return rc;
    hipStream_t s = (hipStream_t)stream;
    CK(hipMemsetAsync(ctx->s_len + slot, 0, 4, s));
    CK(hipMemsetAsync(ctx->s_fin + slot, 0, 4, s));
    CK(hipMemsetAsync(ctx->s_pos + slot, 0, 4, s));
    CK(hipMemsetAsync(ctx->s_ids + (int64_t)slot * ctx->max_new, 0, (size_t)ctx->max_new * 4, s));
    return AUR_OK;
}
extern "C" int aur_slot_retire(aur_ctx* ctx, int32_t slot, void* stream) {
    if (int rc = slot_check(ctx, slot, "aur_slot_retire")) return rc;
    CK(hipMemsetD32Async((hipDeviceptr_t)(ctx->s_fin + slot), 1, 1, (hipStream_t)stream));      // a fill node: no host source, no synchronisation
    return AUR_OK;
}
extern "C" int aur_slot_state(aur_ctx* ctx, int32_t* lens_host, int32_t* finished_host, void* stream) {
    NO_CAPTURE("aur_slot_state");
    if (ctx->batch < 1) return aur_fail(ctx, AUR_ERR_STATE, "aur_slot_state: no active batch");
    hipStream_t s = (hipStream_t)stream;
    if (lens_host) CK(hipMemcpyAsync(lens_host, ctx->s_len, (size_t)ctx->batch * 4, hipMemcpyDeviceToHost, s));
    if (finished_host) CK(hipMemcpyAsync(finished_host, ctx->s_fin, (size_t)ctx->batch * 4, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    return AUR_OK;
}
extern "C" int aur_slot_collect(aur_ctx* ctx, int32_t slot0, int32_t nslots, int32_t* ids_dev, int32_t* lens_dev, void* stream) {
    if (int rc = slot_check(ctx, slot0, "aur_slot_collect")) return rc;
    if (nslots < 1 || slot0 + nslots > ctx->batch || !ids_dev || !lens_dev) return aur_fail(ctx, AUR_ERR_ARG, "aur_slot_collect: slots [%d, %d) outside the batch of %d", slot0, slot0 + nslots, ctx->batch);
    hipStream_t s = (hipStream_t)stream;
    CK(hipMemcpyAsync(ids_dev, ctx->s_ids + (int64_t)slot0 * ctx->max_new, (size_t)nslots * ctx->max_new * 4, hipMemcpyDeviceToDevice, s));
    CK(hipMemcpyAsync(lens_dev, ctx->s_len + slot0, (size_t)nslots * 4, hipMemcpyDeviceToDevice, s));
    return AUR_OK;
}
extern "C" int aur_copy_logits(aur_ctx* ctx, float* dst_dev, void* stream) {
    if (ctx->batch < 1 || !dst_dev) return aur_fail(ctx, AUR_ERR_STATE, "aur_copy_logits: no active batch");
    CK(hipMemcpyAsync(dst_dev, ctx->d_logits, (size_t)ctx->batch * ctx->cfg.llm_vocab * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return AUR_OK;
}

// ------------------------------------------------------------------------------------------ tuning / microbench
extern "C" int aur_set_option(aur_ctx* ctx, const char* name, int64_t value) {
    NO_CAPTURE("aur_set_option");
    if (!strcmp(name, "prefill_prune_last")) {          // 1 (default): the last prefill layer computes Q / attention / MLP for each sequence's last 128 rows only
        ctx->prune_last = value ? 1 : 0;
        return AUR_OK;
    }
    if (!strcmp(name, "tome_fused_ln")) {
        ctx->tome_fused_ln = value ? 1 : 0;
        return AUR_OK;
    }
    if (!strcmp(name, "dec_attn_local")) ctx->attn_local = value ? 1 : 0;
    else if (!strcmp(name, "skinny_row_split_min_k")) ctx->row_split_min_k = (int)value;
    else if (!strncmp(name, "gemm_", 5) || !strcmp(name, "microbench_prefill_nseq")) {
        // GEMM knobs: no GEMM is part of the captured decode step (its projections are the skinny kernels), so the graphs stay
        // valid - the serving schedule changes gemm_max_wgs around every front end (caption_stream) while decode launches are queued
        if (!strcmp(name, "gemm_mode")) ctx->gemm_mode = (int)value;
        else if (!strcmp(name, "gemm_max_wgs")) ctx->gemm_max_wgs = (int)value;
        else if (!strcmp(name, "gemm_nt_out")) ctx->gemm_nt_out = value < 0 ? -1 : (value ? 1 : 0);
        else if (!strcmp(name, "gemm_tail_split")) ctx->gemm_tail_split = value ? 1 : 0;
        else if (!strcmp(name, "microbench_prefill_nseq")) ctx->mb_nseq = (value >= 1 && value <= ctx->cfg.max_batch) ? (int)value : 1;
        else return aur_fail(ctx, AUR_ERR_ARG, "unknown option '%s'", name);
        return AUR_OK;
    }
    else if (!strcmp(name, "dec_attn_pps")) {
        if (value < 1) return aur_fail(ctx, AUR_ERR_ARG, "dec_attn_pps must be >= 1");
        ctx->pps = (int)value;
        ctx->nsplit = (ctx->l_max_pages + (int)value - 1) / (int)value;
    } else if (!strcmp(name, "decode_fused_reduce")) {
        // 1: the split-K residual projections (o, down) sum their partials in the projection kernel (the last-arriving split reduces;
        // decode.hip "split-K with the reduce IN the kernel"); 0: a second launch does (skinny_row_reduce_kernel).  Bitwise the same tokens.
        if (value < 0 || value > 1) return aur_fail(ctx, AUR_ERR_ARG, "decode_fused_reduce must be 0 or 1");
        ctx->fused_reduce = (int)value;
    } else if (!strcmp(name, "decode_stamp_layer")) {
        // value >= 0: two one-thread stamp launches bracket that layer's attention launch in every decode step (aur_decode_stamps_read);
        // -1 (default): none.  The step's graphs are re-captured.
        if (value < -1 || value >= ctx->cfg.llm_layers) return aur_fail(ctx, AUR_ERR_ARG, "decode_stamp_layer must be -1 or a layer index");
        ctx->stamp_layer = (int)value;
    } else if (!strcmp(name, "decode_half_grid")) {
        // the following aur_llm_decode calls go to a stream that owns half of the CUs: QKV / gate-up launch half as many workgroups
        // with twice the tiles (decode.hip launch_skx_nb); bitwise the same tokens.  Keeps both captured graphs.
        ctx->decode_half = value ? 1 : 0;
        return AUR_OK;
    } else return aur_fail(ctx, AUR_ERR_ARG, "unknown option '%s'", name);
    // kernel arguments are frozen in the captured graphs.  A graph may still be queued on a stream this call does not know: wait
    // for the device before destroying it (tuning path only - nothing in the serving schedules changes these knobs)
    if (ctx->graph || ctx->graph_h) (void)hipDeviceSynchronize();
    drop_graphs(ctx);
    return AUR_OK;
}

// Time ONE kernel of the path in isolation on the current generation state (after aur_llm_prefill): `iters`
// back-to-back launches cycling through the layers (so weights stream from HBM, not from the 256 MiB MALL),
// bracketed by HIP events on `stream`.  Synchronises.  us_out = mean microseconds per launch.
extern "C" int aur_microbench(aur_ctx* ctx, const char* kernel, int32_t iters, double* us_out, void* stream) {
    NO_CAPTURE("aur_microbench");
    if (!ctx->finalized || ctx->ll.empty() || ctx->batch < 1 || ctx->last_prefill_len < 1)
        return aur_fail(ctx, AUR_ERR_STATE, "aur_microbench: prefill a batch first");
    hipStream_t s = (hipStream_t)stream;
    const aur_config& g = ctx->cfg;
    const int nl = g.llm_layers, d = g.llm_hidden;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](int i) -> int {
        const int l = i % nl;
        if (!strcmp(kernel, "dec_qkv")) { CK(launch_skinny(mk_dec_qkv(ctx, l), s)); }
        else if (!strcmp(kernel, "dec_attn")) { CK(launch_decode_attention(mk_dec_attn(ctx, l), s)); }
        else if (!strcmp(kernel, "dec_o")) { CK(launch_skinny(mk_dec_o(ctx, l), s)); }
        else if (!strcmp(kernel, "dec_gateup")) { CK(launch_skinny(mk_dec_gateup(ctx, l), s)); }
        else if (!strcmp(kernel, "dec_down")) { CK(launch_skinny(mk_dec_down(ctx, l), s)); }
        else if (!strcmp(kernel, "dec_lm_head")) {
            SkinnyArgs h{};
            h.ssq_in = ctx->s_ssq_mlp; h.norm_eps = g.llm_rms_eps;
            h.xf = ctx->d_x; h.W = ctx->l_head_w; h.B = ctx->batch; h.b_lo = 0; h.b_hi = ctx->batch; h.Npad = ctx->l_vocab_pad; h.K = d;
            h.n_real = g.llm_vocab; h.mode = SK_LOGITS; h.out32 = ctx->d_logits; h.variant = ctx->skinny_variant;
            CK(launch_skinny(h, s));
        } else {
            int bit = !strcmp(kernel, "pre_norm") ? 1 : !strcmp(kernel, "pre_qkv") ? 2 : !strcmp(kernel, "pre_attn") ? 4 : !strcmp(kernel, "pre_o") ? 8
                      : !strcmp(kernel, "pre_gateup") ? 32 : !strcmp(kernel, "pre_down") ? 64 : 0;
            if (!bit) return aur_fail(ctx, AUR_ERR_ARG, "unknown kernel '%s'", kernel);
            // scratch residual stream: l_p1 (projector scratch) is free after the splice
            return prefill_layer(ctx, l, bit, 0, ctx->mb_nseq, ctx->l_p1, ctx->last_prefill_len, s);
        }
        return AUR_OK;
    };
    for (int i = 0; i < 3; ++i) { int rc = run(i); if (rc) return rc; }
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) { int rc = run(i + 3); if (rc) return rc; }
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *us_out = 1e3 * ms / iters;
    return AUR_OK;
}

extern "C" int aur_select_bank(aur_ctx* ctx, int32_t bank) {
    if (bank < 0 || bank >= ctx->nbanks) return aur_fail(ctx, AUR_ERR_ARG, "bank %d outside [0, %d) (aur_config.num_banks)", bank, ctx->nbanks);
    if (!ctx->ws) return aur_fail(ctx, AUR_ERR_STATE, "aur_select_bank: workspace not set");
    aur_ctx::Bank& o = ctx->banks[ctx->cur_bank];
    o.batch = ctx->batch; o.max_new = ctx->max_new; o.eos = ctx->eos; o.graph = ctx->graph; o.graph_h = ctx->graph_h; o.graph_batch = ctx->graph_batch;
    const aur_ctx::Bank& K = ctx->banks[bank];
    ctx->cur_bank = bank;
    ctx->d_x = K.d_x; ctx->s_ssq_mlp = K.s_ssq_mlp; ctx->s_ssq_attn = K.s_ssq_attn; ctx->d_logits = K.d_logits;
    ctx->s_pos = K.s_pos; ctx->s_ids = K.s_ids; ctx->s_len = K.s_len; ctx->s_fin = K.s_fin; ctx->ptab_cur = K.ptab;
    ctx->batch = K.batch; ctx->max_new = K.max_new; ctx->eos = K.eos; ctx->graph = K.graph; ctx->graph_h = K.graph_h; ctx->graph_batch = K.graph_batch;
    return AUR_OK;
}
