// gemv_lab - stand-alone A/B bench for the B=64 decode projections (y[b, n] = sum_k x[b, k] W[n, k], b < 64) on gfx950.
//
// Not part of the product: a lab that times kernel STRUCTURES against each other inside one process (guide rule 24) on the
// real AuroraCap-7B shapes, with weights cycling through > 256 MiB of copies so every launch streams from HBM, and checks
// each variant against a naive reference.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemv_lab/gemv_lab.hip -o tools/gemv_lab/gemv_lab
//   ./tools/gemv_lab/gemv_lab [iters]
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define FRAG 512
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))
#define CKH(x)                                                                      \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

__device__ __forceinline__ f4 mfma16(h8 a, h8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds(GPTR(g), LPTR(l), 16, 0, 0); }
#define BAR()                                  \
    do {                                       \
        asm volatile("" ::: "memory");         \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
        asm volatile("" ::: "memory");         \
    } while (0)

__global__ void fill_kernel(half_t* p, size_t n, uint32_t seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        p[i] = (half_t)(((float)(h & 0xffff) / 32768.0f - 1.0f) * scale);
    }
}

// naive reference straight from the fragment layouts: out_ll[tile][nb][lane][i] = y[n = tile*16 + 4g + i][b = nb*16 + c]
__global__ void ref_kernel(const half_t* W, const half_t* xf, float* out, int N16, int K32, int NB) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (tile, nb, lane, i)
    if (id >= (int64_t)N16 * NB * 64 * 4) return;
    const int i = id & 3, lane = (id >> 2) & 63, nb = (int)((id >> 8) % NB), tile = (int)((id >> 8) / NB);
    const int g = lane >> 4, c = lane & 15;
    const int r = 4 * g + i;
    float s = 0.f;
    for (int k32 = 0; k32 < K32; ++k32)
        for (int gg = 0; gg < 4; ++gg)
            for (int j = 0; j < 8; ++j)
                s += (float)W[(((int64_t)tile * K32 + k32) * 64 + gg * 16 + r) * 8 + j] * (float)xf[(((int64_t)nb * K32 + k32) * 64 + gg * 16 + c) * 8 + j];
    out[id] = s;
}

struct Args {
    const half_t* W;
    const half_t* xf;
    float* out;        // lane-linear [S][N16][4][64][4]
    int N16, K32, S;
    int tiles_lo, n_hi;     // skx: WG c owns tiles_lo + (c < n_hi) consecutive n16 tiles
};

// ------------------------------------------------------------------------------------------------ A: current structure
// one workgroup = NT n16 tiles over the full K; NW waves split K interleaved; x fragments per wave straight from L2.
template <int NT, int NW, int U, int XM, int ROT>
__global__ __launch_bounds__(64 * NW) void cur_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NB = 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K32 = a.K32;
    const int tile0 = blockIdx.x * NT;
    f4 acc[NT][NB];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = f4{0.f, 0.f, 0.f, 0.f};
    const int nit = K32 / NW;                 // multiple of U
    const int rot = ROT ? (int)((blockIdx.x * 5u) % (unsigned)(nit / U)) * U : 0;
    const half_t* wp[NT];
    const half_t* xp[NB];
#pragma unroll
    for (int t = 0; t < NT; ++t) wp[t] = a.W + ((int64_t)(tile0 + t) * K32 + w) * FRAG + lane * 8;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) xp[nb] = a.xf + ((int64_t)nb * K32 + w) * FRAG + lane * 8;
    constexpr int64_t STEP = (int64_t)NW * FRAG;
    auto kix = [&](int i) { i = i < nit ? i : nit - 1; int k = i + rot; return k >= nit ? k - nit : k; };
    h8 cw[U][NT], cx[U][NB], nw[U][NT], nx[U][NB];
    h8 xconst;
#pragma unroll
    for (int j = 0; j < 8; ++j) xconst[j] = (half_t)(0.001f * (lane + j));
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int t = 0; t < NT; ++t) cw[u][t] = __builtin_nontemporal_load((const h8*)(wp[t] + kix(u) * STEP));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) cx[u][nb] = XM ? xconst : *(const h8*)(xp[nb] + kix(u) * STEP);
    }
    for (int i0 = 0; i0 < nit; i0 += U) {
        const bool more = i0 + U < nit;
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int t = 0; t < NT; ++t) nw[u][t] = __builtin_nontemporal_load((const h8*)(wp[t] + kix(i0 + U + u) * STEP));
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) nx[u][nb] = XM ? xconst : *(const h8*)(xp[nb] + kix(i0 + U + u) * STEP);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u >= nit) break;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[t][nb] = mfma16(cw[u][t], cx[u][nb], acc[t][nb]);
        }
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
    float* red = (float*)smem;
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
            *(f4*)(a.out + (((int64_t)(tile0 + t) * NB + nb) * 64 + lane) * 4) = s;
        }
}

// ------------------------------------------------------------------------------------------------ B: x through LDS
// One workgroup per CU.  It owns `ntile` consecutive n16 tiles and the k32 range of split blockIdx.y.  A LOADER wave DMAs the
// x fragments of that range, chunk by chunk (KC k32 tiles x 4 column groups = KC * 4 KiB), into a ring of NBUF LDS buffers -
// once per workgroup, up to NBUF-1 chunks in flight behind a COUNTED vmcnt; T_MAX x KS consumer waves (wave -> tile w / KS,
// k phase w % KS inside every chunk) stream their weight fragments straight into VGPRs (U-deep register pipeline that runs
// across chunk boundaries) and read x with ds_read_b128.  One s_barrier per chunk hands chunk ci to the consumers and chunk
// ci-1's buffer back to the loader.  Chunks are walked in a per-workgroup rotated order so that the 256 workgroups do not hit
// the same L2 lines at the same time.
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int T_MAX, int KS, int KC, int NBUF, int U, int ROT, int NL, int NB, int LM>
__global__ __launch_bounds__(64 * (T_MAX * KS + NL)) void skx_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];         // NBUF x KC x NB KiB
    constexpr int NCW = T_MAX * KS, Q = KC / KS;                        // Q steps per wave per chunk
    constexpr int CHUNK = KC * NB * 1024, LB = LM == 1 ? 2 : NBUF;           // LDS ring depth
    constexpr int PIECES = KC * NB / NL;                                // DMA instructions per chunk per loader wave
    static_assert((KC * NB) % NL == 0, "pieces per chunk must split evenly over the loader waves");
    static_assert(KC % KS == 0 && (Q % U == 0 || U % Q == 0), "chunk steps per wave vs pipeline depth");
    static_assert((NBUF - 1) * PIECES <= 63, "vmcnt is a 6-bit counter");
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = blockIdx.x, s = blockIdx.y;
    const int ntile = a.tiles_lo + (c < a.n_hi ? 1 : 0);
    const int tile0 = c * a.tiles_lo + (c < a.n_hi ? c : a.n_hi);
    const int K32 = a.K32, KE = K32 / a.S, kb = s * KE;
    const int nchunk = (KE + KC - 1) / KC;
    const int rot = ROT ? (int)((c * 3u + s) % (unsigned)nchunk) : 0;
    auto chunk_of = [&](int ci) { int x = ci + rot; return x >= nchunk ? x - nchunk : x; };

    if (w >= NCW && LM == 1) {                                 // ---------------- loader waves, register staged: global_load -> ds_write
        // R chunks in flight in VGPRs (R x PIECES x 4 registers), the LDS ring only double-buffers (any NBUF >= 2)
        constexpr int R = NBUF - 1;
        const int lw = w - NCW;
        h8 rg[R][PIECES];
        auto src = [&](int ci, int q) {
            const int piece = q * NL + lw, kk = piece / NB, nb = piece % NB;
            int kr = chunk_of(ci < nchunk ? ci : nchunk - 1) * KC + kk;
            kr = kr < KE ? kr : KE - 1;
            return (const h8*)(a.xf + ((int64_t)nb * K32 + kb + kr) * FRAG + lane * 8);
        };
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < PIECES; ++q) rg[r][q] = *src(r, q);
        for (int c0 = 0; c0 < nchunk; c0 += R) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int ci = c0 + r;
                if (ci < nchunk) {
                    wait_vm<(R - 1) * PIECES>();
                    char* buf = smem + (ci % LB) * CHUNK + lane * 16;
#pragma unroll
                    for (int q = 0; q < PIECES; ++q) *(h8*)(buf + (q * NL + lw) * 1024) = rg[r][q];
#pragma unroll
                    for (int q = 0; q < PIECES; ++q) rg[r][q] = *src(ci + R, q);        // always issued: the vmcnt above counts them
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    BAR();
                }
            }
        }
        return;
    }
    if (w >= NCW) {                                            // ---------------- loader waves (piece q of a chunk -> loader q % NL)
        const int lw = w - NCW;
        auto issue = [&](int ci) {
            const int ch = chunk_of(ci);
            char* buf = smem + (ci % LB) * CHUNK;
#pragma unroll
            for (int q = 0; q < PIECES; ++q) {
                const int piece = q * NL + lw, kk = piece / NB, nb = piece % NB;
                int kr = ch * KC + kk;
                kr = kr < KE ? kr : KE - 1;                    // always PIECES instructions per chunk: the vmcnt below counts them
                glds16(a.xf + ((int64_t)nb * K32 + kb + kr) * FRAG + lane * 8, buf + piece * 1024);
            }
        };
        int next = 0;
        for (; next < NBUF - 1 && next < nchunk; ++next) issue(next);
        for (int ci = 0; ci < nchunk; ++ci) {
            const int infl = next - ci - 1;                    // chunks that may stay in flight once chunk ci has landed (<= NBUF - 2)
            switch (infl) {
                case 0: wait_vm<0>(); break;
                case 1: wait_vm<PIECES>(); break;
                case 2: wait_vm<(NBUF > 3 ? 2 : 0) * PIECES>(); break;
                case 3: wait_vm<(NBUF > 4 ? 3 : 0) * PIECES>(); break;
                case 4: wait_vm<(NBUF > 5 ? 4 : 0) * PIECES>(); break;
                case 5: wait_vm<(NBUF > 6 ? 5 : 0) * PIECES>(); break;
                case 6: wait_vm<(NBUF > 7 ? 6 : 0) * PIECES>(); break;
                default: wait_vm<0>(); break;
            }
            BAR();                                             // chunk ci is in LDS; chunk ci-1's buffer is free again
            if (next < nchunk) {
                issue(next);
                ++next;
            }
        }
        return;
    }
    const int t = w / KS, p = w % KS;
    if (t >= ntile) return;                                    // idle consumer (ended waves do not count at s_barrier)
    const int tile = tile0 + t;
    f4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = f4{0.f, 0.f, 0.f, 0.f};
    const half_t* wbase = a.W + ((int64_t)tile * K32 + kb) * FRAG + lane * 8;
    const int NS = nchunk * Q;                                 // steps of this wave (the last chunk may hold invalid steps)
    // step j -> chunk slot ci = j / Q, kk = (j % Q) * KS + p; k index (clamped: invalid steps re-read the last fragment, x = 0)
    auto kof = [&](int j, bool& valid) {
        const int ci = j / Q, kk = (j % Q) * KS + p;
        const int kr = chunk_of(ci < nchunk ? ci : nchunk - 1) * KC + kk;
        valid = j < NS && kr < KE;
        return valid ? kr : KE - 1;
    };
    h8 cw[U], nw[U];
    bool vv;
#pragma unroll
    for (int u = 0; u < U; ++u) cw[u] = __builtin_nontemporal_load((const h8*)(wbase + (int64_t)kof(u, vv) * FRAG));
    for (int j0 = 0; j0 < NS; j0 += U) {
        const bool more = j0 + U < NS;
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) nw[u] = __builtin_nontemporal_load((const h8*)(wbase + (int64_t)kof(j0 + U + u, vv) * FRAG));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u;
            if (j < NS && j % Q == 0) BAR();                   // chunk j / Q landed (and chunk j / Q - 1 is released)
            const int ci = j / Q, kk = (j % Q) * KS + p;
            const char* buf = smem + (ci % LB) * CHUNK + lane * 16;
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
            for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma16(cw[u], xr[nb], acc[nb]);
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) cw[u] = nw[u];
        }
    }
    if (KS > 1) {                                              // fixed-order reduction over the KS k phases of a tile
        BAR();                                                 // every consumer is done reading x (the loader has exited)
        float* red = (float*)smem;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) *(f4*)(red + ((w * NB + nb) * 64 + lane) * 4) = acc[nb];
        BAR();
        if (p != 0) return;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f4 sum = acc[nb];
#pragma unroll
            for (int pp = 1; pp < KS; ++pp) {
                const f4 q = *(const f4*)(red + (((w + pp) * NB + nb) * 64 + lane) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) sum[i] += q[i];
            }
            acc[nb] = sum;
        }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) *(f4*)(a.out + ((((int64_t)s * a.N16 + tile) * NB + nb) * 64 + lane) * 4) = acc[nb];
}

// ------------------------------------------------------------------------------------------------ B2: x through LDS, flags instead of barriers
// Same data flow as skx_kernel, but no s_barrier in the K loop: the loader waves publish a chunk by bumping ready[buf] in LDS once
// their DMAs have landed, consumers poll that word before the chunk's first read and bump done[buf] after its last one; a loader
// re-fills buffer b for chunk ci only when done[b] shows that every consumer has left chunk ci - NBUF.  Waves drift apart by up to
// NBUF - 1 chunks instead of meeting 16-32 times per kernel.
__device__ __forceinline__ unsigned lds_load_u32(const unsigned* p) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(uintptr_t)p) : "memory");
    return v;
}
__device__ __forceinline__ void lds_add_u32(unsigned* p, unsigned x) {
    asm volatile("ds_add_u32 %0, %1" ::"v"((unsigned)(uintptr_t)p), "v"(x) : "memory");
}

template <int T_MAX, int KS, int KC, int NBUF, int U, int NL, int NB>
__global__ __launch_bounds__(64 * (T_MAX * KS + NL)) void skf_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];         // NBUF x KC x NB KiB, then the flag words
    constexpr int NCW = T_MAX * KS, Q = KC / KS;
    constexpr int PIECES = KC * NB / NL;
    constexpr int CHUNK = KC * NB * 1024;
    static_assert((KC * NB) % NL == 0 && KC % KS == 0 && (Q % U == 0 || U % Q == 0), "geometry");
    static_assert((NBUF - 1) * PIECES <= 63, "vmcnt is a 6-bit counter");
    unsigned* ready = (unsigned*)(smem + NBUF * CHUNK);                 // [NBUF] loader waves that have landed their pieces, cumulative
    unsigned* done = ready + NBUF;                                      // [NBUF] consumer waves that have left the buffer, cumulative
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = blockIdx.x, s = blockIdx.y;
    const int ntile = a.tiles_lo + (c < a.n_hi ? 1 : 0);
    const int tile0 = c * a.tiles_lo + (c < a.n_hi ? c : a.n_hi);
    const int K32 = a.K32, KE = K32 / a.S, kb = s * KE;
    const int nchunk = (KE + KC - 1) / KC;
    const unsigned nca = (unsigned)(ntile * KS);                        // active consumer waves
    if (tid < 2 * NBUF) ready[tid] = 0u;
    __syncthreads();

    if (w >= NCW) {                                            // ---------------- loader waves
        const int lw = w - NCW;
        auto issue = [&](int ci) {
            const int b = ci % NBUF;
            if (ci >= NBUF) {                                  // the buffer's previous chunk must have been left by every consumer
                const unsigned want = nca * (unsigned)(ci / NBUF);
                while (lds_load_u32(done + b) < want) __builtin_amdgcn_s_sleep(1);
            }
            char* buf = smem + b * CHUNK;
#pragma unroll
            for (int q = 0; q < PIECES; ++q) {
                const int piece = q * NL + lw, kk = piece / NB, nb = piece % NB;
                int kr = ci * KC + kk;
                kr = kr < KE ? kr : KE - 1;
                glds16(a.xf + ((int64_t)nb * K32 + kb + kr) * FRAG + lane * 8, buf + piece * 1024);
            }
        };
        int next = 0;
        for (; next < NBUF - 1 && next < nchunk; ++next) issue(next);
        for (int ci = 0; ci < nchunk; ++ci) {
            switch (next - ci - 1) {
                case 0: wait_vm<0>(); break;
                case 1: wait_vm<PIECES>(); break;
                case 2: wait_vm<(NBUF > 3 ? 2 : 0) * PIECES>(); break;
                case 3: wait_vm<(NBUF > 4 ? 3 : 0) * PIECES>(); break;
                case 4: wait_vm<(NBUF > 5 ? 4 : 0) * PIECES>(); break;
                case 5: wait_vm<(NBUF > 6 ? 5 : 0) * PIECES>(); break;
                case 6: wait_vm<(NBUF > 7 ? 6 : 0) * PIECES>(); break;
                default: wait_vm<0>(); break;
            }
            if (lane == 0) lds_add_u32(ready + ci % NBUF, 1u);
            if (next < nchunk) {
                issue(next);
                ++next;
            }
        }
        return;
    }
    const int t = w / KS, p = w % KS;
    if (t >= ntile) return;
    const int tile = tile0 + t;
    f4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = f4{0.f, 0.f, 0.f, 0.f};
    const half_t* wbase = a.W + ((int64_t)tile * K32 + kb) * FRAG + lane * 8;
    const int NS = nchunk * Q;
    auto kof = [&](int j, bool& valid) {
        const int kr = (j / Q) * KC + (j % Q) * KS + p;
        valid = j < NS && kr < KE;
        return valid ? kr : KE - 1;
    };
    h8 cw[U], nw[U];
    bool vv;
#pragma unroll
    for (int u = 0; u < U; ++u) cw[u] = __builtin_nontemporal_load((const h8*)(wbase + (int64_t)kof(u, vv) * FRAG));
    for (int j0 = 0; j0 < NS; j0 += U) {
        const bool more = j0 + U < NS;
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) nw[u] = __builtin_nontemporal_load((const h8*)(wbase + (int64_t)kof(j0 + U + u, vv) * FRAG));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u;
            const int ci = j / Q, kk = (j % Q) * KS + p;
            if (j < NS && j % Q == 0) {                        // first step of chunk ci: leave the previous buffer, wait for this one
                if (ci > 0 && lane == 0) lds_add_u32(done + (ci - 1) % NBUF, 1u);
                const unsigned want = (unsigned)NL * (unsigned)(ci / NBUF + 1);
                while (lds_load_u32(ready + ci % NBUF) < want) __builtin_amdgcn_s_sleep(0);
            }
            const char* buf = smem + (ci % NBUF) * CHUNK + lane * 16;
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
            for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma16(cw[u], xr[nb], acc[nb]);
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) cw[u] = nw[u];
        }
    }
    if (KS > 1) {                                              // fixed-order reduction over the k phases (the loaders have exited)
        BAR();
        float* red = (float*)smem;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) *(f4*)(red + ((w * NB + nb) * 64 + lane) * 4) = acc[nb];
        BAR();
        if (p != 0) return;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f4 sum = acc[nb];
#pragma unroll
            for (int pp = 1; pp < KS; ++pp) {
                const f4 q = *(const f4*)(red + (((w + pp) * NB + nb) * 64 + lane) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) sum[i] += q[i];
            }
            acc[nb] = sum;
        }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) *(f4*)(a.out + ((((int64_t)s * a.N16 + tile) * NB + nb) * 64 + lane) * 4) = acc[nb];
}

// split-K second stage: out[0] = sum_s part[s] in fixed order (stands in for the residual / norm epilogue of the product)
__global__ __launch_bounds__(256) void reduce_kernel(const float* part, float* out, int64_t n4, int S) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f4 v[16];
#pragma unroll
    for (int s = 0; s < 16; ++s)
        if (s < S) v[s] = *(const f4*)(part + ((int64_t)s * n4 + i) * 4);
    f4 acc = v[0];
#pragma unroll
    for (int s = 1; s < 16; ++s)
        if (s < S) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += v[s][k];
        }
    *(f4*)(out + i * 4) = acc;
}

// ------------------------------------------------------------------------------------------------ C: pure stream ceiling
template <int U>
__global__ __launch_bounds__(512) void stream_kernel(const half_t* W, int64_t frags, float* out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 6), nwave = (int64_t)gridDim.x * 8;
    const int64_t per = frags / nwave;
    const half_t* p = W + wave * per * FRAG + lane * 8;
    f4 acc = f4{0.f, 0.f, 0.f, 0.f};
    h8 xc;
#pragma unroll
    for (int j = 0; j < 8; ++j) xc[j] = (half_t)(0.001f * (lane + j));
    for (int64_t i = 0; i < per; i += U) {
        h8 t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] = __builtin_nontemporal_load((const h8*)(p + ((i + u) < per ? (i + u) : per - 1) * FRAG));
#pragma unroll
        for (int u = 0; u < U; ++u) acc = mfma16(t[u], xc, acc);
    }
    if (acc[0] == 123.456f) out[0] = acc[1];
}

// ------------------------------------------------------------------------------------------------ host
struct Shape {
    const char* name;
    int N16, K32, copies;
};

template <int T_MAX, int KS, int KC, int NBUF, int U, int ROT, int NL = 1, int NB = 4, int LM = 0>
static void launch_skx(dim3 grid, const Args& a, hipStream_t st) {
    constexpr int lds = (LM == 1 ? 2 : NBUF) * KC * NB * 1024 > T_MAX * KS * NB * 1024 ? (LM == 1 ? 2 : NBUF) * KC * NB * 1024 : T_MAX * KS * NB * 1024;
    static bool once = false;
    if (!once) {
        CKH(hipFuncSetAttribute((const void*)skx_kernel<T_MAX, KS, KC, (LM == 1 ? NBUF : NBUF), U, ROT, NL, NB, LM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        once = true;
    }
    hipLaunchKernelGGL((skx_kernel<T_MAX, KS, KC, NBUF, U, ROT, NL, NB, LM>), grid, dim3(64 * (T_MAX * KS + NL)), lds, st, a);
}

template <int T_MAX, int KS, int KC, int NBUF, int U, int NL, int NB>
static void launch_skf(dim3 grid, const Args& a, hipStream_t st) {
    constexpr int ring = NBUF * KC * NB * 1024, red = T_MAX * KS * NB * 1024;
    constexpr int lds = (ring > red ? ring : red) + 64;
    static bool once = false;
    if (!once) {
        CKH(hipFuncSetAttribute((const void*)skf_kernel<T_MAX, KS, KC, NBUF, U, NL, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        once = true;
    }
    hipLaunchKernelGGL((skf_kernel<T_MAX, KS, KC, NBUF, U, NL, NB>), grid, dim3(64 * (T_MAX * KS + NL)), lds, st, a);
}

static int g_mask_cus = 0;          // > 0: run everything on a stream restricted to this many CUs of every XCD

template <int NB>
static void lab(int iters, const char* only) {
    hipStream_t st;
    if (g_mask_cus > 0) {
        uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 256; ++i)
            if (i / 8 < g_mask_cus) m[i >> 5] |= 1u << (i & 31);
        CKH(hipExtStreamCreateWithCUMask(&st, 8, m));
    } else
    CKH(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CKH(hipEventCreate(&e0));
    CKH(hipEventCreate(&e1));
    const Shape shapes[] = {{"o", 256, 128, 16}, {"down", 256, 344, 6}, {"qkv", 768, 128, 6}, {"gateup", 1376, 128, 4}, {"lm_head", 2000, 128, 3}};
    for (const Shape& sh : shapes) {
        if (only[0] && strcmp(only, sh.name)) continue;
        const int64_t wfr = (int64_t)sh.N16 * sh.K32, whalves = wfr * FRAG;
        const double wbytes = (double)whalves * 2;
        half_t* W;
        half_t* xf;
        float *out, *ref, *part;
        CKH(hipMalloc(&W, (size_t)whalves * 2 * sh.copies));
        CKH(hipMalloc(&xf, (size_t)NB * sh.K32 * FRAG * 2));
        const int64_t nout = (int64_t)sh.N16 * NB * 64 * 4;
        CKH(hipMalloc(&out, nout * 4));
        CKH(hipMalloc(&ref, nout * 4));
        CKH(hipMalloc(&part, nout * 4 * 16));
        fill_kernel<<<2048, 256, 0, st>>>(W, (size_t)whalves * sh.copies, 0x1234u, 0.05f);
        fill_kernel<<<256, 256, 0, st>>>(xf, (size_t)NB * sh.K32 * FRAG, 0x77u, 1.0f);
        ref_kernel<<<(unsigned)((nout + 255) / 256), 256, 0, st>>>(W, xf, ref, sh.N16, sh.K32, NB);       // copy 0
        CKH(hipStreamSynchronize(st));
        std::vector<float> href(nout), hout(nout);
        CKH(hipMemcpy(href.data(), ref, nout * 4, hipMemcpyDeviceToHost));
        double refmax = 0;
        for (float v : href) refmax = fmax(refmax, fabs(v));

        auto run = [&](const char* tag, auto launch, bool check, double extra_bytes) {
            // correctness on copy 0
            CKH(hipMemsetAsync(out, 0xff, nout * 4, st));
            launch(0);
            CKH(hipStreamSynchronize(st));
            hipError_t le = hipGetLastError();
            if (le != hipSuccess) {
                printf("%-8s %-40s LAUNCH ERROR %s\n", sh.name, tag, hipGetErrorString(le));
                return;
            }
            double err = 0;
            if (check) {
                CKH(hipMemcpy(hout.data(), out, nout * 4, hipMemcpyDeviceToHost));
                for (int64_t i = 0; i < nout; ++i) {
                    const double d = fabs((double)hout[i] - (double)href[i]);
                    err = fmax(err, std::isnan(d) ? 1e30 : d);
                }
            }
            for (int i = 0; i < 8; ++i) launch(i % sh.copies);
            double best = 1e30, sum = 0;
            const int rounds = 3;
            for (int r = 0; r < rounds; ++r) {
                CKH(hipEventRecord(e0, st));
                for (int i = 0; i < iters; ++i) launch((i + r) % sh.copies);
                CKH(hipEventRecord(e1, st));
                CKH(hipEventSynchronize(e1));
                float ms;
                CKH(hipEventElapsedTime(&ms, e0, e1));
                const double us = 1e3 * ms / iters;
                best = fmin(best, us);
                sum += us;
            }
            printf("%-8s %-44s %8.2f us (min %7.2f)  %7.0f GB/s  err %.2e (max|ref| %.1f)%s\n", sh.name, tag, sum / rounds, best,
                   wbytes / (sum / rounds) / 1e3, err, refmax, (check && err > 2e-3 * refmax) ? "  <-- MISMATCH" : "");
            fflush(stdout);
        };
        auto mk = [&](int copy, float* o, int S, int tiles_lo, int n_hi) {
            Args a;
            a.W = W + (int64_t)copy * whalves; a.xf = xf; a.out = o; a.N16 = sh.N16; a.K32 = sh.K32; a.S = S; a.tiles_lo = tiles_lo; a.n_hi = n_hi;
            return a;
        };
        const std::string nm = sh.name;
        // ---- ceiling: pure streaming of the weights (launch to launch, boundary included)
        run("stream U8 (256 wg x 8 waves)", [&](int cp) { stream_kernel<8><<<256, 512, 0, st>>>(W + (int64_t)cp * whalves, wfr, out); }, false, 0);
        run("stream U8 (512 wg x 8 waves)", [&](int cp) { stream_kernel<8><<<512, 512, 0, st>>>(W + (int64_t)cp * whalves, wfr, out); }, false, 0);
        // ---- A: per-wave x structure (64 rows only)
        if constexpr (NB == 4) {
            if (nm == "o" || nm == "down") {
                run("cur NT1 NW8 U4", [&](int cp) { cur_kernel<1, 8, 4, 0, 0><<<sh.N16, 512, 8 * 1 * 4 * 1024, st>>>(mk(cp, out, 1, 0, 0)); }, true, 0);
                run("cur NT1 NW8 U4 noX", [&](int cp) { cur_kernel<1, 8, 4, 1, 0><<<sh.N16, 512, 8 * 1 * 4 * 1024, st>>>(mk(cp, out, 1, 0, 0)); }, false, 0);
            } else {
                run("cur NT2 NW4 U4", [&](int cp) { cur_kernel<2, 4, 4, 0, 0><<<sh.N16 / 2, 256, 4 * 2 * 4 * 1024, st>>>(mk(cp, out, 1, 0, 0)); }, true, 0);
                run("cur NT2 NW4 U4 noX", [&](int cp) { cur_kernel<2, 4, 4, 1, 0><<<sh.N16 / 2, 256, 4 * 2 * 4 * 1024, st>>>(mk(cp, out, 1, 0, 0)); }, false, 0);
            }
        }
        // ---- B: x through LDS.  dma = LDS-DMA loader waves (the shipped structure), reg = global_load -> ds_write loader waves
        constexpr int KC = NB <= 4 ? 8 : 4;
        auto red = [&](int S) { reduce_kernel<<<(unsigned)((nout / 4 + 255) / 256), 256, 0, st>>>(part, out, nout / 4, S); };
        if (nm == "o" || nm == "down") {
            constexpr int KCR = NB <= 4 ? 6 : 3, NBUFR = NB <= 4 ? 5 : 4, NLR = NB <= 4 ? 2 : 3;
            run("skx T4 KS3 S4 dma (shipped) + reduce", [&](int cp) { launch_skx<4, 3, KCR, NBUFR, 2, 0, NLR, NB, 0>(dim3(sh.N16 / 4, 4), mk(cp, part, 4, 4, 0), st); red(4); }, true, 0);
            run("skx T4 KS3 S4 reg R2 NL2 + reduce", [&](int cp) { launch_skx<4, 3, KCR, 3, 2, 0, (NB <= 4 ? 2 : 3), NB, 1>(dim3(sh.N16 / 4, 4), mk(cp, part, 4, 4, 0), st); red(4); }, true, 0);
            run("skx T4 KS3 S4 reg R3 NL4 KC6 + reduce", [&](int cp) { launch_skx<4, 3, 6, 4, 2, 0, 4, NB, 1>(dim3(sh.N16 / 4, 4), mk(cp, part, 4, 4, 0), st); red(4); }, true, 0);
            run("skx T8 KS1 S8 dma NL4 + reduce", [&](int cp) { launch_skx<8, 1, 4, 4, 4, 0, 4, NB, 0>(dim3(sh.N16 / 8, 8), mk(cp, part, 8, 8, 0), st); red(8); }, true, 0);
            if constexpr (NB > 4) {       // round 3: is the shipped geometry (a barrier per k32 step: KC 3 / KS 3) the best one for the residual projections?
                run("skx T4 KS3 S4 main only (shipped)", [&](int cp) { launch_skx<4, 3, KCR, NBUFR, 2, 0, NLR, NB, 0>(dim3(sh.N16 / 4, 4), mk(cp, part, 4, 4, 0), st); }, false, 0);
                run("skx T4 KS2 KC8 NBUF2 U4 NL4 S4 + reduce", [&](int cp) { launch_skx<4, 2, 8, 2, 4, 0, 4, NB, 0>(dim3(sh.N16 / 4, 4), mk(cp, part, 4, 4, 0), st); red(4); }, true, 0);
                run("skx T4 KS2 KC8 NBUF2 U2 NL4 S4 + reduce", [&](int cp) { launch_skx<4, 2, 8, 2, 2, 0, 4, NB, 0>(dim3(sh.N16 / 4, 4), mk(cp, part, 4, 4, 0), st); red(4); }, true, 0);
                run("skx T4 KS2 KC4 NBUF4 U4 NL4 S4 + reduce", [&](int cp) { launch_skx<4, 2, 4, 4, 4, 0, 4, NB, 0>(dim3(sh.N16 / 4, 4), mk(cp, part, 4, 4, 0), st); red(4); }, true, 0);
                run("skx T4 KS3 KC6 NBUF2 U2 NL3 S4 + reduce", [&](int cp) { launch_skx<4, 3, 6, 2, 2, 0, 3, NB, 0>(dim3(sh.N16 / 4, 4), mk(cp, part, 4, 4, 0), st); red(4); }, true, 0);
                run("skx T4 KS3 KC6 NBUF2 U4 NL3 S4 + reduce", [&](int cp) { launch_skx<4, 3, 6, 2, 4, 0, 3, NB, 0>(dim3(sh.N16 / 4, 4), mk(cp, part, 4, 4, 0), st); red(4); }, true, 0);
                run("skx T2 KS4 KC8 NBUF2 U2 NL4 S2 + reduce", [&](int cp) { launch_skx<2, 4, 8, 2, 2, 0, 4, NB, 0>(dim3(sh.N16 / 2, 2), mk(cp, part, 2, 2, 0), st); red(2); }, true, 0);
                run("skx T2 KS4 KC8 NBUF2 U4 NL4 S2 + reduce", [&](int cp) { launch_skx<2, 4, 8, 2, 4, 0, 4, NB, 0>(dim3(sh.N16 / 2, 2), mk(cp, part, 2, 2, 0), st); red(2); }, true, 0);
                run("skx T2 KS6 KC6 NBUF2 U1 NL4 S2 + reduce", [&](int cp) { launch_skx<2, 6, 6, 2, 1, 0, 4, NB, 0>(dim3(sh.N16 / 2, 2), mk(cp, part, 2, 2, 0), st); red(2); }, true, 0);
                run("skx T8 KS1 KC8 NBUF2 U8 NL4 S8 + reduce", [&](int cp) { launch_skx<8, 1, 8, 2, 8, 0, 4, NB, 0>(dim3(sh.N16 / 8, 8), mk(cp, part, 8, 8, 0), st); red(8); }, true, 0);
                run("skx T8 KS1 KC8 NBUF2 U8 NL4 S8 main only", [&](int cp) { launch_skx<8, 1, 8, 2, 8, 0, 4, NB, 0>(dim3(sh.N16 / 8, 8), mk(cp, part, 8, 8, 0), st); }, false, 0);
            }
            if constexpr (NB <= 4) run("skx T1 KS8 S1 dma NL8", [&](int cp) { launch_skx<1, 8, 8, 4, 4, 0, 8, NB, 0>(dim3(sh.N16, 1), mk(cp, out, 1, 1, 0), st); }, true, 0);
            if constexpr (NB <= 4) run("skx T1 KS8 S1 reg R2 NL4", [&](int cp) { launch_skx<1, 8, 8, 3, 4, 0, 4, NB, 1>(dim3(sh.N16, 1), mk(cp, out, 1, 1, 0), st); }, true, 0);
            if constexpr (NB <= 4) run("skx T1 KS8 S1 reg R2 NL8", [&](int cp) { launch_skx<1, 8, 8, 3, 4, 0, 8, NB, 1>(dim3(sh.N16, 1), mk(cp, out, 1, 1, 0), st); }, true, 0);
        } else if (nm == "qkv") {
            run("skx T3 KS4 dma NBUF4 U4 NL4 (shipped)", [&](int cp) { launch_skx<3, 4, KC, 4, 4, 0, 4, NB, 0>(dim3(256, 1), mk(cp, out, 1, 3, 0), st); }, true, 0);
            if constexpr (NB > 4) {
                run("skx T3 KS4 KC8 NBUF2 U2 (256 wg, engine ring)", [&](int cp) { launch_skx<3, 4, 8, 2, 2, 0, 4, NB, 0>(dim3(256, 1), mk(cp, out, 1, 3, 0), st); }, true, 0);
                run("skx T3 KS2 KC8 NBUF2 U4 (256 wg)", [&](int cp) { launch_skx<3, 2, 8, 2, 4, 0, 4, NB, 0>(dim3(256, 1), mk(cp, out, 1, 3, 0), st); }, true, 0);
                run("skx T3 KS2 KC8 NBUF2 U2 (256 wg)", [&](int cp) { launch_skx<3, 2, 8, 2, 2, 0, 4, NB, 0>(dim3(256, 1), mk(cp, out, 1, 3, 0), st); }, true, 0);
                run("skx T6 KS2 KC8 NBUF2 U4 (128 wg)", [&](int cp) { launch_skx<6, 2, 8, 2, 4, 0, 4, NB, 0>(dim3(128, 1), mk(cp, out, 1, 6, 0), st); }, true, 0);
                run("skx T6 KS2 KC8 NBUF2 U2 (128 wg)", [&](int cp) { launch_skx<6, 2, 8, 2, 2, 0, 4, NB, 0>(dim3(128, 1), mk(cp, out, 1, 6, 0), st); }, true, 0);
            }
            run("skx T3 KS2 dma NBUF4 U8 NL4 (256 wg)", [&](int cp) { launch_skx<3, 2, KC, 4, 8, 0, 4, NB, 0>(dim3(256, 1), mk(cp, out, 1, 3, 0), st); }, true, 0);
            run("skx T3 KS2 dma NBUF4 U4 NL4 (256 wg)", [&](int cp) { launch_skx<3, 2, KC, 4, 4, 0, 4, NB, 0>(dim3(256, 1), mk(cp, out, 1, 3, 0), st); }, true, 0);
            run("skx T6 KS2 dma NBUF4 U4 NL4 (128 wg)", [&](int cp) { launch_skx<6, 2, KC, 4, 4, 0, 4, NB, 0>(dim3(128, 1), mk(cp, out, 1, 6, 0), st); }, true, 0);
            run("skx T6 KS2 dma NBUF4 U8 NL4 (128 wg)", [&](int cp) { launch_skx<6, 2, KC, 4, 8, 0, 4, NB, 0>(dim3(128, 1), mk(cp, out, 1, 6, 0), st); }, true, 0);
            run("skf T3 KS4 flags KCx NBUF4 U4 NL4", [&](int cp) { launch_skf<3, 4, KC, 4, 4, 4, NB>(dim3(256, 1), mk(cp, out, 1, 3, 0), st); }, true, 0);
            run("skf T3 KS4 flags KC4 NBUF4 U4 NL4", [&](int cp) { launch_skf<3, 4, 4, 4, 4, 4, NB>(dim3(256, 1), mk(cp, out, 1, 3, 0), st); }, true, 0);
            run("skf T3 KS4 flags KC4 NBUF(8|4) U4 NL4", [&](int cp) { launch_skf<3, 4, 4, (NB <= 4 ? 8 : 4), 4, 4, NB>(dim3(256, 1), mk(cp, out, 1, 3, 0), st); }, true, 0);
            run("skf T3 KS4 flags KC4 NBUF4 U8 NL4", [&](int cp) { launch_skf<3, 4, 4, 4, 8, 4, NB>(dim3(256, 1), mk(cp, out, 1, 3, 0), st); }, true, 0);
            run("skx T6 KS2 S2 dma NL4 + reduce", [&](int cp) { launch_skx<6, 2, KC, 4, 4, 0, 4, NB, 0>(dim3(128, 2), mk(cp, part, 2, 6, 0), st); red(2); }, true, 0);
            run("skx T12 KS1 S4 dma NL4 + reduce", [&](int cp) { launch_skx<12, 1, KC, 4, 4, 0, 4, NB, 0>(dim3(64, 4), mk(cp, part, 4, 12, 0), st); red(4); }, true, 0);
            run("skx T12 KS1 S4 dma NL4 main only", [&](int cp) { launch_skx<12, 1, KC, 4, 4, 0, 4, NB, 0>(dim3(64, 4), mk(cp, part, 4, 12, 0), st); }, false, 0);
            run("skx T12 KS1 S4 dma NL2 U8 + reduce", [&](int cp) { launch_skx<12, 1, KC, 4, 8, 0, 2, NB, 0>(dim3(64, 4), mk(cp, part, 4, 12, 0), st); red(4); }, true, 0);
            run("skx T6 KS2 S8 dma NL4 + reduce", [&](int cp) { launch_skx<6, 2, KC, 4, 4, 0, 4, NB, 0>(dim3(128, 8), mk(cp, part, 8, 6, 0), st); red(8); }, true, 0);
            run("skx T3 KS4 reg R2 U4 NL4", [&](int cp) { launch_skx<3, 4, KC, 3, 4, 0, 4, NB, 1>(dim3(256, 1), mk(cp, out, 1, 3, 0), st); }, true, 0);
            run("skx T3 KS4 reg R3 U4 NL4", [&](int cp) { launch_skx<3, 4, KC, 4, 4, 0, 4, NB, 1>(dim3(256, 1), mk(cp, out, 1, 3, 0), st); }, true, 0);
            run("skx T3 KS4 reg R2 U4 NL2", [&](int cp) { launch_skx<3, 4, KC, 3, 4, 0, 2, NB, 1>(dim3(256, 1), mk(cp, out, 1, 3, 0), st); }, true, 0);
            run("skx T3 KS4 reg R2 U4 NL4 KC8", [&](int cp) { launch_skx<3, 4, 8, 3, 4, 0, 4, NB, 1>(dim3(256, 1), mk(cp, out, 1, 3, 0), st); }, true, 0);
        } else if (nm == "gateup") {
            run("skx T6 KS2 dma NBUF4 U4 NL4 (shipped)", [&](int cp) { launch_skx<6, 2, KC, 4, 4, 0, 4, NB, 0>(dim3(256, 1), mk(cp, out, 1, 5, 96), st); }, true, 0);
            if constexpr (NB > 4) {
                run("skx T6 KS2 KC8 NBUF2 U4 (256 wg, engine ring)", [&](int cp) { launch_skx<6, 2, 8, 2, 4, 0, 4, NB, 0>(dim3(256, 1), mk(cp, out, 1, 5, 96), st); }, true, 0);
                run("skx T6 KS1 KC8 NBUF2 U4 (256 wg)", [&](int cp) { launch_skx<6, 1, 8, 2, 4, 0, 4, NB, 0>(dim3(256, 1), mk(cp, out, 1, 5, 96), st); }, true, 0);
                run("skx T6 KS1 KC8 NBUF2 U8 (256 wg)", [&](int cp) { launch_skx<6, 1, 8, 2, 8, 0, 4, NB, 0>(dim3(256, 1), mk(cp, out, 1, 5, 96), st); }, true, 0);
                run("skx T11 KS1 KC8 NBUF2 U4 (128 wg)", [&](int cp) { launch_skx<11, 1, 8, 2, 4, 0, 4, NB, 0>(dim3(128, 1), mk(cp, out, 1, 10, 96), st); }, true, 0);
                run("skx T11 KS1 KC4 NBUF4 U2 (128 wg)", [&](int cp) { launch_skx<11, 1, 4, 4, 2, 0, 4, NB, 0>(dim3(128, 1), mk(cp, out, 1, 10, 96), st); }, true, 0);
            }
            run("skx T6 KS1 dma NBUF4 U8 NL4 (256 wg)", [&](int cp) { launch_skx<6, 1, KC, 4, 8, 0, 4, NB, 0>(dim3(256, 1), mk(cp, out, 1, 5, 96), st); }, true, 0);
            run("skx T6 KS1 dma NBUF4 U4 NL4 (256 wg)", [&](int cp) { launch_skx<6, 1, KC, 4, 4, 0, 4, NB, 0>(dim3(256, 1), mk(cp, out, 1, 5, 96), st); }, true, 0);
            run("skx T11 KS1 dma NBUF4 U4 NL4 (128 wg)", [&](int cp) { launch_skx<11, 1, KC, 4, 4, 0, 4, NB, 0>(dim3(128, 1), mk(cp, out, 1, 10, 96), st); }, true, 0);
            run("skx T11 KS1 dma NBUF4 U8 NL4 (128 wg)", [&](int cp) { launch_skx<11, 1, KC, 4, 8, 0, 4, NB, 0>(dim3(128, 1), mk(cp, out, 1, 10, 96), st); }, true, 0);
            run("skf T6 KS2 flags KCx NBUF4 U4 NL4", [&](int cp) { launch_skf<6, 2, KC, 4, 4, 4, NB>(dim3(256, 1), mk(cp, out, 1, 5, 96), st); }, true, 0);
            run("skf T6 KS2 flags KC4 NBUF4 U4 NL4", [&](int cp) { launch_skf<6, 2, 4, 4, 4, 4, NB>(dim3(256, 1), mk(cp, out, 1, 5, 96), st); }, true, 0);
            run("skf T6 KS2 flags KC2 NBUF8 U4 NL4", [&](int cp) { launch_skf<6, 2, 2, 8, 4, 4, NB>(dim3(256, 1), mk(cp, out, 1, 5, 96), st); }, true, 0);
            run("skx T11 KS1 S2 dma NL4 + reduce", [&](int cp) { launch_skx<11, 1, KC, 4, 4, 0, 4, NB, 0>(dim3(128, 2), mk(cp, part, 2, 10, 96), st); red(2); }, true, 0);
            run("skx T11 KS1 S2 dma NL4 main only", [&](int cp) { launch_skx<11, 1, KC, 4, 4, 0, 4, NB, 0>(dim3(128, 2), mk(cp, part, 2, 10, 96), st); }, false, 0);
            run("skx T6 KS2 reg R2 U4 NL4", [&](int cp) { launch_skx<6, 2, KC, 3, 4, 0, 4, NB, 1>(dim3(256, 1), mk(cp, out, 1, 5, 96), st); }, true, 0);
            run("skx T6 KS2 reg R3 U4 NL4", [&](int cp) { launch_skx<6, 2, KC, 4, 4, 0, 4, NB, 1>(dim3(256, 1), mk(cp, out, 1, 5, 96), st); }, true, 0);
            run("skx T6 KS2 reg R2 U4 NL2", [&](int cp) { launch_skx<6, 2, KC, 3, 4, 0, 2, NB, 1>(dim3(256, 1), mk(cp, out, 1, 5, 96), st); }, true, 0);
            run("skx T6 KS2 reg R2 U4 NL4 KC8", [&](int cp) { launch_skx<6, 2, 8, 3, 4, 0, 4, NB, 1>(dim3(256, 1), mk(cp, out, 1, 5, 96), st); }, true, 0);
        } else {
            run("skx T8 KS1 dma NBUF4 U4 NL4 (shipped)", [&](int cp) { launch_skx<8, 1, KC, 4, 4, 0, 4, NB, 0>(dim3(256, 1), mk(cp, out, 1, 7, 208), st); }, true, 0);
            run("skf T8 KS1 flags KCx NBUF4 U4 NL4", [&](int cp) { launch_skf<8, 1, KC, 4, 4, 4, NB>(dim3(256, 1), mk(cp, out, 1, 7, 208), st); }, true, 0);
            run("skx T8 KS1 reg R2 U4 NL4", [&](int cp) { launch_skx<8, 1, KC, 3, 4, 0, 4, NB, 1>(dim3(256, 1), mk(cp, out, 1, 7, 208), st); }, true, 0);
        }
        CKH(hipFree(W)); CKH(hipFree(xf)); CKH(hipFree(out)); CKH(hipFree(ref)); CKH(hipFree(part));
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 96;
    const char* only = argc > 2 ? argv[2] : "";
    const int nb = argc > 3 ? atoi(argv[3]) : 0;            // column groups: 4 (64 rows), 8 (128 rows); 0 = both
    g_mask_cus = argc > 4 ? atoi(argv[4]) : 0;
    if (g_mask_cus > 0) printf("==== stream restricted to %d CUs of every XCD\n", g_mask_cus);
    if (nb == 0 || nb == 4) { printf("---- 64 rows\n"); lab<4>(iters, only); }
    if (nb == 0 || nb == 8) { printf("---- 128 rows\n"); lab<8>(iters, only); }
    return 0;
}
