// Lab: does gfx950 need wait states between a VMEM store of more than 64 bits and a VALU write of the store's DATA registers?
//
// Root-cause probe for round 3's fused split-K reduce: its partial tiles went out through inline asm
// (`global_store_dwordx4 .. sc1`), which hipcc treats as one opaque instruction and does not pad; the NB >= 4 instantiations then
// wrote the next store's address INTO the previous store's data registers one state later.  This kernel does exactly that on purpose:
//     global_store_dwordx4 p, v[20:23], off sc1 ; <W wait states> ; v_mov_b32 v20, 0xdeadbeef ; v_mov_b32 v21, 0xdeadbeef
// for W = 0, 1 (what the round-3 code had: one SALU instruction in between), 2 (`s_nop 1`, what the ISA hazard table asks for), 4,
// alone and beside a bandwidth hog on a second stream, and counts the 16-byte records that reached memory with the poison in them.
//
//   hipcc --offload-arch=gfx950 -O3 -o store_hazard store_hazard.hip && ./store_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int W>
__global__ __launch_bounds__(256) void store_then_clobber(unsigned* out, int iters, int stride_u4) {
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    unsigned* p = out + (size_t)gid * 4;
    for (int i = 0; i < iters; ++i) {
        const unsigned a = gid, b = (unsigned)i, c = gid ^ 0x5a5a5a5au, d = (unsigned)i * 2654435761u;
        unsigned* q = p + (size_t)i * stride_u4 * 4;
        if constexpr (W == 0)
            asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %2\n\tv_mov_b32 v22, %3\n\tv_mov_b32 v23, %4\n\ts_nop 4\n\t"
                         "global_store_dwordx4 %0, v[20:23], off sc1\n\t"
                         "v_mov_b32 v20, 0xdeadbeef\n\tv_mov_b32 v21, 0xdeadbeef\n\tv_mov_b32 v22, 0xdeadbeef\n\tv_mov_b32 v23, 0xdeadbeef"
                         ::"v"(q), "v"(a), "v"(b), "v"(c), "v"(d) : "v20", "v21", "v22", "v23", "memory");
        else if constexpr (W == 1)
            asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %2\n\tv_mov_b32 v22, %3\n\tv_mov_b32 v23, %4\n\ts_nop 4\n\t"
                         "global_store_dwordx4 %0, v[20:23], off sc1\n\ts_nop 0\n\t"
                         "v_mov_b32 v20, 0xdeadbeef\n\tv_mov_b32 v21, 0xdeadbeef\n\tv_mov_b32 v22, 0xdeadbeef\n\tv_mov_b32 v23, 0xdeadbeef"
                         ::"v"(q), "v"(a), "v"(b), "v"(c), "v"(d) : "v20", "v21", "v22", "v23", "memory");
        else if constexpr (W == 2)
            asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %2\n\tv_mov_b32 v22, %3\n\tv_mov_b32 v23, %4\n\ts_nop 4\n\t"
                         "global_store_dwordx4 %0, v[20:23], off sc1\n\ts_nop 1\n\t"
                         "v_mov_b32 v20, 0xdeadbeef\n\tv_mov_b32 v21, 0xdeadbeef\n\tv_mov_b32 v22, 0xdeadbeef\n\tv_mov_b32 v23, 0xdeadbeef"
                         ::"v"(q), "v"(a), "v"(b), "v"(c), "v"(d) : "v20", "v21", "v22", "v23", "memory");
        else
            asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %2\n\tv_mov_b32 v22, %3\n\tv_mov_b32 v23, %4\n\ts_nop 4\n\t"
                         "global_store_dwordx4 %0, v[20:23], off sc1\n\ts_nop 3\n\t"
                         "v_mov_b32 v20, 0xdeadbeef\n\tv_mov_b32 v21, 0xdeadbeef\n\tv_mov_b32 v22, 0xdeadbeef\n\tv_mov_b32 v23, 0xdeadbeef"
                         ::"v"(q), "v"(a), "v"(b), "v"(c), "v"(d) : "v20", "v21", "v22", "v23", "memory");
    }
}

typedef float f4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void hog(const f4v* __restrict__ src, float* sink, size_t n, int reps) {
    float acc = 0.f;
    for (int r = 0; r < reps; ++r)
        for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
            const f4v v = __builtin_nontemporal_load(src + i);
            acc += v.x + v.y + v.z + v.w;
        }
    if (acc == 123.456f) *sink = acc;
}

template <int W>
static long run(unsigned* d_out, std::vector<unsigned>& h, int blocks, int iters, hipStream_t s) {
    const size_t n_u4 = (size_t)blocks * 256 * iters;
    CK(hipMemsetAsync(d_out, 0, n_u4 * 16, s));
    hipLaunchKernelGGL(store_then_clobber<W>, dim3(blocks), dim3(256), 0, s, d_out, iters, blocks * 256);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d_out, n_u4 * 16, hipMemcpyDeviceToHost));
    long bad = 0;
    for (size_t i = 0; i < n_u4; ++i) {
        const unsigned gid = (unsigned)(i % ((size_t)blocks * 256)), it = (unsigned)(i / ((size_t)blocks * 256));
        const unsigned e[4] = {gid, it, gid ^ 0x5a5a5a5au, it * 2654435761u};
        bool ok = true;
        for (int k = 0; k < 4; ++k) ok &= h[i * 4 + k] == e[k];
        bad += ok ? 0 : 1;
    }
    return bad;
}

int main() {
    const int blocks = 2048, iters = 64;
    const size_t n_u4 = (size_t)blocks * 256 * iters;
    unsigned* d_out;
    CK(hipMalloc(&d_out, n_u4 * 16));
    std::vector<unsigned> h(n_u4 * 4);
    hipStream_t s, s2;
    CK(hipStreamCreate(&s));
    CK(hipStreamCreate(&s2));
    const size_t hog_n = (size_t)1 << 26;      // 1 GiB of float4
    f4v* d_hog;
    float* d_sink;
    CK(hipMalloc(&d_hog, hog_n * 16));
    CK(hipMalloc(&d_sink, 4));
    CK(hipMemset(d_hog, 0x11, hog_n * 16));
    printf("records per run: %zu (16 B each)\n", n_u4);
    for (int load = 0; load < 2; ++load) {
        for (int rep = 0; rep < 3; ++rep) {
            long b[4];
            if (load) hipLaunchKernelGGL(hog, dim3(1024), dim3(256), 0, s2, d_hog, d_sink, hog_n, 40);
            b[0] = run<0>(d_out, h, blocks, iters, s);
            b[1] = run<1>(d_out, h, blocks, iters, s);
            b[2] = run<2>(d_out, h, blocks, iters, s);
            b[3] = run<4>(d_out, h, blocks, iters, s);
            CK(hipStreamSynchronize(s2));
            printf("%s rep %d: corrupted records at 0 / 1 / 2 / 4 wait states: %ld / %ld / %ld / %ld\n", load ? "beside a bandwidth hog" : "alone", rep, b[0], b[1], b[2], b[3]);
        }
    }
    return 0;
}
