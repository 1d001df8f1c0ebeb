// cumask_probe - which CUs does a hipExtStreamCreateWithCUMask stream run on?  (lab tool, not part of the product)
//   hipcc --offload-arch=gfx950 -O2 tools/cumask/cumask_probe.hip -o tools/cumask/cumask_probe
// Prints, per mask pattern, the number of workgroups that ran on each XCD (HW_REG_XCC_ID) and the number of distinct
// (XCD, SE, CU) places seen, plus the time of a fixed streaming kernel under that mask.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#define CKH(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void where_kernel(uint32_t* out) {
    // HW_REG_XCC_ID = 20 (bits 3:0), HW_REG_HW_ID = 4 (wave/simd/cu/sh/se ids)
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    // spin a little so that the grid spreads over every available CU
    uint64_t t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 20000) {}
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 28) | (hw & 0x0fffffff);
}

typedef float f4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_kernel(const f4v* p, int64_t n, float* sink) {
    f4v acc = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const f4v v = __builtin_nontemporal_load(p + i);
        acc += v;
    }
    if (acc.x == 1.2345f) sink[0] = acc.y + acc.z + acc.w;
}

int main() {
    const int NW = 8;                                   // 256 bits
    uint32_t* d;
    CKH(hipMalloc(&d, 8192 * 4));
    const int64_t nbytes = 1ll << 30;
    f4v* big;
    float* sink;
    CKH(hipMalloc(&big, nbytes));
    CKH(hipMalloc(&sink, 16));
    CKH(hipMemset(big, 0, nbytes));
    struct Pat { const char* name; std::vector<uint32_t> m; };
    std::vector<Pat> pats;
    auto mk = [&](const char* name, auto pred) {
        Pat p{name, std::vector<uint32_t>(NW, 0)};
        for (int i = 0; i < 256; ++i) if (pred(i)) p.m[i >> 5] |= 1u << (i & 31);
        pats.push_back(p);
    };
    mk("all 256", [](int) { return true; });
    mk("bits i < 128", [](int i) { return i < 128; });
    mk("bits i % 8 < 4", [](int i) { return i % 8 < 4; });
    mk("bits i % 8 < 5", [](int i) { return i % 8 < 5; });
    mk("bits i % 8 < 6", [](int i) { return i % 8 < 6; });
    mk("bits i % 8 >= 4", [](int i) { return i % 8 >= 4; });
    mk("bits i % 8 >= 6", [](int i) { return i % 8 >= 6; });
    mk("bits i % 2 == 0", [](int i) { return i % 2 == 0; });
    mk("bits (i / 8) < 16", [](int i) { return i / 8 < 16; });
    hipEvent_t e0, e1;
    CKH(hipEventCreate(&e0));
    CKH(hipEventCreate(&e1));
    for (auto& p : pats) {
        hipStream_t st;
        CKH(hipExtStreamCreateWithCUMask(&st, NW, p.m.data()));
        where_kernel<<<2048, 64, 0, st>>>(d);
        CKH(hipStreamSynchronize(st));
        std::vector<uint32_t> h(2048);
        CKH(hipMemcpy(h.data(), d, 2048 * 4, hipMemcpyDeviceToHost));
        int per[16] = {0};
        std::set<uint32_t> places;
        for (uint32_t v : h) {
            per[v >> 28]++;
            const uint32_t hw = v & 0x0fffffff;
            places.insert(((v >> 28) << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 7) | ((hw >> 8) & 15));   // xcc, se, sh, cu
        }
        for (int i = 0; i < 4; ++i) stream_kernel<<<2048, 256, 0, st>>>(big, nbytes / 16, sink);
        CKH(hipEventRecord(e0, st));
        for (int i = 0; i < 8; ++i) stream_kernel<<<2048, 256, 0, st>>>(big, nbytes / 16, sink);
        CKH(hipEventRecord(e1, st));
        CKH(hipEventSynchronize(e1));
        float ms;
        CKH(hipEventElapsedTime(&ms, e0, e1));
        printf("%-20s places %3zu  per XCD:", p.name, places.size());
        for (int i = 0; i < 8; ++i) printf(" %4d", per[i]);
        printf("   1 GiB stream %.1f us = %.0f GB/s\n", 1e3 * ms / 8, nbytes / (ms / 8) / 1e6);
        CKH(hipStreamDestroy(st));
    }
    return 0;
}
