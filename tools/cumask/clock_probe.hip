// clock_probe - effective shader clock while other kernels run (lab tool, not part of the product)
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/cumask/clock_probe.hip -o tools/cumask/libclock_probe.so
// One wave spins for `spin_us` of the constant 100 MHz counter (s_memrealtime) and reports how many shader cycles (s_memtime) went by:
// MHz = 100 * cycles / ticks.  Launch it into an unmasked stream next to the kernels under study.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void clock_probe_kernel(uint64_t* out, uint64_t ticks) {
    const uint64_t w0 = wall_clock64();
    const uint64_t c0 = clock64();
    uint64_t w1 = w0;
    while (w1 - w0 < ticks) {
        __builtin_amdgcn_s_sleep(8);
        w1 = wall_clock64();
    }
    const uint64_t c1 = clock64();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = w1 - w0;
    }
}

// out_dev: 2 x uint64 of device memory; returns the hipError_t of the launch
extern "C" int clock_probe_launch(void* stream, uint64_t* out_dev, int spin_us) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_dev, (uint64_t)spin_us * 100);
    return (int)hipGetLastError();
}

// ---- a read streamer for the contention lab: every workgroup sweeps its share of [p, p + n16 * 16) `reps` times with 16-byte
// loads (nt = 1: non-temporal).  A buffer of a few tens of MiB is served by the 256 MiB memory-side cache after the first sweep
// (never by a 4 MiB L2), a multi-GiB one by HBM: the same fabric traffic with and without HBM behind it.
typedef float f4v __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void lab_stream_kernel(const f4v* __restrict__ p, int64_t n16, int reps, float* sink) {
    f4v acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < reps; ++r)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256 * 4) {
            f4v v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t j = i + (int64_t)u * gridDim.x * 256;
                if (j < n16) v[u] = NT ? __builtin_nontemporal_load(p + j) : p[j];
                else v[u] = f4v{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u];
        }
    if (acc.x == 1.2345f) sink[0] = acc.y + acc.z + acc.w;
}
extern "C" int lab_stream_launch(void* stream, const void* p, int64_t bytes, int reps, int blocks, int nt, float* sink) {
    if (nt) hipLaunchKernelGGL(lab_stream_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f4v*)p, bytes / 16, reps, sink);
    else hipLaunchKernelGGL(lab_stream_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f4v*)p, bytes / 16, reps, sink);
    return (int)hipGetLastError();
}

// ---- the same bytes read by all 8 XCDs: does the memory-side cache turn 8 reads into 1 HBM read?  Every XCD's workgroups
// (blockIdx % 8 = XCD) jointly sweep the WHOLE buffer once, XCD x starting `stagger` bytes ahead of XCD x - 1 (wrapping).
// stagger = 0: all XCDs ask for the same lines at the same moment (what a GEMM round shared by the XCDs does); stagger > 0: XCD x
// re-reads what XCD x + 1 fetched `stagger` bytes of streaming earlier (a software-pipelined hand-down of operand tiles).
__global__ __launch_bounds__(256) void lab_share_kernel(const f4v* __restrict__ p, int64_t n16, int64_t stagger16, float* sink) {
    const int xcd = blockIdx.x & 7, wi = blockIdx.x >> 3, nw = gridDim.x >> 3;
    f4v acc = {0.f, 0.f, 0.f, 0.f};
    const int64_t start = (int64_t)xcd * stagger16;
    for (int64_t i = (int64_t)wi * 256 + threadIdx.x; i < n16; i += (int64_t)nw * 256 * 4) {
        f4v v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int64_t j = i + (int64_t)u * nw * 256;
            if (j < n16) {
                j += start;
                j = j >= n16 ? j - n16 : j;
                v[u] = p[j];
            } else v[u] = f4v{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
    }
    if (acc.x == 1.2345f) sink[0] = acc.y + acc.z + acc.w;
}
extern "C" int lab_share_launch(void* stream, const void* p, int64_t bytes, int64_t stagger_bytes, int blocks, float* sink) {
    hipLaunchKernelGGL(lab_share_kernel, dim3(blocks & ~7), dim3(256), 0, (hipStream_t)stream, (const f4v*)p, bytes / 16, stagger_bytes / 16, sink);
    return (int)hipGetLastError();
}

// ---- the decode attention's ADDRESS pattern without its arithmetic: one wave per (slot, head) walks the slot's pages (1 MiB each, all
// heads of 64 tokens) and reads its head's 16 KiB of K fragments and, 512 KiB further, its 16 KiB of V^T fragments - 32 non-temporal
// 1 KiB loads per page, summed into one register.  mode 1 = the same bytes with the heads interleaved at fragment granularity
// ([fragment][head] instead of [head][fragment]): the 32 waves of a slot, which run in near lock-step, then touch one contiguous 32 KiB
// per fragment index instead of 32 pieces 16 KiB apart.  Is the ~25 pJ/B between the decode attention and a bare stream DRAM row locality?
__global__ __launch_bounds__(64) void lab_kvpattern_kernel(const f4v* __restrict__ pool, int pages_per_slot, int mode, float* sink) {
    const int head = blockIdx.y, slot = blockIdx.z, lane = threadIdx.x;
    f4v acc = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < pages_per_slot; ++p) {
        const f4v* page = pool + ((int64_t)slot * pages_per_slot + p) * 65536;            // 1 MiB = 65536 x 16 B
        f4v v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int part = i >> 4, f = i & 15;                                              // K part / V part, fragment
            const int64_t off = mode == 0 ? (int64_t)part * 32768 + head * 1024 + f * 64      // [part][head][fragment][lane]
                                          : (int64_t)part * 32768 + f * 2048 + head * 64;      // [part][fragment][head][lane]
            v[i] = __builtin_nontemporal_load(page + off + lane);
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) acc += v[i];
    }
    if (acc.x == 1.2345f) sink[0] = acc.y + acc.z + acc.w;
}
extern "C" int lab_kvpattern_launch(void* stream, const void* pool, int slots, int pages_per_slot, int mode, float* sink) {
    hipLaunchKernelGGL(lab_kvpattern_kernel, dim3(1, 32, slots), dim3(64), 0, (hipStream_t)stream, (const f4v*)pool, pages_per_slot, mode, sink);
    return (int)hipGetLastError();
}

// ---- the KV walk with the attention's arithmetic added piece by piece (work bits): 1 = 128 v_dot2c per page against a register-resident
// vector (the Q K^T and P V products), 2 = the cross-lane traffic of the softmax (8 + 4 + 4 ds_bpermute-class exchanges per page), 4 = 5 v_exp +
// the probability exchange through 128 B of LDS.  Which piece carries the ~200 W between a bare KV walk and the decode attention?
typedef _Float16 lh2 __attribute__((ext_vector_type(2)));
typedef _Float16 lh8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float lab_dot8(const lh8 a, const lh8 b, float c) {
    c = __builtin_amdgcn_fdot2(lh2{a[0], a[1]}, lh2{b[0], b[1]}, c, false);
    c = __builtin_amdgcn_fdot2(lh2{a[2], a[3]}, lh2{b[2], b[3]}, c, false);
    c = __builtin_amdgcn_fdot2(lh2{a[4], a[5]}, lh2{b[4], b[5]}, c, false);
    c = __builtin_amdgcn_fdot2(lh2{a[6], a[7]}, lh2{b[6], b[7]}, c, false);
    return c;
}
__global__ __launch_bounds__(64, 2) void lab_kvwork_kernel(const lh8* __restrict__ pool, int pages_per_slot, int work, float* sink) {
    __shared__ _Float16 p16[64];
    const int head = blockIdx.y, slot = blockIdx.z, lane = threadIdx.x;
    lh8 q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = pool[lane + i * 64];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, m = 0.f;
    for (int p = 0; p < pages_per_slot; ++p) {
        const lh8* page = pool + ((int64_t)slot * pages_per_slot + p) * 65536;
        lh8 v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = __builtin_nontemporal_load(page + (int64_t)(i >> 4) * 32768 + head * 1024 + (i & 15) * 64 + lane);
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        if (work & 1) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int b = 0; b < 4; ++b) s[kt] = lab_dot8(v[kt * 4 + b], q[b], s[kt]);
        } else {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) s[kt] = (float)v[kt * 4][0] + (float)v[kt * 4 + 1][1] + (float)v[kt * 4 + 2][2] + (float)v[kt * 4 + 3][3];
        }
        if (work & 2) {
            float mx = -1e30f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                s[kt] += __shfl_xor(s[kt], 16, 64);
                s[kt] += __shfl_xor(s[kt], 32, 64);
                mx = fmaxf(mx, s[kt]);
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            m = fmaxf(m, mx);
        }
        lh8 pf[2];
        if (work & 4) {
            float mine = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const float pv = __builtin_amdgcn_exp2f(s[kt] - m);
                mine = kt == (lane >> 4) ? pv : mine;
            }
            p16[lane] = (_Float16)mine;
            const int g = lane >> 4;
#pragma unroll
            for (int b32 = 0; b32 < 2; ++b32) {
                typedef _Float16 lh4 __attribute__((ext_vector_type(4)));
                const lh4 lo = *(const lh4*)(p16 + b32 * 32 + 4 * g), hi = *(const lh4*)(p16 + b32 * 32 + 16 + 4 * g);
                pf[b32] = lh8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        } else {
            pf[0] = q[0];
            pf[1] = q[1];
        }
        if (work & 1) {
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                acc[d] = lab_dot8(v[16 + d * 2], pf[0], acc[d] * 0.999f);
                acc[d] = lab_dot8(v[16 + d * 2 + 1], pf[1], acc[d]);
            }
        } else {
#pragma unroll
            for (int d = 0; d < 8; ++d) acc[d] += (float)v[16 + d * 2][d & 7] + (float)v[16 + d * 2 + 1][d & 7] + s[d & 3];
        }
    }
    float t = m;
#pragma unroll
    for (int d = 0; d < 8; ++d) t += acc[d];
    if (t == 1.2345f) sink[0] = t;
}
extern "C" int lab_kvwork_launch(void* stream, const void* pool, int slots, int pages_per_slot, int work, float* sink) {
    hipLaunchKernelGGL(lab_kvwork_kernel, dim3(1, 32, slots), dim3(64), 0, (hipStream_t)stream, (const lh8*)pool, pages_per_slot, work, sink);
    return (int)hipGetLastError();
}


// ---- the same KV walk through LDS-DMA (round 6): is what a CU keeps in flight on half of the chip (~47 GB/s per CU for register loads)
// a limit of the register-return path, or of the CU's memory pipeline whatever the destination?  One workgroup = WAVES waves, each walking its
// own (slot, head); a wave owns a ring of two 16 KiB halves in LDS and keeps up to 32 one-KiB `global_load_lds` pieces in flight (counted
// vmcnt), consuming nothing.  Same bytes, same addresses as lab_kvpattern_kernel mode 0.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void lab_kvpattern_lds_kernel(const f4v* __restrict__ pool, int pages_per_slot, int heads, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lsm[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int pair = blockIdx.x * WAVES + w;                       // (slot, head)
    const int head = pair % heads, slot = pair / heads;
    char* ring = lsm + w * 32768;
    auto issue = [&](int p, int part) {
        const f4v* src = pool + ((int64_t)slot * pages_per_slot + p) * 65536 + (int64_t)part * 32768 + head * 1024 + lane;
#pragma unroll
        for (int f = 0; f < 16; ++f)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + f * 64),
                                             (__attribute__((address_space(3))) void*)(ring + part * 16384 + f * 1024), 16, 0, 2);
    };
    issue(0, 0);
    issue(0, 1);
    for (int p = 1; p < pages_per_slot; ++p) {
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");           // the K half of page p - 1 has landed: its ring half is free
        issue(p, 0);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        issue(p, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (pages_per_slot < 0) sink[0] = *(float*)ring;
}
extern "C" int lab_kvpattern_lds_launch(void* stream, const void* pool, int slots, int pages_per_slot, int waves, float* sink) {
    const int pairs = slots * 32;
    if (waves == 4) {
        static bool set4 = false;
        if (!set4) { (void)hipFuncSetAttribute((const void*)lab_kvpattern_lds_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768); set4 = true; }
        hipLaunchKernelGGL(lab_kvpattern_lds_kernel<4>, dim3(pairs / 4), dim3(256), 4 * 32768, (hipStream_t)stream, (const f4v*)pool, pages_per_slot, 32, sink);
    } else if (waves == 2) {
        hipLaunchKernelGGL(lab_kvpattern_lds_kernel<2>, dim3(pairs / 2), dim3(128), 2 * 32768, (hipStream_t)stream, (const f4v*)pool, pages_per_slot, 32, sink);
    } else {
        hipLaunchKernelGGL(lab_kvpattern_lds_kernel<1>, dim3(pairs), dim3(64), 32768, (hipStream_t)stream, (const f4v*)pool, pages_per_slot, 32, sink);
    }
    return (int)hipGetLastError();
}
