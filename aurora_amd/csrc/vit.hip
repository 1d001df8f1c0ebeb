// ViT front-end helpers and row gathers (gfx950), all HBM-bound streaming kernels.
//
// Reference: HF CLIPVisionEmbeddings + pre_layrnorm as called from aurora.py:863-866,899 (conv14x14/14
// without bias -> flatten -> prepend class_embedding -> + position_embedding -> LayerNorm); row gathers
// for hidden_states[-2][:, 1:] (aurora.py:253) and the text-embedding lookup of
// prepare_inputs_labels_for_multimodal (model/utils.py:214-216).
#include "kernels.h"

// pixels [frames][chans][img][img] fp16 -> patches-as-rows [frames*gh*gw][kpad]; col = ch*P*P + ky*P + kx
// (the flatten order of the conv weight [D, chans, P, P]); cols >= chans*P*P are zero (K padding).
// height x width need not be the native square: the patch grid is (height / patch) x (width / patch), row-major,
// pixels past the last whole patch are ignored (stride-`patch` convolution without padding).
__global__ void im2col_kernel(const half_t* __restrict__ px, int frames, int chans, int height, int width, int patch, int kpad,
                              half_t* __restrict__ out) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int gw = width / patch, gh = height / patch;
    const int64_t total = (int64_t)frames * gh * gw * kpad;
    if (id >= total) return;
    const int col = id % kpad;
    const int64_t row = id / kpad;
    const int pxi = row % gw, pyi = (row / gw) % gh;
    const int f = row / ((int64_t)gw * gh);
    half_t v = (half_t)0.f;
    if (col < chans * patch * patch) {
        const int ch = col / (patch * patch), rem = col % (patch * patch);
        const int ky = rem / patch, kx = rem % patch;
        v = px[(((int64_t)f * chans + ch) * height + pyi * patch + ky) * width + pxi * patch + kx];
    }
    out[id] = v;
}

hipError_t launch_im2col(const half_t* pixels, int frames, int chans, int height, int width, int patch, int kpad, half_t* out,
                         hipStream_t s) {
    const int64_t total = (int64_t)frames * (height / patch) * (width / patch) * kpad;
    hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pixels, frames, chans, height,
                       width, patch, kpad, out);
    return hipGetLastError();
}

// x[f][0] = LN(cls + pos[0]); x[f][1+p] = LN(patch[f][p] + pos[1+p]); rows >= 1+npatch are zero padding.
#define VA_MAXC 8
__global__ __launch_bounds__(256) void vit_assemble_kernel(const half_t* __restrict__ patches, const half_t* __restrict__ cls,
                                                           const half_t* __restrict__ pos, const float* __restrict__ ln_w,
                                                           const float* __restrict__ ln_b, float eps, int npatch, int d,
                                                           int t_pad, half_t* __restrict__ x) {
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int f = blockIdx.y;
    if (tok >= t_pad) return;
    const int nchunk = d >> 3;
    half_t* xr = x + ((int64_t)f * t_pad + tok) * d;
    if (tok > npatch) {
        for (int c = lane; c < nchunk; c += 64) {
            h8 z;
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = (half_t)0.f;
            *(h8*)(xr + c * 8) = z;
        }
        return;
    }
    const half_t* src = tok == 0 ? cls : patches + ((int64_t)f * npatch + tok - 1) * d;
    const half_t* pr = pos + (int64_t)tok * d;
    float v[VA_MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VA_MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nchunk) {
            const h8 a = *(const h8*)(src + c * 8);
            const h8 p = *(const h8*)(pr + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[i][j] = (float)(half_t)((float)a[j] + (float)p[j]);   // fp16 embedding sum, as the reference stores it
                s += v[i][j];
            }
        }
    }
    s = wave_sum(s);
    const float mean = s / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VA_MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dl = v[i][j] - mean;
                q += dl * dl;
            }
        }
    }
    q = wave_sum(q);
    const float rstd = rsqrtf(q / (float)d + eps);
#pragma unroll
    for (int i = 0; i < VA_MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < nchunk) {
            h8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (half_t)((v[i][j] - mean) * rstd * ln_w[c * 8 + j] + ln_b[c * 8 + j]);
            *(h8*)(xr + c * 8) = o;
        }
    }
}

hipError_t launch_vit_assemble(const half_t* patches, const half_t* cls, const half_t* pos, const float* ln_w,
                               const float* ln_b, float eps, int frames, int npatch, int d, int t_pad, half_t* x,
                               hipStream_t s) {
    if (d > VA_MAXC * 512 || (d & 7)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(vit_assemble_kernel, dim3((t_pad + 3) / 4, frames), dim3(256), 0, s, patches, cls, pos, ln_w, ln_b, eps,
                       npatch, d, t_pad, x);
    return hipGetLastError();
}

// dst[i][:] = src[rows[i]][:]   (rows[i] < 0 -> zeros)
__global__ void gather_rows_kernel(const half_t* __restrict__ src, int ld_src, const int32_t* __restrict__ rows, int nrows,
                                   int d, half_t* __restrict__ dst, int ld_dst) {
    const int i = blockIdx.x;
    if (i >= nrows) return;
    const int r = rows[i];
    for (int c = threadIdx.x; c < (d >> 3); c += blockDim.x) {
        h8 v;
        if (r >= 0) v = *(const h8*)(src + (int64_t)r * ld_src + c * 8);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (half_t)0.f;
        }
        *(h8*)(dst + (int64_t)i * ld_dst + c * 8) = v;
    }
}
hipError_t launch_gather_rows(const half_t* src, int ld_src, const int32_t* rows, int nrows, int d, half_t* dst,
                              int ld_dst, hipStream_t s) {
    if (nrows <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(nrows), dim3(128), 0, s, src, ld_src, rows, nrows, d, dst, ld_dst);
    return hipGetLastError();
}

// dst[dst_rows[i]][:] = table[ids[i]][:]
__global__ void embed_rows_kernel(const half_t* __restrict__ table, int d, const int32_t* __restrict__ ids,
                                  const int32_t* __restrict__ dst_rows, int n, half_t* __restrict__ dst, int ld_dst) {
    const int i = blockIdx.x;
    if (i >= n) return;
    const int id = ids[i], r = dst_rows ? dst_rows[i] : i;
    for (int c = threadIdx.x; c < (d >> 3); c += blockDim.x)
        *(h8*)(dst + (int64_t)r * ld_dst + c * 8) = *(const h8*)(table + (int64_t)id * d + c * 8);
}
hipError_t launch_embed_rows(const half_t* table, int d, const int32_t* ids, const int32_t* dst_rows, int n,
                             half_t* dst, int ld_dst, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(embed_rows_kernel, dim3(n), dim3(128), 0, s, table, d, ids, dst_rows, n, dst, ld_dst);
    return hipGetLastError();
}

// out[f*(t-1) + i][:] = x[f][1 + i][:]   - hidden_states[-2][:, 1:] (aurora.py:253): drop CLS, compact frames
__global__ void strip_cls_kernel(const half_t* __restrict__ x, int t, int t_pad, int d, half_t* __restrict__ out) {
    const int i = blockIdx.x, f = blockIdx.y;
    const half_t* src = x + ((int64_t)f * t_pad + 1 + i) * d;
    half_t* dst = out + ((int64_t)f * (t - 1) + i) * d;
    for (int c = threadIdx.x; c < (d >> 3); c += blockDim.x) *(h8*)(dst + c * 8) = *(const h8*)(src + c * 8);
}
hipError_t launch_strip_cls(const half_t* x, int frames, int t, int t_pad, int d, half_t* out, hipStream_t s) {
    hipLaunchKernelGGL(strip_cls_kernel, dim3(t - 1, frames), dim3(128), 0, s, x, t, t_pad, d, out);
    return hipGetLastError();
}
// dst[f][tok][:] = tok < t ? src[f][tok][:] : 0   (dense [frames, t, d] -> padded [frames, t_pad, d]); sizes likewise
__global__ void pad_rows_kernel(const half_t* __restrict__ src, const float* __restrict__ size_src, int t, int t_pad, int d,
                                half_t* __restrict__ dst, float* __restrict__ size_dst) {
    const int tok = blockIdx.x, f = blockIdx.y;
    half_t* o = dst + ((int64_t)f * t_pad + tok) * d;
    for (int c = threadIdx.x; c < (d >> 3); c += blockDim.x) {
        h8 v;
        if (tok < t) v = *(const h8*)(src + ((int64_t)f * t + tok) * d + c * 8);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (half_t)0.f;
        }
        *(h8*)(o + c * 8) = v;
    }
    if (threadIdx.x == 0 && size_dst) size_dst[(int64_t)f * t_pad + tok] = (tok < t && size_src) ? size_src[(int64_t)f * t + tok] : 1.0f;
}
hipError_t launch_pad_rows(const half_t* src, const float* size_src, int frames, int t, int t_pad, int d, half_t* dst,
                           float* size_dst, hipStream_t s) {
    hipLaunchKernelGGL(pad_rows_kernel, dim3(t_pad, frames), dim3(128), 0, s, src, size_src, t, t_pad, d, dst, size_dst);
    return hipGetLastError();
}
// inverse: padded [frames, t_pad, d] -> dense [frames, t, d]
__global__ void unpad_rows_kernel(const half_t* __restrict__ src, const float* __restrict__ size_src, int t, int t_pad, int d,
                                  half_t* __restrict__ dst, float* __restrict__ size_dst) {
    const int tok = blockIdx.x, f = blockIdx.y;
    const half_t* i = src + ((int64_t)f * t_pad + tok) * d;
    half_t* o = dst + ((int64_t)f * t + tok) * d;
    for (int c = threadIdx.x; c < (d >> 3); c += blockDim.x) *(h8*)(o + c * 8) = *(const h8*)(i + c * 8);
    if (threadIdx.x == 0 && size_dst) size_dst[(int64_t)f * t + tok] = size_src ? size_src[(int64_t)f * t_pad + tok] : 1.0f;
}
hipError_t launch_unpad_rows(const half_t* src, const float* size_src, int frames, int t, int t_pad, int d, half_t* dst,
                             float* size_dst, hipStream_t s) {
    hipLaunchKernelGGL(unpad_rows_kernel, dim3(t, frames), dim3(128), 0, s, src, size_src, t, t_pad, d, dst, size_dst);
    return hipGetLastError();
}

// ---- position table for a non-native patch grid (aurora.py:909-951) --------------------------------------------------------
// The reference resamples the n x n patch rows of the checkpoint's table with
//   F.interpolate(grid, scale_factor = ((gh + 0.1) / n, (gw + 0.1) / n), mode = "bicubic")
// i.e. ATen's upsample_bicubic2d with align_corners = False and the GIVEN scale factors: the source coordinate of output index i is
// (1 / scale) * (i + 0.5) - 0.5 with 1 / scale rounded to float once, taps floor(.) - 1 .. + 2 clamped to the grid, the cubic convolution
// weights with A = -0.75, four taps along x per row first, then along y - the same expressions in the same order here, in fp32 on the fp16
// table (the engine's stored table), one thread per (output row, feature).  Row 0 (class token) is copied.
__device__ __forceinline__ float cubic_conv1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic_conv2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
__device__ __forceinline__ void cubic_coeffs(float (&c)[4], float t) {
    const float A = -0.75f;
    c[0] = cubic_conv2(t + 1.f, A);
    c[1] = cubic_conv1(t, A);
    const float u = 1.f - t;
    c[2] = cubic_conv1(u, A);
    c[3] = cubic_conv2(u + 1.f, A);
}
__global__ void pos_interp_kernel(const half_t* __restrict__ pos, int n, int gh, int gw, int d, float inv_sy, float inv_sx,
                                  half_t* __restrict__ out) {
    const int row = blockIdx.x;                     // 0 = class row, 1 + oy * gw + ox
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        if (row == 0) {
            out[c] = pos[c];
            continue;
        }
        const int oy = (row - 1) / gw, ox = (row - 1) % gw;
        const float ry = inv_sy * ((float)oy + 0.5f) - 0.5f, rx = inv_sx * ((float)ox + 0.5f) - 0.5f;
        const float fy = floorf(ry), fx = floorf(rx);
        const int iy = (int)fy, ix = (int)fx;
        float cy[4], cx[4];
        cubic_coeffs(cy, ry - fy);
        cubic_coeffs(cx, rx - fx);
        float rows[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int y = min(max(iy - 1 + k, 0), n - 1);
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = min(max(ix - 1 + j, 0), n - 1);
                v[j] = (float)pos[(int64_t)(1 + y * n + x) * d + c];
            }
            rows[k] = v[0] * cx[0] + v[1] * cx[1] + v[2] * cx[2] + v[3] * cx[3];
        }
        out[(int64_t)row * d + c] = (half_t)(rows[0] * cy[0] + rows[1] * cy[1] + rows[2] * cy[2] + rows[3] * cy[3]);
    }
}
hipError_t launch_pos_interp(const half_t* pos, int n, int gh, int gw, int d, half_t* out, hipStream_t s) {
    // 1 / scale_factor as ATen takes it: computed in double from the Python floats, rounded to float once
    const float inv_sy = (float)(1.0 / (((double)gh + 0.1) / (double)n)), inv_sx = (float)(1.0 / (((double)gw + 0.1) / (double)n));
    hipLaunchKernelGGL(pos_interp_kernel, dim3(gh * gw + 1), dim3(d < 256 ? 64 * ((d + 63) / 64) : 256), 0, s, pos, n, gh, gw, d, inv_sy, inv_sx, out);
    return hipGetLastError();
}
