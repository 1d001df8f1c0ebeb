// Input stage: decoded RGB frames -> pixel_values, on the GPU.
//
// Restates what the reference does on the host with CLIPImageProcessor(size=378, crop_size=378) followed by
// .to(float16) (inference.py:58-63, 71-75): Pillow's 8-bit BICUBIC resize of the shortest edge (separable, horizontal
// pass first, uint8 intermediate, 22-bit fixed-point taps computed in doubles - Pillow Resample.c), centre crop,
// rescale + normalise.  The taps are computed once per input size on the host (aur_preprocess_plan, doubles, same
// operation order as Pillow so the int32 taps are identical); the device side is pure integer work plus a
// [3][256] fp16 lookup for rescale/normalise, hence bit-exact against the reference processor.
//
// Only the 378 output columns / rows that survive the centre crop are ever computed.
// HBM traffic per frame: H*W*3 (read, only the cropped column range) + 2*H*378*3 (tmp) + 378*378*3*2 (out).
// Plan layout (int32): hdr[8] = {new_h, new_w, top, left, ks_h, ks_v, in_h, in_w}; bounds_h[378][2]; taps_h[ks_h][378]
// (tap-major); bounds_v[378][2]; taps_v[378][ks_v].
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/aurora_hip.h"
#include "kernels.h"

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;
constexpr int kHdr = 8;

inline double bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

inline int axis_ksize(int in_size, int out_size) {
    double scale = (double)in_size / out_size;
    if (scale < 1.0) scale = 1.0;
    return (int)std::ceil(2.0 * scale) * 2 + 1;
}

// taps for output coordinates [first, first+count) of an axis resampled in_size -> out_size
void axis_taps(int in_size, int out_size, int first, int count, int ksize, int32_t* bounds, int32_t* coef) {
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const double ss = 1.0 / filterscale;
    double k[64 * 2 + 1];
    for (int o = 0; o < count; ++o) {
        const int xx = first + o;
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            const double w = bicubic((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        int32_t* c = coef + (int64_t)o * ksize;
        for (int x = 0; x < ksize; ++x) c[x] = 0;
        for (int x = 0; x < xmax; ++x) {
            double v = k[x];
            if (ww != 0.0) v /= ww;
            c[x] = v < 0 ? (int32_t)(-0.5 + v * (double)(1 << kPrecisionBits)) : (int32_t)(0.5 + v * (double)(1 << kPrecisionBits));
        }
        bounds[2 * o] = xmin;
        bounds[2 * o + 1] = xmax;
    }
}

struct Geometry {
    int new_h, new_w, top, left, ks_h, ks_v;
};

bool geometry(int in_h, int in_w, int image, Geometry* g) {
    if (in_h < 1 || in_w < 1 || image < 1 || in_h > (1 << 15) || in_w > (1 << 15)) return false;
    // transformers get_resize_output_image_size(default_to_square=False): int(size * long / short), python doubles
    const int shrt = in_w <= in_h ? in_w : in_h, lng = in_w <= in_h ? in_h : in_w;
    const int new_long = (int)((double)(image * (double)lng) / (double)shrt);
    g->new_h = in_w <= in_h ? new_long : image;
    g->new_w = in_w <= in_h ? image : new_long;
    g->top = (g->new_h - image) / 2;
    g->left = (g->new_w - image) / 2;
    g->ks_h = axis_ksize(in_w, g->new_w);
    g->ks_v = axis_ksize(in_h, g->new_h);
    return g->ks_h <= 129 && g->ks_v <= 129;              // scale <= 32
}

// clip8(acc >> 22) written as clamp-then-shift.  The shift-then-clamp form is selected as v_ashr_pk_u8_i32 by ROCm 7.2's
// hipcc for gfx950, whose result has live garbage in bits 16..31 when it is OR-ed into a packed dword (observed: bytes
// 2/3 of every dword wrong whenever the unrolled tap loop had run) - keep this form.
__device__ __forceinline__ uint32_t clip8(int acc) {
    return (uint32_t)min(max(acc, 0), (255 << kPrecisionBits) | ((1 << kPrecisionBits) - 1)) >> kPrecisionBits;
}

inline int tmp_pitch(int image) { return (image * 3 + 3) & ~3; }

// horizontal pass, one workgroup per input row: the byte range of the row that the 378 surviving columns touch is
// staged in LDS with coalesced dword loads, then thread j forms output column j (3 channels) from LDS.
//   tmp[row][j*3 + c] = clip8((2^21 + sum_x src[row][xmin_j + x][c] * k_j[x]) >> 22),   taps stored [x][j] (coalesced)
__global__ __launch_bounds__(384) void resize_h_kernel(const uint8_t* __restrict__ src, const int32_t* __restrict__ plan,
                                                       uint8_t* __restrict__ tmp, int in_w, int image, int pitch,
                                                       int64_t src_bytes) {
    extern __shared__ uint32_t s_row[];
    const int64_t row = blockIdx.x;                         // f * in_h + y
    const int32_t* bh = plan + kHdr;
    const int32_t* cht = bh + 2 * image;                    // [ks_h][image]
    const int xlo = bh[0], xhi = bh[2 * (image - 1)] + bh[2 * (image - 1) + 1];
    const int64_t byte0 = (row * in_w + xlo) * 3, byte1 = (row * in_w + xhi) * 3;
    const int64_t a0 = byte0 & ~(int64_t)3;
    const int ndw = (int)((byte1 - a0 + 3) >> 2);
    for (int t = threadIdx.x; t < ndw; t += 384) {
        const int64_t a = a0 + 4 * (int64_t)t;
        uint32_t v;
        if (a + 4 <= src_bytes) {
            v = *(const uint32_t*)(src + a);
        } else {                                            // last dword of the whole buffer: never read past its end
            v = 0;
            for (int b = 0; b < 4 && a + b < src_bytes; ++b) v |= (uint32_t)src[a + b] << (8 * b);
        }
        s_row[t] = v;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < image; j += 384) {
        const int xmin = bh[2 * j], n = bh[2 * j + 1];
        const uint8_t* p = (const uint8_t*)s_row + (int)(byte0 - a0) + (xmin - xlo) * 3;
        int c0 = 1 << (kPrecisionBits - 1), c1 = c0, c2 = c0;
#pragma unroll 4
        for (int x = 0; x < n; ++x) {
            const int k = cht[(int64_t)x * image + j];
            c0 += (int)p[3 * x] * k;
            c1 += (int)p[3 * x + 1] * k;
            c2 += (int)p[3 * x + 2] * k;
        }
        uint8_t* o = tmp + row * pitch + j * 3;
        o[0] = (uint8_t)clip8(c0);
        o[1] = (uint8_t)clip8(c1);
        o[2] = (uint8_t)clip8(c2);
    }
}

// vertical pass, one workgroup per output row: each thread owns 4 consecutive bytes (interleaved channels) of the row
// and walks the taps with coalesced dword loads (tap values are wave-uniform); the finished uint8 row goes through LDS
// so that the rescale/normalise lookup can be stored planar and coalesced:  out[f][c][i][j] = lut[c][row[j*3 + c]]
__global__ __launch_bounds__(384) void resize_v_kernel(const uint8_t* __restrict__ tmp, const int32_t* __restrict__ plan,
                                                       const uint16_t* __restrict__ lut, uint16_t* __restrict__ out, int in_h,
                                                       int image, int pitch) {
    extern __shared__ uint32_t s_dyn[];
    uint16_t* s_lut = (uint16_t*)s_dyn;                      // [3][256]
    uint32_t* s_out = s_dyn + 3 * 256 / 2;                   // pitch bytes
    for (int t = threadIdx.x; t < 3 * 256 / 2; t += 384) s_dyn[t] = ((const uint32_t*)lut)[t];
    const int i = blockIdx.x % image, f = blockIdx.x / image;
    const int ks_h = plan[4], ks = plan[5];
    const int32_t* bv = plan + kHdr + 2 * image + (int64_t)image * ks_h;
    const int32_t* cv = bv + 2 * image + (int64_t)i * ks;
    const int ymin = bv[2 * i], n = bv[2 * i + 1];
    const int ndw = pitch >> 2;
    const uint32_t* base = (const uint32_t*)(tmp + ((int64_t)f * in_h + ymin) * pitch);
    for (int t = threadIdx.x; t < ndw; t += 384) {
        const uint32_t* p = base + t;
        int c0 = 1 << (kPrecisionBits - 1), c1 = c0, c2 = c0, c3 = c0;
#pragma unroll 4
        for (int y = 0; y < n; ++y) {
            const uint32_t v = p[(int64_t)y * ndw];
            const int k = cv[y];
            c0 += (int)(v & 255u) * k;
            c1 += (int)((v >> 8) & 255u) * k;
            c2 += (int)((v >> 16) & 255u) * k;
            c3 += (int)(v >> 24) * k;
        }
        s_out[t] = clip8(c0) | (clip8(c1) << 8) | (clip8(c2) << 16) | (clip8(c3) << 24);
    }
    __syncthreads();
    const uint8_t* r = (const uint8_t*)s_out;
    const int64_t plane = (int64_t)image * image;
    uint16_t* o = out + (int64_t)f * 3 * plane + (int64_t)i * image;
    for (int j = threadIdx.x; j < image; j += 384) {
        o[j] = s_lut[r[3 * j]];
        o[plane + j] = s_lut[256 + r[3 * j + 1]];
        o[2 * plane + j] = s_lut[512 + r[3 * j + 2]];
    }
}

}  // namespace

extern "C" int64_t aur_preprocess_plan_len(int32_t in_h, int32_t in_w, int32_t image) {
    Geometry g;
    if (!geometry(in_h, in_w, image, &g)) return -1;
    return kHdr + 2 * (int64_t)image + (int64_t)image * g.ks_h + 2 * (int64_t)image + (int64_t)image * g.ks_v;
}

extern "C" int aur_preprocess_plan(int32_t in_h, int32_t in_w, int32_t image, int32_t* plan_host, int64_t len) {
    Geometry g;
    if (!plan_host || !geometry(in_h, in_w, image, &g) || len != aur_preprocess_plan_len(in_h, in_w, image)) return AUR_ERR_ARG;
    int32_t* p = plan_host;
    p[0] = g.new_h, p[1] = g.new_w, p[2] = g.top, p[3] = g.left, p[4] = g.ks_h, p[5] = g.ks_v, p[6] = in_h, p[7] = in_w;
    int32_t* bh = p + kHdr;
    int32_t* ch = bh + 2 * image;
    int32_t* bv = ch + (int64_t)image * g.ks_h;
    int32_t* cv = bv + 2 * image;
    std::vector<int32_t> rowmajor((size_t)image * g.ks_h);
    axis_taps(in_w, g.new_w, g.left, image, g.ks_h, bh, rowmajor.data());
    for (int j = 0; j < image; ++j)                           // horizontal taps are stored [tap][column]: coalesced on the device
        for (int x = 0; x < g.ks_h; ++x) ch[(int64_t)x * image + j] = rowmajor[(size_t)j * g.ks_h + x];
    axis_taps(in_h, g.new_h, g.top, image, g.ks_v, bv, cv);
    return AUR_OK;
}

extern "C" int64_t aur_preprocess_tmp_bytes(int32_t frames, int32_t in_h, int32_t in_w, int32_t image) {
    if (frames < 1 || in_h < 1 || in_w < 1 || image < 1) return -1;
    return (int64_t)frames * in_h * tmp_pitch(image);
}

extern "C" int aur_preprocess_frames(const uint8_t* frames_dev, int32_t frames, int32_t in_h, int32_t in_w, int32_t image,
                                     const int32_t* plan_dev, const uint16_t* lut_dev, uint8_t* tmp_dev, void* out_pixels,
                                     void* stream) {
    if (!frames_dev || !plan_dev || !lut_dev || !tmp_dev || !out_pixels || frames < 1 || in_h < 1 || in_w < 1 || image < 1)
        return AUR_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    static bool attr_done = false;
    if (!attr_done) {                                          // rows of up to 32768 pixels (96 KiB) are staged in LDS
        if (hipFuncSetAttribute((const void*)resize_h_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
            return AUR_ERR_HIP;
        attr_done = true;
    }
    const int pitch = tmp_pitch(image);
    const size_t lds_h = ((size_t)in_w * 3 + 8 + 3) & ~(size_t)3;
    if ((int64_t)frames * in_h > 0x7fffffffLL || (int64_t)frames * image > 0x7fffffffLL) return AUR_ERR_ARG;
    resize_h_kernel<<<(unsigned)(frames * in_h), 384, lds_h, s>>>(frames_dev, plan_dev, tmp_dev, in_w, image, pitch,
                                                                (int64_t)frames * in_h * in_w * 3);
    resize_v_kernel<<<(unsigned)(frames * image), 384, 3 * 256 * 2 + pitch, s>>>(tmp_dev, plan_dev, lut_dev, (uint16_t*)out_pixels,
                                                                               in_h, image, pitch);
    return hipGetLastError() == hipSuccess ? AUR_OK : AUR_ERR_HIP;
}
