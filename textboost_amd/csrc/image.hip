// Image augmentation + feeder kernels: the byte arithmetic the reference runs in Pillow / torchvision on CPU workers
// (textboost/augment/paired_augmentation.py:20-277, textboost/dataset.py:324-381), on images that stay resident in HBM.
//
// Layout: an image is RGBX u8, one uint32 per pixel ([H][stride] pixels, R in the low byte) -- every access is a naturally aligned dword, so a
// wave reads 256 contiguous bytes per row segment.  All kernels are HBM-bound byte work (no MFMA); results are BIT-EXACT with Pillow:
//   * tb_img_resample      Image.resize (libImaging Resample.c): one separable pass, 22-bit fixed-point coefficients, 8-bit intermediate
//   * tb_img_gather        every separable index map: crop, edge / zero pad, horizontal flip, collage tiling, the NEAREST affine
//                          (ImagingScaleAffine's pretabulated columns / rows), optional convert("L") luma
//   * tb_img_affine_bicubic Image.transform(AFFINE, BICUBIC) (Geometry.c bicubic_filter32RGB) in fp64 with no contraction, fused with the
//                          edge padding before it and the centre crop after it (adjust_scale is one launch)
//   * tb_img_to_pixels     crop + ToImage / ToDtype(scale) / Normalize(0.5, 0.5) -> fp32 NCHW pixel_values
// The coefficient / index tables are O(W + H) host work (libm sin, sequential accumulation) and are computed by the host entry points below,
// exactly as Pillow computes them.
#include <math.h>
#include <stdlib.h>

#include "../../include/textboost_hip.h"
#include "common.h"

#define PRECISION_BITS (32 - 8 - 2)

// ------------------------------------------------------------------------------------------------------------------ host tables
// No FMA contraction anywhere in this file (host or device): Pillow / torch round every operation, and so must these.
#pragma clang fp contract(off)
static inline double bicubic_filter_h(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}
static inline double sinc_filter_h(double x) {
  if (x == 0.0) return 1.0;
  x = x * M_PI;
  return sin(x) / x;
}
static inline double lanczos_filter_h(double x) {
  if (-3.0 <= x && x < 3.0) return sinc_filter_h(x) * sinc_filter_h(x / 3);
  return 0.0;
}

static int resample_ksize(int in_size, int out_size, int filter, double* support_out, double* scale_out) {
  double fsupport = filter == TB_IMG_BICUBIC ? 2.0 : 3.0;
  double scale = (double)((float)in_size - 0.0f) / out_size;
  double filterscale = scale < 1.0 ? 1.0 : scale;
  double support = fsupport * filterscale;
  if (support_out) *support_out = support;
  if (scale_out) *scale_out = scale;
  return (int)ceil(support) * 2 + 1;
}

extern "C" int tb_resample_ksize(int in_size, int out_size, int filter) {
  if (in_size <= 0 || out_size <= 0 || (filter != TB_IMG_BICUBIC && filter != TB_IMG_LANCZOS)) return TB_EINVAL;
  return resample_ksize(in_size, out_size, filter, nullptr, nullptr);
}

extern "C" int tb_resample_coeffs(int in_size, int out_size, int filter, int32_t* bounds, int32_t* kk) {
  if (in_size <= 0 || out_size <= 0 || (filter != TB_IMG_BICUBIC && filter != TB_IMG_LANCZOS) || !bounds || !kk) return TB_EINVAL;
  double support, scale;
  int ksize = resample_ksize(in_size, out_size, filter, &support, &scale);
  double filterscale = scale < 1.0 ? 1.0 : scale;
  double* k = (double*)malloc(sizeof(double) * ksize);
  if (!k) return TB_EINVAL;
  for (int xx = 0; xx < out_size; xx++) {
    double center = 0.0 + (xx + 0.5) * scale;
    double ww = 0.0;
    double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; x++) {
      double arg = (x + xmin - center + 0.5) * ss;
      double w = filter == TB_IMG_BICUBIC ? bicubic_filter_h(arg) : lanczos_filter_h(arg);
      k[x] = w;
      ww += w;
    }
    int32_t* ko = kk + (int64_t)xx * ksize;
    for (int x = 0; x < xmax; x++) {
      double v = ww != 0.0 ? k[x] / ww : k[x];
      ko[x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
    }
    for (int x = xmax; x < ksize; x++) ko[x] = 0;
    bounds[xx * 2 + 0] = xmin;
    bounds[xx * 2 + 1] = xmax;
  }
  free(k);
  return ksize;
}

#define COORD(v) ((v) < 0.0 ? -1 : ((int)(v)))
extern "C" int tb_affine_nearest_tables(const double* a, int in_w, int in_h, int out_w, int out_h, int32_t* xt, int32_t* yt) {
  if (!a || !xt || !yt || in_w <= 0 || in_h <= 0 || out_w <= 0 || out_h <= 0 || a[1] != 0 || a[3] != 0) return TB_EINVAL;
  double xo = a[2] + a[0] * 0.5;
  for (int x = 0; x < out_w; x++) {
    int xin = COORD(xo);
    xt[x] = (xin >= 0 && xin < in_w) ? xin : -1;
    xo += a[0];
  }
  double yo = a[5] + a[4] * 0.5;
  for (int y = 0; y < out_h; y++) {
    int yin = COORD(yo);
    yt[y] = (yin >= 0 && yin < in_h) ? yin : -1;
    yo += a[4];
  }
  return TB_OK;
}

// ---------------------------------------------------------------------------------------------------------------------- kernels
__device__ __forceinline__ uint32_t clip8(int v) {
  v >>= PRECISION_BITS;
  return (uint32_t)min(max(v, 0), 255);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned 16-byte load

// pixel (8 bit) x coefficient: M24 = the coefficient table fits signed 24 bits (the caller checked: always, unless a window's weights nearly
// cancel), so the product is the full-rate v_mad_i32_i24 instead of the quarter-rate 32-bit v_mul_lo -- the bound of the plain loop.
template <int M24>
__device__ __forceinline__ int tap(int acc, uint32_t px, int c) {
  return M24 ? acc + __mul24((int)px, c) : acc + (int)px * c;
}
#define TAP(v, c)                             \
  {                                           \
    s0 = tap<M24>(s0, (v) & 255, (c));        \
    s1 = tap<M24>(s1, ((v) >> 8) & 255, (c)); \
    s2 = tap<M24>(s2, ((v) >> 16) & 255, (c)); \
  }

// One pass of Image.resize.  VERT = 0: out[y][xx] over src[y][xmin .. xmin+cnt); VERT = 1: out[yy][x] over src[ymin .. ymin+cnt)[x].
// Horizontal blocks stage their 64 outputs' coefficient rows in LDS (each lane walks its own row: bank = (lane * ksize + k) % 32 --
// conflict-free because Resample.c's ksize = 2 ceil(support) + 1 is odd); the vertical pass keeps four row loads in flight.
template <int VERT, int M24>
__global__ __launch_bounds__(256) void resample_kernel(const uint32_t* __restrict__ src, int64_t sstride, uint32_t* __restrict__ dst,
                                                       int64_t dstride, int out_w, int out_h, const int32_t* __restrict__ bounds,
                                                       const int32_t* __restrict__ kk, int ksize) {
  extern __shared__ int32_t kk_s[];
  const int x = blockIdx.x * 64 + threadIdx.x;
  if (VERT) {
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= out_w || y >= out_h) return;
    const int ymin = bounds[2 * y], cnt = bounds[2 * y + 1];
    const int32_t* k = kk + (int64_t)y * ksize;  // wave-uniform: scalar loads
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    const uint32_t* p = src + (int64_t)ymin * sstride + x;
    int i = 0;
    for (; i + 4 <= cnt; i += 4) {
      uint32_t v0 = p[(int64_t)i * sstride], v1 = p[(int64_t)(i + 1) * sstride], v2 = p[(int64_t)(i + 2) * sstride],
               v3 = p[(int64_t)(i + 3) * sstride];
      TAP(v0, k[i]) TAP(v1, k[i + 1]) TAP(v2, k[i + 2]) TAP(v3, k[i + 3])
    }
    for (; i < cnt; i++) {
      uint32_t v = p[(int64_t)i * sstride];
      TAP(v, k[i])
    }
    dst[(int64_t)y * dstride + x] = clip8(s0) | (clip8(s1) << 8) | (clip8(s2) << 16);
  } else {
    const int x0 = blockIdx.x * 64;
    const int nx = min(64, out_w - x0);
    for (int i = threadIdx.y * 64 + threadIdx.x; i < nx * ksize; i += 256) kk_s[i] = kk[(int64_t)x0 * ksize + i];
    __syncthreads();
    const int xc = min(x, out_w - 1);
    const int xmin = bounds[2 * xc], cnt = bounds[2 * xc + 1];
    const int32_t* k = kk_s + (xc - x0) * ksize;
    for (int y = blockIdx.y * 4 + threadIdx.y; y < out_h; y += gridDim.y * 4) {
      const uint32_t* p = src + (int64_t)y * sstride;
      int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
      // plain loop (the compiler unrolls it x4 around one 16-byte load, ~9 VALU ops per tap): hand-batching 8 loads per lane (+30 %),
      // staging the row span through LDS (+10 %) and fewer, longer-running blocks (+20 %) all measured SLOWER under rocprofv3
      for (int i = 0; i < cnt; i++) {
        uint32_t v = p[xmin + i];
        TAP(v, k[i])
      }
      if (x < out_w) dst[(int64_t)y * dstride + x] = clip8(s0) | (clip8(s1) << 8) | (clip8(s2) << 16);
    }
  }
}

extern "C" int tb_img_resample(const uint32_t* src, int64_t sstride, int sw, int sh, uint32_t* dst, int64_t dstride, int out_size,
                               const int32_t* bounds, const int32_t* kk, int ksize, int vertical, int coeffs_fit_24bit, tb_stream_t stream) {
  if (!src || !dst || !bounds || !kk || sw <= 0 || sh <= 0 || out_size <= 0 || ksize <= 0 || !(ksize & 1)) return TB_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (vertical) {
    if (dstride < sw || sstride < sw) return TB_EINVAL;
    dim3 grid((sw + 63) / 64, (out_size + 3) / 4);
    if (coeffs_fit_24bit)
      hipLaunchKernelGGL((resample_kernel<1, 1>), grid, dim3(64, 4), 0, st, src, sstride, dst, dstride, sw, out_size, bounds, kk, ksize);
    else
      hipLaunchKernelGGL((resample_kernel<1, 0>), grid, dim3(64, 4), 0, st, src, sstride, dst, dstride, sw, out_size, bounds, kk, ksize);
  } else {
    if (dstride < out_size || sstride < sw) return TB_EINVAL;
    size_t lds = (size_t)64 * ksize * sizeof(int32_t);
    if (lds > 64 * 1024) return TB_EINVAL;  // ksize <= 255: down-scaling by up to ~42x with Lanczos
    // rows are strided over beyond 256 row groups (the coefficient tile is staged once per block); 128 / 256 / 384 groups for 1536 rows
    // measured 16.0 / 13.1 / 14.5 us
    const int gx = (out_size + 63) / 64;
    int gy = (sh + 3) / 4;
    if (gy > 256) gy = 256;
    dim3 grid(gx, gy);
    if (coeffs_fit_24bit)
      hipLaunchKernelGGL((resample_kernel<0, 1>), grid, dim3(64, 4), lds, st, src, sstride, dst, dstride, out_size, sh, bounds, kk, ksize);
    else
      hipLaunchKernelGGL((resample_kernel<0, 0>), grid, dim3(64, 4), lds, st, src, sstride, dst, dstride, out_size, sh, bounds, kk, ksize);
  }
  TB_CHECK_LAUNCH();
  return TB_OK;
}

// out[y][x] = (xt[x] >= 0 && yt[y] >= 0) ? f(src[yt[y]][xt[x]]) : 0 ;  f = identity or Pillow's convert("L") luma replicated to RGB
__global__ __launch_bounds__(256) void gather_kernel(const uint32_t* __restrict__ src, int64_t sstride, uint32_t* __restrict__ dst,
                                                     int64_t dstride, int dw, int dh, const int32_t* __restrict__ xt,
                                                     const int32_t* __restrict__ yt, int gray) {
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int y = blockIdx.y * 4 + threadIdx.y;
  if (x >= dw || y >= dh) return;
  const int sx = xt[x], sy = yt[y];
  uint32_t v = 0;
  if (sx >= 0 && sy >= 0) {
    v = src[(int64_t)sy * sstride + sx] & 0xffffffu;
    if (gray) {
      uint32_t l = ((v & 255) * 19595u + ((v >> 8) & 255) * 38470u + ((v >> 16) & 255) * 7471u + 0x8000u) >> 16;
      v = l | (l << 8) | (l << 16);
    }
  }
  dst[(int64_t)y * dstride + x] = v;
}

extern "C" int tb_img_gather(const uint32_t* src, int64_t sstride, uint32_t* dst, int64_t dstride, int dw, int dh, const int32_t* xt,
                             const int32_t* yt, int gray, tb_stream_t stream) {
  if (!src || !dst || !xt || !yt || dw <= 0 || dh <= 0 || dstride < dw) return TB_EINVAL;
  hipLaunchKernelGGL(gather_kernel, dim3((dw + 63) / 64, (dh + 3) / 4), dim3(64, 4), 0, (hipStream_t)stream, src, sstride, dst, dstride, dw, dh,
                     xt, yt, gray);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

// Geometry.c BICUBIC(): p1 = v2; p2 = -v1 + v3; p3 = 2 (v1 - v2) + v3 - v4; p4 = -v1 + v2 - v3 + v4; v = p1 + d (p2 + d (p3 + d p4)) -- every
// operation rounded separately (Pillow's x86-64 build has no FMA).  HIP's __dadd_rn / __dmul_rn are plain operators defined in a header that is
// compiled with contraction allowed, so they do NOT prevent fusion; the file-scope `#pragma clang fp contract(off)` above is what does.
__device__ __forceinline__ double bicubic_poly(double v1, double v2, double v3, double v4, double d) {
  double p1 = v2;
  double p2 = -v1 + v3;
  double p3 = 2 * (v1 - v2) + v3 - v4;
  double p4 = -v1 + v2 - v3 + v4;
  return p1 + d * (p2 + d * (p3 + d * p4));
}

// dst[y][x] = crop(affine_bicubic(edge_pad(src, pad_x, pad_y)))[y + oy][x + ox]; zero outside the affine output (center_crop's zero padding)
// and where the source coordinate leaves the padded image (transform's fill).
__global__ __launch_bounds__(256) void affine_bicubic_kernel(const uint32_t* __restrict__ src, int64_t sstride, int sw, int sh, int pad_x,
                                                             int pad_y, uint32_t* __restrict__ dst, int64_t dstride, int dw, int dh, int ox,
                                                             int oy, double a0, double a2, double a4, double a5) {
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int y = blockIdx.y * 4 + threadIdx.y;
  if (x >= dw || y >= dh) return;
  const int pw = sw + 2 * pad_x, ph = sh + 2 * pad_y;  // the padded image: input and output size of the transform
  const int X = x + ox, Y = y + oy;
  uint32_t out = 0;
  if (X >= 0 && X < pw && Y >= 0 && Y < ph) {
    // affine_transform(): xin = a0 * (x + 0.5) + a1 * (y + 0.5) + a2 with a1 = 0 (adds an exact zero)
    double xin = a0 * ((double)X + 0.5) + a2;
    double yin = a4 * ((double)Y + 0.5) + a5;
    if (!(xin < 0.0 || xin >= (double)pw || yin < 0.0 || yin >= (double)ph)) {
      xin -= 0.5;
      yin -= 0.5;
      const double fx = floor(xin), fy = floor(yin);
      const double dx = xin - fx, dy = yin - fy;
      const int bx = (int)fx - 1, by = (int)fy - 1;
      int cx[4], cy[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        cx[i] = min(max(min(max(bx + i, 0), pw - 1) - pad_x, 0), sw - 1);  // XCLIP on the padded image, then the edge padding
        cy[i] = min(max(min(max(by + i, 0), ph - 1) - pad_y, 0), sh - 1);
      }
      double r[4][3];
      const bool interior = cx[3] - cx[0] == 3;  // no clamping in x: the four taps are one 16-byte load
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t* row = src + (int64_t)cy[j] * sstride;
        uint32_t p0, p1, p2, p3;
        if (interior) {
          u32x4 v = *(const u32x4_u*)(row + cx[0]);
          p0 = v.x, p1 = v.y, p2 = v.z, p3 = v.w;
        } else {
          p0 = row[cx[0]], p1 = row[cx[1]], p2 = row[cx[2]], p3 = row[cx[3]];
        }
#pragma unroll
        for (int c = 0; c < 3; c++)
          r[j][c] = bicubic_poly((double)((p0 >> (8 * c)) & 255), (double)((p1 >> (8 * c)) & 255), (double)((p2 >> (8 * c)) & 255),
                                 (double)((p3 >> (8 * c)) & 255), dx);
      }
#pragma unroll
      for (int c = 0; c < 3; c++) {
        double v = bicubic_poly(r[0][c], r[1][c], r[2][c], r[3][c], dy);
        uint32_t q = v <= 0.0 ? 0u : v >= 255.0 ? 255u : (uint32_t)v;  // (UINT8) v: truncation
        out |= q << (8 * c);
      }
    }
  }
  dst[(int64_t)y * dstride + x] = out;
}

extern "C" int tb_img_affine_bicubic(const uint32_t* src, int64_t sstride, int sw, int sh, int pad_x, int pad_y, uint32_t* dst, int64_t dstride,
                                     int dw, int dh, int ox, int oy, const double* a, tb_stream_t stream) {
  if (!src || !dst || !a || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || pad_x < 0 || pad_y < 0 || dstride < dw || a[1] != 0 || a[3] != 0)
    return TB_EINVAL;
  hipLaunchKernelGGL(affine_bicubic_kernel, dim3((dw + 63) / 64, (dh + 3) / 4), dim3(64, 4), 0, (hipStream_t)stream, src, sstride, sw, sh, pad_x,
                     pad_y, dst, dstride, dw, dh, ox, oy, a[0], a[2], a[4], a[5]);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

// crop [y0, y0+R) x [x0, x0+R) + ToImage + ToDtype(float32, scale=True) + Normalize(0.5, 0.5): fl(fl(fl(v) * fl(1/255)) - 0.5) / 0.5, NCHW
__global__ __launch_bounds__(256) void to_pixels_kernel(const uint32_t* __restrict__ src, int64_t sstride, int x0, int y0,
                                                        float* __restrict__ dst, int R) {
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int y = blockIdx.y * 4 + threadIdx.y;
  if (x >= R || y >= R) return;
  const uint32_t v = src[(int64_t)(y + y0) * sstride + x + x0];
  const float k = (float)(1.0 / 255.0);
#pragma unroll
  for (int c = 0; c < 3; c++) {
    float f = (float)((v >> (8 * c)) & 255) * k;  // no contraction (file-scope pragma): the subtraction below rounds on its own
    dst[((int64_t)c * R + y) * R + x] = (f - 0.5f) / 0.5f;
  }
}

extern "C" int tb_img_to_pixels(const uint32_t* src, int64_t sstride, int sw, int sh, int x0, int y0, float* dst, int R, tb_stream_t stream) {
  if (!src || !dst || R <= 0 || x0 < 0 || y0 < 0 || x0 + R > sw || y0 + R > sh || sstride < sw) return TB_EINVAL;
  hipLaunchKernelGGL(to_pixels_kernel, dim3((R + 63) / 64, (R + 3) / 4), dim3(64, 4), 0, (hipStream_t)stream, src, sstride, x0, y0, dst, R);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

// RGB u8 [n][3] (what a decoded JPEG is) <-> RGBX u32 [n]
__global__ __launch_bounds__(256) void pack_rgb_kernel(const uint8_t* __restrict__ rgb, uint32_t* __restrict__ dst, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = (uint32_t)rgb[3 * i] | ((uint32_t)rgb[3 * i + 1] << 8) | ((uint32_t)rgb[3 * i + 2] << 16);
}
__global__ __launch_bounds__(256) void unpack_rgb_kernel(const uint32_t* __restrict__ src, int64_t sstride, int w, uint8_t* __restrict__ rgb,
                                                         int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t v = src[(i / w) * sstride + (i % w)];
  rgb[3 * i] = v & 255;
  rgb[3 * i + 1] = (v >> 8) & 255;
  rgb[3 * i + 2] = (v >> 16) & 255;
}

extern "C" int tb_img_pack_rgb(const uint8_t* rgb, uint32_t* dst, int64_t n_pixels, tb_stream_t stream) {
  if (!rgb || !dst || n_pixels <= 0) return TB_EINVAL;
  hipLaunchKernelGGL(pack_rgb_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rgb, dst, n_pixels);
  TB_CHECK_LAUNCH();
  return TB_OK;
}

extern "C" int tb_img_unpack_rgb(const uint32_t* src, int64_t sstride, int w, int h, uint8_t* rgb, tb_stream_t stream) {
  if (!src || !rgb || w <= 0 || h <= 0 || sstride < w) return TB_EINVAL;
  int64_t n = (int64_t)w * h;
  hipLaunchKernelGGL(unpack_rgb_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, sstride, w, rgb, n);
  TB_CHECK_LAUNCH();
  return TB_OK;
}
