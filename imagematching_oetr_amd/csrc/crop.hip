// Device-side box -> crop step of the matching pipeline, gfx950.
//
// Reference: the overlap branch of Matching.forward (evaluation.py:82-170) and
// tensor_overlap_crop / patch_resize (dloc/core/utils/utils.py:476-564).  There every
// pair round-trips through the host: boxes are read back (.int(), Python slicing), the
// crops go to numpy, cv2.resize(INTER_CUBIC) runs on the CPU and the result is uploaded
// again.  Here the whole step is three launches on the caller's stream, with NO
// device-to-host copy: the boxes never leave the GPU.
//
//   k_crop_geometry (1 thread)   boxes x overlap_scales, int truncation, the gate,
//                                patch_resize in double (Python floats are doubles),
//                                rounding to size_divisor -> oetr_crop_info in HBM
//   k_crop_resize   (pass 1)     crop x255 -> bicubic -> tmp   [C][new_h][new_w]
//   k_crop_resize   (pass 2)     tmp -> bicubic -> /255 -> out [C][out_h][out_w]
//                                (the reference resizes a second time when the size is
//                                rounded up to a multiple of size_divisor; with equal
//                                sizes the cubic weights are exactly (0,1,0,0): a copy)
// When the gate fails the full images are passed through (evaluation.py:142-170): copied
// bit for bit, no resize passes.
//
// The sizes are data dependent, so the grids cover the caller's capacity and every
// thread reads the geometry first.  Bicubic = OpenCV's float path (a = -0.75, pixel
// centres (d + 0.5) * scale - 0.5, taps clamped to the border, horizontal then
// vertical); cv2 is not available in the build image: resize numerics are
// parity-unpinned (oracle/crop_oracle.py restates the same published algorithm).
#include "../../include/oetr_hip.h"
#include "common.h"

namespace oetr {

struct CropLaunch {
  const float* image[2];   // [C][h][w]
  int h[2], w[2];
  int channels;
  const float* box[2];     // device [4] xyxy in the OETR input frame
  float scale[2][2];       // overlap_scales (sx, sy) per image
  int keep_aspect, size_divisor, gate_mode;
  int cap_h, cap_w;        // capacity of tmp / out per channel plane
  float* tmp[2];
  float* out[2];
  oetr_crop_info* info;
};

__global__ void k_crop_geometry(CropLaunch p) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  oetr_crop_info g;
  int bw[2], bh[2];
  for (int i = 0; i < 2; ++i) {
    for (int j = 0; j < 4; ++j) {
      // bbox * overlap_scales in float32 (torch tensor product), then .int() truncation
      const float v = p.box[i][j] * p.scale[i][j & 1];
      g.sbox[i][j] = v;
      g.box[i][j] = (int)v;
    }
    bw[i] = g.box[i][2] - g.box[i][0];
    bh[i] = g.box[i][3] - g.box[i][1];
  }
  int mn = min(min(bw[0], bh[0]), min(bw[1], bh[1]));
  bool valid = mn > 1;
  if (valid && p.gate_mode == 1) {   // 'pragueparks-val': integer floor_divide scores
    const int score = max(max(bw[0] / bw[1], bh[0] / bh[1]), max(bw[1] / bw[0], bh[1] / bh[0]));
    valid = score > 2;
  }
  // the larger-area image provides the target size (utils.py:525-534)
  const long a0 = (long)p.w[0] * p.h[0], a1 = (long)p.w[1] * p.h[1];
  const int ow = a0 >= a1 ? p.w[0] : p.w[1], oh = a0 >= a1 ? p.h[0] : p.h[1];
  for (int i = 0; i < 2; ++i) {
    if (!valid) {
      g.box[i][0] = 0; g.box[i][1] = 0; g.box[i][2] = p.w[i]; g.box[i][3] = p.h[i];
      g.sbox[i][0] = 0.f; g.sbox[i][1] = 0.f; g.sbox[i][2] = (float)p.w[i]; g.sbox[i][3] = (float)p.h[i];
      g.crop_w[i] = g.new_w[i] = g.out_w[i] = p.w[i];
      g.crop_h[i] = g.new_h[i] = g.out_h[i] = p.h[i];
      g.ratio[i][0] = g.ratio[i][1] = 1.0;
      continue;
    }
    // python slicing image[:, y1:y2, x1:x2] clamps at the border
    const int x1 = min(g.box[i][0], p.w[i]), x2 = min(g.box[i][2], p.w[i]);
    const int y1 = min(g.box[i][1], p.h[i]), y2 = min(g.box[i][3], p.h[i]);
    const int cw = max(0, x2 - x1), ch = max(0, y2 - y1);
    g.crop_w[i] = cw; g.crop_h[i] = ch;
    double rx, ry, nw, nh;
    if (p.keep_aspect) {   // patch_resize, extractor != 'disk'
      if ((double)ow / (double)cw > (double)oh / (double)ch) {
        rx = (double)oh / (double)ch; nw = rx * (double)cw; nh = (double)oh;
      } else {
        rx = (double)ow / (double)cw; nw = (double)ow; nh = rx * (double)ch;
      }
      ry = rx;
    } else {
      rx = (double)ow / (double)cw; ry = (double)oh / (double)ch; nw = (double)ow; nh = (double)oh;
    }
    g.ratio[i][0] = rx; g.ratio[i][1] = ry;
    g.new_w[i] = (int)nw; g.new_h[i] = (int)nh;
    g.out_w[i] = g.new_w[i]; g.out_h[i] = g.new_h[i];
    if (p.size_divisor > 1) {   // math.ceil(new / d) * d in double
      g.out_w[i] = (int)ceil((double)g.new_w[i] / p.size_divisor) * p.size_divisor;
      g.out_h[i] = (int)ceil((double)g.new_h[i] / p.size_divisor) * p.size_divisor;
    }
  }
  // A crop that does not fit the caller's capacity, or a degenerate one (the reference would
  // raise from cv2.resize there), cannot be produced: valid = -1, sizes zeroed - NOT the same
  // thing as a failed gate (valid = 0: the images pass through untouched).
  bool fits = true;
  for (int i = 0; i < 2; ++i)
    if (g.out_w[i] > p.cap_w || g.out_h[i] > p.cap_h || g.new_w[i] > p.cap_w || g.new_h[i] > p.cap_h ||
        g.new_w[i] <= 0 || g.new_h[i] <= 0)
      fits = false;
  if (!fits)
    for (int i = 0; i < 2; ++i) g.out_w[i] = g.out_h[i] = g.new_w[i] = g.new_h[i] = 0;
  g.valid = !fits ? -1 : (valid ? 1 : 0);
  *p.info = g;
}

// OpenCV interpolateCubic (imgproc/resize.cpp), float32
__device__ __forceinline__ void cubic_weights(float t, float (&w)[4]) {
#pragma clang fp contract(off)   // separate multiplies and adds, like the C++ it restates
  const float a = -0.75f;
  w[0] = ((a * (t + 1.f) - 5.f * a) * (t + 1.f) + 8.f * a) * (t + 1.f) - 4.f * a;
  w[1] = ((a + 2.f) * t - (a + 3.f)) * t * t + 1.f;
  const float u = 1.f - t;
  w[2] = ((a + 2.f) * u - (a + 3.f)) * u * u + 1.f;
  w[3] = 1.f - w[0] - w[1] - w[2];
}

// One bicubic pass for both images (blockIdx.y = image, blockIdx.z = channel).
// PASS 1: src = the crop rectangle of the image, x255;  PASS 2: src = tmp, result /255.
template <int PASS>
__global__ __launch_bounds__(256) void k_crop_resize(CropLaunch p) {
  const int im = blockIdx.y, c = blockIdx.z;
  const oetr_crop_info& g = *p.info;
  if (g.valid != 1) {
    // gate failed: the reference hands data['image0'/'image1'] back untouched
    // (evaluation.py:142-170) - a plain copy, bit for bit, in the second pass only
    if (PASS == 2 && g.valid == 0) {
      const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
      const long npix = (long)p.w[im] * p.h[im];
      if (idx < npix) p.out[im][(size_t)c * npix + idx] = p.image[im][(size_t)c * npix + idx];
    }
    return;
  }
  int sw, sh, dw, dh, x0, y0, spitch;
  const float* src;
  float* dst;
  if (PASS == 1) {
    x0 = min(g.box[im][0], p.w[im]); y0 = min(g.box[im][1], p.h[im]);
    sw = g.crop_w[im]; sh = g.crop_h[im];
    dw = g.new_w[im]; dh = g.new_h[im];
    spitch = p.w[im];
    src = p.image[im] + (size_t)c * p.h[im] * p.w[im];
    dst = p.tmp[im] + (size_t)c * dw * dh;
  } else {
    x0 = y0 = 0;
    sw = g.new_w[im]; sh = g.new_h[im];
    dw = g.out_w[im]; dh = g.out_h[im];
    spitch = sw;
    src = p.tmp[im] + (size_t)c * sw * sh;
    dst = p.out[im] + (size_t)c * dw * dh;
  }
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)dw * dh || sw <= 0 || sh <= 0) return;
  const int dy = (int)(idx / dw), dx = (int)(idx - (long)dy * dw);
  // fx = (float)((dx + 0.5) * scale_x - 0.5) with scale in double, as OpenCV computes it
  const float fx = (float)(((double)dx + 0.5) * ((double)sw / (double)dw) - 0.5);
  const float fy = (float)(((double)dy + 0.5) * ((double)sh / (double)dh) - 0.5);
  const int sx = (int)floorf(fx), sy = (int)floorf(fy);
  float wx[4], wy[4];
  cubic_weights(fx - (float)sx, wx);
  cubic_weights(fy - (float)sy, wy);
  const float in_scale = PASS == 1 ? 255.0f : 1.0f;
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int yy = min(max(sy - 1 + j, 0), sh - 1);
    const float* row = src + (size_t)(y0 + yy) * spitch + x0;
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int xx = min(max(sx - 1 + i, 0), sw - 1);
      r = __fadd_rn(r, __fmul_rn(row[xx] * in_scale, wx[i]));   // hresize: S[..]*a0 + ... left to right
    }
    acc = __fadd_rn(acc, __fmul_rn(r, wy[j]));                  // vresize: S0*b0 + S1*b1 + ...
  }
  dst[idx] = PASS == 1 ? acc : acc / 255.0f;
}

hipError_t launch_overlap_crop(const CropLaunch& p, hipStream_t s) {
  hipLaunchKernelGGL(k_crop_geometry, dim3(1), dim3(64), 0, s, p);
  const long cap = (long)p.cap_h * p.cap_w;
  const dim3 grid((unsigned)((cap + 255) / 256), 2, p.channels);
  hipLaunchKernelGGL((k_crop_resize<1>), grid, dim3(256), 0, s, p);
  hipLaunchKernelGGL((k_crop_resize<2>), grid, dim3(256), 0, s, p);
  return hipGetLastError();
}

}  // namespace oetr

using namespace oetr;

extern "C" {

size_t oetr_overlap_crop_capacity(int channels, int h1, int w1, int h2, int w2, int size_divisor,
                                  int* cap_h, int* cap_w) {
  if (channels <= 0 || h1 <= 0 || w1 <= 0 || h2 <= 0 || w2 <= 0 || size_divisor < 1) return 0;
  const int d = size_divisor;
  const int ch = ((max(h1, h2) + d - 1) / d) * d, cw = ((max(w1, w2) + d - 1) / d) * d;
  if (cap_h) *cap_h = ch;
  if (cap_w) *cap_w = cw;
  return (size_t)channels * ch * cw;
}

oetr_status oetr_overlap_crop(const float* image1, const float* image2, int channels, int h1, int w1,
                              int h2, int w2, const float* box1, const float* box2,
                              const float scale1[2], const float scale2[2], int keep_aspect,
                              int size_divisor, int gate_mode, float* tmp, float* out1, float* out2,
                              size_t capacity_floats, oetr_crop_info* info, void* stream) {
  if (!image1 || !image2 || !box1 || !box2 || !scale1 || !scale2 || !tmp || !out1 || !out2 || !info)
    return (oetr_status)set_last_error(OETR_ERR_BAD_ARG, "oetr_overlap_crop: NULL argument");
  int cap_h = 0, cap_w = 0;
  const size_t need = oetr_overlap_crop_capacity(channels, h1, w1, h2, w2, size_divisor, &cap_h, &cap_w);
  if (need == 0 || (gate_mode != 0 && gate_mode != 1))
    return (oetr_status)set_last_error(OETR_ERR_BAD_ARG, "oetr_overlap_crop: bad shape / size_divisor / gate_mode");
  if (capacity_floats < need)
    return (oetr_status)set_last_error(OETR_ERR_WORKSPACE, "oetr_overlap_crop: buffers smaller than oetr_overlap_crop_capacity()");
  CropLaunch p;
  p.image[0] = image1; p.image[1] = image2;
  p.h[0] = h1; p.w[0] = w1; p.h[1] = h2; p.w[1] = w2;
  p.channels = channels;
  p.box[0] = box1; p.box[1] = box2;
  p.scale[0][0] = scale1[0]; p.scale[0][1] = scale1[1];
  p.scale[1][0] = scale2[0]; p.scale[1][1] = scale2[1];
  p.keep_aspect = keep_aspect; p.size_divisor = size_divisor; p.gate_mode = gate_mode;
  p.cap_h = cap_h; p.cap_w = cap_w;
  p.tmp[0] = tmp; p.tmp[1] = tmp + need;
  p.out[0] = out1; p.out[1] = out2;
  p.info = info;
  const hipError_t e = launch_overlap_crop(p, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return (oetr_status)set_last_error(OETR_ERR_HIP, hipGetErrorString(e));
  return OETR_OK;
}

}  // extern "C"
