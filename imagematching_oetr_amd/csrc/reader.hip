// Device-side image reader of the overlap pipeline (the READER half), gfx950.
//
// Reference: read_overlap_image (dloc/core/utils/utils.py:271-343) + process_resize (:248-265):
// after cv2.imread the picture is resized on the HOST twice with cv2.resize (INTER_LINEAR) - to
// the matcher's frame (sizes rounded up to a multiple of 32 / 8 for 'disk' / 'loftr'), and that
// result to the OETR input frame (resize[0] x resize[0]) - divided by 255 and uploaded, one
// image per call.  Here the decoded bytes are uploaded once (the caller batches the copy) and
// both resizes, the grey conversion, the /255 and the layout changes run on the GPU, writing
// straight into a slot of the [N,S,S,3] batch OETR.forward_dummy takes:
//
//   k_read_resize<U8>   decoded [h][w][3] (u8 or f32) -> bilinear -> tmp [h_new][w_new][3], 0..255
//   k_read_outputs      tmp -> bilinear -> /255 -> overlap_out [h_ov][w_ov][3]
//                       tmp -> grey (0.114 B + 0.587 G + 0.299 R) or planar -> /255 -> inp_out
//
// Bilinear = OpenCV's float32 INTER_LINEAR (pixel centres (d + 0.5) * scale - 0.5, two taps
// clamped to the border, horizontal then vertical, separate multiplies and adds).  cv2 is not
// available in the build image: the PIXELS are parity-unpinned (oracle/reader_oracle.py restates
// the same published algorithm); sizes and scale factors are pinned to the reference
// (tests/golden/reader.npz).
#include <math.h>

#include "../../include/oetr_hip.h"
#include "common.h"

namespace oetr {

struct ReadLaunch {
  const void* src;   // [h][w][3]
  int h, w, h_new, w_new, h_ov, w_ov, grayscale, swap_rb;
  float* tmp;        // [h_new][w_new][3]
  float* overlap_out;
  float* inp_out;
};

struct Tap { int i0, i1; float f; };
__device__ __forceinline__ Tap tap(int d, int src, int dst) {
  // fx = (float)((d + 0.5) * scale - 0.5) with scale in double, as OpenCV computes it
  const float fx = (float)(((double)d + 0.5) * ((double)src / (double)dst) - 0.5);
  int s = (int)floorf(fx);
  float t = fx - (float)s;
  if (s < 0) { s = 0; t = 0.f; }
  if (s >= src - 1) { s = src - 1; t = 0.f; }
  return Tap{s, min(s + 1, src - 1), t};
}
__device__ __forceinline__ float lerp2(float a, float b, float t) {
#pragma clang fp contract(off)   // a * (1 - t) + b * t with separate roundings, like the C++ it restates
  return a * (1.0f - t) + b * t;
}

template <bool U8>
__global__ __launch_bounds__(256) void k_read_resize(ReadLaunch p) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)p.h_new * p.w_new) return;
  const int dy = (int)(idx / p.w_new), dx = (int)(idx - (long)dy * p.w_new);
  float* dst = p.tmp + idx * 3;
  auto px = [&](int y, int x, int c) -> float {
    // swap_rb: image[:, :, ::-1] of the unaligned reader (utils.py:283-284)
    const size_t o = ((size_t)y * p.w + x) * 3 + (p.swap_rb ? 2 - c : c);
    return U8 ? (float)static_cast<const uint8_t*>(p.src)[o] : static_cast<const float*>(p.src)[o];
  };
  if (p.h_new == p.h && p.w_new == p.w) {   // equal sizes: a copy
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[c] = px(dy, dx, c);
    return;
  }
  const Tap tx = tap(dx, p.w, p.w_new), ty = tap(dy, p.h, p.h_new);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float r0 = lerp2(px(ty.i0, tx.i0, c), px(ty.i0, tx.i1, c), tx.f);
    const float r1 = lerp2(px(ty.i1, tx.i0, c), px(ty.i1, tx.i1, c), tx.f);
    dst[c] = lerp2(r0, r1, ty.f);
  }
}

// blockIdx.y = 0: the OETR frame; 1: the matcher's input
__global__ __launch_bounds__(256) void k_read_outputs(ReadLaunch p) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const float* t = p.tmp;
  if (blockIdx.y == 0) {
    if (idx >= (long)p.h_ov * p.w_ov) return;
    const int dy = (int)(idx / p.w_ov), dx = (int)(idx - (long)dy * p.w_ov);
    float* dst = p.overlap_out + idx * 3;
    if (p.h_ov == p.h_new && p.w_ov == p.w_new) {
#pragma unroll
      for (int c = 0; c < 3; ++c) dst[c] = t[idx * 3 + c] / 255.0f;
      return;
    }
    const Tap tx = tap(dx, p.w_new, p.w_ov), ty = tap(dy, p.h_new, p.h_ov);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float r0 = lerp2(t[((size_t)ty.i0 * p.w_new + tx.i0) * 3 + c], t[((size_t)ty.i0 * p.w_new + tx.i1) * 3 + c], tx.f);
      const float r1 = lerp2(t[((size_t)ty.i1 * p.w_new + tx.i0) * 3 + c], t[((size_t)ty.i1 * p.w_new + tx.i1) * 3 + c], tx.f);
      dst[c] = lerp2(r0, r1, ty.f) / 255.0f;
    }
  } else {
    const long npix = (long)p.h_new * p.w_new;
    if (idx >= npix) return;
    const float b = t[idx * 3], g = t[idx * 3 + 1], r = t[idx * 3 + 2];
    if (p.grayscale) {
#pragma clang fp contract(off)
      const float grey = b * 0.114f + g * 0.587f + r * 0.299f;
      p.inp_out[idx] = grey / 255.0f;
    } else {   // image.transpose((2, 0, 1)): planar, still BGR
      p.inp_out[idx] = b / 255.0f;
      p.inp_out[npix + idx] = g / 255.0f;
      p.inp_out[2 * npix + idx] = r / 255.0f;
    }
  }
}

}  // namespace oetr

using namespace oetr;

extern "C" {

oetr_status oetr_overlap_frame(int w, int h, int resize, oetr_align align, int* w_new, int* h_new,
                               int* w_ov, int* h_ov, double scales[2], double overlap_scales[2]) {
  if (w <= 0 || h <= 0 || !w_new || !h_new || !w_ov || !h_ov || !scales || !overlap_scales ||
      (resize != -1 && resize <= 0))
    return (oetr_status)set_last_error(OETR_ERR_BAD_ARG, "oetr_overlap_frame: bad argument");
  // math.ceil(w / 32) * 32 on Python floats: exact for any image size
  const int d = align == OETR_ALIGN_DISK ? 32 : align == OETR_ALIGN_LOFTR ? 8 : 1;
  *w_new = (w + d - 1) / d * d;
  *h_new = (h + d - 1) / d * d;
  *w_ov = resize == -1 ? w : resize;
  *h_ov = resize == -1 ? h : resize;
  scales[0] = (double)w / (double)*w_new;
  scales[1] = (double)h / (double)*h_new;
  overlap_scales[0] = (double)*w_new / (double)*w_ov;
  overlap_scales[1] = (double)*h_new / (double)*h_ov;
  return OETR_OK;
}

oetr_status oetr_read_overlap_image(const void* image_bgr, int is_u8, int h, int w, int h_new, int w_new,
                                    int h_ov, int w_ov, int grayscale, int swap_rb, float* tmp,
                                    float* overlap_out, float* inp_out, void* stream) {
  if (!image_bgr || !tmp || !overlap_out || !inp_out)
    return (oetr_status)set_last_error(OETR_ERR_BAD_ARG, "oetr_read_overlap_image: NULL argument");
  if (h <= 0 || w <= 0 || h_new <= 0 || w_new <= 0 || h_ov <= 0 || w_ov <= 0 ||
      (long)h * w > (1L << 28) || (long)h_new * w_new > (1L << 28) || (long)h_ov * w_ov > (1L << 28))
    return (oetr_status)set_last_error(OETR_ERR_BAD_SHAPE, "oetr_read_overlap_image: bad image / frame size");
  ReadLaunch p;
  p.src = image_bgr; p.h = h; p.w = w; p.h_new = h_new; p.w_new = w_new; p.h_ov = h_ov; p.w_ov = w_ov;
  p.grayscale = grayscale; p.swap_rb = swap_rb != 0; p.tmp = tmp; p.overlap_out = overlap_out; p.inp_out = inp_out;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned g1 = (unsigned)(((long)h_new * w_new + 255) / 256);
  if (is_u8) hipLaunchKernelGGL((k_read_resize<true>), dim3(g1), dim3(256), 0, s, p);
  else hipLaunchKernelGGL((k_read_resize<false>), dim3(g1), dim3(256), 0, s, p);
  const long m = (long)h_new * w_new > (long)h_ov * w_ov ? (long)h_new * w_new : (long)h_ov * w_ov;
  hipLaunchKernelGGL(k_read_outputs, dim3((unsigned)((m + 255) / 256), 2), dim3(256), 0, s, p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (oetr_status)set_last_error(OETR_ERR_HIP, hipGetErrorString(e));
  return OETR_OK;
}

}  // extern "C"
