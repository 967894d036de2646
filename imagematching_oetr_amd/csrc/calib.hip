// oetr_debug_mfma_rate (include/oetr_hip.h, ABI 6; bench / tests only): what this chip SUSTAINS on
// back-to-back dense f16 MFMAs right now - the yardstick bench.py quotes beside the nominal peak
// (`roofline.sustained_peak_measured`).  MI355X hits its socket power cap on this loop and settles at
// the shader clock the cap leaves (round 5 measured 1 642 TFLOP/s at 1.69 GHz on one box with a
// stand-alone probe; the number belongs to the box and the moment, so it is measured in the run that
// quotes it).  Every CU gets 8 waves x 8 independent 32x32x16 accumulators, operands are pseudo-random
// f16 values in (-1, 1) (data toggling is part of the power), nothing but MFMAs in the loop.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "common.h"

namespace oetr {
namespace {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f16x8 rand_f16x8(uint32_t seed) {
  f16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    seed = seed * 1664525u + 1013904223u;
    v[i] = (_Float16)((float)(int)(seed >> 8) * (1.0f / 8388608.0f) - 1.0f);
  }
  return v;
}

__global__ __launch_bounds__(512) void k_mfma_rate(int iters) {
  const uint32_t t = threadIdx.x + blockIdx.x * 512u;
  f16x8 a[4], b[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = rand_f16x8(t * 4u + i);
#pragma unroll
  for (int i = 0; i < 2; ++i) b[i] = rand_f16x8(t * 2u + i + 777u);
  f32x16 acc[8] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[i & 1], acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
  asm volatile("" ::"v"(s));   // (consumes the accumulators: keeps the loop alive without a store)
}
}  // namespace

// Launches of ~2 ms each, back to back on `s`, until `seconds` have passed on the host clock; the rate
// is counted MFMA FLOP / the HIP-event time around the launches of the SECOND half (the first half lets
// the power controller settle).  Synchronises `s`.
hipError_t measure_mfma_rate(int num_cus, double seconds, double* tflops, hipStream_t s) {
  const int wgs = num_cus > 0 ? num_cus : 256, iters = 4096;   // 8 waves x 8 MFMAs x 4096 x 32 cycles ~ 2 ms at 2 GHz... per SIMD: 2 waves
  const double flop_per_launch = (double)wgs * 8 * iters * 8 * (2.0 * 32 * 32 * 16);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  if (e != hipSuccess) return e;
  // one launch to learn its duration, then the number of launches that fill the two halves
  hipLaunchKernelGGL(k_mfma_rate, dim3(wgs), dim3(512), 0, s, iters);
  (void)hipEventRecord(e0, s);
  hipLaunchKernelGGL(k_mfma_rate, dim3(wgs), dim3(512), 0, s, iters);
  (void)hipEventRecord(e1, s);
  e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  if (e == hipSuccess) {
    const int half = (int)fmax(1.0, seconds * 500.0 / fmax(ms, 1e-3));
    for (int i = 0; i < half; ++i) hipLaunchKernelGGL(k_mfma_rate, dim3(wgs), dim3(512), 0, s, iters);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < half; ++i) hipLaunchKernelGGL(k_mfma_rate, dim3(wgs), dim3(512), 0, s, iters);
    (void)hipEventRecord(e1, s);
    e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e == hipSuccess) *tflops = flop_per_launch * half / (ms * 1e-3) / 1e12;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (e == hipSuccess) e = hipGetLastError();
  return e;
}

}  // namespace oetr
