// Probe: the encoder's OWN split-f16 KV-state routine (encoder.hip: kv_state_64 with
// -DOETR_SPLIT_STATE=1, the form that gave timing-dependent states inside the encoder) called in
// isolation - waves 0-3 of a workgroup repeat it on fixed register inputs while the sibling
// waves 4-7 of their SIMDs idle / stream MFMAs / stream global loads / both.  Every repetition
// of every workgroup must produce the same bits.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DOETR_SPLIT_STATE=1 -DOETR_SPLIT3_PAD_OFF=1 -I imagematching_oetr_amd/csrc -o tools/bin/kv_state_probe tools/kv_state_probe.hip
#include "../imagematching_oetr_amd/csrc/encoder.hip"
#include <cstdio>
#include <vector>
using namespace oetr;

__global__ __launch_bounds__(512) void k_probe(float* out, int iters, int aggr, const float* lbuf, int nvalid) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int half = lane >> 5;
  if (wave < 4) {
    f32x16 accK[2], accV[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        accK[mt][r] = 0.01f * (float)((lane * 7 + r * 13 + mt * 5) % 97) - 0.4f;
        accV[mt][r] = 0.02f * (float)((lane * 11 + r * 3 + mt * 17) % 89) - 0.8f;
      }
    Range rg;
    f32x16 ref;
    float kref = 0.f;
    int bad = 0;
    for (int it = 0; it < iters; ++it) {
      f32x16 kv;
      float ksum;
      // (keep the inputs opaque so the call is not hoisted out of the loop)
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(accK[0][r]), "+v"(accV[0][r]), "+v"(accK[1][r]), "+v"(accV[1][r]));
      kv_state_64<GM_SPLIT>(accK, accV, 400, nvalid, half, nvalid > 32, kv, ksum, rg);
      if (it == 0) { ref = kv; kref = ksum; }
      else {
#pragma unroll
        for (int r = 0; r < 16; ++r) bad += __float_as_uint(kv[r]) != __float_as_uint(ref[r]);
        bad += __float_as_uint(ksum) != __float_as_uint(kref);
      }
    }
    out[((size_t)blockIdx.x * 4 + wave) * 64 + lane] = (float)bad;
    if (rg.fm == 12345.f) out[0] = 1.f;
  } else if (aggr) {
    // sibling waves: 1 = MFMA stream, 2 = global-load stream, 3 = both
    f32x16 a0 = {0}, a1 = {0};
    f32x4 x = {1.f, 2.f, 3.f, 4.f}, y = {0.5f, 0.25f, 0.125f, 1.f};
    const f32x4* src = reinterpret_cast<const f32x4*>(lbuf) + lane;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < 3 * iters; ++it) {
      if (aggr & 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a0 = mma16<GM_SPLIT>(x, y, a0);
          a1 = mma16<GM_SPLIT>(y, x, a1);
        }
      }
      if (aggr & 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += src[((it * 4 + j) & 4095) * 64];
      }
    }
    if (a0[0] + a1[0] + acc[0] == 12345.f) out[1] = 1.f;
  }
}

int main() {
  const int blocks = 256, iters = 400;
  float *d, *lbuf;
  hipMalloc(&d, (size_t)blocks * 4 * 64 * sizeof(float));
  hipMalloc(&lbuf, (size_t)8 << 20);
  hipMemset(lbuf, 0, (size_t)8 << 20);
  std::vector<float> h((size_t)blocks * 4 * 64);
  for (int nvalid : {64, 49, 16})
    for (int aggr = 0; aggr < 4; ++aggr) {
      long bad = 0, lanes[4] = {0, 0, 0, 0};
      for (int rep = 0; rep < 5; ++rep) {
        hipMemset(d, 0, h.size() * sizeof(float));
        hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(512), 0, 0, d, iters, aggr, lbuf, nvalid);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        hipMemcpy(h.data(), d, h.size() * sizeof(float), hipMemcpyDeviceToHost);
        for (size_t i = 0; i < h.size(); ++i) if (h[i] != 0.f) { bad += (long)h[i]; ++lanes[(i % 64) / 16]; }
      }
      printf("kv_state_64 (split f16), %2d valid rows, sibling stream %d (0 none, 1 MFMA, 2 loads, 3 both): %ld differing values in %d x %d repetitions (lanes 0-15: %ld, 16-31: %ld, 32-47: %ld, 48-63: %ld)\n",
             nvalid, aggr, bad, 5 * blocks * 4, iters, lanes[0], lanes[1], lanes[2], lanes[3]);
    }
  return 0;
}
