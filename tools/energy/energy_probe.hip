// Energy price list of MI355X for the operations the encoder kernels are made of (round 5; DESIGN 3.14: the hot path runs
// the socket at its power cap, so time follows energy per pair).  Every kernel fills all 256 CUs with 8 waves each and runs
// for a few seconds while a host thread samples `rocm-smi` (socket W, sclk); rate = counted operations / wall time.
//   mfma   : back-to-back v_mfma_f32_32x32x16_f16, 8 independent accumulators per wave, random f16 operands
//   mfma50 : the same with the MFMA pipe half idle (s_nop padding): what a 50 %-busy kernel pays for its MFMAs
//   l2     : every workgroup streams the SAME 2 MB (L2-resident) with global_load_dwordx4 - the weight stream
//   lds    : ds_read_b128 of a 64-KB LDS tile by all eight waves - the A-operand reads
//   valu   : dependent-free v_fma_f32 / v_cvt chains - LayerNorm / GELU / conversion work
//   idle   : resident, sleeping (s_sleep) - the floor
// hipcc --offload-arch=gfx950 -O3 -o energy_probe energy_probe.hip -lpthread ; ./energy_probe [seconds per kernel]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(512) void k_mfma(const f16x8* src, float* sink, int iters, int pad) {
  const int t = threadIdx.x + blockIdx.x * 512;
  f16x8 a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = src[(t * 4 + i) & 65535];
  for (int i = 0; i < 2; ++i) b[i] = src[(t * 2 + i + 777) & 65535];
  f32x16 acc[8] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[i & 1], acc[i], 0, 0, 0);
      if (pad) asm volatile("s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7" ::: "memory");   // 32 idle issue cycles per MFMA
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
  if (s == 12345.678f) sink[t] = s;
}
__global__ __launch_bounds__(512) void k_l2(const f32x4* buf, float* sink, int iters) {   // buf: 2 MB = 131072 f32x4
  const int t = threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it)
#pragma unroll 8
    for (int i = 0; i < 256; ++i) {   // 256 x 512 threads x 16 B = 2 MB per sweep
      const f32x4 v = buf[i * 512 + t];
      acc += v;
    }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[blockIdx.x * 512 + t] = acc[0];
}
__global__ __launch_bounds__(512) void k_lds(float* sink, int iters) {
  __shared__ f32x4 tile[4096];   // 64 KB
  const int t = threadIdx.x;
  for (int i = t; i < 4096; i += 512) tile[i] = f32x4{(float)i, 1.f, 2.f, 3.f};
  __syncthreads();
  f32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it)
#pragma unroll 8
    for (int i = 0; i < 64; ++i) acc += tile[(i * 64 + (t & 63)) & 4095];   // every wave reads the whole tile, 1 KB per instruction
  if (acc[0] + acc[3] == 12345.678f) sink[blockIdx.x * 512 + t] = acc[0];
}
__global__ __launch_bounds__(512) void k_valu(float* sink, int iters, float seed) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = seed + threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = fmaf(x[i], 0.999f, 0.5f);
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += x[i];
  if (s == 12345.678f) sink[blockIdx.x * 512 + threadIdx.x] = s;
}
__global__ __launch_bounds__(512) void k_idle(float* sink, int iters) {
  for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_sleep(127);
  if (iters == -1) sink[0] = 1.f;
}

static std::atomic<bool> g_stop{false};
static std::vector<std::pair<double, int>> g_samples;
static void sampler() {
  while (!g_stop) {
    FILE* f = popen("rocm-smi -d 0 --showpower --showclocks 2>/dev/null", "r");
    if (!f) return;
    char line[512]; double w = -1; int mhz = -1;
    while (fgets(line, sizeof line, f)) {
      const char* p;
      if ((p = strstr(line, "Socket Graphics Package Power (W):"))) w = atof(p + 35);
      if ((p = strstr(line, "sclk clock level:"))) { const char* q = strchr(p, '('); if (q) mhz = atoi(q + 1); }
    }
    pclose(f);
    if (w > 0 && mhz > 0) g_samples.push_back({w, mhz});
  }
}
template <class F>
static void run(const char* name, double secs, double ops_per_launch, const char* unit, F launch) {
  launch(); CK(hipDeviceSynchronize());
  g_samples.clear(); g_stop = false;
  std::thread th(sampler);
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0; double dt = 0;
  while (dt < secs) {
    for (int i = 0; i < 8; ++i) launch();
    CK(hipDeviceSynchronize());
    launches += 8;
    dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  g_stop = true; th.join();
  std::vector<double> w; std::vector<int> c;
  for (size_t i = 1; i < g_samples.size(); ++i) { w.push_back(g_samples[i].first); c.push_back(g_samples[i].second); }
  std::sort(w.begin(), w.end()); std::sort(c.begin(), c.end());
  const double W = w.empty() ? -1 : w[w.size() / 2]; const int C = c.empty() ? -1 : c[c.size() / 2];
  const double rate = ops_per_launch * launches / dt;
  printf("%-7s %7.0f W  %5d MHz  %10.2f %s  (%zu samples)", name, W, C, rate / 1e12, unit, w.size());
  if (W > 0 && rate > 0) printf("  -> %.2f pJ per unit above the 280-W idle floor", (W - 280.0) / rate * 1e12);
  printf("\n");
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 4.0;
  const int WGS = 256;
  f16x8* src; float* sink; f32x4* buf;
  CK(hipMalloc(&src, 65536 * sizeof(f16x8))); CK(hipMalloc(&sink, WGS * 512 * 4 * 8)); CK(hipMalloc(&buf, 2 << 20));
  std::vector<_Float16> h(65536 * 8);
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
  CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  std::vector<float> hb((2 << 20) / 4);
  for (auto& v : hb) v = rand() / (float)RAND_MAX;
  CK(hipMemcpy(buf, hb.data(), 2 << 20, hipMemcpyHostToDevice));
  printf("# %d workgroups x 512 threads, %.1f s per kernel; unit of the rate column in its name (T = 1e12)\n", WGS, secs);
  run("idle", secs, 1, "T launches/s", [&] { k_idle<<<WGS, 512>>>(sink, 20000); });
  const int IT = 4000;
  const double mfma_flop = (double)WGS * 8 * IT * 8 * (2.0 * 32 * 32 * 16);
  run("mfma", secs, mfma_flop, "TFLOP/s (f16 MFMA)", [&] { k_mfma<<<WGS, 512>>>(src, sink, IT, 0); });
  run("mfma50", secs, mfma_flop / 4, "TFLOP/s (f16 MFMA)", [&] { k_mfma<<<WGS, 512>>>(src, sink, IT / 4, 1); });
  run("l2", secs, (double)WGS * 200 * (2 << 20), "TB/s (L2 -> CU)", [&] { k_l2<<<WGS, 512>>>(buf, sink, 200); });
  run("lds", secs, (double)WGS * 512 * 16.0 * 64 * 2000, "TB/s (LDS reads)", [&] { k_lds<<<WGS, 512>>>(sink, 2000); });
  run("valu", secs, (double)WGS * 512 * 2000 * 16 * 8 * 2, "TFLOP/s (VALU fma)", [&] { k_valu<<<WGS, 512>>>(sink, 2000, 1.0f); });
  return 0;
}
