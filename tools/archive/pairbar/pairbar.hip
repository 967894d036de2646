// What does an in-kernel barrier between the workgroups of one image pair cost against a launch boundary?
// 208 workgroups (8 pairs x 26 = 2 images x 13 tiles, one workgroup per CU through 80 KB of LDS).  A "layer": every workgroup
// reads the 13 32-KB partial states of one image of its pair (written in the previous layer), spins for WORK clocks, writes its
// own state for the next layer (double-buffered) - the memory traffic of the serial encoder launch at 8 pairs @640x640.
//   mode 0: one launch per layer (the shipped structure)
//   mode 1: ONE launch, pair barrier with agent-scope release / acquire fences (buffer_wbl2 sc1 ... buffer_inv sc1)
//   mode 2: ONE launch, pair barrier for workgroups that share an XCD's L2 (block % 8): stores + s_waitcnt, relaxed L2
//           atomics, state loads with sc0 (bypass the CU's L1)
//   mode 3: ONE launch, sc1 stores + sc1 loads (device-coherent accesses), relaxed atomics
//   mode 4: ONE launch, plain stores + sc1 loads (agent scope: past the L1 for certain - sc0 = workgroup scope may hit it)
//   mode 5: mode 4 with the arrival atomics in the XCD's own L2 (no scope bits), polled with sc1 loads
// Every value read is checked against the tag its writer stored.   hipcc --offload-arch=gfx950 -O3 -o pairbar pairbar.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int WGS = 208, PAIRS = 8, PER = 26, TILES = 13, STATE_F = 32 * 256, THREADS = 512;

template <int MODE>
__device__ __forceinline__ void layer(float* state, unsigned* cnt, unsigned* bad, int it, int work) {
  extern __shared__ float lds[];
  const int b = blockIdx.x, pair = b % PAIRS, member = b / PAIRS, t = threadIdx.x;
  const float* rd = state + (size_t)(it & 1) * WGS * STATE_F;
  float* wr = state + (size_t)((it + 1) & 1) * WGS * STATE_F;
  // ---- read the 13 states of "my" image (members of my parity), written in layer it - 1
  unsigned wrong = 0;
  if (it > 0)
    for (int m = 0; m < TILES; ++m) {
      const int src_member = 2 * m + (member & 1);
      const float* src = rd + (size_t)(src_member * PAIRS + pair) * STATE_F + t * 4;
      const float want = (float)((it - 1) * 1000 + src_member);
      f4 v0, v1, v2, v3;
      if (MODE == 2) {
        asm volatile("global_load_dwordx4 %0, %4, off sc0\n global_load_dwordx4 %1, %5, off sc0\n"
                     "global_load_dwordx4 %2, %6, off sc0\n global_load_dwordx4 %3, %7, off sc0\n s_waitcnt vmcnt(0)"
                     : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(src), "v"(src + 2048), "v"(src + 4096), "v"(src + 6144) : "memory");
      } else if (MODE >= 3) {
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n global_load_dwordx4 %1, %5, off sc1\n"
                     "global_load_dwordx4 %2, %6, off sc1\n global_load_dwordx4 %3, %7, off sc1\n s_waitcnt vmcnt(0)"
                     : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(src), "v"(src + 2048), "v"(src + 4096), "v"(src + 6144) : "memory");
      } else {
        v0 = *reinterpret_cast<const f4*>(src); v1 = *reinterpret_cast<const f4*>(src + 2048);
        v2 = *reinterpret_cast<const f4*>(src + 4096); v3 = *reinterpret_cast<const f4*>(src + 6144);
      }
      wrong += (v0.x != want) + (v0.w != want) + (v1.y != want) + (v2.z != want) + (v3.x != want) + (v3.w != want);
    }
  if (wrong) atomicAdd(bad, wrong);
  // ---- "compute"
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < work) __builtin_amdgcn_s_sleep(8);
  // ---- write my state for the next layer
  const float tag = (float)(it * 1000 + member);
  float* mine = wr + (size_t)b * STATE_F + t * 4;
  const f4 v = {tag, tag, tag, tag};
  if (MODE == 3) {
    asm volatile("global_store_dwordx4 %0, %4, off sc1\n global_store_dwordx4 %1, %4, off sc1\n"
                 "global_store_dwordx4 %2, %4, off sc1\n global_store_dwordx4 %3, %4, off sc1"
                 :: "v"(mine), "v"(mine + 2048), "v"(mine + 4096), "v"(mine + 6144), "v"(v) : "memory");
  } else {
    *reinterpret_cast<f4*>(mine) = v; *reinterpret_cast<f4*>(mine + 2048) = v;
    *reinterpret_cast<f4*>(mine + 4096) = v; *reinterpret_cast<f4*>(mine + 6144) = v;
  }
  if (MODE >= 1) {   // ---- barrier among the pair's 26 workgroups
    if (MODE == 1) __threadfence();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
      const long long t0 = wall_clock64(); (void)t0;
      if (MODE == 1) {
        __hip_atomic_fetch_add(&cnt[pair * 32], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(&cnt[pair * 32], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)PER * (it + 1)) __builtin_amdgcn_s_sleep(1);
      } else if (MODE == 5) {   // arrivals in the XCD's own L2 (no scope bits), polled with sc1 loads
        asm volatile("global_atomic_add %0, %1, off" ::"v"(&cnt[pair * 32]), "v"(1u) : "memory");
        unsigned v;
        do {
          asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(&cnt[pair * 32]) : "memory");
          if (v < (unsigned)PER * (it + 1)) __builtin_amdgcn_s_sleep(1);
        } while (v < (unsigned)PER * (it + 1) && wall_clock64() - t0 < 300000000);
      } else {
        __hip_atomic_fetch_add(&cnt[pair * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(&cnt[pair * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)PER * (it + 1)) __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (MODE == 1) __threadfence();
  }
  lds[t] = (float)wrong;
}

template <int MODE>
__global__ __launch_bounds__(THREADS) void k_layers(float* state, unsigned* cnt, unsigned* bad, unsigned* xcc_mix, int first, int layers, int work) {
  if (threadIdx.x == 0 && first == 0) atomicOr(&xcc_mix[blockIdx.x % PAIRS], 1u << (__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u));   // HW_REG_XCC_ID
  for (int it = first; it < first + layers; ++it) layer<MODE>(state, cnt, bad, it, work);
}

int main(int argc, char** argv) {
  float* state; unsigned *cnt, *bad, *mix;
  CK(hipMalloc(&state, sizeof(float) * STATE_F * WGS * 2));
  CK(hipMalloc(&cnt, 4 * 32 * PAIRS)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&mix, 4 * PAIRS));
  const size_t LDS = 80 * 1024;
  CK(hipFuncSetAttribute((const void*)k_layers<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CK(hipFuncSetAttribute((const void*)k_layers<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CK(hipFuncSetAttribute((const void*)k_layers<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CK(hipFuncSetAttribute((const void*)k_layers<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CK(hipFuncSetAttribute((const void*)k_layers<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CK(hipFuncSetAttribute((const void*)k_layers<5>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int LAYERS = 64;
  for (int work : {0, 2500})          // ticks of "compute" per layer (100-MHz wall clock: 25 us)
    for (int rep = 0; rep < 2; ++rep)
      for (int mode = 0; mode < 6; ++mode) {
        CK(hipMemset(cnt, 0, 4 * 32 * PAIRS)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(mix, 0, 4 * PAIRS));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        if (mode == 0) for (int it = 0; it < LAYERS; ++it) k_layers<0><<<WGS, THREADS, LDS>>>(state, cnt, bad, mix, it, 1, work);
        else if (mode == 1) k_layers<1><<<WGS, THREADS, LDS>>>(state, cnt, bad, mix, 0, LAYERS, work);
        else if (mode == 2) k_layers<2><<<WGS, THREADS, LDS>>>(state, cnt, bad, mix, 0, LAYERS, work);
        else if (mode == 3) k_layers<3><<<WGS, THREADS, LDS>>>(state, cnt, bad, mix, 0, LAYERS, work);
        else if (mode == 4) k_layers<4><<<WGS, THREADS, LDS>>>(state, cnt, bad, mix, 0, LAYERS, work);
        else k_layers<5><<<WGS, THREADS, LDS>>>(state, cnt, bad, mix, 0, LAYERS, work);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned hbad, hmix[PAIRS]; CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hmix, mix, 4 * PAIRS, hipMemcpyDeviceToHost));
        printf("work %5d mode %d: %6.2f us per layer, %u wrong values, XCC masks per pair:", work, mode, ms * 1e3 / LAYERS, hbad);
        for (int p = 0; p < PAIRS; ++p) printf(" %02x", hmix[p]);
        printf("\n");
      }
  return 0;
}
