// Kernel lab: throughput of LDS float atomics (ds_add_f32, no return) next to the filter's ds_read_b64 stream.
// Question: can the dual-reverse message kernel add its rbf_proj-gradient contributions (13 taps x 3 parts x CH
// columns per pair and lane) into an LDS-resident accumulator instead of writing g_phi / g_psi pair rows to HBM?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/lab/lds_lab.hip -o scripts/lab/_bin/lds_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ROWS = 100, COLS = 192;   // one 64-channel slice: [100 taps][3 parts][64 channels]

// MODE 0: 39 ds_add_f32 per "pair" (13 taps x 3 parts, one channel per lane)
// MODE 1: 39 ds_write_b32 (store rate reference)
// MODE 2: 39 ds_read_b32 + 39 ds_add_f32 (filter reads of a CH = 1 slice next to the adds)
// MODE 3: reads only
template <int MODE>
__global__ __launch_bounds__(1024) void k_lds(float* out, int iters, const int* k0s) {
  extern __shared__ float lds[];
  float* wrt = lds;                       // [ROWS + 13][COLS]
  float* acc = lds + ROWS * COLS;  // [ROWS + 13][COLS]
  for (int i = threadIdx.x; i < 2 * ROWS * COLS; i += blockDim.x) lds[i] = (i < ROWS * COLS) ? 1e-3f * (i % 97) : 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float g0 = 1.0f + lane * 1e-3f, g1 = 0.5f, g2 = 0.25f, s = 0.f;
  for (int it = 0; it < iters; ++it) {
    const int k0 = __builtin_amdgcn_readfirstlane(k0s[(it * 16 + wave) & 4095]);
    float* a = acc + k0 * COLS + lane;
    const float* w = wrt + k0 * COLS + lane;
    if (MODE == 2 || MODE == 3) {
#pragma unroll
      for (int t = 0; t < 13; ++t) { s += w[t * COLS] * g0; s += w[t * COLS + 64] * g1; s += w[t * COLS + 128] * g2; }
    }
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int t = 0; t < 13; ++t) {
        const float r = 0.1f * t + 0.01f * it;
        __hip_atomic_fetch_add(a + t * COLS, g0 * r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(a + t * COLS + 64, g1 * r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(a + t * COLS + 128, g2 * r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    if (MODE == 1) {
#pragma unroll
      for (int t = 0; t < 13; ++t) {
        const float r = 0.1f * t + 0.01f * it;
        a[t * COLS] = g0 * r; a[t * COLS + 64] = g1 * r; a[t * COLS + 128] = g2 * r;
      }
    }
    g0 += 1e-6f;
  }
  __syncthreads();
  float tot = s;
  for (int i = threadIdx.x; i < ROWS * COLS; i += blockDim.x) tot += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = tot;
}

template <int MODE>
void run(const char* name, int waves, int iters, float* out, const int* k0s) {
  const size_t lds = 2 * ROWS * COLS * sizeof(float);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k_lds<MODE><<<256, waves * 64, lds>>>(out, iters / 4, k0s);
  CK(hipEventRecord(e0));
  k_lds<MODE><<<256, waves * 64, lds>>>(out, iters, k0s);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double instr = (double)waves * iters * 39.0;   // wave-instructions of the measured kind per CU
  printf("%-34s waves/CU %2d  %8.3f ms  %6.2f ns per (wave-pair of 39)  ~%5.2f cycles per wave-instruction per CU at 2.4 GHz\n", name, waves, ms,
         1e6 * ms / ((double)waves * iters), ms * 1e-3 * 2.4e9 / instr);
}

int main() {
  float* out; int* k0s;
  CK(hipMalloc(&out, 256 * 1024 * sizeof(float)));
  std::vector<int> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = (int)(((unsigned)i * 2654435761u >> 8) % 88);
  CK(hipMalloc(&k0s, 4096 * sizeof(int)));
  CK(hipMemcpy(k0s, h.data(), 4096 * sizeof(int), hipMemcpyHostToDevice));
  for (int waves : {4, 8, 16}) {
    run<0>("ds_add_f32 x39", waves, 20000, out, k0s);
    run<1>("ds_write_b32 x39", waves, 20000, out, k0s);
    run<3>("ds_read_b32 x39", waves, 20000, out, k0s);
    run<2>("ds_read_b32 x39 + ds_add_f32 x39", waves, 20000, out, k0s);
  }
  return 0;
}
