// Kernel lab (round 5, second question): the first lab (mfma_overlap_lab.hip) found time(matrix waves + VALU waves) = sum when the matrix wavefronts issue their
// instructions back to back.  Does the matrix core run in the background when a wavefront SPACES its matrix instructions with independent VALU work
// (MFMA, 16 v_fma, MFMA, 16 v_fma ...) -- the way hand-scheduled GEMM loops do -- or is the VALU port held for the whole matrix instruction?
//   A  matrix only: 2 MFMA per iteration (alternating accumulators)       B  VALU only: 32 v_fma per iteration (16 independent chains)
//   C  interleaved: MFMA, 16 v_fma, MFMA, 16 v_fma                         D  clumped: MFMA, MFMA, 32 v_fma
//   E  interleaved fine: MFMA, 8 v_fma, ... (4 groups; same totals as C with MFMA count doubled -> matrix-bound)
// Everything is inline assembly (the order in the binary is the order written).  WAVES wavefronts per SIMD run the same code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s8v __attribute__((ext_vector_type(8)));

#define MFMA_F32(c) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(m), "v"(z))
#define MFMA_BF(c) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %1, %0" : "+v"(c) : "v"(ab))
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(z))
#define FMA8(o) FMA(o); FMA(o + 1); FMA(o + 2); FMA(o + 3); FMA(o + 4); FMA(o + 5); FMA(o + 6); FMA(o + 7)
#define FMA16 FMA8(0); FMA8(8)

template <int MODE, int BF>
__global__ __launch_bounds__(1024) void k_lab(float* out, int iters) {
  f16v c0, c1;
  for (int i = 0; i < 16; ++i) { c0[i] = threadIdx.x * 1e-3f; c1[i] = 1.f; }
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;
  const float m = 1.0001f, z = 1e-9f;
  s8v ab; for (int i = 0; i < 8; ++i) ab[i] = (short)(0x3f80 + i);
#define MF(c) do { if (BF) MFMA_BF(c); else MFMA_F32(c); } while (0)
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) { MF(c0); MF(c1); }
    if (MODE == 1) { FMA16; FMA16; }
    if (MODE == 2) { MF(c0); FMA16; MF(c1); FMA16; }
    if (MODE == 3) { MF(c0); MF(c1); FMA16; FMA16; }
    if (MODE == 4) { MF(c0); FMA8(0); MF(c1); FMA8(8); MF(c0); FMA8(0); MF(c1); FMA8(8); }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int BF>
static float run(float* out, int iters, int threads) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_lab<MODE, BF>), dim3(256), dim3(threads), 0, 0, out, 10);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_lab<MODE, BF>), dim3(256), dim3(threads), 0, 0, out, iters);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}
template <int BF>
static void table(float* out, const char* name) {
  const int it = 4000;
  for (int threads : {256, 512, 1024}) {
    const int wps = threads / 256;
    const float A = run<0, BF>(out, it, threads), B = run<1, BF>(out, it, threads), C = run<2, BF>(out, it, threads), D = run<3, BF>(out, it, threads),
                E = run<4, BF>(out, it, threads);
    const double cyc = 1e-3 * 2.4e9 / (double)it / wps;   // cycles per iteration and wavefront of a SIMD
    printf("%s, %d wavefront(s) per SIMD: cycles per iteration per wavefront  A matrix only (2) %.1f | B VALU only (32) %.1f | C interleaved %.1f | D clumped %.1f | "
           "E 4 matrix + 32 VALU interleaved %.1f   (sum A+B %.1f, max %.1f)\n", name, wps, A * cyc, B * cyc, C * cyc, D * cyc, E * cyc, (A + B) * cyc,
           (A > B ? A : B) * cyc);
  }
}
int main() {
  float* out; CK(hipMalloc(&out, 256 * 1024 * sizeof(float)));
  table<0>(out, "f32 32x32x2  ");
  table<1>(out, "bf16 32x32x16");
  return 0;
}
