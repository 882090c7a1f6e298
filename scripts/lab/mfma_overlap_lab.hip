// Kernel lab (round 5): does v_mfma_f32_32x32x2_f32 (f32 inputs) run beside the VALU of the other wavefronts of its SIMD, or instead of it?
// 16 wavefronts per CU (4 per SIMD).  ROLE bit 0: wavefronts 0-7 issue matrix-core chains; bit 1: wavefronts 8-15 issue v_fma_f32 chains.
// Compared per matrix flavour (f32 32x32x2, bf16 32x32x16): matrix only, VALU only, both.  If the two pipes overlap, time(both) ~ max, else ~ sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s8v __attribute__((ext_vector_type(8)));

template <int FLAVOUR>   // 0: f32 32x32x2, 1: bf16 32x32x16
__global__ __launch_bounds__(1024) void k_mix(float* out, int iters, int role, int split) {
  const int wave = threadIdx.x >> 6;
  // split = 0: wavefronts 0-7 matrix, 8-15 VALU -> every SIMD (wave % 4) hosts two of each kind; split = 1 (round 6): the kind follows the SIMD -- wavefronts with
  // (wave & 3) < 2 are matrix wavefronts, so SIMDs 0 / 1 issue only matrix instructions and SIMDs 2 / 3 only VALU (MI355X_MICROARCH.md: "a MFMA-only wave and a
  // VALU-only wave on the same CU run concurrently" is a statement per CU; the round-5 lab put both kinds on every SIMD)
  const bool matrix_wave = split ? ((wave & 3) < 2) : (wave < 8);
  f16v c0, c1;
  for (int i = 0; i < 16; ++i) { c0[i] = threadIdx.x * 1e-3f; c1[i] = 1.f; }
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;
  const float m = 1.0001f, z = 1e-9f;
  s8v ab; for (int i = 0; i < 8; ++i) ab[i] = (short)(0x3f80 + i);
  if (matrix_wave) {
    if (role & 1)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (FLAVOUR == 0) { c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(m, z, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(z, m, c1, 0, 0, 0); }
          else { c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, c1, 0, 0, 0); }
        }
      }
  } else {
    if (role & 2)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(z));
      }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int FLAVOUR>
static float run(float* out, int iters, int role, int split = 0) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_mix<FLAVOUR>), dim3(256), dim3(1024), 0, 0, out, 10, role, split);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_mix<FLAVOUR>), dim3(256), dim3(1024), 0, 0, out, iters, role, split);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}
int main() {
  float* out; CK(hipMalloc(&out, 256 * 1024 * sizeof(float)));
  const int it = 4000;
  // per iteration: 8 matrix instructions per matrix wavefront (2 per SIMD -> 16 per SIMD and iteration), 128 v_fma per VALU wavefront (2 per SIMD -> 256 per SIMD)
  const float f_m = run<0>(out, it, 1), f_v = run<0>(out, it, 2), f_b = run<0>(out, it, 3);
  printf("f32  32x32x2 : matrix only %.3f ms (%.1f cycles per instruction per SIMD at 2.4 GHz), VALU only %.3f ms, both %.3f ms  (sum %.3f, max %.3f)\n", f_m,
         f_m * 1e-3 * 2.4e9 / (it * 16.0), f_v, f_b, f_m + f_v, f_m > f_v ? f_m : f_v);
  const float b_m = run<1>(out, it, 1), b_v = run<1>(out, it, 2), b_b = run<1>(out, it, 3);
  printf("bf16 32x32x16: matrix only %.3f ms (%.1f cycles per instruction per SIMD at 2.4 GHz), VALU only %.3f ms, both %.3f ms  (sum %.3f, max %.3f)\n", b_m,
         b_m * 1e-3 * 2.4e9 / (it * 16.0), b_v, b_b, b_m + b_v, b_m > b_v ? b_m : b_v);
  // round 6: the two kinds on DIFFERENT SIMDs of the CU (same totals per CU; per SIMD the matrix SIMDs carry twice the matrix work, the VALU SIMDs twice the VALU work)
  for (int fl = 0; fl < 2; ++fl) {
    const float m_ = fl ? run<1>(out, it, 1, 1) : run<0>(out, it, 1, 1), v_ = fl ? run<1>(out, it, 2, 1) : run<0>(out, it, 2, 1), b_ = fl ? run<1>(out, it, 3, 1) : run<0>(out, it, 3, 1);
    printf("%s, kinds on different SIMDs: matrix only %.3f ms, VALU only %.3f ms, both %.3f ms  (sum %.3f, max %.3f)\n", fl ? "bf16 32x32x16" : "f32  32x32x2 ", m_, v_, b_, m_ + v_,
           m_ > v_ ? m_ : v_);
  }
  return 0;
}
