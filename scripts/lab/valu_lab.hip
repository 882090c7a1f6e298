// Kernel lab (round 5): issue rates that decide the lane mapping of the molecule-per-workgroup message kernels.
//   - v_fma_f32 vs v_pk_fma_f32 per wave64 instruction (is the packed form worth keeping two channels per lane?)
//   - the same FMA with EXEC = one half-wave (does a half-masked VALU instruction cost half?)
//   - v_cndmask with an SGPR source, v_readlane, v_permlane32_swap
//   - ds_read_b128 / ds_read_b64 / ds_read_b32 wave-instruction rates, contiguous and half-wave-broadcast addresses
//   - FMA stream next to a ds_read_b128 stream (do LDS reads and VALU overlap inside one wave / across waves?)
// One workgroup per CU, W waves per SIMD; s_memtime around the loop of every wave, maximum over waves.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/lab/valu_lab.hip -o scripts/lab/_bin/valu_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// single (never paired) LDS reads: a volatile access through an LDS-address-space pointer stays one ds_read_b64 / ds_read_b32
typedef __attribute__((address_space(3))) const volatile f2* lds_f2p;
typedef __attribute__((address_space(3))) const volatile float* lds_f1p;
__device__ __forceinline__ f2 lds_ld2(const float* p) { return *(lds_f2p)(p); }
__device__ __forceinline__ float lds_ld1(const float* p) { return *(lds_f1p)(p); }

// MODE: 0 v_fma_f32 x16 independent, 1 v_pk_fma_f32 x16, 2 v_fma_f32 with exec = low half, 3 v_fma with SGPR multiplier,
// 4 v_cndmask(sgpr) + v_fma, 5 v_readlane x16, 6 v_permlane32_swap x16, 7 fma with exec = low 16 lanes
template <int MODE>
__global__ __launch_bounds__(1024) void k_valu(float* out, long long* cyc, int iters, float sm) {
  float a[16];
  f2 p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 1e-3f + i; p[i] = f2{a[i], a[i] + 1.f}; }
  const float m = 1.0001f + sm;
  const f2 pm = f2{m, m};
  int sel = (threadIdx.x & 32) ? -1 : 0;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(sm));
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pm));
    } else if (MODE == 2 || MODE == 7) {
      unsigned long long saved;
      if (MODE == 2) asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, 0xffffffff" : "=s"(saved));
      else asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, 0xffff" : "=s"(saved));
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(sm));
      asm volatile("s_mov_b64 exec, %0" :: "s"(saved));
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(m), "v"(sm));
    } else if (MODE == 4) {
      float r[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        asm volatile("v_cmp_ne_u32 vcc, 0, %1\n v_cndmask_b32 %0, %2, %3, vcc" : "=v"(r[i]) : "v"(sel), "s"(m), "v"(sm) : "vcc");
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(r[i]), "v"(sm));
      }
    } else if (MODE == 5) {
      int s[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(s[i]) : "v"(a[i]), "i"(i));
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "s"(s[i]));
    } else if (MODE == 6) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) asm volatile("s_nop 1\n v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[i + 1]));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// LDS modes: 0 ds_read_b128 contiguous (lane*16), 1 ds_read_b128 two half-waves at different rows (each half contiguous 512 B),
// 2 ds_read_b128 half-wave broadcast (all lanes of a half read the same 16 B), 3 ds_read_b64 contiguous, 4 ds_read_b64 halves at different rows,
// 5 ds_read_b32 contiguous, 6 ds_read_b128 (mode 1) + 2 FMAs per read in the same wave, 7 ds_read_b64 (mode 4) + 3 FMA per read,
// 8 ds_write_b64 contiguous, 9 ds_write_b128 contiguous
template <int MODE>
__global__ __launch_bounds__(1024) void k_ldsr(float* out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = 1e-3f * (i % 97);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l32 = lane & 31;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const int r0 = ((it * 7 + wave * 3) & 63), r1 = ((it * 13 + wave * 5 + 17) & 63);
    const int row = half ? r1 : r0;
    if (MODE == 0) {
#pragma unroll
      for (int t = 0; t < 16; ++t) { const f4 v = *reinterpret_cast<const f4*>(lds + (r0 + t) * 256 + lane * 4); acc[t & 7] += v.x + v.w; }
    } else if (MODE == 1 || MODE == 6) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const f4 v = *reinterpret_cast<const f4*>(lds + (row + t) * 128 + l32 * 4);
        if (MODE == 1) acc[t & 7] += v.x + v.w;
        else { acc[t & 7] = fmaf(v.x, 1.0001f, acc[t & 7]); acc[(t + 4) & 7] = fmaf(v.y, v.z, acc[(t + 4) & 7]); }
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int t = 0; t < 16; ++t) { const f4 v = *reinterpret_cast<const f4*>(lds + (row + t) * 128); acc[t & 7] += v.x + v.w; }
    } else if (MODE == 3) {
#pragma unroll
      for (int t = 0; t < 16; ++t) { const f2 v = *reinterpret_cast<const f2*>(lds + (r0 + t) * 128 + lane * 2); acc[t & 7] += v.x + v.y; }
    } else if (MODE == 4 || MODE == 7) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const f2 v = *reinterpret_cast<const f2*>(lds + (row + t) * 64 + l32 * 2);
        if (MODE == 4) acc[t & 7] += v.x + v.y;
        else { acc[t & 7] = fmaf(v.x, 1.0001f, acc[t & 7]); acc[(t + 4) & 7] = fmaf(v.y, 1.0002f, acc[(t + 4) & 7]); acc[(t + 2) & 7] = fmaf(v.y, v.x, acc[(t + 2) & 7]); }
      }
    } else if (MODE == 5) {
#pragma unroll
      for (int t = 0; t < 16; ++t) { acc[t & 7] += lds[(r0 + t) * 64 + lane]; }
    } else if (MODE == 10) {   // single ds_read_b64 (volatile: the compiler may not pair them into ds_read2_b64)
#pragma unroll
      for (int t = 0; t < 16; ++t) { const f2 v = lds_ld2(lds + (r0 + t) * 128 + lane * 2); acc[t & 7] += v.x + v.y; }
    } else if (MODE == 11) {
#pragma unroll
      for (int t = 0; t < 16; ++t) { const f2 v = lds_ld2(lds + (row + t) * 64 + l32 * 2); acc[t & 7] += v.x + v.y; }
    } else if (MODE == 12) {
#pragma unroll
      for (int t = 0; t < 16; ++t) { acc[t & 7] += lds_ld1(lds + (r0 + t) * 64 + lane); }
    } else if (MODE == 13 || MODE == 14) {   // the filter of the message kernels: 13 taps x 3 parts, two channels per lane, packed FMAs; 13 = paired reads, 14 = single reads
      const float* wk = lds + r0 * 384 + lane * 2;
      f2 va = f2{acc[0], acc[1]}, vb = f2{acc[2], acc[3]}, vc = f2{acc[4], acc[5]};
#pragma unroll
      for (int t = 0; t < 13; ++t) {
        f2 wa, wb, wc;
        if (MODE == 13) { wa = *reinterpret_cast<const f2*>(wk + t * 384); wb = *reinterpret_cast<const f2*>(wk + t * 384 + 128); wc = *reinterpret_cast<const f2*>(wk + t * 384 + 256); }
        else { wa = lds_ld2(wk + t * 384); wb = lds_ld2(wk + t * 384 + 128); wc = lds_ld2(wk + t * 384 + 256); }
        const float r = 1.0f + 1e-3f * t;
        va = __builtin_elementwise_fma(wa, f2{r, r}, va); vb = __builtin_elementwise_fma(wb, f2{r, r}, vb); vc = __builtin_elementwise_fma(wc, f2{r, r}, vc);
      }
      acc[0] = va.x; acc[1] = va.y; acc[2] = vb.x; acc[3] = vb.y; acc[4] = vc.x; acc[5] = vc.y;
    } else if (MODE == 15) {   // b128 filter: [tap][32 ch][a, b, c, pad], half-wave per edge, 13 taps
      const float* wk = lds + row * 128 + l32 * 4;
      float a = acc[0], b = acc[1], c = acc[2];
#pragma unroll
      for (int t = 0; t < 13; ++t) {
        const f4 w = *reinterpret_cast<const f4*>(wk + t * 128);
        const float r = 1.0f + 1e-3f * t;
        a = fmaf(w.x, r, a); b = fmaf(w.y, r, b); c = fmaf(w.z, r, c);
      }
      acc[0] = a; acc[1] = b; acc[2] = c;
    } else if (MODE == 8) {
#pragma unroll
      for (int t = 0; t < 16; ++t) *reinterpret_cast<f2*>(lds + (r0 + t) * 128 + lane * 2) = f2{acc[0] + t, acc[1]};
    } else if (MODE == 9) {
#pragma unroll
      for (int t = 0; t < 16; ++t) *reinterpret_cast<f4*>(lds + (r0 + t) * 256 + lane * 4) = f4{acc[0] + t, acc[1], acc[2], acc[3]};
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i];
  if (MODE >= 8) { __syncthreads(); s += lds[threadIdx.x]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

static float* d_out; static long long* d_cyc;

template <typename F>
static void report(const char* name, int waves_per_simd, int iters, int instr_per_iter, F launch) {
  const int threads = waves_per_simd * 256, nw = 256 * waves_per_simd * 4;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(threads, 10);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  launch(threads, iters);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> c(nw);
  CK(hipMemcpy(c.data(), d_cyc, nw * sizeof(long long), hipMemcpyDeviceToHost));
  const long long mx = *std::max_element(c.begin(), c.end());
  double avg = 0; for (auto v : c) avg += v; avg /= nw;
  const double n = (double)iters * instr_per_iter;
  // s_memtime ticks at a fixed 100 MHz on some parts; report both tick- and wall-derived figures (wall at the measured kernel time, per SIMD)
  printf("%-58s w/SIMD %d: ticks/instr/wave %.3f (max %.3f)  wall ns per wave-instr per SIMD %.3f  (%.3f ms)\n", name, waves_per_simd, avg / n, mx / n,
         ms * 1e6 / (n * waves_per_simd), ms);
}

int main() {
  CK(hipMalloc(&d_out, 256 * 1024 * sizeof(float)));
  CK(hipMalloc(&d_cyc, 256 * 16 * sizeof(long long)));
  const int it = 4000;
#define RUNV(M, NAME, IPI) for (int w : {1, 2, 4}) report(NAME, w, it, IPI, [&](int th, int n) { hipLaunchKernelGGL((k_valu<M>), dim3(256), dim3(th), 0, 0, d_out, d_cyc, n, 0.f); });
  RUNV(0, "v_fma_f32 x16 (vgpr operands)", 16)
  RUNV(1, "v_pk_fma_f32 x16", 16)
  RUNV(2, "v_fma_f32 x16, exec = lanes 0-31", 16)
  RUNV(7, "v_fma_f32 x16, exec = lanes 0-15", 16)
  RUNV(3, "v_fma_f32 x16, sgpr multiplier", 16)
  RUNV(4, "(v_cmp + v_cndmask(sgpr) + v_fma) x16  [48 instr]", 48)
  RUNV(5, "(v_readlane + v_add sgpr) x16 [32 instr]", 32)
  RUNV(6, "v_permlane32_swap x8", 8)
#define RUNL(M, NAME, IPI) for (int w : {1, 2, 4}) report(NAME, w, it / 2, IPI, [&](int th, int n) { hipLaunchKernelGGL((k_ldsr<M>), dim3(256), dim3(th), 131072, 0, d_out, d_cyc, n); });
  CK(hipFuncSetAttribute((const void*)k_ldsr<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CK(hipFuncSetAttribute((const void*)k_ldsr<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CK(hipFuncSetAttribute((const void*)k_ldsr<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CK(hipFuncSetAttribute((const void*)k_ldsr<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CK(hipFuncSetAttribute((const void*)k_ldsr<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CK(hipFuncSetAttribute((const void*)k_ldsr<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CK(hipFuncSetAttribute((const void*)k_ldsr<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CK(hipFuncSetAttribute((const void*)k_ldsr<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CK(hipFuncSetAttribute((const void*)k_ldsr<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CK(hipFuncSetAttribute((const void*)k_ldsr<9>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  RUNL(0, "ds_read_b128 contiguous x16", 16)
  RUNL(1, "ds_read_b128 halves at different rows x16", 16)
  RUNL(2, "ds_read_b128 half-wave broadcast x16", 16)
  RUNL(3, "ds_read_b64 contiguous x16", 16)
  RUNL(4, "ds_read_b64 halves at different rows x16", 16)
  RUNL(5, "ds_read_b32 contiguous x16", 16)
  RUNL(6, "ds_read_b128 halves + 2 fma each x16 (per read)", 16)
  RUNL(7, "ds_read_b64 halves + 3 fma each x16 (per read)", 16)
  CK(hipFuncSetAttribute((const void*)k_ldsr<10>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CK(hipFuncSetAttribute((const void*)k_ldsr<11>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CK(hipFuncSetAttribute((const void*)k_ldsr<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CK(hipFuncSetAttribute((const void*)k_ldsr<13>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CK(hipFuncSetAttribute((const void*)k_ldsr<14>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CK(hipFuncSetAttribute((const void*)k_ldsr<15>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  RUNL(10, "single ds_read_b64 (volatile) contiguous x16", 16)
  RUNL(11, "single ds_read_b64 (volatile) halves at different rows x16", 16)
  RUNL(12, "single ds_read_b32 (volatile) contiguous x16", 16)
  RUNL(13, "filter 13 taps x 3 parts b64 PAIRED reads + 39 pk_fma [per edge, 128 ch]", 1)
  RUNL(14, "filter 13 taps x 3 parts b64 SINGLE reads + 39 pk_fma [per edge, 128 ch]", 1)
  RUNL(15, "filter 13 taps b128 [a,b,c,pad] + 39 fma, half-wave per edge [per 2 edges x 32 ch]", 1)
  RUNL(8, "ds_write_b64 contiguous x16", 16)
  RUNL(9, "ds_write_b128 contiguous x16", 16)
  printf("note: per-CU wave-instruction time = (wall ns per wave-instr per SIMD) / 4 for VALU (4 SIMDs issue in parallel); LDS is one pipe per CU: per-CU ns per LDS wave-instr = value / 4 as well\n");
  return 0;
}
