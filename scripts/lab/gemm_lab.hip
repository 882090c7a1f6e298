// Kernel lab for the fp32 MFMA GEMM (no torch: starts in milliseconds on the GPU box).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I nabladft_amd/csrc scripts/lab/gemm_lab.hip -o scripts/lab/_bin/gemm_lab -L nabladft_amd -lnablaq -Wl,-rpath,'$ORIGIN/../../../nabladft_amd'
// Times every tile variant of gemm_tile.h (and the shipped library entry points) on the shapes of the five models, interleaved rounds, HIP
// events on the launch stream, uniform random operands; checks a row sample of every result against an fmaf-chain reference kernel.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "gemm_tile.h"
#include "gemm_split.h"

extern "C" {
int nq_linear_forward(const float* A, const float* W, const float* bias, float* C, float* C_silu, int32_t M, int32_t N, int32_t K, void* stream);
int nq_linear_input_grad(const float* G, const float* W, float* C, int32_t M, int32_t N, int32_t K, int32_t accumulate, void* stream);
size_t nq_weight_grad_scratch_floats(int64_t rows, int32_t N, int32_t K);
int nq_linear_weight_grad(const float* G, const float* X, float* gW, int64_t rows, int32_t N, int32_t K, float* scratch, void* stream);
const char* nq_last_error();
void nq_set_gemm_variant(int32_t v);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_fill(float* p, long n, uint32_t seed, float scale, int positive = 0) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t x = (uint32_t)i * 2654435761u + seed;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  const float u = (float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;
  p[i] = scale * (positive ? fabsf(u) : u);
}
// reference on a row sample: C[m][n] = sum_k a(m,k) b(k,n) in double
__global__ void k_ref(const float* A, const float* B, double* C, int M, int N, int K, long lda, long ldb, int a_kc, int b_kc, const int* rows, int nrows) {
  int n = blockIdx.x * blockDim.x + threadIdx.x, ri = blockIdx.y;
  if (n >= N || ri >= nrows) return;
  int m = rows[ri];
  double s = 0;
  for (int k = 0; k < K; ++k) {
    float a = a_kc ? A[(long)m * lda + k] : A[(long)k * lda + m];
    float b = b_kc ? B[(long)n * ldb + k] : B[(long)k * ldb + n];
    s += (double)a * b;
  }
  C[(long)ri * N + n] = s;
}
__global__ void k_reduce(const float* part, int nsplit, long stride, long count, float* out) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = 0;
  for (; k + 3 < nsplit; k += 4) {
    s0 += part[(long)k * stride + i]; s1 += part[(long)(k + 1) * stride + i]; s2 += part[(long)(k + 2) * stride + i]; s3 += part[(long)(k + 3) * stride + i];
  }
  for (; k < nsplit; ++k) s0 += part[(long)k * stride + i];
  out[i] = (s0 + s1) + (s2 + s3);
}

// calibration: back-to-back v_mfma_f32_32x32x2_f32 on registers only (no LDS, no memory): the matrix-pipe ceiling of THIS box under sustained load (DVFS included)
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma_peak(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[(long)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the three-way split on the device: h + m + l must reproduce x to 2^-25 |x| (every exponent where the third piece is still a normal number, both signs)
__global__ void k_split_check(unsigned long long* bad, unsigned long long* inexact, int rounds) {
  uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  for (int r = 0; r < rounds; ++r) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; x += 0x9e3779b9u;
    uint32_t y = x * 747796405u + 2891336453u;
    uint32_t e0 = (x >> 23) & 0xff, e1 = (y >> 23) & 0xff;
    if (e0 < 40 || e0 > 215 || e1 < 40 || e1 > 215) continue;   // normal numbers whose third piece is still normal, no overflow
    const float a = __uint_as_float(x), b = __uint_as_float(y);
    unsigned h1, m1, l1;
    sp_split2(a, b, h1, m1, l1);
    atomicAdd(bad, 1ull);   // pairs checked
    const double ra = (double)__uint_as_float(h1 << 16) + (double)__uint_as_float(m1 << 16) + (double)__uint_as_float(l1 << 16);
    const double rb = (double)__uint_as_float(h1 & 0xffff0000u) + (double)__uint_as_float(m1 & 0xffff0000u) + (double)__uint_as_float(l1 & 0xffff0000u);
    if (fabs(ra - (double)a) > ldexp(fabs((double)a), -25) || fabs(rb - (double)b) > ldexp(fabs((double)b), -25)) atomicAdd(inexact, 1ull);
  }
}

struct Shape { const char* kind; const char* name; int M, N, K; };   // nt/nn: C[M,N], contraction K.  tn: out[M=Mo, N=No], K = rows

template <bool A_KC, bool B_KC, int EPI, int BM, int BN, int BK, int NWM, int NWN, int WPE, int ABL = 0>
static void launch2(hipStream_t st, GemmArgs p, int splits, int wg_per_cu) {   // wg_per_cu 0: one workgroup per tile
  int ntiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * splits;
  dim3 grid(wg_per_cu ? std::min(ntiles, wg_per_cu * 256) : ntiles, 1, 1);
  hipLaunchKernelGGL((k_gemm2<A_KC, B_KC, EPI, BM, BN, BK, NWM, NWN, WPE, ABL>), grid, dim3(NWM * NWN * 64), 0, st, p);
}

template <bool A_KC, bool B_KC, int EPI, int WPE, int TERMS, int ABL = 0, bool KTAIL = true>
static void launch3(hipStream_t st, GemmArgs p, int splits, int wg_per_cu) {
  int ntiles = ((p.M + 127) / 128) * ((p.N + 127) / 128) * splits;
  dim3 grid(std::min(ntiles, wg_per_cu * 256), 1, 1);
  hipLaunchKernelGGL((k_gemm3<A_KC, B_KC, EPI, WPE, KTAIL, false, TERMS, ABL>), grid, dim3(256), 0, st, p);
}

struct Variant { std::string name; std::function<void()> run; bool check; };

int main(int argc, char** argv) {
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  const bool pos = argc > 1 && !strcmp(argv[1], "pos");   // one-signed operands: every product positive (shows a biased accumulation)
  const bool abl = argc > 1 && !strcmp(argv[1], "abl");
  const bool split_only = abl || pos || (argc > 1 && !strcmp(argv[1], "split"));
  hipStream_t st; CK(hipStreamCreate(&st));
  {   // matrix-pipe ceiling: 256 CUs x (1, 2, 4) workgroups of 4 waves, 4 independent accumulators, ~0.2 s of sustained MFMA issue in total
    float* sink; CK(hipMalloc(&sink, 1024 * 256 * 4 * sizeof(float)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wg = 1; wg <= 4; wg *= 2) {
      const int iters = 4000;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k_mfma_peak<4>, dim3(256 * wg), dim3(256), 0, st, sink, iters, 0.5f, 0.25f);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double fl = 2.0 * 32 * 32 * 2 * 4.0 * iters * 4 * 256 * wg;
        if (rep == 2) printf("# pure MFMA 32x32x2 f32, %d workgroup(s) of 4 waves per CU, %.2f ms: %.1f TFLOP/s  (157.3 = 2.4 GHz x 64 flop/clk/SIMD x 1024 SIMDs)\n", wg, ms, fl / ms * 1e-9);
      }
    }
    CK(hipFree(sink));
  }
  {
    unsigned long long* cnt; CK(hipMalloc(&cnt, 16)); CK(hipMemsetAsync(cnt, 0, 16, st));
    hipLaunchKernelGGL(k_split_check, dim3(1024), dim3(256), 0, st, cnt, cnt + 1, 256);
    unsigned long long hc[2]; CK(hipMemcpyAsync(hc, cnt, 16, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
    printf("# three-way bf16 split on the device: %llu random value pairs over 176 binades, |h + m + l - x| > 2^-25 |x| in %llu of them\n", hc[0], hc[1]);
    CK(hipFree(cnt));
  }
  std::vector<Shape> shapes = {
      // PaiNN step at B = 2048 (N = 85576 atoms)
      {"nt", "painn W1", 85576, 128, 128}, {"nt", "painn W2", 85576, 384, 128}, {"nt", "painn U", 256728, 256, 128}, {"nt", "painn V1", 85576, 128, 256},
      {"nn", "painn U", 256728, 256, 128}, {"nn", "painn U2x", 513456, 256, 128}, {"nn", "painn W2", 171152, 128, 384}, {"nn", "painn V1", 171152, 256, 128},
      {"tn", "painn W2", 384, 128, 171152}, {"tn", "painn U", 256, 128, 513456}, {"tn", "painn V1", 128, 256, 171152}, {"tn", "painn W1", 128, 128, 171152},
      // GemNet-OC / eSCN / EquiformerV2 at B = 16
      {"nt", "gemnet 512", 20480, 512, 512}, {"nt", "gemnet 256", 20480, 256, 256}, {"nt", "gemnet atom", 700, 256, 256}, {"nt", "escn so2", 28000, 1536, 1536},
      {"nt", "eq 896", 20000, 896, 896}, {"nn", "gemnet 512", 20480, 512, 512}, {"nn", "escn so2", 28000, 1536, 1536}, {"tn", "gemnet 512", 512, 512, 20480},
      {"tn", "escn so2", 1536, 1536, 28000},
      // QHNet weight generators
      {"nt", "qh pair", 27552, 8320, 128}, {"nt", "qh edge", 27000, 5376, 32}, {"nt", "4096^3", 4096, 4096, 4096},
  };
  if (quick) shapes.resize(2);
  if (pos) shapes = {{"nt", "eq 896", 20000, 896, 896}, {"nn", "painn U", 256728, 256, 128}, {"tn", "painn U", 256, 128, 513456}, {"nt", "4096^3", 4096, 4096, 4096}};
  if (abl) shapes = {{"nt", "painn U", 256728, 256, 128}, {"nt", "eq 896", 20000, 896, 896}, {"nt", "4096^3", 4096, 4096, 4096}};
  for (const Shape& s : shapes) {
    const bool nt = !strcmp(s.kind, "nt"), nn = !strcmp(s.kind, "nn"), tn = !strcmp(s.kind, "tn");
    const int M = s.M, N = s.N, K = s.K;
    // operand storage
    const bool a_kc = !tn, b_kc = nt;
    const long a_elems = (long)M * K, b_elems = (long)N * K, c_elems = (long)M * N;
    const long lda = a_kc ? K : M, ldb = b_kc ? K : N;
    float *A, *B, *C, *bias, *part = nullptr, *scr = nullptr;
    CK(hipMalloc(&A, a_elems * 4)); CK(hipMalloc(&B, b_elems * 4)); CK(hipMalloc(&C, c_elems * 4)); CK(hipMalloc(&bias, N * 4));
    hipLaunchKernelGGL(k_fill, dim3((a_elems + 255) / 256), dim3(256), 0, st, A, a_elems, 1u, 1.0f, (int)pos);
    hipLaunchKernelGGL(k_fill, dim3((b_elems + 255) / 256), dim3(256), 0, st, B, b_elems, 2u, tn ? 1.0f : 0.1f, (int)pos);
    hipLaunchKernelGGL(k_fill, dim3((N + 255) / 256), dim3(256), 0, st, bias, (long)N, 3u, 1.0f);
    // split count for tn, as the library chooses it: ~768 workgroups, >= 128 rows per split
    int splits = 1, kper = K;
    if (tn) {
      const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
      long sp = (768 + tiles - 1) / tiles;
      sp = std::min<long>(sp, (K + 127) / 128); sp = std::max<long>(1, std::min<long>(sp, 512));
      splits = (int)sp; kper = (int)((K + splits - 1) / splits); kper = (kper + 31) / 32 * 32;
      CK(hipMalloc(&part, (long)splits * c_elems * 4));
      CK(hipMalloc(&scr, nq_weight_grad_scratch_floats(K, M, N) * 4));
    }
    // reference rows
    std::vector<int> rows;
    for (int i = 0; i < 64 && i < M; ++i) rows.push_back((int)(((long)i * 7919) % M));
    for (int i = std::max(0, M - 40); i < M; ++i) rows.push_back(i);
    int* drows; double* dref;
    CK(hipMalloc(&drows, rows.size() * 4)); CK(hipMalloc(&dref, rows.size() * (long)N * 8));
    CK(hipMemcpyAsync(drows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_ref, dim3((N + 127) / 128, rows.size()), dim3(128), 0, st, A, B, dref, M, N, K, lda, ldb, (int)a_kc, (int)b_kc, drows, (int)rows.size());
    std::vector<double> ref(rows.size() * (long)N);
    CK(hipMemcpyAsync(ref.data(), dref, ref.size() * 8, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    double refmax = 0; for (double v : ref) refmax = std::max(refmax, std::fabs(v));

    GemmArgs p{};
    p.A = A; p.B = B; p.C = tn ? part : C; p.bias = nullptr; p.M = M; p.N = N; p.K = K; p.lda = (int)lda; p.ldb = (int)ldb; p.ldc = N;
    p.k_per_split = kper; p.part_stride = c_elems;
    auto reduce = [&]() { if (tn) hipLaunchKernelGGL(k_reduce, dim3((c_elems + 63) / 64), dim3(64), 0, st, part, splits, c_elems, c_elems, C); };

    std::vector<Variant> vs;
    if (nt) vs.push_back({"lib exact-f32 engine", [&]() { nq_set_gemm_variant(1 | 32); nq_linear_forward(A, B, nullptr, C, nullptr, M, N, K, st); nq_set_gemm_variant(1); }, true});
    if (nn) vs.push_back({"lib exact-f32 engine", [&]() { nq_set_gemm_variant(1 | 32); nq_linear_input_grad(A, B, C, M, K, N, 0, st); nq_set_gemm_variant(1); }, true});
    if (tn) vs.push_back({"lib exact-f32 engine", [&]() { nq_set_gemm_variant(1 | 32); nq_linear_weight_grad(A, B, C, K, M, N, scr, st); nq_set_gemm_variant(1); }, true});
    if (nt) vs.push_back({"lib", [&]() { nq_linear_forward(A, B, nullptr, C, nullptr, M, N, K, st); }, true});
    if (nn) vs.push_back({"lib", [&]() { nq_linear_input_grad(A, B, C, M, K, N, 0, st); }, true});   // (G[M,Nout=K], W[Nout=K, Kin=N]) -> C[M, Kin=N]
    if (tn) vs.push_back({"lib", [&]() { nq_linear_weight_grad(A, B, C, K, M, N, scr, st); }, true});
#define V(name, AKC, BKC, EPI, BM, BN, BK, WM, WN, ABL, chk, PC) vs.push_back({name, [&]() { launch2<AKC, BKC, EPI, BM, BN, BK, WM, WN, ((PC) ? (PC) : 2) * WM * WN / 4, ABL>(st, p, splits, PC); reduce(); }, chk})
#define S(name, AKC, BKC, EPI, WPE, TERMS, PC) vs.push_back({name, [&]() { launch3<AKC, BKC, EPI, WPE, TERMS>(st, p, splits, PC); reduce(); }, true})
    if (nt) {
      S("split6 p3", true, true, EPI_STORE, 3, 6, 3);
      if (K % 16 == 0) vs.push_back({"split6 p3 no k-tail code", [&]() { launch3<true, true, EPI_STORE, 3, 6, 0, false>(st, p, splits, 3); }, true});
      S("split6 p2", true, true, EPI_STORE, 2, 6, 2);
      S("split3 p3", true, true, EPI_STORE, 3, 3, 3);
#define SA_(name, ABL) vs.push_back({name, [&]() { launch3<true, true, EPI_STORE, 3, 6, ABL>(st, p, splits, 3); }, false})
      if (abl) { SA_("  abl1 no split arith", 1); SA_("  abl2 no loads", 2); SA_("  abl3 no split, no loads", 3); SA_("  abl4 no ds_read", 4); SA_("  abl7 mfma+ds_write+barrier only", 7); }
    }
    if (nn) {
      S("split6 p3", true, false, EPI_STORE, 3, 6, 3);
      if (K % 16 == 0) vs.push_back({"split6 p3 no k-tail code", [&]() { launch3<true, false, EPI_STORE, 3, 6, 0, false>(st, p, splits, 3); }, true});
      if (K % 16 == 0) vs.push_back({"split6 p2 no k-tail code", [&]() { launch3<true, false, EPI_STORE, 2, 6, 0, false>(st, p, splits, 2); }, true});
      S("split6 p2", true, false, EPI_STORE, 2, 6, 2);
    }
    if (tn) {
      S("split6 p3", false, false, EPI_PARTIAL, 3, 6, 3);
      S("split6 p2", false, false, EPI_PARTIAL, 2, 6, 2);
    }
    if (nt && !split_only) {
      V("128x128x32 w4x2 p2", true, true, EPI_STORE, 128, 128, 32, 4, 2, 0, true, 2);
      V("128x128x32 w2x2 p2", true, true, EPI_STORE, 128, 128, 32, 2, 2, 0, true, 2);
      V("128x64x32 w2x2 p4", true, true, EPI_STORE, 128, 64, 32, 2, 2, 0, true, 4);
      V("128x64x32 w2x2 p2", true, true, EPI_STORE, 128, 64, 32, 2, 2, 0, true, 2);
      V("128x64x16 w2x2 p4", true, true, EPI_STORE, 128, 64, 16, 2, 2, 0, true, 4);
      V("64x64x32 w2x2 p4", true, true, EPI_STORE, 64, 64, 32, 2, 2, 0, true, 4);
      V("64x64x32 w2x2 p6", true, true, EPI_STORE, 64, 64, 32, 2, 2, 0, true, 6);
      V("  abl nostore 128x128x32 w4x2 p2", true, true, EPI_STORE, 128, 128, 32, 4, 2, 1, false, 2);
      V("  abl neither 128x128x32 w4x2 p2", true, true, EPI_STORE, 128, 128, 32, 4, 2, 3, false, 2);
    }
    if (nn && !split_only) {
      V("128x128x32 w4x2 p2", true, false, EPI_STORE, 128, 128, 32, 4, 2, 0, true, 2);
      V("128x64x32 w2x2 p4", true, false, EPI_STORE, 128, 64, 32, 2, 2, 0, true, 4);
      V("64x64x32 w2x2 p4", true, false, EPI_STORE, 64, 64, 32, 2, 2, 0, true, 4);
      V("  abl neither 128x128x32 w4x2 p2", true, false, EPI_STORE, 128, 128, 32, 4, 2, 3, false, 2);
    }
    if (tn && !split_only) {
      V("128x128x32 w4x2 p2", false, false, EPI_PARTIAL, 128, 128, 32, 4, 2, 0, true, 2);
      V("128x128x32 w2x2 p2", false, false, EPI_PARTIAL, 128, 128, 32, 2, 2, 0, true, 2);
      V("128x64x32 w2x2 p4", false, false, EPI_PARTIAL, 128, 64, 32, 2, 2, 0, true, 4);
      V("64x64x32 w2x2 p4", false, false, EPI_PARTIAL, 64, 64, 32, 2, 2, 0, true, 4);
      V("  abl neither 128x128x32 w4x2 p2", false, false, EPI_PARTIAL, 128, 128, 32, 4, 2, 3, false, 2);
    }
    const double flops = 2.0 * M * N * K;
    printf("== %s %-12s M=%d N=%d K=%d  (%.2f GFLOP, splits %d)\n", s.kind, s.name, M, N, K, flops * 1e-9, splits);
    std::vector<std::vector<float>> times(vs.size());
    std::vector<double> errs(vs.size(), -1.0);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> got(rows.size() * (long)N);
    for (size_t v = 0; v < vs.size(); ++v) {   // correctness (also the warm-up)
      CK(hipMemsetAsync(C, 0, c_elems * 4, st));
      vs[v].run();
      CK(hipStreamSynchronize(st));
      hipError_t le = hipGetLastError();
      if (le != hipSuccess) { printf("   %-34s LAUNCH ERROR %s\n", vs[v].name.c_str(), hipGetErrorString(le)); continue; }
      if (!vs[v].check) continue;
      double err = 0;
      for (size_t ri = 0; ri < rows.size(); ++ri) {
        CK(hipMemcpy(got.data() + ri * N, C + (long)rows[ri] * N, (long)N * 4, hipMemcpyDeviceToHost));
        for (int n = 0; n < N; ++n) {
          err = std::max(err, std::fabs((double)got[ri * N + n] - ref[ri * N + n]));
        }
      }
      errs[v] = err / std::max(refmax, 1e-30);
    }
    const int rounds = quick ? 2 : 4, reps = 5;
    for (int r = 0; r < rounds; ++r)
      for (size_t v = 0; v < vs.size(); ++v) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) vs[v].run();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        times[v].push_back(ms / reps);
      }
    for (size_t v = 0; v < vs.size(); ++v) {
      std::sort(times[v].begin(), times[v].end());
      const float med = times[v][times[v].size() / 2], mn = times[v][0];
      printf("   %-34s med %8.4f ms %7.1f TF | min %8.4f ms %7.1f TF | err %.1e\n", vs[v].name.c_str(), med, flops / med * 1e-9, mn, flops / mn * 1e-9, errs[v]);
    }
    fflush(stdout);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(bias)); CK(hipFree(drows)); CK(hipFree(dref));
    if (part) CK(hipFree(part));
    if (scr) CK(hipFree(scr));
  }
  return 0;
}
