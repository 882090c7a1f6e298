// bf16 MFMA GEMM for the GemNet-OC Dense layers (BASELINE.json configs[2]: "bf16 with fp32 scatter-accumulate").
//   C[M, N] = A[M, K] (fp32 in HBM, rounded to bf16 while it is staged into LDS) x B[N, K]^T (bf16, packed once per optimizer step), fp32 accumulation
// on v_mfma_f32_32x32x16_bf16 (16x the issue rate of the exact-f32 MFMA the rest of the engine uses).  One kernel serves the forward product (B = W) and the
// input gradient (B = W^T, packed next to W); the weight gradient stays on the fp32 split-K kernel (its contraction runs over the rows of both operands).
// 128x128x32 tiles, 4 wavefronts (2x2, each 64x64 = 2x2 MFMA tiles), register prefetch of the next k-tile under the MFMAs, two LDS buffers, one barrier
// per k-tile.  LDS rows are K-contiguous with a 16-byte pad (80-byte stride): every lane fetches its 8 consecutive k of one row with a single ds_read_b128.
// Epilogues as in gemm.hip: store / accumulate / ScaledSiLU + residual (gemnet_oc/layers/base_layers.py:11-97).
#include "common.h"

typedef __bf16 hb8 __attribute__((ext_vector_type(8)));
typedef __bf16 hb4 __attribute__((ext_vector_type(4)));
typedef float hf16 __attribute__((ext_vector_type(16)));

#define HB_BM 128
#define HB_BN 128
#define HB_BK 32
#define HB_LD (HB_BK + 8)

struct HbArgs {
  const float* A; const __bf16* B; float* C; float* C2; const float* resid;
  float ea, eb;
  int M, N, K, lda, ldb, ldc;
  const __bf16* Ab;          // A_BF16: A already packed in bf16 (weight gradient: transposed activations)
  int k_split; long part_stride;   // split over the contraction: blockIdx.z covers [z * k_split, (z + 1) * k_split), output slab z
};
enum { HB_STORE = 0, HB_ACC = 1, HB_SILU_RES = 2, HB_DSILU = 3, HB_RES = 4 };

template <int EPI, bool A_BF16 = false>
__global__ __launch_bounds__(256) void k_gemm_bf16_nt(HbArgs p) {
  __shared__ __attribute__((aligned(16))) __bf16 sA[2][HB_BM * HB_LD];
  __shared__ __attribute__((aligned(16))) __bf16 sB[2][HB_BN * HB_LD];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * HB_BM, n0 = blockIdx.y * HB_BN;
  hf16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 ra[4];
  uint4 rb[2], rab[2];
  const int kbeg = A_BF16 ? blockIdx.z * p.k_split : 0, kend = A_BF16 ? min(p.K, kbeg + p.k_split) : p.K;
  const int ar = t >> 3, ak = (t & 7) * 4;      // A: 8 threads x float4 per row of 32 k
  const int br = t >> 2, bk = (t & 3) * 8;      // B: 4 threads x 8 bf16 per row
  auto fetch = [&](int k0) {
    if (A_BF16) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = m0 + br + 64 * i;
        rab[i] = row < p.M ? *reinterpret_cast<const uint4*>(p.Ab + (long)row * p.lda + k0 + bk) : make_uint4(0u, 0u, 0u, 0u);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = m0 + ar + 32 * i;
        ra[i] = row < p.M ? *reinterpret_cast<const float4*>(p.A + (long)row * p.lda + k0 + ak) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = n0 + br + 64 * i;
      rb[i] = row < p.N ? *reinterpret_cast<const uint4*>(p.B + (long)row * p.ldb + k0 + bk) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto stash = [&](int buf) {
    if (A_BF16) {
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4*>(&sA[buf][(br + 64 * i) * HB_LD + bk]) = rab[i];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        hb4 v;
        v[0] = (__bf16)ra[i].x; v[1] = (__bf16)ra[i].y; v[2] = (__bf16)ra[i].z; v[3] = (__bf16)ra[i].w;
        *reinterpret_cast<hb4*>(&sA[buf][(ar + 32 * i) * HB_LD + ak]) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4*>(&sB[buf][(br + 64 * i) * HB_LD + bk]) = rb[i];
  };
  fetch(kbeg);
  stash(0);
  __syncthreads();
  int buf = 0;
  const int lr = lane & 31, lk = lane >> 5;
  for (int k0 = kbeg; k0 < kend; k0 += HB_BK) {
    const bool more = k0 + HB_BK < kend;
    if (more) fetch(k0 + HB_BK);
#pragma unroll
    for (int kk = 0; kk < HB_BK; kk += 16) {
      hb8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const hb8*>(&sA[buf][(wm * 64 + i * 32 + lr) * HB_LD + kk + 8 * lk]);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const hb8*>(&sB[buf][(wn * 64 + j * 32 + lr) * HB_LD + kk + 8 * lk]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) stash(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + lr;
      if (col >= p.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row >= p.M) continue;
        const long off = (long)row * p.ldc + col + (A_BF16 ? (long)blockIdx.z * p.part_stride : 0L);
        const float v = acc[i][j][r];
        if (EPI == HB_ACC) p.C[off] += v;
        else if (EPI == HB_DSILU) p.C[off] = p.eb * v * nq_dsilu_fast(p.resid[off]);
        else if (EPI == HB_RES) p.C[off] = p.ea * p.resid[off] + v;
        else p.C[off] = v;
        if (EPI == HB_SILU_RES) p.C2[off] = p.resid ? p.ea * p.resid[off] + p.eb * nq_silu_fast(v) : p.eb * nq_silu_fast(v);
      }
    }
}

// W [N][K] fp32 -> Wb [N][K] and WbT [K][N] in bf16 (round to nearest even)
__global__ void k_bf16_pack(const float* __restrict__ W, int N, int K, __bf16* __restrict__ Wb, __bf16* __restrict__ WbT) {
  __shared__ float tile[32][33];
  const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int n = n0 + r, k = k0 + tx;
    const float v = (n < N && k < K) ? W[(long)n * K + k] : 0.f;
    tile[r][tx] = v;
    if (n < N && k < K) Wb[(long)n * K + k] = (__bf16)v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, n = n0 + tx;
    if (n < N && k < K) WbT[(long)k * N + n] = (__bf16)tile[tx][r];
  }
}

// x [M][C] fp32 -> xT [C][Mp] bf16, zero for m >= M (Mp = M rounded up to the k-tile).  64 (rows) x 32 (columns) tiles through LDS: reads are 128-byte row
// segments, writes are 128-byte segments of 64 consecutive m
__global__ __launch_bounds__(256) void k_transpose_bf16(const float* __restrict__ x, long M, int C, long Mp, __bf16* __restrict__ xT) {
  __shared__ float tile[64][33];
  const long m0 = (long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 64; r += 8) {
    const long m = m0 + r; const int c = c0 + tx;
    tile[r][tx] = (m < M && c < C) ? x[m * C + c] : 0.f;
  }
  __syncthreads();
  const int mx = (threadIdx.x & 31) * 2, cy = threadIdx.x >> 5;   // each thread writes two consecutive m (4 bytes)
  for (int r = cy; r < 32; r += 8) {
    const int c = c0 + r; const long m = m0 + mx;
    if (c < C && m + 1 < Mp + 1 && m < Mp) {
      typedef __bf16 hb2 __attribute__((ext_vector_type(2)));
      hb2 v; v[0] = (__bf16)tile[mx][r]; v[1] = (__bf16)tile[mx + 1][r];
      *reinterpret_cast<hb2*>(&xT[(long)c * Mp + m]) = v;
    }
  }
}
__global__ void k_hb_reduce(const float* __restrict__ part, int nsplit, long stride, long count, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float s = 0.f;
  for (int k = 0; k < nsplit; ++k) s += part[(long)k * stride + i];      // fixed order
  out[i] = s;
}

static void hb_wgrad_plan(long rows, int N, int K, long* Mp, int* nsplit, int* ksplit) {
  *Mp = (rows + HB_BK - 1) / HB_BK * HB_BK;
  const long tiles = (long)nq_cdiv(N, HB_BM) * nq_cdiv(K, HB_BN);
  long s = 256 / tiles; if (s < 1) s = 1;       // one workgroup per CU: every extra split is one more fp32 slab written and re-read by the reduction
  const long maxs = (*Mp + 255) / 256; if (s > maxs) s = maxs;
  long ks = (*Mp + s - 1) / s; ks = (ks + HB_BK - 1) / HB_BK * HB_BK;
  *ksplit = (int)ks;
  *nsplit = (int)((*Mp + ks - 1) / ks);
}

static int hb_launch(hipStream_t st, const float* A, const void* B, float* C, float* C2, const float* resid, float ea, float eb, int M, int N, int K,
                     int accumulate, const char* kind, int mode = 0) {
  char nm__[48]; if (nq_profile_on) snprintf(nm__, sizeof nm__, "gemm_bf16_%s:[n=%d,k=%d]", kind, N, K); else nm__[0] = 0;
  NQ_PROF(st, nm__);
  NQ_PROF_FLOPS(2.0 * M * N * K);
  if (M <= 0) return NQ_OK;
  if (K % HB_BK != 0 || N <= 0) return nq_fail(NQ_ERR_ARG, "bf16 gemm: K = %d must be a multiple of %d", K, HB_BK);
  HbArgs p{A, (const __bf16*)B, C, C2, resid, ea, eb, M, N, K, K, K, N, nullptr, 0, 0};
  dim3 grid(nq_cdiv(M, HB_BM), nq_cdiv(N, HB_BN), 1);
  if (mode == 1) hipLaunchKernelGGL((k_gemm_bf16_nt<HB_DSILU>), grid, dim3(256), 0, st, p);
  else if (mode == 2) hipLaunchKernelGGL((k_gemm_bf16_nt<HB_RES>), grid, dim3(256), 0, st, p);
  else if (C2) hipLaunchKernelGGL((k_gemm_bf16_nt<HB_SILU_RES>), grid, dim3(256), 0, st, p);
  else if (accumulate) hipLaunchKernelGGL((k_gemm_bf16_nt<HB_ACC>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((k_gemm_bf16_nt<HB_STORE>), grid, dim3(256), 0, st, p);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

extern "C" {

int nq_bf16_pack(const float* W, int32_t N, int32_t K, void* Wb, void* WbT, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "bf16_pack");
  if (!W || !Wb || !WbT || N <= 0 || K <= 0) return nq_fail(NQ_ERR_ARG, "bad argument");
  hipLaunchKernelGGL(k_bf16_pack, dim3(nq_cdiv(K, 32), nq_cdiv(N, 32)), dim3(256), 0, st, W, N, K, (__bf16*)Wb, (__bf16*)WbT);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_linear_forward_bf16(const float* A, const void* Wb, float* C, float* C_act, const float* resid, float alpha, float beta, int32_t M, int32_t N,
                           int32_t K, void* stream) {
  if (!A || !Wb || !C) return nq_fail(NQ_ERR_ARG, "null argument");
  return hb_launch((hipStream_t)stream, A, Wb, C, C_act, resid, alpha, beta, M, N, K, 0, "nt");
}

int nq_linear_input_grad_bf16(const float* G, const void* WbT, float* C, int32_t M, int32_t N, int32_t K, int32_t accumulate, void* stream) {
  if (!G || !WbT || !C) return nq_fail(NQ_ERR_ARG, "null argument");
  return hb_launch((hipStream_t)stream, G, WbT, C, nullptr, nullptr, 0.f, 0.f, M, K, N, accumulate, "nn");
}

int nq_linear_input_grad_bf16_epi(const float* G, const void* WbT, float* C, int32_t M, int32_t N, int32_t K, const float* aux, float alpha, float beta,
                                  int32_t mode, void* stream) {
  if (!G || !WbT || !C || !aux || (mode != 1 && mode != 2)) return nq_fail(NQ_ERR_ARG, "bad argument");
  return hb_launch((hipStream_t)stream, G, WbT, C, nullptr, aux, alpha, beta, M, K, N, 0, "nn", mode);
}

/* Weight gradient gW[N][K] = G[rows][N]^T X[rows][K] on the bf16 MFMA: both operands are transposed into bf16 [features][rows] copies (the contraction
 * must be the contiguous index of both MFMA operands), multiplied by the NT kernel split over the rows, partial slabs reduced in a fixed order. */
size_t nq_weight_grad_bf16_scratch_bytes(int64_t rows, int32_t N, int32_t K) {
  long Mp; int ns, ks;
  hb_wgrad_plan(rows, N, K, &Mp, &ns, &ks);
  return (size_t)(N + K) * Mp * 2 + (size_t)ns * N * K * 4 + 512;
}
int nq_linear_weight_grad_bf16(const float* G, const float* X, float* gW, int64_t rows, int32_t N, int32_t K, void* scratch, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  char nm__[48]; if (nq_profile_on) snprintf(nm__, sizeof nm__, "gemm_bf16_tn:[%dx%d]", N, K); else nm__[0] = 0;
  NQ_PROF(st, nm__);
  NQ_PROF_FLOPS(2.0 * rows * N * K);
  if (!G || !X || !gW || !scratch || rows <= 0) return nq_fail(NQ_ERR_ARG, "bad argument");
  long Mp; int ns, ks;
  hb_wgrad_plan(rows, N, K, &Mp, &ns, &ks);
  __bf16* GT = (__bf16*)scratch;
  __bf16* XT = GT + (size_t)N * Mp;
  float* part = (float*)(((uintptr_t)(XT + (size_t)K * Mp) + 255) & ~(uintptr_t)255);
  hipLaunchKernelGGL(k_transpose_bf16, dim3((unsigned)((Mp + 63) / 64), nq_cdiv(N, 32)), dim3(256), 0, st, G, (long)rows, N, Mp, GT);
  hipLaunchKernelGGL(k_transpose_bf16, dim3((unsigned)((Mp + 63) / 64), nq_cdiv(K, 32)), dim3(256), 0, st, X, (long)rows, K, Mp, XT);
  NQ_LAUNCH_CHECK();
  HbArgs p{nullptr, XT, ns > 1 ? part : gW, nullptr, nullptr, 0.f, 0.f, N, K, (int)Mp, (int)Mp, (int)Mp, K, GT, ks, (long)N * K};
  dim3 grid(nq_cdiv(N, HB_BM), nq_cdiv(K, HB_BN), ns);
  hipLaunchKernelGGL((k_gemm_bf16_nt<HB_STORE, true>), grid, dim3(256), 0, st, p);
  NQ_LAUNCH_CHECK();
  if (ns > 1) {
    const long cnt = (long)N * K;
    hipLaunchKernelGGL(k_hb_reduce, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, part, ns, cnt, cnt, gW);
    NQ_LAUNCH_CHECK();
  }
  return NQ_OK;
}

}  // extern "C"
