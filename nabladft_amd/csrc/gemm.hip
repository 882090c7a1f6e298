// Dense fp32 GEMMs of the PaiNN path on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact f32,
// bitwise an fmaf chain, 157 TFLOP/s chip peak).  These are the filter-generating / node MLP
// contractions the reference runs through ATen mm (nn.Linear in painn.py:459-464,520-525,79-83)
// and, in the backward sweeps, their input- and weight-gradient forms.
//
//   C[M,N] = sum_k A(m,k) * B(k,n)   (+ bias[n])  with three operand layouts
//     NT  A[m*lda+k], B = W[n*ldb+k]   y = x W^T + b        (forward / tangent)
//     NN  A[m*lda+k], B = W[k*ldb+n]   gx = gy W            (input gradient)
//     TN  A[k*lda+m], B[k*ldb+n]       gW = gy^T x          (weight gradient; K = rows, split over
//                                                            workgroups, partials reduced in a
//                                                            fixed order -> run-to-run deterministic)
// Tile: 128x128x32 per 256-thread workgroup; 4 wavefronts as 2x2, each owning a 64x64 sub-tile =
// 2x2 MFMA 32x32 accumulators (64 accumulator VGPRs).  Both operands are staged in LDS k-major
// ([BK][BM+pad]) so that the MFMA operand fetch (lane l: row l&31, k = l>>5) is a conflict-free
// ds_read_b32 of 32 consecutive floats per half-wave.
#include <algorithm>
#include "gemm_tile.h"   // round 3: the persistent, double-buffered tile engine k_gemm2 (every launch whose operands are 16-byte friendly)
#include "gemm_split.h"  // round 3: the same skeleton on the bf16 matrix pipe, every f32 value split exactly into three bf16 pieces (large launches)

// ---- generic kernels (round 1): operands that are not 16-byte friendly (odd K / leading dimensions) and the row-mapped spherical launches ----
#define BM 128
#define BN 128
// BKT (template parameter of the kernels) = K depth of one staged tile: 32 by default; the input-gradient (NN) launches use 16 -- half the
// LDS and 64 VGPRs give four workgroups per CU instead of three, measured +10-20 % for that layout (profiles/r01_gemm_variants_microbench.txt)
#define GEMM_THREADS 256
#ifdef NQ_GEMM_WAVES8
#define GEMM_OCC __attribute__((amdgpu_waves_per_eu(8, 8)))
#else
#define GEMM_OCC
#endif

__device__ __forceinline__ long rm_row(int q, int w, int ncomp, int base) { return (long)(q / w) * ncomp + base + q % w; }

// K-contiguous source (element (r, k) at src[r*ld + k]) -> LDS tile[k][r], stride LDS_KC
#define LDS_KC (BM + 1)
// MN-contiguous source (element (k, r) at src[k*ld + r]) -> LDS tile[k][r], stride LDS_MC (16-B aligned rows)
#define LDS_MC (BM + 4)

// ---- tile staging, split into fetch (global -> registers) and stash (registers -> LDS) so that the fetch of
// k-tile t+1 can be issued before the MFMAs of k-tile t (the global latency then hides under the matrix pipe).
// KC source: element (r, k) at src[r*ld + k]; MC source: element (k, r) at src[k*ld + r].  NT = threads per workgroup.
template <bool KC, int NT, int BKT, bool MAP = false>
__device__ __forceinline__ void fetch_tile(float4 (&v)[BM * BKT / 4 / NT], const float* __restrict__ src, int ld, int r0, int R, int k0, int Kend,
                                           bool vec_ok, int mw = 1, int mn = 1, int mb = 0) {
  const int t = threadIdx.x;
#pragma unroll
  for (int it = 0; it < BM * BKT / 4 / NT; ++it) {
    const int idx = t + NT * it;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KC) {  // 128 rows x BK k: consecutive lanes walk k (128-B row segments at BK = 32)
      const int row = idx / (BKT / 4), kq = (idx % (BKT / 4)) * 4;
      const int gr = r0 + row, gk = k0 + kq;
      if (gr < R) {
        const float* ptr = src + (MAP ? rm_row(gr, mw, mn, mb) : (long)gr) * ld + gk;   // MAP: K-contiguous operand, logical row -> packed row
        if (vec_ok && gk + 3 < Kend) x = *reinterpret_cast<const float4*>(ptr);
        else {
          if (gk + 0 < Kend) x.x = ptr[0];
          if (gk + 1 < Kend) x.y = ptr[1];
          if (gk + 2 < Kend) x.z = ptr[2];
          if (gk + 3 < Kend) x.w = ptr[3];
        }
      }
    } else {   // 32 k x 128 cols: consecutive lanes walk the contiguous dimension (512-B rows)
      const int k = idx >> 5, rq = (idx & 31) * 4;
      const int gk = k0 + k, gr = r0 + rq;
      if (gk < Kend) {
        const float* ptr = src + (MAP ? rm_row(gk, mw, mn, mb) : (long)gk) * ld + gr;   // MAP: the reduction index is the logical row
        if (vec_ok && gr + 3 < R) x = *reinterpret_cast<const float4*>(ptr);
        else {
          if (gr + 0 < R) x.x = ptr[0];
          if (gr + 1 < R) x.y = ptr[1];
          if (gr + 2 < R) x.z = ptr[2];
          if (gr + 3 < R) x.w = ptr[3];
        }
      }
    }
    v[it] = x;
  }
}

template <bool KC, int NT, int BKT>
__device__ __forceinline__ void stash_tile(float* __restrict__ tile, const float4 (&v)[BM * BKT / 4 / NT]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int it = 0; it < BM * BKT / 4 / NT; ++it) {
    const int idx = t + NT * it;
    if (KC) {
      const int row = idx / (BKT / 4), kq = (idx % (BKT / 4)) * 4;
      tile[(kq + 0) * LDS_KC + row] = v[it].x;
      tile[(kq + 1) * LDS_KC + row] = v[it].y;
      tile[(kq + 2) * LDS_KC + row] = v[it].z;
      tile[(kq + 3) * LDS_KC + row] = v[it].w;
    } else {
      const int k = idx >> 5, rq = (idx & 31) * 4;
      *reinterpret_cast<float4*>(tile + k * LDS_MC + rq) = v[it];
    }
  }
}

// NW = wavefronts per 128x128 tile: 4 -> each wave owns 64x64 (2x2 MFMA tiles), 8 -> 64x32 (2x1).
// PF = register prefetch of the next k-tile (one extra barrier-free overlap of global latency with MFMAs).
template <bool A_KC, bool B_KC, int EPI, int NW, bool PF, int BKT = 32, int RM = 0>
__global__ __launch_bounds__(NW * 64) GEMM_OCC void k_gemm(GemmArgs p) {
  constexpr int NT = NW * 64;
  constexpr int TN = NW == 4 ? 2 : 1;
  constexpr int LDA_S = A_KC ? LDS_KC : LDS_MC;
  constexpr int LDB_S = B_KC ? LDS_KC : LDS_MC;
  __shared__ __attribute__((aligned(16))) float As[BKT * LDA_S];
  __shared__ __attribute__((aligned(16))) float Bs[BKT * LDB_S];

  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  // row maps (see GemmArgs): RM 1 = all orders in one launch (blockIdx.z = L; A rows and C rows mapped), RM 2 = one order, reduction index mapped
  const int zc = RM == 3 ? (int)blockIdx.z / p.rm_s : 0;                                   // RM 3: packed component this split belongs to
  const int zL = RM == 3 ? (zc >= 36 ? 6 : zc >= 25 ? 5 : zc >= 16 ? 4 : zc >= 9 ? 3 : zc >= 4 ? 2 : zc >= 1 ? 1 : 0) : (int)blockIdx.z;
  const int mw = RM == 1 || RM == 3 ? 2 * zL + 1 : p.rm_w, mn = p.rm_ncomp, mb = RM == 1 || RM == 3 ? zL * zL : p.rm_base;
  const int Meff = RM == 1 ? p.rm_rows * mw : p.M;
  const float* const Bsrc = RM == 1 ? p.Bz[blockIdx.z] : p.B;
  if (RM == 1 && m0 >= Meff) return;   // workgroup-uniform: the grid is sized for the largest order
  int kbeg = 0, kend = p.K;
  if (EPI == EPI_PARTIAL) {
    const int split = RM == 3 ? (zc - zL * zL) * p.rm_s + (int)blockIdx.z % p.rm_s : (int)blockIdx.z;   // RM 3: order L owns (2L+1) * rm_s splits
    const int Ktot = RM == 3 ? p.rm_rows * mw : p.K;
    kbeg = min(Ktot, split * p.k_per_split);
    kend = min(Ktot, kbeg + p.k_per_split);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = NW == 4 ? (wave >> 1) : (wave >> 2), wn = NW == 4 ? (wave & 1) : (wave & 3);
  const int wcol0 = NW == 4 ? wn * 64 : wn * 32;
  const int lr = lane & 31, lk = lane >> 5;

  const bool a_vec = ((p.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0);
  const bool b_vec = ((p.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(Bsrc) & 15) == 0);

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float bsum = 0.f;
  float4 ra[BM * BKT / 4 / NT], rb[BM * BKT / 4 / NT];
  if (PF) {
    fetch_tile<A_KC, NT, BKT, RM != 0>(ra, p.A, p.lda, m0, Meff, kbeg, kend, a_vec, mw, mn, mb);
    fetch_tile<B_KC, NT, BKT, RM >= 2>(rb, Bsrc, p.ldb, n0, p.N, kbeg, kend, b_vec, mw, mn, mb);
  }
  for (int k0 = kbeg; k0 < kend; k0 += BKT) {
    if (!PF) {
      fetch_tile<A_KC, NT, BKT, RM != 0>(ra, p.A, p.lda, m0, Meff, k0, kend, a_vec, mw, mn, mb);
      fetch_tile<B_KC, NT, BKT, RM >= 2>(rb, Bsrc, p.ldb, n0, p.N, k0, kend, b_vec, mw, mn, mb);
    }
    stash_tile<A_KC, NT, BKT>(As, ra);
    stash_tile<B_KC, NT, BKT>(Bs, rb);
    __syncthreads();
    if (EPI == EPI_PARTIAL && !A_KC) {
      // bias gradient for free: column sums of the A tile (= gy rows) that is already in LDS, primal rows only
      if (p.bpart && blockIdx.y == 0 && threadIdx.x < BM && (RM != 3 || zc == 0)) {   // RM 3: the scalar rows (component 0) carry the bias
        const int kmax = min(BKT, p.brows - k0);
        for (int kk = 0; kk < kmax; ++kk) bsum += As[kk * LDA_S + threadIdx.x];
      }
    }
    if (PF && k0 + BKT < kend) {  // issue the next tile's global loads; they complete under the MFMAs below
      fetch_tile<A_KC, NT, BKT, RM != 0>(ra, p.A, p.lda, m0, Meff, k0 + BKT, kend, a_vec, mw, mn, mb);
      fetch_tile<B_KC, NT, BKT, RM >= 2>(rb, Bsrc, p.ldb, n0, p.N, k0 + BKT, kend, b_vec, mw, mn, mb);
    }
#pragma unroll
    for (int kk = 0; kk < BKT; kk += 2) {
      const float a0 = As[(kk + lk) * LDA_S + wm * 64 + lr];
      const float a1 = As[(kk + lk) * LDA_S + wm * 64 + 32 + lr];
      const float b0 = Bs[(kk + lk) * LDB_S + wcol0 + lr];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      if (TN == 2) {
        const float b1 = Bs[(kk + lk) * LDB_S + wcol0 + 32 + lr];
        acc[0][TN - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][TN - 1], 0, 0, 0);
        acc[1][TN - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][TN - 1], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  if (EPI == EPI_PARTIAL && !A_KC) {
    if (p.bpart && blockIdx.y == 0 && threadIdx.x < BM && m0 + (int)threadIdx.x < p.M && (RM != 3 || zc == 0))
      p.bpart[(long)blockIdx.z * p.M + m0 + threadIdx.x] = bsum;   // RM 3: component 0 owns the first rm_s values of blockIdx.z
  }
  // epilogue. C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  float* Cout = p.C;
  if (EPI == EPI_PARTIAL) Cout += (long)blockIdx.z * p.part_stride;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wcol0 + j * 32 + lr;
      if (col >= p.N) continue;
      const float bv = (EPI != EPI_PARTIAL && p.bias && (RM != 1 || blockIdx.z == 0)) ? p.bias[col] : 0.f;   // spherical: bias on the scalars only
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row >= Meff) continue;
        const long off = (RM == 1 ? rm_row(row, mw, mn, mb) : (long)row) * p.ldc + col;
        const float v = acc[i][j][r] + bv;
        if (EPI == EPI_ACC) Cout[off] += v;
        else if (EPI == EPI_DSILU) Cout[off] = p.eb * v * nq_dsilu_fast(p.resid[off]);       // 5: C = eb * v * silu'(aux)   (adjoint of the activation of the layer below)
        else if (EPI == EPI_RES) Cout[off] = p.ea * p.resid[off] + v;                 // 6: C = ea * aux + v          (skip connection of the adjoint)
#ifdef NQ_GEMM_NT_STORE
        else if (EPI == EPI_STORE) __builtin_nontemporal_store(v, &Cout[off]);
#endif
        else Cout[off] = v;
        if (EPI == EPI_SILU) p.C2[off] = nq_silu(v);
        if (EPI == EPI_SILU_RES) p.C2[off] = p.resid ? p.ea * p.resid[off] + p.eb * nq_silu_fast(v) : p.eb * nq_silu_fast(v);
        if (EPI == EPI_DSILU2) p.C2[off] = v * nq_dsilu_fast(p.resid[off]);
      }
    }
}

// ---- small problems (few 128x128 tiles: batch sizes of 32-256 conformers, PhiSNet's per-order layers) -----------------------------------------
// Same contraction on 64x64x32 tiles with 256 threads (2x2 wavefronts, one 32x32 MFMA accumulator each): four times the workgroups, a quarter of
// the MFMA work per k-tile and workgroup, so the serial K loop is bound by the (prefetched) global-load latency only.  Row-major A (NT / NN).
#define SM 64
#define SLDS_KC (SM + 1)
#define SLDS_MC (SM + 4)
template <bool KC>
__device__ __forceinline__ void fetch_small(float4 (&v)[2], const float* __restrict__ src, int ld, int r0, int R, int k0, int Kend, bool vec_ok) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = threadIdx.x + 256 * it;     // 512 float4 = 64 x 32 floats
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    const int row = KC ? idx >> 3 : (idx & 15) * 4, kq = KC ? (idx & 7) * 4 : idx >> 4;   // KC: (row, 4 k's); MC: (k, 4 cols)
    const int gr = r0 + row, gk = k0 + kq;
    if (KC) {
      if (gr < R) {
        const float* ptr = src + (long)gr * ld + gk;
        if (vec_ok && gk + 3 < Kend) x = *reinterpret_cast<const float4*>(ptr);
        else { if (gk < Kend) x.x = ptr[0]; if (gk + 1 < Kend) x.y = ptr[1]; if (gk + 2 < Kend) x.z = ptr[2]; if (gk + 3 < Kend) x.w = ptr[3]; }
      }
    } else if (gk < Kend) {
      const float* ptr = src + (long)gk * ld + gr;
      if (vec_ok && gr + 3 < R) x = *reinterpret_cast<const float4*>(ptr);
      else { if (gr < R) x.x = ptr[0]; if (gr + 1 < R) x.y = ptr[1]; if (gr + 2 < R) x.z = ptr[2]; if (gr + 3 < R) x.w = ptr[3]; }
    }
    v[it] = x;
  }
}
template <bool KC>
__device__ __forceinline__ void stash_small(float* __restrict__ tile, const float4 (&v)[2]) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = threadIdx.x + 256 * it;
    if (KC) {
      const int row = idx >> 3, kq = (idx & 7) * 4;
      tile[(kq + 0) * SLDS_KC + row] = v[it].x; tile[(kq + 1) * SLDS_KC + row] = v[it].y;
      tile[(kq + 2) * SLDS_KC + row] = v[it].z; tile[(kq + 3) * SLDS_KC + row] = v[it].w;
    } else {
      *reinterpret_cast<float4*>(tile + (idx >> 4) * SLDS_MC + (idx & 15) * 4) = v[it];
    }
  }
}
template <bool B_KC, int EPI>
__global__ __launch_bounds__(256) void k_gemm_small(GemmArgs p) {
  constexpr int LDB_S = B_KC ? SLDS_KC : SLDS_MC;
  __shared__ __attribute__((aligned(16))) float As[32 * SLDS_KC];
  __shared__ __attribute__((aligned(16))) float Bs[32 * LDB_S];
  const int m0 = blockIdx.x * SM, n0 = blockIdx.y * SM;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lk = lane >> 5;
  const bool a_vec = ((p.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0);
  const bool b_vec = ((p.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.B) & 15) == 0);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float4 ra[2], rb[2];
  fetch_small<true>(ra, p.A, p.lda, m0, p.M, 0, p.K, a_vec);
  fetch_small<B_KC>(rb, p.B, p.ldb, n0, p.N, 0, p.K, b_vec);
  for (int k0 = 0; k0 < p.K; k0 += 32) {
    stash_small<true>(As, ra);
    stash_small<B_KC>(Bs, rb);
    __syncthreads();
    if (k0 + 32 < p.K) {   // next k-tile's global loads complete under the MFMAs below
      fetch_small<true>(ra, p.A, p.lda, m0, p.M, k0 + 32, p.K, a_vec);
      fetch_small<B_KC>(rb, p.B, p.ldb, n0, p.N, k0 + 32, p.K, b_vec);
    }
#pragma unroll
    for (int kk = 0; kk < 32; kk += 2) {
      const float a = As[(kk + lk) * SLDS_KC + wm * 32 + lr];
      const float b = Bs[(kk + lk) * LDB_S + wn * 32 + lr];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int col = n0 + wn * 32 + lr;
  if (col >= p.N) return;
  const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
    if (row >= p.M) continue;
    const long off = (long)row * p.ldc + col;
    const float v = acc[r] + bv;
    if (EPI == EPI_ACC) p.C[off] += v;
    else if (EPI == EPI_DSILU) p.C[off] = p.eb * v * nq_dsilu_fast(p.resid[off]);
    else if (EPI == EPI_RES) p.C[off] = p.ea * p.resid[off] + v;
    else p.C[off] = v;
    if (EPI == EPI_SILU) p.C2[off] = nq_silu(v);
    if (EPI == EPI_SILU_RES) p.C2[off] = p.resid ? p.ea * p.resid[off] + p.eb * nq_silu_fast(v) : p.eb * nq_silu_fast(v);
    if (EPI == EPI_DSILU2) p.C2[off] = v * nq_dsilu_fast(p.resid[off]);
  }
}
// fewer than one 128x128 tile per CU -> the small-tile kernel
static bool gemm_is_small(int M, int N) { return (long)nq_cdiv(M, BM) * nq_cdiv(N, BN) < 256; }

// process-global kernel variant (tuning / A-B measurements): bit0 = 8 waves per tile, bit1 = register prefetch, bit3 = never use the small-tile kernel
static int g_gemm_variant = 1;  // measured best on MI355X (scripts/gemm_bench.py): 8 wavefronts per tile, no prefetch
extern "C" void nq_set_gemm_variant(int32_t v) { g_gemm_variant = v; }

template <bool A_KC, bool B_KC, int EPI, int BKT = 32>
static void launch_gemm(hipStream_t st, dim3 grid, const GemmArgs& p) {
  int variant = g_gemm_variant & 3;
  // latency-bound regime (few workgroups, e.g. batch_size 32): nothing else hides the global-load latency of the serial
  // K loop, so the register-prefetch form pays there (it is neutral-to-slightly-negative on full grids)
  if (variant == 1 && (long)grid.x * grid.y * grid.z < 512) variant = 3;
  // row-major-A layouts (forward, input gradient): the prefetch form also wins on full grids since the input-gradient launches use 16-deep
  // tiles (+14 % on the U / 2x shapes, +3-5 % forward); the weight-gradient layout (A_KC = false) loses 15 % with it and stays on variant 1
  if (variant == 1 && A_KC) variant = 3;
  switch (variant) {
    case 0: hipLaunchKernelGGL((k_gemm<A_KC, B_KC, EPI, 4, false, BKT>), grid, dim3(256), 0, st, p); break;
    case 1: hipLaunchKernelGGL((k_gemm<A_KC, B_KC, EPI, 8, false, BKT>), grid, dim3(512), 0, st, p); break;
    case 2: hipLaunchKernelGGL((k_gemm<A_KC, B_KC, EPI, 4, true, BKT>), grid, dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL((k_gemm<A_KC, B_KC, EPI, 8, true, BKT>), grid, dim3(512), 0, st, p); break;
  }
}

// out[i] = sum_s part[s*stride + i], fixed order -> deterministic.  A thread that walks all splits of one element is a serial chain of loads (512 splits of a
// 128 x 128 weight-gradient tile: 128 trips of ~0.3 us with 4 loads in flight, one wavefront per CU: 30 us for 33 MB), so the splits of an element are cut
// into RP_CHUNKS contiguous ranges summed by 8 threads (each as before: 4 interleaved accumulators), combined through LDS as ((c0+c1)+(c2+c3))+((c4+c5)+(c6+c7)).
// Few splits (< RP_MIN_SPLITS): one thread per element as before.  The order depends on (nsplit) only, never on the launch geometry.
#define RP_CHUNKS 8
#define RP_ELEMS 32          // elements per workgroup of RP_CHUNKS * RP_ELEMS threads: a half-wavefront reads one 128-byte segment of a split
#define RP_MIN_SPLITS 32
__device__ __forceinline__ float reduce_range(const float* __restrict__ part, long stride, long i, int k, int kend) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (; k + 3 < kend; k += 4) {   // 4 independent loads in flight; the combination order is fixed
    s0 += part[(long)k * stride + i]; s1 += part[(long)(k + 1) * stride + i];
    s2 += part[(long)(k + 2) * stride + i]; s3 += part[(long)(k + 3) * stride + i];
  }
  for (; k < kend; ++k) s0 += part[(long)k * stride + i];
  return (s0 + s1) + (s2 + s3);
}
__device__ __forceinline__ void reduce_partials_body(const float* __restrict__ part, int nsplit, long stride, long count, float* __restrict__ out, long block) {
  if (nsplit < RP_MIN_SPLITS) {
    const long i = block * (RP_CHUNKS * RP_ELEMS) + threadIdx.x;
    if (i < count) out[i] = reduce_range(part, stride, i, 0, nsplit);
    return;
  }
  __shared__ float red[RP_CHUNKS][RP_ELEMS];
  const int e = threadIdx.x & (RP_ELEMS - 1), c = threadIdx.x / RP_ELEMS;
  const long i = block * RP_ELEMS + e;
  const int per = (nsplit + RP_CHUNKS - 1) / RP_CHUNKS;
  red[c][e] = i < count ? reduce_range(part, stride, i, min(nsplit, c * per), min(nsplit, (c + 1) * per)) : 0.f;
  __syncthreads();
  if (c == 0 && i < count) out[i] = ((red[0][e] + red[1][e]) + (red[2][e] + red[3][e])) + ((red[4][e] + red[5][e]) + (red[6][e] + red[7][e]));
}
__global__ __launch_bounds__(RP_CHUNKS * RP_ELEMS) void k_reduce_partials(const float* __restrict__ part, int nsplit, long stride, long count, float* __restrict__ out) {
  reduce_partials_body(part, nsplit, stride, count, out, (long)blockIdx.x);
}
// two reductions of the same split count in one launch (a weight-gradient product and its bias gradient): blocks [0, blocks_a) take the first
__global__ __launch_bounds__(RP_CHUNKS * RP_ELEMS) void k_reduce_partials2(const float* __restrict__ part_a, long count_a, float* __restrict__ out_a, int blocks_a,
                                                                           const float* __restrict__ part_b, long count_b, float* __restrict__ out_b, int nsplit) {
  if ((int)blockIdx.x < blocks_a) reduce_partials_body(part_a, nsplit, count_a, count_a, out_a, (long)blockIdx.x);
  else reduce_partials_body(part_b, nsplit, count_b, count_b, out_b, (long)blockIdx.x - blocks_a);
}
static unsigned reduce_partials_blocks(int nsplit, long count) { return (unsigned)nq_cdiv(count, nsplit < RP_MIN_SPLITS ? (long)RP_CHUNKS * RP_ELEMS : (long)RP_ELEMS); }
#define LAUNCH_REDUCE_PARTIALS(st, part, nsplit, stride, count, out) \
  hipLaunchKernelGGL(k_reduce_partials, dim3(reduce_partials_blocks(nsplit, count)), dim3(RP_CHUNKS * RP_ELEMS), 0, st, part, nsplit, stride, count, out)

// column sums of A[rows][cols] (bias gradients): 256 threads = 64 columns x 4 row lanes, one 2048-row chunk per
// workgroup (coalesced 256-B row segments, 4 independent accumulators per thread), lanes combined through LDS;
// the per-chunk partials are then summed in fixed order by k_reduce_partials -> deterministic.
#define CS_ROWS 2048
__global__ __launch_bounds__(256) void k_colsum_partial(const float* __restrict__ A, long rows, int cols, int lda, float* __restrict__ part, int cs_rows) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const long r0 = (long)blockIdx.y * cs_rows, r1 = min(rows, r0 + cs_rows);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < cols) {
    long r = r0 + rl;
    for (; r + 12 < r1; r += 16) {
      s0 += A[r * lda + c]; s1 += A[(r + 4) * lda + c]; s2 += A[(r + 8) * lda + c]; s3 += A[(r + 12) * lda + c];
    }
    for (; r < r1; r += 4) s0 += A[r * lda + c];
  }
  red[rl][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rl == 0 && c < cols) part[(long)blockIdx.y * cols + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// ---- round-3 tile engine: eligibility, tile choice, launch -----------------------------------------------------------------------------
// k_gemm2 reads both operands with 16-byte buffer loads and 32-bit offsets from the tile origin.
template <bool A_KC, bool B_KC>
static bool gemm2_ok(const GemmArgs& p, long kspan) {
  if (g_gemm_variant & 16) return false;   // A/B switch: force the generic kernels
  if ((reinterpret_cast<uintptr_t>(p.A) & 15) || (reinterpret_cast<uintptr_t>(p.B) & 15) || (p.lda & 3) || (p.ldb & 3)) return false;
  if ((A_KC || B_KC) && (p.K & 3)) return false;   // K-contiguous operand: whole float4s along k
  if (!A_KC && (p.M & 3)) return false;            // MN-contiguous operands: whole float4s along m / n
  if (!B_KC && (p.N & 3)) return false;
  // 32-bit byte offsets inside one tile / k span
  if (!A_KC && kspan * p.lda * 4 >= 0x7fffffffL) return false;
  if (!B_KC && kspan * p.ldb * 4 >= 0x7fffffffL) return false;
  if ((A_KC && 128L * p.lda * 4 >= 0x7fffffffL) || (B_KC && 128L * p.ldb * 4 >= 0x7fffffffL) || 128L * p.ldc * 4 >= 0x7fffffffL) return false;
  return true;
}
// Tile choice by the load of the busiest CU (tiles are dealt round-robin): 128x128 (8 wavefronts, 2 workgroups per CU) runs ~10 % more flops per
// CU-cycle than 64x64 (4 wavefronts, 4 per CU) but quantises coarser -- [20480 x 256] x [256 x 256] is 320 big tiles = 2 on the busiest CU, or
// 1280 small ones = 5 quarter-size tiles (measured 0.043 vs 0.032 ms, scripts/lab/gemm_lab.hip).
static bool gemm2_small_tiles(int M, int N, int splits) {
  const long t128 = (long)nq_cdiv(M, 128) * nq_cdiv(N, 128) * splits, t64 = (long)nq_cdiv(M, 64) * nq_cdiv(N, 64) * splits;
  const double c128 = (double)((t128 + 255) / 256), c64 = (double)((t64 + 255) / 256) / (4.0 * 0.9);
  return c64 < c128;
}
template <bool A_KC, bool B_KC, int EPI>
static void launch_gemm2(hipStream_t st, const GemmArgs& p, int splits) {
  if (gemm2_small_tiles(p.M, p.N, splits)) {
    const long tiles = (long)nq_cdiv(p.M, 64) * nq_cdiv(p.N, 64) * splits;
    // (the two-output tangent epilogue needs 3 workgroups per CU worth of registers: at 4 it spilled 124 bytes per lane)
    constexpr int WPE64 = EPI == EPI_DSILU2 ? 3 : 4;
    hipLaunchKernelGGL((k_gemm2<A_KC, B_KC, EPI, 64, 64, 32, 2, 2, WPE64>), dim3((unsigned)(tiles < 1024 ? tiles : 1024)), dim3(256), 0, st, p);
  } else {
    const long tiles = (long)nq_cdiv(p.M, 128) * nq_cdiv(p.N, 128) * splits;
    hipLaunchKernelGGL((k_gemm2<A_KC, B_KC, EPI, 128, 128, 32, 4, 2, 4>), dim3((unsigned)(tiles < 512 ? tiles : 512)), dim3(512), 0, st, p);
  }
}

// ---- split-bf16 engine (gemm_split.h): f32-accurate products on the bf16 matrix pipe, 128 x 128 tiles only -------------------------------------------
// Taken when the problem gives every CU work in 128 x 128 tiles (measured, scripts/lab/gemm_lab.hip: 1.3-1.7x the exact-f32 engine there); small
// problems stay on k_gemm2's 64 x 64 tiles.  nq_set_gemm_variant(32) (or NQ_GEMM_F32=1 in the environment) keeps every product on v_mfma_f32_32x32x2_f32.
static bool gemm3_disabled() {
  static const int env = [] { const char* e = getenv("NQ_GEMM_F32"); return (e && e[0] && e[0] != '0') ? 1 : 0; }();
  return env || (g_gemm_variant & 32);
}
bool nq_gemm_exact_f32_requested() { return gemm3_disabled(); }   // NQ_GEMM_F32=1 / nq_set_gemm_variant(32): every product on the exact-f32 instruction
template <bool A_KC, bool B_KC>
static bool gemm3_ok(const GemmArgs& p, long kspan, int splits) {
  if (gemm3_disabled() || !gemm2_ok<A_KC, B_KC>(p, kspan)) return false;
  return (g_gemm_variant & 64) || (long)nq_cdiv(p.M, 128) * nq_cdiv(p.N, 128) * splits >= 192;   // bit 6: every eligible launch (tests)
}
template <bool A_KC, bool B_KC, int EPI>
static int launch_gemm3(hipStream_t st, const GemmArgs& p, int splits) {
  // 168 registers (3 workgroups per CU) hold the plain epilogues; one that reads a second tile (+=, SiLU' / residual operands: 64 more registers) gets 256
  constexpr bool AUX = EPI == EPI_ACC || EPI == EPI_DSILU || EPI == EPI_RES || EPI == EPI_SILU_RES || EPI == EPI_DSILU2;
  constexpr int WPE = AUX ? 2 : 3;
  const long tiles = (long)nq_cdiv(p.M, 128) * nq_cdiv(p.N, 128) * splits;
  const dim3 grid((unsigned)(tiles < 256 * WPE ? tiles : 256 * WPE));   // (measured round 6: capping the grid at 2 / 1 workgroups per CU costs +0.2 / +1.4 ms per step)
  if constexpr (EPI == EPI_PARTIAL) {   // weight gradient: no K-contiguous operand, so no k-tail code; the bias gradient is a second instantiation
    if (p.bpart) hipLaunchKernelGGL((k_gemm3<A_KC, B_KC, EPI, 2, false, true>), dim3((unsigned)(tiles < 512 ? tiles : 512)), dim3(256), 0, st, p);   // (one register short of 168)
    else hipLaunchKernelGGL((k_gemm3<A_KC, B_KC, EPI, WPE, false, false>), grid, dim3(256), 0, st, p);
  } else if (p.K & 15) {
    constexpr int WT = (A_KC && B_KC) ? WPE : 2;   // (the input-gradient layout spills at 168 registers with the tail code)
    const dim3 gt((unsigned)(tiles < 256 * WT ? tiles : 256 * WT));
    hipLaunchKernelGGL((k_gemm3<A_KC, B_KC, EPI, WT, true>), gt, dim3(256), 0, st, p);
  } else {
    if constexpr (A_KC && !B_KC) {   // input-gradient layout: the weight operand may come pre-split (GemmArgs::Bpre)
      if (p.Bpre) { hipLaunchKernelGGL((k_gemm3<A_KC, B_KC, EPI, WPE, false, false, NQ_GEMM3_TERMS, 0, false, true>), grid, dim3(256), 0, st, p); return NQ_OK; }
    }
    hipLaunchKernelGGL((k_gemm3<A_KC, B_KC, EPI, WPE, false>), grid, dim3(256), 0, st, p);
  }
  return NQ_OK;
}

// ---- batched launches (round 4): nbatch independent products that differ by operand offsets only (GemmArgs::nbatch, bsA / bsB / bsC, Bz), grid.y = batch.
// The spherical linears of QHNet / PhiSNet / EquiformerV2 act on packed irreps tensors [rows][ncomp][F]; for ONE packed component the rows form a plain
// strided matrix (leading dimension ncomp * F), so the whole layer is ncomp plain products = one batched launch of the tile engines instead of the
// row-mapped launch of the generic round-1 kernel (32-48 TFLOP/s).
template <bool A_KC, bool B_KC, int EPI>
static bool launch_batched(hipStream_t st, const GemmArgs& p, int splits, long kspan) {
  if (g_gemm_variant & 256) return false;                                   // bit 8: keep the row-mapped generic launches (A/B switch)
  if (!gemm2_ok<A_KC, B_KC>(p, kspan) || p.nbatch < 1) return false;
  if ((p.bsA & 3) || (p.bsB & 3) || (p.bsC & 3) || (p.ldc & 3) || (reinterpret_cast<uintptr_t>(p.C) & 15)) return false;
  if (p.bz) for (int L = 0; L < 7; ++L) if (p.Bz[L] && (reinterpret_cast<uintptr_t>(p.Bz[L]) & 15)) return false;
  const long t128 = (long)nq_cdiv(p.M, 128) * nq_cdiv(p.N, 128) * splits, t64 = (long)nq_cdiv(p.M, 64) * nq_cdiv(p.N, 64) * splits;
  const bool use3 = !gemm3_disabled() && !p.bpart && EPI != EPI_PARTIAL && ((g_gemm_variant & 64) || t128 * p.nbatch >= 192) && !(p.K & 15);
  if (use3) {
    const long cap = 256 * 3;
    const unsigned gx = (unsigned)std::min<long>(t128, std::max<long>(1, (cap + p.nbatch - 1) / p.nbatch));
    hipLaunchKernelGGL((k_gemm3<A_KC, B_KC, EPI, 3, false, false, NQ_GEMM3_TERMS, 0, true>), dim3(gx, p.nbatch), dim3(256), 0, st, p);
  } else if (gemm2_small_tiles(p.M * p.nbatch, p.N, splits)) {
    const unsigned gx = (unsigned)std::min<long>(t64, std::max<long>(1, (1024 + p.nbatch - 1) / p.nbatch));
    hipLaunchKernelGGL((k_gemm2<A_KC, B_KC, EPI, 64, 64, 32, 2, 2, 4, 0, true>), dim3(gx, p.nbatch), dim3(256), 0, st, p);
  } else {
    const unsigned gx = (unsigned)std::min<long>(t128, std::max<long>(1, (512 + p.nbatch - 1) / p.nbatch));
    hipLaunchKernelGGL((k_gemm2<A_KC, B_KC, EPI, 128, 128, 32, 4, 2, 4, 0, true>), dim3(gx, p.nbatch), dim3(512), 0, st, p);
  }
  return true;
}

// ---- split-K for long contractions with few output tiles (round 4) -----------------------------------------------------------------------------------
// QHNet's generator adjoints are [rows x 5376] x [5376 x 32] and [rows x 8320] x [8320 x 128] at 17-28 k rows: 136-215 output tiles of 128 x 128, each a
// 1.4-4.3 MB serial stream of the long operand -- one tile per CU at best, 2.2-2.5 TB/s.  With the contraction cut into S ranges (S x tiles workgroups,
// partial slabs in a per-stream scratch of the library, fixed-order reduction that also applies bias / accumulation) the same bytes are streamed by the whole chip.
#include <map>
#include <mutex>
#include <vector>
static int splitk_count(int M, int N, int K) {
  const long tiles = (long)nq_cdiv(M, 128) * nq_cdiv(N, 128);
  if (K < 2048 || (K & 31) || tiles >= 256 || (g_gemm_variant & 128)) return 1;      // bit 7: never split (A/B switch)
  long s = 768 / tiles;
  if (s > K / 512) s = K / 512;          // >= 512 k per range
  if (s > 16) s = 16;
  return s < 2 ? 1 : (int)s;
}
struct SplitkScratch {
  float* ptr = nullptr; size_t floats = 0;
  bool captured = false;       // handed out while its stream was capturing: a HIP graph holds this pointer, it must outlive the library's own use
  size_t failed = 0;           // smallest request hipMalloc has refused: not retried on every call
  std::vector<float*> retired; // buffers replaced after a capture used them: kept for the graphs that replay into them (nq_gemm_splitk_release frees them)
};
static std::mutex g_splitk_mu;
static std::map<std::pair<int, hipStream_t>, SplitkScratch> g_splitk;
// Grow-only buffer per (device, stream): a launch on stream s may only reuse what earlier launches of the SAME stream wrote (stream order protects it).
// Lifetime rule: a buffer a capture has seen is never freed or moved (a captured graph replays into the raw pointer); when a later eager call needs more, the old
// buffer is retired, not freed.  No allocation happens inside a capture (too small -> the caller issues the plain launch).
static float* splitk_scratch(hipStream_t st, size_t floats) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(g_splitk_mu);
  SplitkScratch& b = g_splitk[{dev, st}];
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
  if (b.floats >= floats) { if (capturing) b.captured = true; return b.ptr; }
  if (capturing) return nullptr;
  if (b.failed && floats >= b.failed) return nullptr;
  const size_t want = floats + floats / 4;
  float* fresh = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&fresh), want * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); b.failed = floats; return b.floats >= floats ? b.ptr : nullptr; }
  if (b.ptr) {
    if (b.captured) b.retired.push_back(b.ptr);
    else { (void)hipStreamSynchronize(st); (void)hipFree(b.ptr); }
  }
  b.ptr = fresh; b.floats = want; b.captured = false;
  return b.ptr;
}
// frees every split-K buffer of the calling thread's device, retired ones included: only when no captured graph that used them will be replayed again
extern "C" void nq_gemm_splitk_release(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  (void)hipDeviceSynchronize();
  std::lock_guard<std::mutex> lock(g_splitk_mu);
  for (auto it = g_splitk.begin(); it != g_splitk.end();) {
    if (it->first.first != dev) { ++it; continue; }
    if (it->second.ptr) (void)hipFree(it->second.ptr);
    for (float* r : it->second.retired) (void)hipFree(r);
    it = g_splitk.erase(it);
  }
}
// test hook: {current pointer, floats, captured flag, retired count} of the (device, stream) buffer
extern "C" int nq_gemm_splitk_state(void* stream, uint64_t* ptr, uint64_t* floats, int32_t* captured, int32_t* retired) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return NQ_ERR_HIP;
  std::lock_guard<std::mutex> lock(g_splitk_mu);
  auto it = g_splitk.find({dev, (hipStream_t)stream});
  const bool have = it != g_splitk.end();
  if (ptr) *ptr = have ? (uint64_t)(uintptr_t)it->second.ptr : 0;
  if (floats) *floats = have ? it->second.floats : 0;
  if (captured) *captured = have && it->second.captured;
  if (retired) *retired = have ? (int32_t)it->second.retired.size() : 0;
  return NQ_OK;
}
// out[i] = (acc ? out[i] : 0) + bias[col] + sum_s part[s * stride + i]   (fixed order)
__global__ void k_reduce_splitk(const float* __restrict__ part, int nsplit, long stride, long count, int N, const float* __restrict__ bias, int accumulate,
                                float* __restrict__ out) {
  const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= count) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < nsplit; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(part + (long)k * stride + i4);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  if (bias) { const int c = (int)(i4 % N); s.x += bias[c]; s.y += bias[c + 1]; s.z += bias[c + 2]; s.w += bias[c + 3]; }
  float4* o = reinterpret_cast<float4*>(out + i4);
  if (accumulate) { const float4 v = *o; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  *o = s;
}
// returns true if the product was issued as a split-K launch (C contiguous, N % 4 == 0)
template <bool A_KC, bool B_KC>
static bool try_splitk(hipStream_t st, GemmArgs p, const float* bias, int accumulate, int& rc) {
  rc = NQ_OK;
  if (p.ldc != p.N || (p.N & 3) || (reinterpret_cast<uintptr_t>(p.C) & 15)) return false;
  const int S = splitk_count(p.M, p.N, p.K);
  if (S < 2) return false;
  int kper = nq_cdiv(p.K, S);
  kper = (kper + 31) / 32 * 32;
  const int nse = nq_cdiv(p.K, kper);
  if (nse < 2) return false;
  const long cnt = (long)p.M * p.N;
  float* scratch = splitk_scratch(st, (size_t)nse * cnt);
  if (!scratch) return false;
  float* out = p.C;
  p.C = scratch; p.bias = nullptr; p.C2 = nullptr; p.k_per_split = kper; p.part_stride = cnt; p.bpart = nullptr; p.brows = 0;
  if (gemm3_ok<A_KC, B_KC>(p, kper, nse)) { rc = launch_gemm3<A_KC, B_KC, EPI_PARTIAL>(st, p, nse); if (rc != NQ_OK) return true; }
  else if (gemm2_ok<A_KC, B_KC>(p, kper)) launch_gemm2<A_KC, B_KC, EPI_PARTIAL>(st, p, nse);
  else return false;
  hipLaunchKernelGGL(k_reduce_splitk, dim3((unsigned)nq_cdiv(cnt / 4, 256)), dim3(256), 0, st, scratch, nse, cnt, cnt, p.N, bias, accumulate, out);
  return true;
}

// ---- host launchers ------------------------------------------------------------------------
int nq_gemm_nt(hipStream_t st, const float* A, const float* W, float* C, const float* bias, float* C2_silu, int M, int N, int K,
               int lda, int ldw, int ldc, const char* tag) {
  char nm__[48]; if (nq_profile_on) snprintf(nm__, sizeof nm__, "gemm_nt:%s[n=%d,k=%d]", tag ? tag : "", N, K); else nm__[0] = 0;
  NQ_PROF(st, nm__);
  NQ_PROF_FLOPS(2.0 * M * N * K);
  if (M <= 0) return NQ_OK;
  GemmArgs p{A, W, C, bias, C2_silu, M, N, K, lda, ldw, ldc, 0, 0, nullptr, 0};
  if (!C2_silu) {
    int rc;
    if (try_splitk<true, true>(st, p, bias, 0, rc)) { NQ_TRY(rc); NQ_LAUNCH_CHECK(); return NQ_OK; }
  }
  if (gemm3_ok<true, true>(p, K, 1)) {
    if (C2_silu) NQ_TRY((launch_gemm3<true, true, EPI_SILU>(st, p, 1)));
    else NQ_TRY((launch_gemm3<true, true, EPI_STORE>(st, p, 1)));
    NQ_LAUNCH_CHECK();
    return NQ_OK;
  }
  if (gemm2_ok<true, true>(p, K)) {
    if (C2_silu) launch_gemm2<true, true, EPI_SILU>(st, p, 1);
    else launch_gemm2<true, true, EPI_STORE>(st, p, 1);
    NQ_LAUNCH_CHECK();
    return NQ_OK;
  }
  if (gemm_is_small(M, N) && !(g_gemm_variant & 8)) {
    dim3 gs(nq_cdiv(M, SM), nq_cdiv(N, SM), 1);
    if (C2_silu) hipLaunchKernelGGL((k_gemm_small<true, EPI_SILU>), gs, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_gemm_small<true, EPI_STORE>), gs, dim3(256), 0, st, p);
    NQ_LAUNCH_CHECK();
    return NQ_OK;
  }
  dim3 grid(nq_cdiv(M, BM), nq_cdiv(N, BN), 1);
  if (C2_silu) launch_gemm<true, true, EPI_SILU>(st, grid, p);
  else launch_gemm<true, true, EPI_STORE>(st, grid, p);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

// C[M, N] = A W^T and C2 = ea * resid + eb * silu(C)  (Dense + ScaledSiLU + residual of the GemNet-OC blocks in one pass)
int nq_gemm_nt_act(hipStream_t st, const float* A, const float* W, float* C, float* C2, const float* resid, float ea, float eb, int M, int N, int K,
                   const char* tag) {
  char nm__[48]; if (nq_profile_on) snprintf(nm__, sizeof nm__, "gemm_nt:%s[n=%d,k=%d]", tag ? tag : "", N, K); else nm__[0] = 0;
  NQ_PROF(st, nm__);
  NQ_PROF_FLOPS(2.0 * M * N * K);
  if (M <= 0) return NQ_OK;
  GemmArgs p{A, W, C, nullptr, C2, M, N, K, K, K, N, 0, 0, nullptr, 0};
  p.resid = resid; p.ea = ea; p.eb = eb;
  if (gemm3_ok<true, true>(p, K, 1)) {
    NQ_TRY((launch_gemm3<true, true, EPI_SILU_RES>(st, p, 1)));
    NQ_LAUNCH_CHECK();
    return NQ_OK;
  }
  if (gemm2_ok<true, true>(p, K)) {
    launch_gemm2<true, true, EPI_SILU_RES>(st, p, 1);
    NQ_LAUNCH_CHECK();
    return NQ_OK;
  }
  if (gemm_is_small(M, N) && !(g_gemm_variant & 8)) {
    dim3 gs(nq_cdiv(M, SM), nq_cdiv(N, SM), 1);
    hipLaunchKernelGGL((k_gemm_small<true, EPI_SILU_RES>), gs, dim3(256), 0, st, p);
  } else {
    dim3 grid(nq_cdiv(M, BM), nq_cdiv(N, BN), 1);
    launch_gemm<true, true, EPI_SILU_RES>(st, grid, p);
  }
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

// C[M, N] = A W^T (no bias) and C2 = C * silu'(aux): the tangent of a Linear + SiLU layer in one pass (aux = the primal pre-activation Z, C = the tangent
// pre-activation, C2 = the tangent of the activation); leading dimensions = the matrix widths
int nq_gemm_nt_dsilu(hipStream_t st, const float* A, const float* W, float* C, float* C2, const float* aux, int M, int N, int K, const char* tag) {
  char nm__[48]; if (nq_profile_on) snprintf(nm__, sizeof nm__, "gemm_nt:%s[n=%d,k=%d]", tag ? tag : "", N, K); else nm__[0] = 0;
  NQ_PROF(st, nm__);
  NQ_PROF_FLOPS(2.0 * M * N * K);
  if (M <= 0) return NQ_OK;
  GemmArgs p{A, W, C, nullptr, C2, M, N, K, K, K, N, 0, 0, nullptr, 0};
  p.resid = aux; p.ea = 0.f; p.eb = 1.f;
  if (gemm3_ok<true, true>(p, K, 1)) {
    NQ_TRY((launch_gemm3<true, true, EPI_DSILU2>(st, p, 1)));
  } else if (gemm2_ok<true, true>(p, K)) {
    launch_gemm2<true, true, EPI_DSILU2>(st, p, 1);
  } else if (gemm_is_small(M, N) && !(g_gemm_variant & 8)) {
    hipLaunchKernelGGL((k_gemm_small<true, EPI_DSILU2>), dim3(nq_cdiv(M, SM), nq_cdiv(N, SM), 1), dim3(256), 0, st, p);
  } else {
    launch_gemm<true, true, EPI_DSILU2>(st, dim3(nq_cdiv(M, BM), nq_cdiv(N, BN), 1), p);
  }
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

// C[M, N] = ea * aux + A W^T   (aux [M, N], must not alias C): the second product of a two-term sum (the +-m pairs of the SO(2) convolutions:
// out_p = x_p W_r^T - x_m W_i^T is a plain product into t followed by this one with aux = t, ea = -1 -- no separate linear-combination pass)
int nq_gemm_nt_res(hipStream_t st, const float* A, const float* W, float* C, const float* aux, float ea, int M, int N, int K) {
  char nm__[48]; if (nq_profile_on) snprintf(nm__, sizeof nm__, "gemm_nt:[n=%d,k=%d]", N, K); else nm__[0] = 0;
  NQ_PROF(st, nm__);
  NQ_PROF_FLOPS(2.0 * M * N * K);
  if (M <= 0) return NQ_OK;
  GemmArgs p{A, W, C, nullptr, nullptr, M, N, K, K, K, N, 0, 0, nullptr, 0};
  p.resid = aux; p.ea = ea; p.eb = 0.f;
  if (gemm3_ok<true, true>(p, K, 1)) {
    NQ_TRY((launch_gemm3<true, true, EPI_RES>(st, p, 1)));
  } else if (gemm2_ok<true, true>(p, K)) {
    launch_gemm2<true, true, EPI_RES>(st, p, 1);
  } else if (gemm_is_small(M, N) && !(g_gemm_variant & 8)) {
    hipLaunchKernelGGL((k_gemm_small<true, EPI_RES>), dim3(nq_cdiv(M, SM), nq_cdiv(N, SM), 1), dim3(256), 0, st, p);
  } else {
    launch_gemm<true, true, EPI_RES>(st, dim3(nq_cdiv(M, BM), nq_cdiv(N, BN), 1), p);
  }
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

// C[M, Kin] (+)= G[M, Nout] * W[Nout, Kin]
int nq_gemm_nn(hipStream_t st, const float* G, const float* W, float* C, int M, int Nout, int Kin, int ldg, int ldw, int ldc,
               int accumulate, const char* tag, const void* Bpre) {
  char nm__[48]; if (nq_profile_on) snprintf(nm__, sizeof nm__, "gemm_nn:%s[n=%d,k=%d]", tag ? tag : "", Kin, Nout); else nm__[0] = 0;
  NQ_PROF(st, nm__);
  NQ_PROF_FLOPS(2.0 * M * Nout * Kin);
  if (M <= 0) return NQ_OK;
  GemmArgs p{G, W, C, nullptr, nullptr, M, Kin, Nout, ldg, ldw, ldc, 0, 0, nullptr, 0};
  if (Bpre && !(Kin & 127) && !(Nout & 15) && ldw == Kin) { p.Bpre = Bpre; p.ldbpre = Nout; p.bpre_plane_bytes = Kin * Nout * 2; }
  {
    int rc;
    if (!p.Bpre && try_splitk<true, false>(st, p, nullptr, accumulate, rc)) { NQ_TRY(rc); NQ_LAUNCH_CHECK(); return NQ_OK; }
  }
  if (gemm3_ok<true, false>(p, Nout, 1)) {
    if (accumulate) NQ_TRY((launch_gemm3<true, false, EPI_ACC>(st, p, 1)));
    else NQ_TRY((launch_gemm3<true, false, EPI_STORE>(st, p, 1)));
    NQ_LAUNCH_CHECK();
    return NQ_OK;
  }
  if (gemm2_ok<true, false>(p, Nout)) {
    if (accumulate) launch_gemm2<true, false, EPI_ACC>(st, p, 1);
    else launch_gemm2<true, false, EPI_STORE>(st, p, 1);
    NQ_LAUNCH_CHECK();
    return NQ_OK;
  }
  if (gemm_is_small(M, Kin) && !(g_gemm_variant & 8)) {
    dim3 gs(nq_cdiv(M, SM), nq_cdiv(Kin, SM), 1);
    if (accumulate) hipLaunchKernelGGL((k_gemm_small<false, EPI_ACC>), gs, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_gemm_small<false, EPI_STORE>), gs, dim3(256), 0, st, p);
    NQ_LAUNCH_CHECK();
    return NQ_OK;
  }
  dim3 grid(nq_cdiv(M, BM), nq_cdiv(Kin, BN), 1);
  if (accumulate) launch_gemm<true, false, EPI_ACC, 16>(st, grid, p);
  else launch_gemm<true, false, EPI_STORE, 16>(st, grid, p);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

// C[M, Kin] = epilogue(G[M, Nout] * W[Nout, Kin]): mode 1: eb * v * silu'(aux), mode 2: ea * aux + v   (aux [M, Kin])
int nq_gemm_nn_epi(hipStream_t st, const float* G, const float* W, float* C, int M, int Nout, int Kin, const float* aux, float ea, float eb, int mode, const char* tag,
                   const void* Bpre) {
  char nm__[48]; if (nq_profile_on) snprintf(nm__, sizeof nm__, "gemm_nn:%s[n=%d,k=%d]", tag ? tag : "", Kin, Nout); else nm__[0] = 0;
  NQ_PROF(st, nm__);
  NQ_PROF_FLOPS(2.0 * M * Nout * Kin);
  if (M <= 0) return NQ_OK;
  GemmArgs p{G, W, C, nullptr, nullptr, M, Kin, Nout, Nout, Kin, Kin, 0, 0, nullptr, 0};
  p.resid = aux; p.ea = ea; p.eb = eb;
  if (Bpre && !(Kin & 127) && !(Nout & 15)) { p.Bpre = Bpre; p.ldbpre = Nout; p.bpre_plane_bytes = Kin * Nout * 2; }
  if (gemm3_ok<true, false>(p, Nout, 1)) {
    if (mode == 1) NQ_TRY((launch_gemm3<true, false, EPI_DSILU>(st, p, 1)));
    else NQ_TRY((launch_gemm3<true, false, EPI_RES>(st, p, 1)));
    NQ_LAUNCH_CHECK();
    return NQ_OK;
  }
  if (gemm2_ok<true, false>(p, Nout)) {
    if (mode == 1) launch_gemm2<true, false, EPI_DSILU>(st, p, 1);
    else launch_gemm2<true, false, EPI_RES>(st, p, 1);
    NQ_LAUNCH_CHECK();
    return NQ_OK;
  }
  if (gemm_is_small(M, Kin) && !(g_gemm_variant & 8)) {
    dim3 gs(nq_cdiv(M, SM), nq_cdiv(Kin, SM), 1);
    if (mode == 1) hipLaunchKernelGGL((k_gemm_small<false, EPI_DSILU>), gs, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_gemm_small<false, EPI_RES>), gs, dim3(256), 0, st, p);
  } else {
    dim3 grid(nq_cdiv(M, BM), nq_cdiv(Kin, BN), 1);
    if (mode == 1) launch_gemm<true, false, EPI_DSILU, 16>(st, grid, p);
    else launch_gemm<true, false, EPI_RES, 16>(st, grid, p);
  }
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

// C = G W (the adjoint of a SiLU layer's OUTPUT) and C2 = C * silu'(aux) (the adjoint of its pre-activation) in one pass: the force sweep keeps both, because the
// second-order sweep needs the first for its silu'' term and the second as an operand of the weight gradient (engine.hip: per-layer adjoint store)
int nq_gemm_nn_dsilu2(hipStream_t st, const float* G, const float* W, float* C, float* C2, const float* aux, int M, int Nout, int Kin, const char* tag, const void* Bpre) {
  char nm__[48]; if (nq_profile_on) snprintf(nm__, sizeof nm__, "gemm_nn:%s[n=%d,k=%d]", tag ? tag : "", Kin, Nout); else nm__[0] = 0;
  NQ_PROF(st, nm__);
  NQ_PROF_FLOPS(2.0 * M * Nout * Kin);
  if (M <= 0) return NQ_OK;
  GemmArgs p{G, W, C, nullptr, C2, M, Kin, Nout, Nout, Kin, Kin, 0, 0, nullptr, 0};
  p.resid = aux; p.ea = 0.f; p.eb = 1.f;
  if (Bpre && !(Kin & 127) && !(Nout & 15)) { p.Bpre = Bpre; p.ldbpre = Nout; p.bpre_plane_bytes = Kin * Nout * 2; }
  if (gemm3_ok<true, false>(p, Nout, 1)) NQ_TRY((launch_gemm3<true, false, EPI_DSILU2>(st, p, 1)));
  else if (gemm2_ok<true, false>(p, Nout)) launch_gemm2<true, false, EPI_DSILU2>(st, p, 1);
  else if (gemm_is_small(M, Kin) && !(g_gemm_variant & 8)) hipLaunchKernelGGL((k_gemm_small<false, EPI_DSILU2>), dim3(nq_cdiv(M, SM), nq_cdiv(Kin, SM), 1), dim3(256), 0, st, p);
  else launch_gemm<true, false, EPI_DSILU2, 16>(st, dim3(nq_cdiv(M, BM), nq_cdiv(Kin, BN), 1), p);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

// out[Mo, No] = sum_{r<rows} GY[r, Mo] * X[r, No];  scratch must hold nq_gemm_tn_scratch_floats()
// Split count for the weight-gradient contraction: enough workgroups to fill 256 CUs (~768) given the number
// of 128x128 output tiles, but at least 512 rows per split so the 64-KB partial slab stays amortised.
// Split count of the weight-gradient contraction: the persistent engine keeps 2 workgroups of 8 wavefronts per CU = 512 slots; as many
// (output tile, row range) items as fit WITHOUT exceeding them (one item more would cost a whole extra round), at least 128 rows per split.
static int tn_splits(long rows, int Mo, int No, int slots) {
  const long tiles = (long)nq_cdiv(Mo, BM) * nq_cdiv(No, BN);
  long s = slots / tiles;
  const long by_rows = (rows + 127) / 128;   // >= 128 rows (4 k-tiles) per split
  if (s > by_rows) s = by_rows;
  if (s < 1) s = 1;
  return (int)s;
}
// workgroup slots of the engine that will run: 2 x 256 for the exact-f32 engine (and the split engine's bias-gradient flavour), 3 x 256 for the split engine
static int tn_slots(bool with_bias) { return (gemm3_disabled() || with_bias) ? 512 : 768; }
// (sized for the larger split count: the engine choice may change between the sizing call and the product)
size_t nq_gemm_tn_scratch_floats(long rows, int Mo, int No) { return (size_t)tn_splits(rows, Mo, No, 768) * ((size_t)Mo * No + Mo); }

int nq_gemm_tn(hipStream_t st, const float* GY, const float* X, float* out, long rows, int Mo, int No, int ldg, int ldx, float* scratch,
               const char* tag, float* bias_out, long bias_rows) {
  char nm__[48]; if (nq_profile_on) snprintf(nm__, sizeof nm__, "gemm_tn:%s[%dx%d]", tag ? tag : "", Mo, No); else nm__[0] = 0;
  NQ_PROF(st, nm__);
  NQ_PROF_FLOPS(2.0 * rows * Mo * No);
  if (rows <= 0) {
    NQ_HIP(hipMemsetAsync(out, 0, sizeof(float) * Mo * No, st));
    if (bias_out) NQ_HIP(hipMemsetAsync(bias_out, 0, sizeof(float) * Mo, st));
    return NQ_OK;
  }
  if (rows > 2000000000L) return nq_fail(NQ_ERR_ARG, "gemm_tn: too many rows");
  // The split count belongs to the engine that RUNS: sized for the split engine's 3 x 256 workgroup slots first; if that launch turns out not to be
  // eligible for it (operand alignment, M / N not multiples of 4, too few tiles) the count is taken again for the 2 x 256 slots of the exact / generic
  // engines (they used to inherit a split count tuned for the other engine).
  int ns = 0, kper = 0, nse = 0;
  float* bpart = nullptr;
  GemmArgs p{};
  auto plan = [&](int slots) {
    ns = tn_splits(rows, Mo, No, slots);
    kper = (int)((rows + ns - 1) / ns);
    kper = (kper + 31) / 32 * 32;   // whole k-tiles (BKT = 32 for the TN launches)
    bpart = bias_out ? scratch + (size_t)ns * Mo * No : nullptr;
    p = GemmArgs{GY, X, scratch, nullptr, nullptr, Mo, No, (int)rows, ldg, ldx, No, kper, (long)Mo * No, bpart, (int)bias_rows};
    nse = nq_cdiv(rows, kper);      // splits that actually hold rows (<= ns); only their slabs are reduced
  };
  const int slots0 = tn_slots(bias_out != nullptr);
  plan(slots0);
  if (slots0 != 512 && !gemm3_ok<false, false>(p, kper, nse)) plan(512);
  if (gemm3_ok<false, false>(p, kper, nse)) NQ_TRY((launch_gemm3<false, false, EPI_PARTIAL>(st, p, nse)));
  else if (gemm2_ok<false, false>(p, kper)) launch_gemm2<false, false, EPI_PARTIAL>(st, p, nse);
  else launch_gemm<false, false, EPI_PARTIAL>(st, dim3(nq_cdiv(Mo, BM), nq_cdiv(No, BN), nse), p);
  NQ_LAUNCH_CHECK();
  const long cnt = (long)Mo * No;
  if (bias_out) {   // weights and bias in one launch (the reduction order of each is the one k_reduce_partials uses)
    const unsigned ba = reduce_partials_blocks(nse, cnt), bb = reduce_partials_blocks(nse, (long)Mo);
    hipLaunchKernelGGL(k_reduce_partials2, dim3(ba + bb), dim3(RP_CHUNKS * RP_ELEMS), 0, st, scratch, cnt, out, (int)ba, bpart, (long)Mo, bias_out, nse);
  } else {
    LAUNCH_REDUCE_PARTIALS(st, scratch, nse, cnt, cnt, out);
  }
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

// ---- weights pre-split for the input-gradient products (GemmArgs::Bpre) -----------------------------------------------------------------------
// W [Kc][N] row-major (Kc = the contracted index of C = G W) -> planes [piece][N][Kc] bf16, Kc-contiguous: what PreStageB (gemm_split.h) loads
struct PresplitArgs { const float* W[5]; int Kc[5]; int N[5]; unsigned short* out[5]; int first[6]; int n; };
__global__ __launch_bounds__(256) void k_presplit_kn(PresplitArgs a) {
  const int b = (int)blockIdx.x;
  int g = 0;
#pragma unroll
  for (int i = 1; i < 5; ++i) g += (i < a.n && b >= a.first[i]) ? 1 : 0;
  const float* W = nullptr; int Kc = 0, N = 0, first = 0; unsigned short* out = nullptr;
#pragma unroll
  for (int i = 0; i < 5; ++i)
    if (g == i) { W = a.W[i]; Kc = a.Kc[i]; N = a.N[i]; out = a.out[i]; first = a.first[i]; }
  const long idx = (long)(b - first) * 256 + threadIdx.x;     // (n, k octet), n fastest: coalesced reads of W rows
  const int n = (int)(idx % N), oct = (int)(idx / N);
  if (oct >= Kc / 8) return;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = W[(long)(oct * 8 + i) * N + n];
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) sp_split2(v[2 * i], v[2 * i + 1], h[i], m[i], l[i]);
  const long plane = (long)N * Kc;
  gemm_u32x4* o = reinterpret_cast<gemm_u32x4*>(out + (long)n * Kc + oct * 8);
  *o = gemm_u32x4{h[0], h[1], h[2], h[3]};
  *reinterpret_cast<gemm_u32x4*>(out + plane + (long)n * Kc + oct * 8) = gemm_u32x4{m[0], m[1], m[2], m[3]};
  *reinterpret_cast<gemm_u32x4*>(out + 2 * plane + (long)n * Kc + oct * 8) = gemm_u32x4{l[0], l[1], l[2], l[3]};
}
// up to five matrices in one launch; out[i]: 3 * Kc[i] * N[i] bf16 (16-byte aligned)
int nq_gemm_presplit_kn(hipStream_t st, int n, const float* const* W, const int* Kc, const int* N, void* const* out) {
  if (n < 1 || n > 5) return nq_fail(NQ_ERR_ARG, "presplit: 1-5 matrices");
  NQ_PROF(st, "gemm_presplit");
  PresplitArgs a{};
  int blocks = 0;
  for (int i = 0; i < 5; ++i) {
    a.first[i] = blocks;
    if (i >= n) continue;
    if ((Kc[i] & 7) || !W[i] || !out[i]) return nq_fail(NQ_ERR_ARG, "presplit: bad matrix %d", i);
    a.W[i] = W[i]; a.Kc[i] = Kc[i]; a.N[i] = N[i]; a.out[i] = reinterpret_cast<unsigned short*>(out[i]);
    blocks += nq_cdiv((long)N[i] * (Kc[i] / 8), 256);
  }
  a.first[5] = blocks; a.n = n;
  hipLaunchKernelGGL(k_presplit_kn, dim3(blocks), dim3(256), 0, st, a);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

// ---- several weight-gradient products in one launch (k_gemm3_tn_group, gemm_split.h) + one reduction of all their partial tiles ----------------
#define RG_MAX (2 * GEMM_GROUP_MAX)
struct ReduceGroupArgs { const float* part[RG_MAX]; float* out[RG_MAX]; int nsplit[RG_MAX]; int count[RG_MAX]; int first[RG_MAX + 1]; int n; };
__global__ __launch_bounds__(256) void k_reduce_partials_group(ReduceGroupArgs q) {
  const int b = (int)blockIdx.x;
  int g = 0;
#pragma unroll
  for (int i = 1; i < RG_MAX; ++i) g += (i < q.n && b >= q.first[i]) ? 1 : 0;
  // static member indices only (see k_gemm3_tn_group)
  const float* part = nullptr; float* out = nullptr; int nsplit = 0, count = 0, first = 0;
#pragma unroll
  for (int i = 0; i < RG_MAX; ++i)
    if (g == i) { part = q.part[i]; out = q.out[i]; nsplit = q.nsplit[i]; count = q.count[i]; first = q.first[i]; }
  reduce_partials_body(part, nsplit, (long)count, (long)count, out, (long)(b - first));   // same combination order as k_reduce_partials
}

static const int TN_GROUP_SLOTS = 512;   // 2 workgroups per CU (the bias-gradient flavour of the split engine)
struct TnGroupPlan { int ns[GEMM_GROUP_MAX], kper[GEMM_GROUP_MAX], nse[GEMM_GROUP_MAX]; size_t off[GEMM_GROUP_MAX], boff[GEMM_GROUP_MAX], total; };
static void tn_group_plan(const NqTnSpec* sp, int n, TnGroupPlan* P) {
  double wsum = 0.0;
  long tiles[GEMM_GROUP_MAX];
  for (int g = 0; g < n; ++g) { tiles[g] = (long)nq_cdiv(sp[g].Mo, 128) * nq_cdiv(sp[g].No, 128); wsum += (double)sp[g].rows * tiles[g]; }
  size_t o = 0;
  for (int g = 0; g < n; ++g) {
    long s = (long)(TN_GROUP_SLOTS * ((double)sp[g].rows * tiles[g] / (wsum > 0 ? wsum : 1.0)) / tiles[g]);   // slots in proportion to rows x tiles, floor: never more than the slots
    const long by_rows = (sp[g].rows + 127) / 128;
    if (s > by_rows) s = by_rows;
    if (s < 1) s = 1;
    int kper = (int)((sp[g].rows + s - 1) / s);
    kper = (kper + 31) / 32 * 32;
    P->ns[g] = (int)s; P->kper[g] = kper; P->nse[g] = nq_cdiv(sp[g].rows, kper);
    P->off[g] = o; o += (size_t)s * sp[g].Mo * sp[g].No;
    P->boff[g] = o; o += sp[g].bias_out ? (size_t)s * sp[g].Mo : 0;
    o = (o + 3) & ~(size_t)3;
  }
  P->total = o;
}
size_t nq_gemm_tn_group_scratch_floats(const NqTnSpec* sp, int n) {
  if (n < 1 || n > GEMM_GROUP_MAX) return 0;
  TnGroupPlan P;
  tn_group_plan(sp, n, &P);
  return P.total;
}
// All products through the split engine in one launch + one reduction; returns NQ_ERR_ARG (nothing launched) when a product is not eligible
// (the caller then issues them one by one with nq_gemm_tn).
int nq_gemm_tn_group(hipStream_t st, const NqTnSpec* sp, int n, float* scratch) {
  if (n < 1 || n > GEMM_GROUP_MAX || gemm3_disabled()) return nq_fail(NQ_ERR_ARG, "gemm_tn_group: not eligible");
  TnGroupPlan P;
  tn_group_plan(sp, n, &P);
  GemmGroupArgs q{};
  ReduceGroupArgs r{};
  int wg = 0, rb = 0, nr = 0;
  double flops = 0.0;
  for (int g = 0; g < GEMM_GROUP_MAX; ++g) {
    q.first[g] = wg;
    if (g >= n) continue;
    const NqTnSpec& x = sp[g];
    if (x.rows <= 0 || x.rows > 2000000000L) return nq_fail(NQ_ERR_ARG, "gemm_tn_group: bad row count");
    float* bpart = x.bias_out ? scratch + P.boff[g] : nullptr;
    q.a[g] = GemmArgs{x.G, x.X, scratch + P.off[g], nullptr, nullptr, x.Mo, x.No, (int)x.rows, x.ldg, x.ldx, x.No, P.kper[g], (long)x.Mo * x.No, bpart, (int)x.bias_rows};
    if (!gemm3_ok<false, false>(q.a[g], P.kper[g], P.nse[g])) return nq_fail(NQ_ERR_ARG, "gemm_tn_group: product %d is not eligible for the split engine", g);
    wg += nq_cdiv(x.Mo, 128) * nq_cdiv(x.No, 128) * P.nse[g];
    flops += 2.0 * x.rows * x.Mo * x.No;
    r.part[nr] = scratch + P.off[g]; r.out[nr] = x.out; r.nsplit[nr] = P.nse[g]; r.count[nr] = x.Mo * x.No; r.first[nr] = rb; rb += (int)reduce_partials_blocks(P.nse[g], (long)x.Mo * x.No); ++nr;
    if (x.bias_out) { r.part[nr] = bpart; r.out[nr] = x.bias_out; r.nsplit[nr] = P.nse[g]; r.count[nr] = x.Mo; r.first[nr] = rb; rb += (int)reduce_partials_blocks(P.nse[g], (long)x.Mo); ++nr; }
  }
  q.first[GEMM_GROUP_MAX] = wg; q.n = n;
  for (int i = nr; i <= RG_MAX; ++i) r.first[i] = rb;
  r.n = nr;
  NQ_PROF(st, "gemm_tn_group");
  NQ_PROF_FLOPS(flops);
  hipLaunchKernelGGL(k_gemm3_tn_group<0>, dim3(wg), dim3(256), 0, st, q);
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_reduce_partials_group, dim3(rb), dim3(256), 0, st, r);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

// rows per workgroup: 2048 for large inputs, shorter chunks (more workgroups, shorter serial loops) for small ones; multiple of 16
static int cs_rows_for(long rows) { long c = (rows + 127) / 128; c = (c + 15) / 16 * 16; return (int)(c < 64 ? 64 : (c > CS_ROWS ? CS_ROWS : c)); }
// (the chunk count is not monotonic in `rows` below 128 chunks: callers size one scratch for several row counts, so never report fewer than 128)
size_t nq_colsum_scratch_floats(long rows, int cols) { const long c = nq_cdiv(rows, cs_rows_for(rows)); return (size_t)(c < 128 ? 128 : c) * cols; }

int nq_colsum(hipStream_t st, const float* A, long rows, int cols, int lda, float* out, float* scratch) {
  NQ_PROF(st, "colsum");
  if (rows <= 0) {
    NQ_HIP(hipMemsetAsync(out, 0, sizeof(float) * cols, st));
    return NQ_OK;
  }
  const int csr = cs_rows_for(rows), chunks = nq_cdiv(rows, csr);
  hipLaunchKernelGGL(k_colsum_partial, dim3(nq_cdiv(cols, 64), chunks), dim3(256), 0, st, A, rows, cols, lda, scratch, csr);
  NQ_LAUNCH_CHECK();
  LAUNCH_REDUCE_PARTIALS(st, scratch, chunks, (long)cols, (long)cols, out);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_reduce_partials(hipStream_t st, const float* part, int nsplit, long stride, long count, float* out) {
  LAUNCH_REDUCE_PARTIALS(st, part, nsplit, stride, count, out);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

// ---- spherical linear layers on packed irreps tensors (PhiSNet SphericalLinear, spherical_linear.py:50-59) -----------------------------------
// x, y: [rows][(order+1)^2][F]; one weight matrix per order.  Forward and input gradient: ONE launch for all orders (blockIdx.z = L, rows of
// order L found through the row map) instead of order+1 GEMMs on gathered copies; weight gradient: one split-K launch per order.
static int sph_check(long rows, int order, int Fin, int Fout) {
  if (order < 0 || order > 6) return nq_fail(NQ_ERR_ARG, "spherical linear: order must be in 0..6");
  if (rows < 0 || rows * (2L * order + 1) > 2000000000L || Fin <= 0 || Fout <= 0) return nq_fail(NQ_ERR_ARG, "spherical linear: bad sizes");
  return NQ_OK;
}

// weight gradients of all orders in one launch: every packed component gets `s` splits of the logical rows (order L: (2L+1) * s splits)
static int sph_tn_splits(long rows, int ncomp) {
  long s = 768 / ncomp;
  const long by_rows = (rows + 127) / 128;
  if (s > by_rows) s = by_rows;
  return (int)(s < 1 ? 1 : s);
}
struct SphOut { float* out[7]; };
// out[L][i] = sum over the (2L+1) * s partial slabs of order L, in slab order (deterministic)
__global__ void k_reduce_grouped(const float* __restrict__ part, int s, long cnt, SphOut o) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  const int L = blockIdx.y;
  const float* src = part + (long)L * L * s * cnt + i;
  const int n = (2 * L + 1) * s;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int k = 0;
  for (; k + 3 < n; k += 4) { a0 += src[(long)k * cnt]; a1 += src[(long)(k + 1) * cnt]; a2 += src[(long)(k + 2) * cnt]; a3 += src[(long)(k + 3) * cnt]; }
  for (; k < n; ++k) a0 += src[(long)k * cnt];
  o.out[L][i] = (a0 + a1) + (a2 + a3);
}

extern "C" {

int nq_sph_linear_forward(const float* x, const float* const* W_host, const float* bias0, float* y, int64_t rows, int32_t order, int32_t Fin,
                          int32_t Fout, void* stream) {
  NQ_TRY(sph_check(rows, order, Fin, Fout));
  if (!x || !W_host || !y) return nq_fail(NQ_ERR_ARG, "null argument");
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "sph_linear_fwd");
  NQ_PROF_FLOPS(2.0 * rows * (order + 1) * (order + 1) * Fin * Fout);
  if (rows == 0) return NQ_OK;
  const int ncomp = (order + 1) * (order + 1);
  GemmArgs p{};
  p.A = x; p.C = y; p.bias = bias0; p.M = (int)rows * (2 * order + 1); p.N = Fout; p.K = Fin; p.lda = Fin; p.ldb = Fin; p.ldc = Fout;
  p.rm_rows = (int)rows; p.rm_ncomp = ncomp;
  for (int L = 0; L <= order; ++L) { if (!W_host[L]) return nq_fail(NQ_ERR_ARG, "null weight"); p.Bz[L] = W_host[L]; }
  p.B = p.Bz[0];
  {   // one plain strided product per packed component on the tile engines (rows x Fin, leading dimension ncomp * Fin)
    GemmArgs b = p;
    b.M = (int)rows; b.lda = ncomp * Fin; b.ldc = ncomp * Fout; b.nbatch = ncomp; b.bz = 1; b.bsA = Fin; b.bsB = 0; b.bsC = Fout;
    if (launch_batched<true, true, EPI_STORE>(st, b, 1, Fin)) { NQ_LAUNCH_CHECK(); return NQ_OK; }
  }
  dim3 grid(nq_cdiv(p.M, BM), nq_cdiv(Fout, BN), order + 1);
  hipLaunchKernelGGL((k_gemm<true, true, EPI_STORE, 8, true, 32, 1>), grid, dim3(512), 0, st, p);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

// gx_L = gy_L W_L   (W_L: [Fout][Fin])
int nq_sph_linear_input_grad(const float* gy, const float* const* W_host, float* gx, int64_t rows, int32_t order, int32_t Fin, int32_t Fout,
                             void* stream) {
  NQ_TRY(sph_check(rows, order, Fin, Fout));
  if (!gy || !W_host || !gx) return nq_fail(NQ_ERR_ARG, "null argument");
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "sph_linear_bwd_x");
  NQ_PROF_FLOPS(2.0 * rows * (order + 1) * (order + 1) * Fin * Fout);
  if (rows == 0) return NQ_OK;
  const int ncomp = (order + 1) * (order + 1);
  GemmArgs p{};
  p.A = gy; p.C = gx; p.M = (int)rows * (2 * order + 1); p.N = Fin; p.K = Fout; p.lda = Fout; p.ldb = Fin; p.ldc = Fin;
  p.rm_rows = (int)rows; p.rm_ncomp = ncomp;
  for (int L = 0; L <= order; ++L) { if (!W_host[L]) return nq_fail(NQ_ERR_ARG, "null weight"); p.Bz[L] = W_host[L]; }
  p.B = p.Bz[0];
  {
    GemmArgs b = p;
    b.M = (int)rows; b.lda = ncomp * Fout; b.ldc = ncomp * Fin; b.nbatch = ncomp; b.bz = 1; b.bsA = Fout; b.bsB = 0; b.bsC = Fin;
    if (launch_batched<true, false, EPI_STORE>(st, b, 1, Fout)) { NQ_LAUNCH_CHECK(); return NQ_OK; }
  }
  dim3 grid(nq_cdiv(p.M, BM), nq_cdiv(Fin, BN), order + 1);
  hipLaunchKernelGGL((k_gemm<true, false, EPI_STORE, 8, true, 16, 1>), grid, dim3(512), 0, st, p);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

size_t nq_sph_weight_grad_scratch_floats(int64_t rows, int32_t order, int32_t Fin, int32_t Fout) {
  const size_t ncomp = (size_t)(order + 1) * (order + 1);
  const size_t sps = sph_tn_splits(rows, (int)ncomp);
  const size_t part = ncomp * sps * (size_t)Fout * Fin + sps * (size_t)Fout, cs = nq_colsum_scratch_floats(rows, Fout);   // + the bias partials of the scalar rows
  return part > cs ? part : cs;
}

// gW_L[Fout][Fin] = sum over the rows of order L of gy^T x; gbias0[Fout] = column sums of the scalar rows of gy (or null)
int nq_sph_linear_weight_grad(const float* gy, const float* x, float* const* gW_host, float* gbias0, int64_t rows, int32_t order, int32_t Fin,
                              int32_t Fout, float* scratch, void* stream) {
  NQ_TRY(sph_check(rows, order, Fin, Fout));
  if (!gy || !x || !gW_host || !scratch) return nq_fail(NQ_ERR_ARG, "null argument");
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "sph_linear_bwd_w");
  NQ_PROF_FLOPS(2.0 * rows * (order + 1) * (order + 1) * Fin * Fout);
  const int ncomp = (order + 1) * (order + 1);
  GemmArgs p{};
  for (int L = 0; L <= order; ++L) {
    if (!gW_host[L]) return nq_fail(NQ_ERR_ARG, "null weight gradient");
    if (rows == 0) NQ_HIP(hipMemsetAsync(gW_host[L], 0, sizeof(float) * Fout * Fin, st));
  }
  if (rows > 0) {
    const int sps = sph_tn_splits(rows, ncomp);
    int kper = (int)((rows + sps - 1) / sps);
    kper = (kper + 31) / 32 * 32;
    p.A = gy; p.B = x; p.C = scratch; p.M = Fout; p.N = Fin; p.K = (int)rows * (2 * order + 1); p.lda = Fout; p.ldb = Fin; p.ldc = Fin;
    p.k_per_split = kper; p.part_stride = (long)Fout * Fin;
    p.rm_rows = (int)rows; p.rm_ncomp = ncomp; p.rm_s = sps;
    float* bpart = scratch + (size_t)ncomp * sps * Fout * Fin;
    if (gbias0) { p.bpart = bpart; p.brows = (int)rows; }      // bias gradient = column sums of the scalar rows of gy, taken from the staged A tiles of component 0
    bool done = false;
    if (nq_cdiv(rows, kper) == sps) {   // every split holds rows: the slab layout [component][split] is the same for both kernels
      GemmArgs b = p;                   // per packed component: gW partial[split] = gy_c[rows, Fout]^T x_c[rows, Fin] over the split's rows
      b.K = (int)rows; b.lda = ncomp * Fout; b.ldb = ncomp * Fin; b.nbatch = ncomp; b.bz = 0; b.bsA = Fout; b.bsB = Fin; b.bsC = (long)sps * Fout * Fin;
      done = launch_batched<false, false, EPI_PARTIAL>(st, b, sps, kper);
    }
    if (!done) {
      dim3 grid(nq_cdiv(Fout, BM), nq_cdiv(Fin, BN), ncomp * sps);
      hipLaunchKernelGGL((k_gemm<false, false, EPI_PARTIAL, 8, true, 32, 3>), grid, dim3(512), 0, st, p);
    }
    NQ_LAUNCH_CHECK();
    SphOut outs{};
    for (int L = 0; L <= order; ++L) outs.out[L] = gW_host[L];
    const long cnt = (long)Fout * Fin;
    hipLaunchKernelGGL(k_reduce_grouped, dim3(nq_cdiv(cnt, 64), order + 1), dim3(64), 0, st, scratch, sps, cnt, outs);
    NQ_LAUNCH_CHECK();
  }
  if (gbias0) {
    if (rows == 0) NQ_HIP(hipMemsetAsync(gbias0, 0, sizeof(float) * Fout, st));
    else {   // fixed-order sum of the per-split partials written by the contraction above (no separate pass over gy)
      const int sps = sph_tn_splits(rows, ncomp);
      LAUNCH_REDUCE_PARTIALS(st, scratch + (size_t)ncomp * sps * Fout * Fin, sps, (long)Fout, (long)Fout, gbias0);
      NQ_LAUNCH_CHECK();
    }
  }
  return NQ_OK;
}

}  // extern "C"

