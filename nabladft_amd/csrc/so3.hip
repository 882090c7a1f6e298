// SO(3) Clebsch-Gordan mixing: the contraction at the heart of PhiSNet's PairMixing / SelfMixing
// (/root/reference/nablaDFT/phisnet/nn/modules/pair_mixing.py:47-69, self_mixing.py:55-83)
//
//   y_L[r, M, f] += c_path[r, f] * sum_{m1, m2} CG_path[m1, m2, M] * x1_{l1}[r, m1, f] * x2_{l2}[r, m2, f]      path = (l1, l2, L), l <= 4
//
// The reference materialises the 5-D broadcast product cg * x1 * x2 for each of the 65 paths (12 s per call on CPU).  Here
// one thread owns one (row, feature): its 25 + 25 input components and 25 outputs stay in registers and the 2052 non-zero coefficients
// are compile-time constants (cg_l4.inc, generated from nabladft_amd/cg.py), so the contraction is straight-line FMA code with no table
// look-ups; features are the fastest axis, so every load/store is coalesced across the wavefront.
//   x1, x2, y: [rows][(order+1)^2][F] (components of all orders concatenated, offset l*l + m + l)
//   coefficients c: [rows][n_enabled][F] (PairMixing: rbf @ W^T, an MFMA GEMM) or [n_enabled][F] broadcast over rows (SelfMixing)
// The sign convention of a model's own CG table is folded into c by the host (cg.py:path_signs).
#include "common.h"
#include "../../include/nablaq.h"

#define SO3_NCOMP 25
#define SO3_NPATHS 65

struct So3Args {
  const float* x1; const float* x2; const float* c; const float* keep;   // keep: [o_keep+1][F] or null (SelfMixing: y_L += keep_L * x1_L)
  float* y;
  const float* gy; float* gx1; float* gx2; float* gc;                     // backward
  long rows; int F;
  int n1, n2, ny;                                                         // components present: (order+1)^2
  long c_row_stride;                                                      // n_enabled * F, or 0 for row-broadcast coefficients
  long gc_row_stride;                                                     // n_enabled * F: the coefficient gradient is always written per row
  int keep_orders;                                                        // number of orders with a keep coefficient
  float* gc_part; int rows_per_block;                                     // RED kernels: per-workgroup partial sums of dL/dc, [blocks][n_enabled][F]
  int n_enabled;
  signed char cidx[SO3_NPATHS];                                           // path id -> index among the enabled paths, -1 = disabled
};

// RED (reverse pass with coefficients shared by all rows, SelfMixing): a workgroup walks rows_per_block rows, every thread adds its dL/dc
// contributions into ITS OWN slot of an LDS slab [256 / F][n_enabled][F] (no conflicts, fixed order), and the workgroup writes one partial
// [n_enabled][F] at the end -- instead of a [rows][n_enabled][F] array for the host to sum (PhiSNet calls this ~100 times per step).
template <bool BWD, bool RED = false>
__global__ __launch_bounds__(256) void k_so3_mix(So3Args a) {
  extern __shared__ float so3_slab[];
  const int rpp = RED ? 256 / a.F : 1;                       // rows per pass of the workgroup
  const int sub = RED ? (int)threadIdx.x / a.F : 0;
  float* const sg = RED ? so3_slab + (long)sub * a.n_enabled * a.F + (threadIdx.x % a.F) : nullptr;
  if (RED) {
    for (int i = threadIdx.x; i < rpp * a.n_enabled * a.F; i += blockDim.x) so3_slab[i] = 0.f;
    __syncthreads();
  }
  const int passes = RED ? a.rows_per_block / rpp : 1;
  for (int pass = 0; pass < passes; ++pass) {
  const long idx = RED ? ((long)blockIdx.x * a.rows_per_block + (long)pass * rpp) * a.F + threadIdx.x : (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.rows * a.F) { if (RED) break; else return; }
  const long r = idx / a.F;
  const int f = (int)(idx % a.F);
  float x1[SO3_NCOMP], x2[SO3_NCOMP], y[SO3_NCOMP], gx1[SO3_NCOMP], gx2[SO3_NCOMP];
#pragma unroll
  for (int k = 0; k < SO3_NCOMP; ++k) {
    x1[k] = k < a.n1 ? a.x1[(r * a.n1 + k) * a.F + f] : 0.f;
    x2[k] = k < a.n2 ? a.x2[(r * a.n2 + k) * a.F + f] : 0.f;
    if (BWD) { y[k] = k < a.ny ? a.gy[(r * a.ny + k) * a.F + f] : 0.f; gx1[k] = 0.f; gx2[k] = 0.f; }   // in the reverse kernel y[] holds dL/dy
    else y[k] = 0.f;
  }
  const float* crow = a.c + r * a.c_row_stride + f;
  float* gcrow = (BWD && !RED) ? a.gc + r * a.gc_row_stride + f : nullptr;   // per-row coefficient gradients (PairMixing: every row has its own)

#define CG_PATH_BEGIN(pid, l1, l2, L)                       \
  if (a.cidx[pid] >= 0) {                                   \
    const int ci = a.cidx[pid];                             \
    const float cc = crow[(long)ci * a.F];                  \
    float t[2 * L + 1];                                     \
    _Pragma("unroll") for (int M = 0; M < 2 * L + 1; ++M) t[M] = 0.f; \
    constexpr int YO = L * L;
#define CG_NZ(ia, ib, Mi, v)                                                                   \
    {                                                                                          \
      t[Mi] = fmaf(v, x1[ia] * x2[ib], t[Mi]);                                                 \
      if (BWD) { const float w = cc * v * y[YO + Mi]; gx1[ia] = fmaf(w, x2[ib], gx1[ia]); gx2[ib] = fmaf(w, x1[ia], gx2[ib]); } \
    }
#define CG_PATH_END(pid, l1, l2, L)                                                            \
    if (BWD) {                                                                                 \
      float g = 0.f;                                                                           \
      _Pragma("unroll") for (int M = 0; M < 2 * L + 1; ++M) g = fmaf(y[YO + M], t[M], g);      \
      if (RED) sg[(long)ci * a.F] += g; else gcrow[(long)ci * a.F] = g;                        \
    } else {                                                                                   \
      _Pragma("unroll") for (int M = 0; M < 2 * L + 1; ++M) y[YO + M] = fmaf(cc, t[M], y[YO + M]); \
    }                                                                                          \
  }
#include "cg_l4.inc"
#undef CG_PATH_BEGIN
#undef CG_NZ
#undef CG_PATH_END

  if (a.keep) {   // y_L += keep_L * x1_L for the orders both sides have
#pragma unroll
    for (int L = 0; L <= 4; ++L) {
      if (L < a.keep_orders) {
        const float kc = a.keep[(long)L * a.F + f];
#pragma unroll
        for (int M = 0; M < 2 * L + 1; ++M) {
          if (BWD) gx1[L * L + M] = fmaf(kc, y[L * L + M], gx1[L * L + M]);
          else y[L * L + M] = fmaf(kc, x1[L * L + M], y[L * L + M]);
        }
      }
    }
  }
  if (BWD) {
#pragma unroll
    for (int k = 0; k < SO3_NCOMP; ++k) {
      if (k < a.n1) a.gx1[(r * a.n1 + k) * a.F + f] = gx1[k];
      if (k < a.n2) a.gx2[(r * a.n2 + k) * a.F + f] = gx2[k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < SO3_NCOMP; ++k)
      if (k < a.ny) a.y[(r * a.ny + k) * a.F + f] = y[k];
  }
  }   // passes
  if (RED) {
    __syncthreads();
    float* out = a.gc_part + (long)blockIdx.x * a.n_enabled * a.F;
    for (int i = threadIdx.x; i < a.n_enabled * a.F; i += blockDim.x) {
      float s = 0.f;
      for (int q = 0; q < rpp; ++q) s += so3_slab[(long)q * a.n_enabled * a.F + i];
      out[i] = s;
    }
  }
}

// gradient of the keep coefficients per row: gk[r][L][f] = sum_M gy[r][L*L+M][f] * x[r][L*L+M][f]   (summed over rows by the host GEMM-free colsum)
__global__ void k_so3_keep_grad(const float* __restrict__ x, const float* __restrict__ gy, long rows, int F, int nx, int ny, int keep_orders,
                                float* __restrict__ gk) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * F) return;
  const long r = idx / F;
  const int f = (int)(idx % F);
  for (int L = 0; L < keep_orders; ++L) {
    float s = 0.f;
    for (int M = 0; M < 2 * L + 1; ++M) s = fmaf(gy[(r * ny + L * L + M) * F + f], x[(r * nx + L * L + M) * F + f], s);
    gk[(r * keep_orders + L) * F + f] = s;
  }
}

static int so3_fill(So3Args* a, const float* x1, const float* x2, const float* c, const float* keep, long rows, int F, int o1, int o2, int oy,
                    const int8_t* path_index, long c_row_stride, int keep_orders) {
  if (!x1 || !x2 || !c || !path_index) return nq_fail(NQ_ERR_ARG, "null argument");
  if (o1 < 0 || o1 > 4 || o2 < 0 || o2 > 4 || oy < 0 || oy > 4) return nq_fail(NQ_ERR_ARG, "orders must be in 0..4");
  if (F <= 0 || rows < 0) return nq_fail(NQ_ERR_ARG, "bad sizes");
  a->x1 = x1; a->x2 = x2; a->c = c; a->keep = keep; a->rows = rows; a->F = F;
  a->n1 = (o1 + 1) * (o1 + 1); a->n2 = (o2 + 1) * (o2 + 1); a->ny = (oy + 1) * (oy + 1);
  a->c_row_stride = c_row_stride; a->keep_orders = keep ? keep_orders : 0;
  for (int p = 0; p < SO3_NPATHS; ++p) a->cidx[p] = path_index[p];
  return NQ_OK;
}

extern "C" {

int nq_so3_mix_forward(const float* x1, const float* x2, const float* coeff, const float* keep, int64_t rows, int32_t F, int32_t order1,
                       int32_t order2, int32_t order_out, const int8_t* path_index_host, int64_t coeff_row_stride, int32_t keep_orders, float* y,
                       void* stream) {
  So3Args a{};
  NQ_TRY(so3_fill(&a, x1, x2, coeff, keep, rows, F, order1, order2, order_out, path_index_host, coeff_row_stride, keep_orders));
  if (!y) return nq_fail(NQ_ERR_ARG, "null argument");
  a.y = y;
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "so3_mix_fwd");
  if (rows * F > 0) hipLaunchKernelGGL((k_so3_mix<false>), dim3((unsigned)((rows * F + 255) / 256)), dim3(256), 0, st, a);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_so3_mix_backward(const float* x1, const float* x2, const float* coeff, const float* keep, const float* grad_y, int64_t rows, int32_t F,
                        int32_t order1, int32_t order2, int32_t order_out, const int8_t* path_index_host, int64_t coeff_row_stride,
                        int32_t keep_orders, float* grad_x1, float* grad_x2, float* grad_coeff_rows, float* grad_keep_rows, void* stream) {
  So3Args a{};
  NQ_TRY(so3_fill(&a, x1, x2, coeff, keep, rows, F, order1, order2, order_out, path_index_host, coeff_row_stride, keep_orders));
  if (!grad_y || !grad_x1 || !grad_x2 || !grad_coeff_rows) return nq_fail(NQ_ERR_ARG, "null argument");
  a.gy = grad_y; a.gx1 = grad_x1; a.gx2 = grad_x2; a.gc = grad_coeff_rows;
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "so3_mix_bwd");
  if (rows * F > 0) {
    int n_enabled = 0;
    for (int p = 0; p < SO3_NPATHS; ++p) n_enabled += path_index_host[p] >= 0;
    a.gc_row_stride = (long)n_enabled * F;
    hipLaunchKernelGGL((k_so3_mix<true>), dim3((unsigned)((rows * F + 255) / 256)), dim3(256), 0, st, a);
    NQ_LAUNCH_CHECK();
    if (keep && grad_keep_rows && keep_orders > 0) {
      hipLaunchKernelGGL(k_so3_keep_grad, dim3((unsigned)((rows * F + 255) / 256)), dim3(256), 0, st, x1, grad_y, (long)rows, F, a.n1, a.ny, keep_orders,
                         grad_keep_rows);
      NQ_LAUNCH_CHECK();
    }
  }
  return NQ_OK;
}

/* rows a workgroup of the reduced reverse kernel walks: ~1024 workgroups for large inputs, never fewer rows than one pass */
static int so3_rows_per_block(int64_t rows, int32_t F) {
  const int rpp = 256 / F;
  long k = (rows + 1023) / 1024;
  k = (k + rpp - 1) / rpp * rpp;
  if (k < rpp) k = rpp;
  if (k > 128) k = 128 / rpp * rpp;
  return (int)k;
}
int64_t nq_so3_mix_partial_blocks(int64_t rows, int32_t F) {
  if (F <= 0 || 256 % F != 0 || rows <= 0) return 0;
  const int k = so3_rows_per_block(rows, F);
  return (rows + k - 1) / k;
}
int nq_so3_mix_backward_shared(const float* x1, const float* x2, const float* coeff, const float* keep, const float* grad_y, int64_t rows, int32_t F,
                               int32_t order1, int32_t order2, int32_t order_out, const int8_t* path_index_host, int32_t keep_orders, float* grad_x1,
                               float* grad_x2, float* grad_coeff_partials, float* grad_keep_rows, void* stream) {
  So3Args a{};
  NQ_TRY(so3_fill(&a, x1, x2, coeff, keep, rows, F, order1, order2, order_out, path_index_host, 0, keep_orders));
  if (!grad_y || !grad_x1 || !grad_x2 || !grad_coeff_partials) return nq_fail(NQ_ERR_ARG, "null argument");
  if (256 % F != 0) return nq_fail(NQ_ERR_ARG, "reduced coefficient gradient: F must divide 256");
  a.gy = grad_y; a.gx1 = grad_x1; a.gx2 = grad_x2; a.gc_part = grad_coeff_partials;
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "so3_mix_bwd_shared");
  if (rows > 0) {
    int n_enabled = 0;
    for (int p = 0; p < SO3_NPATHS; ++p) n_enabled += path_index_host[p] >= 0;
    a.n_enabled = n_enabled;
    a.rows_per_block = so3_rows_per_block(rows, F);
    const long blocks = (rows + a.rows_per_block - 1) / a.rows_per_block;
    const size_t lds = sizeof(float) * (size_t)(256 / F) * n_enabled * F;
    if (lds > 64 * 1024) NQ_HIP(hipFuncSetAttribute((const void*)k_so3_mix<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_so3_mix<true, true>), dim3((unsigned)blocks), dim3(256), lds, st, a);
    NQ_LAUNCH_CHECK();
    if (keep && grad_keep_rows && keep_orders > 0) {
      hipLaunchKernelGGL(k_so3_keep_grad, dim3((unsigned)((rows * F + 255) / 256)), dim3(256), 0, st, x1, grad_y, (long)rows, F, a.n1, a.ny, keep_orders,
                         grad_keep_rows);
      NQ_LAUNCH_CHECK();
    }
  }
  return NQ_OK;
}

}  // extern "C"
