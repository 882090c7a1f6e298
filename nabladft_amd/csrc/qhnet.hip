// QHNet building blocks on gfx950: the SO(3) tensor-product layers of
// /root/reference/nablaDFT/qhnet/layers.py (ConvLayer :234-274, InnerProduct :277-294, NormGate :123-147, PairNetLayer :465-492,
// SelfNetLayer :565-578, Expansion :598-662), which the reference runs through e3nn's TorchScript einsums and torch_scatter.
//
// Layout: an irreps feature "C x (0 + 1 + ... + lmax)" is stored [rows][(lmax+1)^2][C], component offset l*l + m + l, channel fastest (every
// load / store is a coalesced row over the channels; e3nn's own [mul][2l+1] layout never appears on the device).  The real-basis
// Clebsch-Gordan tensors are the compile-time constants of cg_l4.inc (canonical sign); e3nn's sign and path normalisation are folded into
// the path weights by the host (nabladft_amd/qhnet.py).
//
// Graph: CSR by owner atom ("src" / row 1 of the reference's edge_index, qhnet.py:262), neighbours ascending ("dst" / row 0), rev = slot of
// the reverse edge.  Every reduction over the edges of an atom is a loop over its own row -- messages arriving at n travel along the
// reverse slots of n's row (the neighbour relation is symmetric) -- so there are no atomics and the summation order is the reference's
// sequential scatter order.
#include "common.h"
#include "../../include/nablaq.h"

#define QH_NCOMP 25
#define QH_NPATHS 65
// Paths per prefetched weight chunk and register budget, per kernel flavour (round 4, scripts/bench_qh_tp.py at 27.5 k rows, profiles/r04_qh_tp_variants.txt):
//   uuu forward 4 (0.81 -> 0.67 ms), uvu forward 8, uuu reverse 8 at TWO wavefronts per SIMD (21 spilled registers, 1.94 -> 1.62 ms), uvu reverse 16 (0.77 -> 0.68 ms)
#ifdef QH_WCH
#define QH_WCH_OF(UVU, BWD) (QH_WCH)
#else
#define QH_WCH_OF(UVU, BWD) ((UVU) ? ((BWD) ? 16 : 8) : ((BWD) ? 8 : 4))
#endif

// ---------------------------------------------------------------------------------------------------------------------------------------
// invariants of an edge / pair (layers.py:236-258 and :466-476): s0 = [x0[dst] | x0[dst or src] | <x_l[dst], x_l[src]> / (2l+1), l = 1..lmax]
__global__ __launch_bounds__(256) void k_qh_inv_fwd(const float* __restrict__ x, const int* __restrict__ own, const int* __restrict__ col, long R, int C,
                                                    int ncomp, int lmax, int second_from_owner, float* __restrict__ s0) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * C) return;
  const long r = idx / C;
  const int u = (int)(idx % C);
  const long a = col[r], b = own[r];                      // a = dst (row 0 of edge_index), b = src (row 1)
  const float* xa = x + a * ncomp * C + u;
  const float* xb = x + b * ncomp * C + u;
  float* o = s0 + r * (long)(2 + lmax) * C + u;
  o[0] = xa[0];
  o[C] = second_from_owner ? xb[0] : xa[0];
  for (int l = 1; l <= lmax; ++l) {
    float s = 0.f;
    for (int m = 0; m < 2 * l + 1; ++m) s = fmaf(xa[(long)(l * l + m) * C], xb[(long)(l * l + m) * C], s);
    o[(long)(1 + l) * C] = s / (float)(2 * l + 1);
  }
}

// reverse: one thread per (atom n, channel); n is "src" of its own rows r and "dst" of their reverse rows
__global__ __launch_bounds__(256) void k_qh_inv_bwd(const float* __restrict__ x, const float* __restrict__ gs, const int* __restrict__ row_ptr,
                                                    const int* __restrict__ col, const int* __restrict__ rev, int N, int C, int ncomp, int lmax,
                                                    int second_from_owner, float* __restrict__ gx) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)N * C) return;
  const int n = (int)(idx / C), u = (int)(idx % C);
  float acc[QH_NCOMP];
#pragma unroll
  for (int k = 0; k < QH_NCOMP; ++k) acc[k] = 0.f;
  const long W = (long)(2 + lmax) * C;
  for (int r = row_ptr[n]; r < row_ptr[n + 1]; ++r) {
    const long j = col[r], rr = rev[r];
    const float* g1 = gs + (long)r * W + u;     // row r: src = n, dst = j
    const float* g2 = gs + rr * W + u;          // row rev[r]: src = j, dst = n
    acc[0] += g2[0] + (second_from_owner ? g1[C] : g2[C]);
    const float* xj = x + j * ncomp * C + u;
#pragma unroll
    for (int l = 1; l <= 4; ++l) {
      if (l <= lmax) {
        const float g = (g1[(long)(1 + l) * C] + g2[(long)(1 + l) * C]) / (float)(2 * l + 1);
#pragma unroll
        for (int m = 0; m < 2 * l + 1; ++m) acc[l * l + m] = fmaf(g, xj[(long)(l * l + m) * C], acc[l * l + m]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < QH_NCOMP; ++k)
    if (k < ncomp) gx[((long)n * ncomp + k) * C + u] = acc[k];
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Tensor products per ROW (edge or ordered pair), one thread per (row, channel):
//   UVU (ConvLayer.tp_node, layers.py:262):        y[r] = sum_paths w1 w2 * CG . x[i1[r]] . sh[r]          (sh: one scalar per component and row)
//   UUU (PairNetLayer.tp_node_pair, layers.py:481): y[r] = sum_paths w1 w2 * CG . x[i1[r]] . x[i2[r]]
// The set of enabled paths is a template parameter (0: all 65; 1: the 42 with even l1+l2+L; 2: the 5 with l1 = 0 of the first conv layer), so the
// path loop is branch-free straight-line code and the weight loads of later paths are issued ahead of the FMAs of earlier ones.
// Reverse: per-row adjoints of the gathered operands (summed over each atom's rows by k_qh_pair_reduce -- parallelism over rows instead of atoms is
// what fills the chip at the reference's batch size of 2 molecules) and of both weight factors.
__host__ __device__ constexpr bool qh_path_on(int set, int l1, int l2, int L) { return set == 0 ? true : set == 1 ? ((l1 + l2 + L) % 2 == 0) : (l1 == 0); }
__host__ __device__ constexpr int qh_path_slot(int set, int pid) {
  int id = 0, slot = 0;
  for (int l1 = 0; l1 <= 4; ++l1)
    for (int l2 = 0; l2 <= 4; ++l2)
      for (int L = (l1 > l2 ? l1 - l2 : l2 - l1); L <= (l1 + l2 < 4 ? l1 + l2 : 4); ++L) {
        if (id == pid) return slot;
        if (qh_path_on(set, l1, l2, L)) ++slot;
        ++id;
      }
  return slot;
}
__host__ __device__ constexpr int qh_path_count(int set) { return qh_path_slot(set, QH_NPATHS); }

struct QhTpArgs {
  const float* x; const int* i1; const int* i2; const float* sh; const float* w1; const float* w2;
  float* y;
  const float* gy; const int* ig; float* gx1; float* gx2; float* gw1; float* gw2;
  long R; int C, n1, np_rt;
};

#ifndef QH_BWD_WPE
#define QH_BWD_WPE 2   // wavefronts per SIMD the reverse kernels are compiled for (register budget 512 / QH_BWD_WPE): 2 since round 4 (see QH_WCH_OF)
#endif
template <int SET, bool UVU, bool BWD, int VAR>
__global__ __launch_bounds__(256, (BWD || VAR == 0) ? QH_BWD_WPE : (VAR == 1 ? 4 : 3)) void k_qh_tp(QhTpArgs a) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.R * a.C) return;
  const long r = idx / a.C;
  const int u = (int)(idx % a.C);
  const long C = a.C;
  constexpr int NP = qh_path_count(SET);
  constexpr int N1 = SET == 2 ? 1 : QH_NCOMP;                                 // components of x1 that can be non-zero
  float x1[QH_NCOMP], x2[QH_NCOMP], y[QH_NCOMP], gx1[QH_NCOMP], gx2[QH_NCOMP];
  const float* p1 = a.x + (long)a.i1[r] * a.n1 * C + u;
  const float* p2 = UVU ? a.sh + r * QH_NCOMP : a.x + (long)a.i2[r] * QH_NCOMP * C + u;
  const long gr = BWD ? (a.ig ? (long)a.ig[r] : r) : 0;
#pragma unroll
  for (int k = 0; k < QH_NCOMP; ++k) {
    x1[k] = k < N1 ? p1[k * C] : 0.f;
    x2[k] = UVU ? p2[k] : p2[k * C];
    if (BWD) { y[k] = a.gy[(gr * QH_NCOMP + k) * C + u]; gx1[k] = 0.f; gx2[k] = 0.f; }
    else y[k] = 0.f;
  }
  const float* w1r = a.w1 + r * NP * C + u;
  const float* w2r = a.w2 ? a.w2 + r * NP * C + u : nullptr;
  float* g1r = BWD ? a.gw1 + r * NP * C + u : nullptr;
  float* g2r = (BWD && a.w2) ? a.gw2 + r * NP * C + u : nullptr;

  // VAR 0: each path loads its own weights; VAR 1 / 2: weights fetched in chunks of WCHK paths, double buffered (the loads of chunk k+1 are
  // issued before the arithmetic of chunk k), with (1) / without (2) scheduling barriers.  The runtime test on np_rt is always true: it keeps one
  // basic block per path -- as one block the reverse kernel spills 4 kB per lane.
  constexpr bool CHUNK = VAR != 0, BAR = VAR == 1;
  constexpr int WCHK = QH_WCH_OF(UVU, BWD);
  float wq1[2][WCHK], wq2[2][WCHK];
  if constexpr (CHUNK) {
#pragma unroll
    for (int i = 0; i < WCHK; ++i) {
      wq1[0][i] = i < NP ? w1r[(long)i * C] : 0.f;
      wq2[0][i] = (w2r && i < NP) ? w2r[(long)i * C] : 1.f;
    }
  }
#define CG_PATH_BEGIN(pid, l1, l2, L)                                   \
  if constexpr (qh_path_on(SET, l1, l2, L)) {                           \
    constexpr int slot = qh_path_slot(SET, pid);                        \
    if (slot < a.np_rt) {                                               \
    constexpr long ci = slot;                                           \
    constexpr int cb = (slot / WCHK) % 2, ck = slot % WCHK;               \
    if constexpr (CHUNK && ck == 0) {                                   \
      _Pragma("unroll") for (int i = 0; i < WCHK; ++i) {                \
        constexpr int nb = 1 - cb;                                      \
        const int sl = slot + WCHK + i;                                 \
        wq1[nb][i] = sl < NP ? w1r[(long)sl * C] : 0.f;                 \
        wq2[nb][i] = (w2r && sl < NP) ? w2r[(long)sl * C] : 1.f;        \
      }                                                                 \
      if constexpr (BAR) __builtin_amdgcn_sched_barrier(0);             \
    }                                                                   \
    float wa, wb;                                                       \
    if constexpr (CHUNK) { wa = wq1[cb][ck]; wb = wq2[cb][ck]; }        \
    else { wa = w1r[ci * C]; wb = w2r ? w2r[ci * C] : 1.f; }            \
    const float cc = wa * wb;                                           \
    float t[2 * L + 1];                                                 \
    _Pragma("unroll") for (int M = 0; M < 2 * L + 1; ++M) t[M] = 0.f;   \
    constexpr int YO = L * L;
#define CG_NZ(ia, ib, Mi, v)                                            \
    {                                                                   \
      t[Mi] = fmaf(v, x1[ia] * x2[ib], t[Mi]);                          \
      if (BWD) {                                                        \
        const float w = cc * v * y[YO + Mi];                            \
        gx1[ia] = fmaf(w, x2[ib], gx1[ia]);                             \
        if (!UVU) gx2[ib] = fmaf(w, x1[ia], gx2[ib]);                   \
      }                                                                 \
    }
#define CG_PATH_END(pid, l1, l2, L)                                     \
    if (BWD) {                                                          \
      float g = 0.f;                                                    \
      _Pragma("unroll") for (int M = 0; M < 2 * L + 1; ++M) g = fmaf(y[YO + M], t[M], g); \
      g1r[ci * C] = g * wb;                                             \
      if (g2r) g2r[ci * C] = g * wa;                                    \
    } else {                                                            \
      _Pragma("unroll") for (int M = 0; M < 2 * L + 1; ++M) y[YO + M] = fmaf(cc, t[M], y[YO + M]); \
    }                                                                   \
    if constexpr (BAR) __builtin_amdgcn_sched_barrier(0);               \
  } }
#include "cg_l4.inc"
#undef CG_PATH_BEGIN
#undef CG_NZ
#undef CG_PATH_END

  if (BWD) {
#pragma unroll
    for (int k = 0; k < QH_NCOMP; ++k) {
      if (k < N1) a.gx1[(r * N1 + k) * C + u] = gx1[k];
      if (!UVU) a.gx2[(r * QH_NCOMP + k) * C + u] = gx2[k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < QH_NCOMP; ++k) a.y[(r * QH_NCOMP + k) * C + u] = y[k];
  }
}

// out[n] = base[n] + sum_{r in row n} (a[r] + b[rev[r]]) (each operand nullable): per-row messages / adjoints of the two gathered operands back to the atoms (fixed order, no atomics)
__global__ __launch_bounds__(256) void k_qh_pair_reduce(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ base,
                                                        const int* __restrict__ row_ptr, const int* __restrict__ rev, int N, int W, float* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)N * W) return;
  const int n = (int)(idx / W), k = (int)(idx % W);
  float s = base ? base[idx] : 0.f;
  for (int r = row_ptr[n]; r < row_ptr[n + 1]; ++r) s += (a ? a[(long)r * W + k] : 0.f) + (b ? b[(long)rev[r] * W + k] : 0.f);
  out[idx] = s;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// NormGate pieces (layers.py:141-147): f0 = [x_0 | ||x_1|| | ... | ||x_lmax||] and y = [gates_0 | x_l * gates_l]
__global__ __launch_bounds__(256) void k_qh_normcat(const float* __restrict__ x, const float* __restrict__ gf, long rows, int C, int lmax,
                                                    float* __restrict__ out, int bwd) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * C) return;
  const long r = idx / C;
  const int u = (int)(idx % C);
  const int ncomp = (lmax + 1) * (lmax + 1);
  const float* xr = x + r * ncomp * C + u;
  if (!bwd) {
    float* o = out + r * (long)(lmax + 1) * C + u;
    o[0] = xr[0];
    for (int l = 1; l <= lmax; ++l) {
      float s = 0.f;
      for (int m = 0; m < 2 * l + 1; ++m) { const float v = xr[(long)(l * l + m) * C]; s = fmaf(v, v, s); }
      o[(long)l * C] = sqrtf(s);
    }
  } else {
    const float* g = gf + r * (long)(lmax + 1) * C + u;
    float* o = out + r * ncomp * C + u;
    o[0] = g[0];
    for (int l = 1; l <= lmax; ++l) {
      float s = 0.f;
      for (int m = 0; m < 2 * l + 1; ++m) { const float v = xr[(long)(l * l + m) * C]; s = fmaf(v, v, s); }
      const float f = s > 0.f ? g[(long)l * C] / sqrtf(s) : 0.f;
      for (int m = 0; m < 2 * l + 1; ++m) o[(long)(l * l + m) * C] = f * xr[(long)(l * l + m) * C];
    }
  }
}

// forward: y from (x, gates); backward: gx (l >= 1; 0 for the scalars) and ggates from (x, gates, gy)
__global__ __launch_bounds__(256) void k_qh_gate(const float* __restrict__ x, const float* __restrict__ gates, const float* __restrict__ gy, long rows, int C,
                                                 int lmax, float* __restrict__ y, float* __restrict__ gx, float* __restrict__ gg) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * C) return;
  const long r = idx / C;
  const int u = (int)(idx % C);
  const int ncomp = (lmax + 1) * (lmax + 1);
  const float* xr = x + r * ncomp * C + u;
  const float* gt = gates + r * (long)(lmax + 1) * C + u;
  if (!gy) {
    float* o = y + r * ncomp * C + u;
    o[0] = gt[0];
    for (int l = 1; l <= lmax; ++l) {
      const float g = gt[(long)l * C];
      for (int m = 0; m < 2 * l + 1; ++m) o[(long)(l * l + m) * C] = xr[(long)(l * l + m) * C] * g;
    }
  } else {
    const float* g = gy + r * ncomp * C + u;
    float* ox = gx + r * ncomp * C + u;
    float* og = gg + r * (long)(lmax + 1) * C + u;
    ox[0] = 0.f;
    og[0] = g[0];
    for (int l = 1; l <= lmax; ++l) {
      const float gate = gt[(long)l * C];
      float s = 0.f;
      for (int m = 0; m < 2 * l + 1; ++m) {
        const float gv = g[(long)(l * l + m) * C];
        s = fmaf(gv, xr[(long)(l * l + m) * C], s);
        ox[(long)(l * l + m) * C] = gv * gate;
      }
      og[(long)l * C] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// y = cst * act(x) (gy == null) or gx = gy * cst * act'(x); kind 0 = SiLU, 1 = shifted softplus (layers.py:21-22) -- the activations of
// e3nn's FullyConnectedNet (cst = its second-moment normalisation) and of the torch.nn.Sequential heads
__device__ __forceinline__ float qh_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__global__ __launch_bounds__(256) void k_qh_act(const float* __restrict__ x, const float* __restrict__ gy, int kind, float cst, long count, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float v = x[i];
  float o;
  if (!gy) o = cst * (kind == 0 ? nq_silu(v) : qh_softplus(v) - 0.69314718055994530942f);
  else o = gy[i] * cst * (kind == 0 ? nq_dsilu(v) : nq_sigmoid(v));
  out[i] = o;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Expansion (layers.py:598-662): per row (atom or ordered pair) the irreps x [25][Cb] and the per-row path weights [nw] (+ bias [nb] on the
// l_in = 0 paths) become one padded S x S block:  res[u,v,k] = sum_w W[w,u,v] x[l_in, k, w] (+ b[u,v]);  blk[(u,i),(v,j)] += w3j[i,j,k] res[u,v,k] / Cb.
// One workgroup per row: the weight row (33 kB for the def2-SVP layout) is staged in LDS by coalesced loads, then one thread per
// (l1, l2, u, v) accumulates its (2 l1 + 1) x (2 l2 + 1) sub-block over the admissible l_in -- every output element has one owner.
#define QH_EXP_MAXINS 19
struct QhExpArgs {
  const float* x; const float* W; const float* bias; float* out;
  const float* gout; float* gx; float* gW; float* gbias;
  const float* w3j;                 // [n_ins][5][5][9] (zero padded)
  long R; int Cb, S, nw, nb;
  int cnt[3], roff[3];
  int combo_start[10];              // prefix of cnt[l1] * cnt[l2] over (l1, l2) row-major
  int ins_of[5][3][3];              // instruction index of (l_in, l1, l2) or -1
  int woff[QH_EXP_MAXINS], boff[QH_EXP_MAXINS], roff_res[QH_EXP_MAXINS];   // offsets into W row / bias row / the LDS result slab
  float scale;
};

// coalesced float4 copy of n floats (n % 4 == 0 and 16-byte aligned source assumed by the caller) into LDS, four independent loads in flight per thread
__device__ __forceinline__ void qh_stage4(float* __restrict__ dst, const float* __restrict__ src, int n) {
  const float4* s4 = reinterpret_cast<const float4*>(src);
  float4* d4 = reinterpret_cast<float4*>(dst);
  const int n4 = n >> 2, T = blockDim.x;
  int i = threadIdx.x;
  for (; i + 3 * T < n4; i += 4 * T) {
    const float4 a = s4[i], b = s4[i + T], c = s4[i + 2 * T], d = s4[i + 3 * T];
    d4[i] = a; d4[i + T] = b; d4[i + 2 * T] = c; d4[i + 3 * T] = d;
  }
  for (; i < n4; i += T) d4[i] = s4[i];
}

template <int L1, int L2>
__device__ __forceinline__ void qh_exp_fwd_body(const QhExpArgs& a, const float* sW, const float* sX, const float* sB, int u, int v, float* orow) {
  constexpr int D1 = 2 * L1 + 1, D2 = 2 * L2 + 1;
  const int n1 = a.cnt[L1], n2 = a.cnt[L2];
  float blk[D1 * D2];
#pragma unroll
  for (int q = 0; q < D1 * D2; ++q) blk[q] = 0.f;
#pragma unroll
  for (int LI = (L1 > L2 ? L1 - L2 : L2 - L1); LI <= (L1 + L2 < 4 ? L1 + L2 : 4); ++LI) {
    const int ins = a.ins_of[LI][L1][L2];
    if (ins < 0) continue;
    float res[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) res[k] = 0.f;
    const float* wp = sW + a.woff[ins] + u * n2 + v;
    for (int w = 0; w < a.Cb; ++w) {
      const float wv = wp[w * n1 * n2];
#pragma unroll
      for (int k = 0; k < 9; ++k)
        if (k < 2 * LI + 1) res[k] = fmaf(wv, sX[(LI * LI + k) * a.Cb + w], res[k]);
    }
    if (a.boff[ins] >= 0) res[0] += sB[a.boff[ins] + u * n2 + v];
    const float* c3 = a.w3j + (long)ins * 225;
#pragma unroll
    for (int i = 0; i < D1; ++i)
#pragma unroll
      for (int j = 0; j < D2; ++j) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k)
          if (k < 2 * LI + 1) s = fmaf(c3[(i * 5 + j) * 9 + k], res[k], s);
        blk[i * D2 + j] = fmaf(s, a.scale, blk[i * D2 + j]);
      }
  }
  const int r0 = a.roff[L1] + u * D1, c0 = a.roff[L2] + v * D2;
#pragma unroll
  for (int i = 0; i < D1; ++i)
#pragma unroll
    for (int j = 0; j < D2; ++j) orow[(long)(r0 + i) * a.S + c0 + j] = blk[i * D2 + j];
}

__global__ __launch_bounds__(256) void k_qh_exp_fwd(QhExpArgs a) {
  extern __shared__ float lds[];
  float* sW = lds;
  float* sX = sW + ((a.nw + 3) & ~3);
  float* sB = sX + QH_NCOMP * a.Cb;
  const long r = blockIdx.x;
  qh_stage4(sW, a.W + r * a.nw, a.nw);                                   // nw = Cb * (sum of shell products): a multiple of 4 (checked by the launcher)
  qh_stage4(sX, a.x + r * QH_NCOMP * a.Cb, QH_NCOMP * a.Cb);
  for (int i = threadIdx.x; i < a.nb; i += blockDim.x) sB[i] = a.bias ? a.bias[r * a.nb + i] : 0.f;
  __syncthreads();
  float* orow = a.out + r * (long)a.S * a.S;
  for (int t = threadIdx.x; t < a.combo_start[9]; t += blockDim.x) {
    int c = 0;
    while (t >= a.combo_start[c + 1]) ++c;
    const int uv = t - a.combo_start[c];
    const int n2 = a.cnt[c % 3];
    const int u = uv / n2, v = uv % n2;
    switch (c) {
      case 0: qh_exp_fwd_body<0, 0>(a, sW, sX, sB, u, v, orow); break;
      case 1: qh_exp_fwd_body<0, 1>(a, sW, sX, sB, u, v, orow); break;
      case 2: qh_exp_fwd_body<0, 2>(a, sW, sX, sB, u, v, orow); break;
      case 3: qh_exp_fwd_body<1, 0>(a, sW, sX, sB, u, v, orow); break;
      case 4: qh_exp_fwd_body<1, 1>(a, sW, sX, sB, u, v, orow); break;
      case 5: qh_exp_fwd_body<1, 2>(a, sW, sX, sB, u, v, orow); break;
      case 6: qh_exp_fwd_body<2, 0>(a, sW, sX, sB, u, v, orow); break;
      case 7: qh_exp_fwd_body<2, 1>(a, sW, sX, sB, u, v, orow); break;
      default: qh_exp_fwd_body<2, 2>(a, sW, sX, sB, u, v, orow); break;
    }
  }
}

// reverse, phase 1: gres[ins][u,v,k] = scale * sum_{i,j} w3j[i,j,k] gout[(u,i),(v,j)] into the LDS slab sR (and the bias gradient)
template <int L1, int L2>
__device__ __forceinline__ void qh_exp_bwd_body(const QhExpArgs& a, const float* sG, float* sR, float* gbrow, int u, int v) {
  constexpr int D1 = 2 * L1 + 1, D2 = 2 * L2 + 1;
  const int n2 = a.cnt[L2];
  float g[D1 * D2];
  const int r0 = a.roff[L1] + u * D1, c0 = a.roff[L2] + v * D2;
#pragma unroll
  for (int i = 0; i < D1; ++i)
#pragma unroll
    for (int j = 0; j < D2; ++j) g[i * D2 + j] = sG[(r0 + i) * a.S + c0 + j];
#pragma unroll
  for (int LI = (L1 > L2 ? L1 - L2 : L2 - L1); LI <= (L1 + L2 < 4 ? L1 + L2 : 4); ++LI) {
    const int ins = a.ins_of[LI][L1][L2];
    if (ins < 0) continue;
    const float* c3 = a.w3j + (long)ins * 225;
    float* rp = sR + a.roff_res[ins] + (u * n2 + v) * (2 * LI + 1);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      if (k < 2 * LI + 1) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < D1; ++i)
#pragma unroll
          for (int j = 0; j < D2; ++j) s = fmaf(c3[(i * 5 + j) * 9 + k], g[i * D2 + j], s);
        s *= a.scale;
        rp[k] = s;
        if (k == 0 && a.boff[ins] >= 0 && gbrow) gbrow[a.boff[ins] + u * n2 + v] = s;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_qh_exp_bwd(QhExpArgs a, int n_ins, int res_total) {
  extern __shared__ float lds[];
  float* sW = lds;
  float* sX = sW + ((a.nw + 3) & ~3);
  float* sG = sX + QH_NCOMP * a.Cb;
  float* sR = sG + a.S * a.S;
  const long r = blockIdx.x;
  qh_stage4(sW, a.W + r * a.nw, a.nw);
  qh_stage4(sX, a.x + r * QH_NCOMP * a.Cb, QH_NCOMP * a.Cb);
  for (int i = threadIdx.x; i < a.S * a.S; i += blockDim.x) sG[i] = a.gout[r * (long)a.S * a.S + i];
  __syncthreads();
  float* gbrow = a.gbias ? a.gbias + r * a.nb : nullptr;
  for (int t = threadIdx.x; t < a.combo_start[9]; t += blockDim.x) {
    int c = 0;
    while (t >= a.combo_start[c + 1]) ++c;
    const int uv = t - a.combo_start[c];
    const int n2 = a.cnt[c % 3];
    const int u = uv / n2, v = uv % n2;
    switch (c) {
      case 0: qh_exp_bwd_body<0, 0>(a, sG, sR, gbrow, u, v); break;
      case 1: qh_exp_bwd_body<0, 1>(a, sG, sR, gbrow, u, v); break;
      case 2: qh_exp_bwd_body<0, 2>(a, sG, sR, gbrow, u, v); break;
      case 3: qh_exp_bwd_body<1, 0>(a, sG, sR, gbrow, u, v); break;
      case 4: qh_exp_bwd_body<1, 1>(a, sG, sR, gbrow, u, v); break;
      case 5: qh_exp_bwd_body<1, 2>(a, sG, sR, gbrow, u, v); break;
      case 6: qh_exp_bwd_body<2, 0>(a, sG, sR, gbrow, u, v); break;
      case 7: qh_exp_bwd_body<2, 1>(a, sG, sR, gbrow, u, v); break;
      default: qh_exp_bwd_body<2, 2>(a, sG, sR, gbrow, u, v); break;
    }
  }
  __syncthreads();
  // phase 2: weight gradient gW[ins][w,u,v] = sum_k gres[u,v,k] x[l_in,k,w]  (coalesced stores)
  float* gWr = a.gW + r * a.nw;
  for (int li = 0; li < 5; ++li)
    for (int l1 = 0; l1 < 3; ++l1)
      for (int l2 = 0; l2 < 3; ++l2) {
        const int ins = a.ins_of[li][l1][l2];
        if (ins < 0) continue;
        const int nuv = a.cnt[l1] * a.cnt[l2], d = 2 * li + 1;
        const float* rp = sR + a.roff_res[ins];
        for (int i = threadIdx.x; i < a.Cb * nuv; i += blockDim.x) {
          const int w = i / nuv, uv = i % nuv;
          float s = 0.f;
          for (int k = 0; k < d; ++k) s = fmaf(rp[uv * d + k], sX[(li * li + k) * a.Cb + w], s);
          gWr[a.woff[ins] + i] = s;
        }
      }
  // phase 3: input gradient gx[l_in,k,w] = sum_{ins with that l_in} sum_{u,v} W[w,u,v] gres[u,v,k]
  for (int i = threadIdx.x; i < QH_NCOMP * a.Cb; i += blockDim.x) {
    const int comp = i / a.Cb, w = i % a.Cb;
    const int li = comp >= 16 ? 4 : comp >= 9 ? 3 : comp >= 4 ? 2 : comp >= 1 ? 1 : 0;
    const int k = comp - li * li, d = 2 * li + 1;
    float s = 0.f;
    for (int l1 = 0; l1 < 3; ++l1)
      for (int l2 = 0; l2 < 3; ++l2) {
        const int ins = a.ins_of[li][l1][l2];
        if (ins < 0) continue;
        const int nuv = a.cnt[l1] * a.cnt[l2];
        const float* wp = sW + a.woff[ins] + w * nuv;
        const float* rp = sR + a.roff_res[ins] + k;
        for (int uv = 0; uv < nuv; ++uv) s = fmaf(wp[uv], rp[uv * d], s);
      }
    a.gx[r * QH_NCOMP * a.Cb + i] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
static int qh_exp_fill(QhExpArgs* a, const float* x, const float* W, const float* bias, int64_t R, int32_t Cb, const int32_t* shells_host, int32_t nw, int32_t nb,
                       const float* w3j, int* n_ins, int* res_total) {
  if (!x || !W || !shells_host || !w3j) return nq_fail(NQ_ERR_ARG, "null argument");
  if (Cb <= 0 || Cb > 64 || Cb % 4 != 0 || R < 0) return nq_fail(NQ_ERR_ARG, "expansion: bottleneck channels must be a multiple of 4, <= 64");
  a->x = x; a->W = W; a->bias = bias; a->R = R; a->Cb = Cb; a->nw = nw; a->nb = nb; a->w3j = w3j; a->scale = 1.0f / (float)Cb;
  int ro = 0;
  for (int l = 0; l < 3; ++l) {
    if (shells_host[l] <= 0) return nq_fail(NQ_ERR_ARG, "expansion: every shell kind (s, p, d) must occur at least once");
    a->cnt[l] = shells_host[l]; a->roff[l] = ro; ro += shells_host[l] * (2 * l + 1);
  }
  a->S = ro;
  int cs = 0;
  for (int c = 0; c < 9; ++c) { a->combo_start[c] = cs; cs += a->cnt[c / 3] * a->cnt[c % 3]; }
  a->combo_start[9] = cs;
  // instruction order of Expansion.get_expansion_path (layers.py:664-671): l_in outermost, then l1, then l2
  int ins = 0, wo = 0, bo = 0, rs = 0;
  for (int li = 0; li < 5; ++li)
    for (int l1 = 0; l1 < 3; ++l1)
      for (int l2 = 0; l2 < 3; ++l2) {
        const bool ok = li >= (l1 > l2 ? l1 - l2 : l2 - l1) && li <= l1 + l2;
        a->ins_of[li][l1][l2] = ok ? ins : -1;
        if (!ok) continue;
        if (ins >= QH_EXP_MAXINS) return nq_fail(NQ_ERR_ARG, "expansion: too many instructions");
        const int nuv = a->cnt[l1] * a->cnt[l2];
        a->woff[ins] = wo; wo += Cb * nuv;
        a->boff[ins] = li == 0 ? bo : -1;
        if (li == 0) bo += nuv;
        a->roff_res[ins] = rs; rs += nuv * (2 * li + 1);
        ++ins;
      }
  if (wo != nw || bo != nb) return nq_fail(NQ_ERR_ARG, "expansion: weight count %d / bias count %d do not match the shell layout (%d / %d)", nw, nb, wo, bo);
  *n_ins = ins; *res_total = rs;
  return NQ_OK;
}

static int qh_tp_variant = 2;   // measured (scripts/bench_qh_tp.py, profiles/r02_qh_tp_variants.txt): chunked weight prefetch without scheduling barriers
template <int SET, bool UVU, int VAR>
static void qh_tp_launch_v(const QhTpArgs& a, bool bwd, hipStream_t st) {
  const unsigned grid = (unsigned)((a.R * a.C + 255) / 256);
  if (bwd) hipLaunchKernelGGL((k_qh_tp<SET, UVU, true, VAR>), dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((k_qh_tp<SET, UVU, false, VAR>), dim3(grid), dim3(256), 0, st, a);
}
template <int SET, bool UVU>
static void qh_tp_launch(const QhTpArgs& a, bool bwd, hipStream_t st) {
  if (qh_tp_variant == 1) qh_tp_launch_v<SET, UVU, 1>(a, bwd, st);
  else if (qh_tp_variant == 2) qh_tp_launch_v<SET, UVU, 2>(a, bwd, st);
  else qh_tp_launch_v<SET, UVU, 0>(a, bwd, st);
}

static int qh_tp_dispatch(QhTpArgs& a, int path_set, bool bwd, hipStream_t st) {
  const bool uvu = a.sh != nullptr;
  if (uvu ? (path_set != 1 && path_set != 2) : path_set != 0) return nq_fail(NQ_ERR_ARG, "path set: 0 (all 65, gathered second operand), 1 (42 even) or 2 (first layer) with spherical harmonics");
  if (a.n1 != (path_set == 2 ? 1 : QH_NCOMP)) return nq_fail(NQ_ERR_ARG, "first operand must have %d components for this path set", path_set == 2 ? 1 : QH_NCOMP);
  if (a.R * a.C <= 0) return NQ_OK;
  a.np_rt = qh_path_count(path_set);
  if (!uvu) qh_tp_launch<0, false>(a, bwd, st);
  else if (path_set == 1) qh_tp_launch<1, true>(a, bwd, st);
  else qh_tp_launch<2, true>(a, bwd, st);
  return NQ_OK;
}

extern "C" {

int nq_qh_invariants_forward(const float* x, int64_t N, int32_t ncomp, int32_t C, const int32_t* own, const int32_t* col, int64_t R, int32_t second_from_owner,
                             float* s0, void* stream) {
  if (!x || !own || !col || !s0) return nq_fail(NQ_ERR_ARG, "null argument");
  if (ncomp != 1 && ncomp != 4 && ncomp != 9 && ncomp != 16 && ncomp != 25) return nq_fail(NQ_ERR_ARG, "components must be (lmax+1)^2, lmax <= 4");
  int lmax = 0; while ((lmax + 1) * (lmax + 1) < ncomp) ++lmax;
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "qh_inv_fwd");
  if (R * C > 0) hipLaunchKernelGGL(k_qh_inv_fwd, dim3((unsigned)((R * C + 255) / 256)), dim3(256), 0, st, x, own, col, (long)R, C, ncomp, lmax, second_from_owner, s0);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_qh_invariants_backward(const float* x, const float* grad_s0, int64_t N, int32_t ncomp, int32_t C, const int32_t* row_ptr, const int32_t* col,
                              const int32_t* rev, int32_t second_from_owner, float* grad_x, void* stream) {
  if (!x || !grad_s0 || !row_ptr || !col || !rev || !grad_x) return nq_fail(NQ_ERR_ARG, "null argument");
  if (ncomp != 1 && ncomp != 4 && ncomp != 9 && ncomp != 16 && ncomp != 25) return nq_fail(NQ_ERR_ARG, "components must be (lmax+1)^2, lmax <= 4");
  int lmax = 0; while ((lmax + 1) * (lmax + 1) < ncomp) ++lmax;
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "qh_inv_bwd");
  if (N * C > 0) hipLaunchKernelGGL(k_qh_inv_bwd, dim3((unsigned)((N * C + 255) / 256)), dim3(256), 0, st, x, grad_s0, row_ptr, col, rev, (int)N, C, ncomp, lmax,
                                    second_from_owner, grad_x);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

void nq_qh_set_tp_variant(int32_t v) { qh_tp_variant = v; }
int nq_qh_tp_num_paths(int32_t path_set) { return path_set == 0 ? qh_path_count(0) : path_set == 1 ? qh_path_count(1) : path_set == 2 ? qh_path_count(2) : -1; }

int nq_qh_tp_forward(const float* x, int32_t ncomp1, const int32_t* idx1, const float* sh, const int32_t* idx2, const float* w1, const float* w2, int64_t R,
                     int32_t C, int32_t path_set, float* y_rows, void* stream) {
  QhTpArgs a{};
  if (!x || !idx1 || (!sh && !idx2) || !w1 || !y_rows) return nq_fail(NQ_ERR_ARG, "null argument");
  a.x = x; a.i1 = idx1; a.i2 = idx2; a.sh = sh; a.w1 = w1; a.w2 = w2; a.y = y_rows; a.R = R; a.C = C; a.n1 = ncomp1;
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, sh ? "qh_tp_uvu_fwd" : "qh_tp_uuu_fwd");
  NQ_TRY(qh_tp_dispatch(a, path_set, false, st));
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_qh_tp_backward(const float* x, int32_t ncomp1, const int32_t* idx1, const float* sh, const int32_t* idx2, const float* w1, const float* w2,
                      const float* grad_y, const int32_t* idx_gy, int64_t R, int32_t C, int32_t path_set, float* grad_x1_rows, float* grad_x2_rows,
                      float* grad_w1, float* grad_w2, void* stream) {
  QhTpArgs a{};
  if (!x || !idx1 || (!sh && !idx2) || !w1 || !grad_y || !grad_x1_rows || (!sh && !grad_x2_rows) || !grad_w1 || (w2 && !grad_w2))
    return nq_fail(NQ_ERR_ARG, "null argument");
  a.x = x; a.i1 = idx1; a.i2 = idx2; a.sh = sh; a.w1 = w1; a.w2 = w2; a.gy = grad_y; a.ig = idx_gy; a.gx1 = grad_x1_rows; a.gx2 = grad_x2_rows;
  a.gw1 = grad_w1; a.gw2 = grad_w2; a.R = R; a.C = C; a.n1 = ncomp1;
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, sh ? "qh_tp_uvu_bwd" : "qh_tp_uuu_bwd");
  NQ_TRY(qh_tp_dispatch(a, path_set, true, st));
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_qh_pair_reduce(const float* rows_own, const float* rows_nbr, const float* base, const int32_t* row_ptr, const int32_t* rev, int64_t N, int32_t width, float* out,
                      void* stream) {
  if ((!rows_own && !rows_nbr) || !row_ptr || !rev || !out) return nq_fail(NQ_ERR_ARG, "null argument");
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "qh_pair_reduce");
  if (N * width > 0) hipLaunchKernelGGL(k_qh_pair_reduce, dim3((unsigned)((N * width + 255) / 256)), dim3(256), 0, st, rows_own, rows_nbr, base, row_ptr, rev, (int)N, width, out);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_qh_normcat(const float* x, const float* grad_f0, int64_t rows, int32_t C, int32_t lmax, float* out, void* stream) {
  if (!x || !out || lmax < 0 || lmax > 4) return nq_fail(NQ_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, grad_f0 ? "qh_normcat_bwd" : "qh_normcat_fwd");
  if (rows * C > 0) hipLaunchKernelGGL(k_qh_normcat, dim3((unsigned)((rows * C + 255) / 256)), dim3(256), 0, st, x, grad_f0, (long)rows, C, lmax, out, grad_f0 ? 1 : 0);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_qh_gate(const float* x, const float* gates, const float* grad_y, int64_t rows, int32_t C, int32_t lmax, float* y, float* grad_x, float* grad_gates, void* stream) {
  if (!x || !gates || lmax < 0 || lmax > 4) return nq_fail(NQ_ERR_ARG, "bad argument");
  if (grad_y ? (!grad_x || !grad_gates) : !y) return nq_fail(NQ_ERR_ARG, "null output");
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, grad_y ? "qh_gate_bwd" : "qh_gate_fwd");
  if (rows * C > 0) hipLaunchKernelGGL(k_qh_gate, dim3((unsigned)((rows * C + 255) / 256)), dim3(256), 0, st, x, gates, grad_y, (long)rows, C, lmax, y, grad_x, grad_gates);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_qh_act(const float* x, const float* grad_y, int32_t kind, float cst, int64_t count, float* out, void* stream) {
  if (!x || !out || kind < 0 || kind > 1) return nq_fail(NQ_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "qh_act");
  if (count > 0) hipLaunchKernelGGL(k_qh_act, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, x, grad_y, kind, cst, (long)count, out);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_qh_expansion_forward(const float* x, const float* weights, const float* bias, int64_t R, int32_t Cb, const int32_t* shells_host, int32_t n_weights,
                            int32_t n_bias, const float* w3j, float* out, void* stream) {
  QhExpArgs a{};
  int n_ins, res_total;
  NQ_TRY(qh_exp_fill(&a, x, weights, bias, R, Cb, shells_host, n_weights, n_bias, w3j, &n_ins, &res_total));
  if (!out) return nq_fail(NQ_ERR_ARG, "null argument");
  a.out = out;
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "qh_exp_fwd");
  const size_t lds = sizeof(float) * (((size_t)n_weights + 3) / 4 * 4 + QH_NCOMP * Cb + n_bias + 4);
  if (lds > 160 * 1024) return nq_fail(NQ_ERR_ARG, "expansion: weight row does not fit the LDS");
  if (lds > 64 * 1024) NQ_HIP(hipFuncSetAttribute((const void*)k_qh_exp_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (R > 0) hipLaunchKernelGGL(k_qh_exp_fwd, dim3((unsigned)R), dim3(256), lds, st, a);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_qh_expansion_backward(const float* x, const float* weights, const float* grad_out, int64_t R, int32_t Cb, const int32_t* shells_host, int32_t n_weights,
                             int32_t n_bias, const float* w3j, float* grad_x, float* grad_weights, float* grad_bias, void* stream) {
  QhExpArgs a{};
  int n_ins, res_total;
  NQ_TRY(qh_exp_fill(&a, x, weights, nullptr, R, Cb, shells_host, n_weights, n_bias, w3j, &n_ins, &res_total));
  if (!grad_out || !grad_x || !grad_weights) return nq_fail(NQ_ERR_ARG, "null argument");
  a.gout = grad_out; a.gx = grad_x; a.gW = grad_weights; a.gbias = grad_bias;
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "qh_exp_bwd");
  const size_t lds = sizeof(float) * (((size_t)n_weights + 3) / 4 * 4 + QH_NCOMP * Cb + (size_t)a.S * a.S + res_total + 4);
  if (lds > 160 * 1024) return nq_fail(NQ_ERR_ARG, "expansion: weight row does not fit the LDS");
  if (lds > 64 * 1024) NQ_HIP(hipFuncSetAttribute((const void*)k_qh_exp_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (R > 0) hipLaunchKernelGGL(k_qh_exp_bwd, dim3((unsigned)R), dim3(256), lds, st, a, n_ins, res_total);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

}  // extern "C"
