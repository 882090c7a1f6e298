// Node-level (per atom x channel) kernels of the PaiNN path: update block (painn.py:535-548),
// SiLU derivative steps, readout (painn.py:79-83,127-128), embedding (layers.py:215-222), loss
// (painn.py:741-745 with L1Loss + gemnet_oc/loss.py:5-22), optimizer.  Math: oracle/painn_sweeps.py.
// Thread mapping everywhere: channel index fastest -> fully coalesced rows of F floats.
#include "common.h"

// ---------------------------------------------------------------------------------------------

// Vector width of the update-block kernels: every thread owns VW consecutive channels of one atom (VW = 1 shipped, VW = 4 measured slower, see NQ_NODE_V below;
// F is a multiple of 64).  The arithmetic per channel does not depend on VW.
typedef float nq_f4 __attribute__((ext_vector_type(4)));
template <typename V> struct NqVW;
template <> struct NqVW<float> { static constexpr int w = 1; };
template <> struct NqVW<nq_f4> { static constexpr int w = 4; };
template <typename V> __device__ __forceinline__ V nq_ld(const float* p) { return *reinterpret_cast<const V*>(p); }
template <typename V> __device__ __forceinline__ void nq_st(float* p, V v) { *reinterpret_cast<V*>(p) = v; }
__device__ __forceinline__ float nq_vsqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ nq_f4 nq_vsqrt(nq_f4 x) { return nq_f4{sqrtf(x[0]), sqrtf(x[1]), sqrtf(x[2]), sqrtf(x[3])}; }
#define NQ_NODE_INDEX                                                            \
  constexpr int VW = NqVW<V>::w;                                                 \
  const int F = q.F, FV = F / VW;                                                \
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;                  \
  if (idx >= (long)q.N * FV) return;                                             \
  const long n = idx / FV; const int f = (int)(idx % FV) * VW;                   \
  const long nf = n * F + f;

// s = sum_c v1 v2 ; n = sqrt(sum_c v2^2 + 1e-8) ; cat = [x_msg | n]
template <bool TAN, typename V>
__global__ void k_upd_a(UpdArgs q) {
  NQ_NODE_INDEX
  const float* u = q.U + n * 6 * F;
  V a[3], b[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { a[c] = nq_ld<V>(u + c * 2 * F + f); b[c] = nq_ld<V>(u + c * 2 * F + F + f); }
  if (!TAN) {
    const V s = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
    const V nn = nq_vsqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + 1e-8f);
    nq_st<V>(q.S + nf, s);
    nq_st<V>(q.CAT + n * 2 * F + f, nq_ld<V>(q.XM + nf));
    nq_st<V>(q.CAT + n * 2 * F + F + f, nn);
  } else {
    const float* tu = q.TU + n * 6 * F;
    V ta[3], tb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { ta[c] = nq_ld<V>(tu + c * 2 * F + f); tb[c] = nq_ld<V>(tu + c * 2 * F + F + f); }
    const V nn = nq_ld<V>(q.CAT + n * 2 * F + F + f);
    const V ts = ta[0] * b[0] + a[0] * tb[0] + ta[1] * b[1] + a[1] * tb[1] + ta[2] * b[2] + a[2] * tb[2];
    const V tn = (b[0] * tb[0] + b[1] * tb[1] + b[2] * tb[2]) / nn;
    nq_st<V>(q.TS + nf, ts);
    nq_st<V>(q.TCAT + n * 2 * F + f, nq_ld<V>(q.TXM + nf));
    nq_st<V>(q.TCAT + n * 2 * F + F + f, tn);
  }
}

// x_upd = x_msg + y_a + y_b s ; vec_upd[c] = vec_msg[c] + y_c v1[c]
template <bool TAN, typename V>
__global__ void k_upd_b(UpdArgs q) {
  NQ_NODE_INDEX
  const float* u = q.U + n * 6 * F;
  const float* y = q.Y + n * 3 * F;
  const V s = nq_ld<V>(q.S + nf);
  if (!TAN) {
    nq_st<V>(q.X1 + nf, nq_ld<V>(q.XM + nf) + nq_ld<V>(y + f) + nq_ld<V>(y + F + f) * s);
#pragma unroll
    for (int c = 0; c < 3; ++c)
      nq_st<V>(q.V1 + n * 3 * F + c * F + f, nq_ld<V>(q.VM + n * 3 * F + c * F + f) + nq_ld<V>(y + 2 * F + f) * nq_ld<V>(u + c * 2 * F + f));
  } else {
    const float* tu = q.TU + n * 6 * F;
    const float* ty = q.TY + n * 3 * F;
    nq_st<V>(q.TX1 + nf, nq_ld<V>(q.TXM + nf) + nq_ld<V>(ty + f) + nq_ld<V>(ty + F + f) * s + nq_ld<V>(y + F + f) * nq_ld<V>(q.TS + nf));
#pragma unroll
    for (int c = 0; c < 3; ++c)
      nq_st<V>(q.TV1 + n * 3 * F + c * F + f, nq_ld<V>(q.TVM + n * 3 * F + c * F + f) + nq_ld<V>(ty + 2 * F + f) * nq_ld<V>(u + c * 2 * F + f) +
                                                   nq_ld<V>(y + 2 * F + f) * nq_ld<V>(tu + c * 2 * F + f));
  }
}

// tangent of SiLU: TH = dsilu(Z) * TZ
__global__ void k_silu_tan(const float* __restrict__ Z, const float* __restrict__ TZ, float* __restrict__ TH, long count) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) TH[i] = nq_dsilu(Z[i]) * TZ[i];
}

// reverse of SiLU, in place on the adjoint(s):  G <- G dsilu(Z) (+ GT d2silu(Z) TZ) ;  GT <- GT dsilu(Z).  Four consecutive elements per thread (count % 4 == 0: N x F)
// LITE (dual): GT holds the tangent adjoint of the layer's OUTPUT (from the force sweep's store) and is only read; its pre-activation form is in the store already
template <bool DUAL, bool LITE = false>
__global__ void k_silu_rev(const float* __restrict__ Z, const float* __restrict__ TZ, float* __restrict__ G, float* __restrict__ GT, long count) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= count) return;
  const nq_f4 z = nq_ld<nq_f4>(Z + i);
  nq_f4 g = nq_ld<nq_f4>(G + i), gt = g, tz = g;
  if (DUAL) { gt = nq_ld<nq_f4>(GT + i); tz = nq_ld<nq_f4>(TZ + i); }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float d1 = nq_dsilu(z[c]);
    float gc = g[c] * d1;
    if (DUAL) {
      gc += gt[c] * nq_d2silu(z[c]) * tz[c];
      gt[c] = gt[c] * d1;
    }
    g[c] = gc;
  }
  nq_st<nq_f4>(G + i, g);
  if (DUAL && !LITE) nq_st<nq_f4>(GT + i, gt);
}

// ---------------------------------------------------------------------------------------------

// rev1: adjoints of y = (ya, yb, yc) from x_upd = x_msg + ya + yb s, vec_upd = vec_msg + yc v1
template <bool DUAL, typename V>
__global__ void k_upd_rev1(UpdRevArgs q) {
  NQ_NODE_INDEX
  const float* u = q.U + n * 6 * F;
  const V gx = nq_ld<V>(q.GX + nf), s = nq_ld<V>(q.S + nf);
  const float* gv = q.GV + n * 3 * F;
  const V gv0 = nq_ld<V>(gv + f), gv1 = nq_ld<V>(gv + F + f), gv2 = nq_ld<V>(gv + 2 * F + f);
  const V u0 = nq_ld<V>(u + f), u1 = nq_ld<V>(u + 2 * F + f), u2 = nq_ld<V>(u + 4 * F + f);
  V gyb = gx * s;
  V gyc = gv0 * u0 + gv1 * u1 + gv2 * u2;
  if (DUAL) {
    const float* tu = q.TU + n * 6 * F;
    const V gtx = nq_ld<V>(q.GTX + nf);
    const float* gtv = q.GTV + n * 3 * F;
    const V gtv0 = nq_ld<V>(gtv + f), gtv1 = nq_ld<V>(gtv + F + f), gtv2 = nq_ld<V>(gtv + 2 * F + f);
    gyb += gtx * nq_ld<V>(q.TS + nf);
    gyc += gtv0 * nq_ld<V>(tu + f) + gtv1 * nq_ld<V>(tu + 2 * F + f) + gtv2 * nq_ld<V>(tu + 4 * F + f);
    if (!q.lite) {
      float* gty = q.GTY + n * 3 * F;
      nq_st<V>(gty + f, gtx);
      nq_st<V>(gty + F + f, gtx * s);
      nq_st<V>(gty + 2 * F + f, gtv0 * u0 + gtv1 * u1 + gtv2 * u2);
    }
  }
  float* gy = q.GY + n * 3 * F;
  nq_st<V>(gy + f, gx); nq_st<V>(gy + F + f, gyb); nq_st<V>(gy + 2 * F + f, gyc);
}

// rev2: given gcat = adjoint of [x_msg | n]: adjoints of u = (v1, v2) and gx_msg = gx_upd + gcat[:F]
template <bool DUAL, typename V>
__global__ void k_upd_rev2(UpdRevArgs q) {
  NQ_NODE_INDEX
  const float* u = q.U + n * 6 * F;
  const float* y = q.Y + n * 3 * F;
  const V yb = nq_ld<V>(y + F + f), yc = nq_ld<V>(y + 2 * F + f);
  const V nn = nq_ld<V>(q.CAT + n * 2 * F + F + f);
  const V gx = nq_ld<V>(q.GX + nf);
  const float* gv = q.GV + n * 3 * F;
  const V gn = nq_ld<V>(q.GCAT + n * 2 * F + F + f);
  V gs = gx * yb;
  V gts = gx * 0.f, gtn = gts, tn = gts, tyc = gts, gtx = gts;
  const float* tu = nullptr; const float* gtv = nullptr;
  if (DUAL) {
    tu = q.TU + n * 6 * F;
    gtv = q.GTV + n * 3 * F;
    gtx = nq_ld<V>(q.GTX + nf);
    const float* ty = q.TY + n * 3 * F;
    gs += gtx * nq_ld<V>(ty + F + f);
    gts = gtx * yb;
    tyc = nq_ld<V>(ty + 2 * F + f);
    gtn = nq_ld<V>(q.GTCAT + n * 2 * F + F + f);
    tn = nq_ld<V>(q.TCAT + n * 2 * F + F + f);
  }
  const V gn_n = gn / nn, gtn_n = gtn / nn, tn_n = tn / nn;
  float* gu = q.GU + n * 6 * F;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const V a = nq_ld<V>(u + c * 2 * F + f), b = nq_ld<V>(u + c * 2 * F + F + f);
    V g1 = nq_ld<V>(gv + c * F + f) * yc + gs * b;
    V g2 = gn_n * b + gs * a;
    if (DUAL) {
      const V ta = nq_ld<V>(tu + c * 2 * F + f), tb = nq_ld<V>(tu + c * 2 * F + F + f);
      const V gtvc = nq_ld<V>(gtv + c * F + f);
      g1 += gtvc * tyc + gts * tb;
      g2 += gtn_n * (tb - tn_n * b) + gts * ta;
      if (!q.lite) {
        float* gtu = q.GTU + n * 6 * F;
        nq_st<V>(gtu + c * 2 * F + f, gtvc * yc + gts * b);
        nq_st<V>(gtu + c * 2 * F + F + f, gtn_n * b + gts * a);
      }
    }
    nq_st<V>(gu + c * 2 * F + f, g1);
    nq_st<V>(gu + c * 2 * F + F + f, g2);
  }
  nq_st<V>((q.GX_out ? q.GX_out : q.GX) + nf, gx + nq_ld<V>(q.GCAT + n * 2 * F + f));
  if (DUAL && !q.lite) nq_st<V>(q.GTX + nf, gtx + nq_ld<V>(q.GTCAT + n * 2 * F + f));
}

// ---------------------------------------------------------------------------------------------
// embedding gather: X0[n][f] = Emb[z[n]-1][f]
__global__ void k_embed(const int* __restrict__ z, const float* __restrict__ emb, int N, int F, float* __restrict__ X0) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)N * F) return;
  const long n = idx / F; const int f = (int)(idx % F);
  X0[idx] = emb[(long)(z[n] - 1) * F + f];
}

// embedding gradient, deterministic: one workgroup per chunk of atoms, thread f owns column f of an LDS
// accumulator [T][F]; partial slabs are then reduced in fixed order by k_reduce_partials.
__global__ void k_embed_grad_partial(const int* __restrict__ z, const float* __restrict__ GX, int N, int F, int T, int chunk,
                                     float* __restrict__ part) {
  extern __shared__ float acc[];
  const int f = threadIdx.x;
  for (int t = 0; t < T; ++t) acc[t * F + f] = 0.f;
  const int n0 = blockIdx.x * chunk, n1 = min(N, n0 + chunk);
  for (int n = n0; n < n1; ++n) acc[(z[n] - 1) * F + f] += GX[(long)n * F + f];
  for (int t = 0; t < T; ++t) part[((long)blockIdx.x * T + t) * F + f] = acc[t * F + f];
}

// readout: e[n] = w2 . silu(zo[n]) + o2 (one wavefront per atom), tangent, and reverse

template <int MODE>  // 0 fwd, 1 tangent
__global__ void k_readout(ReadoutArgs q) {
  const int lane = threadIdx.x & 63;
  const long n = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (n >= q.N) return;
  float acc = 0.f;
  for (int h = lane; h < q.H; h += 64) {
    const float z = q.ZO[n * q.H + h];
    acc += q.w2[h] * (MODE == 0 ? nq_silu(z) : nq_dsilu(z) * q.TZO[n * q.H + h]);
  }
  acc = nq_wave_sum(acc);
  if (lane == 0) {
    if (MODE == 0) q.e_atom[n] = acc + q.o2[0];
    else q.te_atom[n] = acc;
  }
}

template <bool DUAL>
__global__ void k_readout_rev(ReadoutArgs q) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)q.N * q.H) return;
  const long n = idx / q.H; const int h = (int)(idx % q.H);
  const float z = q.ZO[idx], w = q.w2[h];
  const float d1 = nq_dsilu(z);
  const float ge = q.ge[n];
  float g = ge * w * d1;
  if (DUAL) {
    const float gte = q.gte[n], tz = q.TZO[idx];
    g += gte * w * nq_d2silu(z) * tz;
    q.GTZO[idx] = gte * w * d1;
    q.TMPW[idx] = ge * nq_silu(z) + gte * d1 * tz;
  }
  q.GZO[idx] = g;
}

// energy[b] = sum of e_atom over the molecule's atoms, sequential order (== index_add order)
__global__ void k_mol_sum(const float* __restrict__ e_atom, const int* __restrict__ mol_ptr, int B, float* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s = 0.f;
  for (int n = mol_ptr[b]; n < mol_ptr[b + 1]; ++n) s += e_atom[n];
  out[b] = s;
}

// per-atom seeds for the reverse sweeps: ge[n] = gE[mol(n)] (or 1), gte[n] = 1
__global__ void k_atom_seeds(const float* __restrict__ gE, const int* __restrict__ atom_mol, int N, float* __restrict__ ge, float* __restrict__ gte) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  ge[n] = gE ? gE[atom_mol[n]] : 1.0f;
  if (gte) gte[n] = 1.0f;
}

__global__ void k_scale_neg(const float* __restrict__ in, float* __restrict__ out, long count) {  // out = -in
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = -in[i];
}

// ---------------------------------------------------------------------------------------------
// MSE = false: loss = ce * mean_b |E_b - y_b| + cf * mean_i ||F_i - Ft_i||_2                       (painn.py:741-745)
// MSE = true : loss = ce * mean_b (E_b - y_b)^2 + cf * mean_{i,c} (F_ic - Ft_ic)^2  (torch.nn.MSELoss, config/model/painn.yaml:30-46)
// and the seeds dL/dE, dL/dF   (single workgroup)
template <bool MSE>
__global__ __launch_bounds__(1024) void k_loss(const float* __restrict__ E, const float* __restrict__ y, int B, const float* __restrict__ Fc,
                                               const float* __restrict__ Ft, int N, float ce, float cf, float* __restrict__ loss,
                                               float* __restrict__ gE, float* __restrict__ gF) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float d = E[b] - y[b];
    if (MSE) {
      acc += ce * d * d / (float)B;
      gE[b] = 2.f * ce * d / (float)B;
    } else {
      acc += ce * fabsf(d) / (float)B;
      gE[b] = ce * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / (float)B;
    }
  }
#pragma unroll 4   // (one workgroup: the trips are a serial chain of load latencies unless several are in flight)
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    const float dx = Fc[3 * (long)n] - Ft[3 * (long)n], dy = Fc[3 * (long)n + 1] - Ft[3 * (long)n + 1], dz = Fc[3 * (long)n + 2] - Ft[3 * (long)n + 2];
    float sc;
    if (MSE) {
      acc += cf * (dx * dx + dy * dy + dz * dz) / (3.f * (float)N);
      sc = 2.f * cf / (3.f * (float)N);
    } else {
      const float nr = sqrtf(dx * dx + dy * dy + dz * dz);
      acc += cf * nr / (float)N;
      sc = nr > 0.f ? cf / (nr * (float)N) : 0.f;
    }
    gF[3 * (long)n] = dx * sc; gF[3 * (long)n + 1] = dy * sc; gF[3 * (long)n + 2] = dz * sc;
  }
  acc = nq_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    loss[0] = s;
  }
}

// ---- optimizer: squared grad norm (two stage), clip + AdamW in one pass ----------------------
__global__ void k_sqnorm_partial(const float* __restrict__ g, long count, float* __restrict__ part) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) acc += g[i] * g[i];
  acc = nq_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void k_sqnorm_final(const float* __restrict__ part, int n, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += part[i];
    out[0] = s;
  }
}
// torch.optim.AdamW semantics (decoupled weight decay) with clip_grad_norm_(max_norm) folded in:
//   g <- g * min(1, max_norm / (||g|| + 1e-6))
__global__ void k_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long count,
                        const float* __restrict__ sqnorm, float max_norm, float lr, float beta1, float beta2, float eps, float wd,
                        float bc1, float bc2) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float scale = 1.0f;
  if (max_norm > 0.f) {
    const float nrm = sqrtf(sqnorm[0]);
    scale = fminf(1.0f, max_norm / (nrm + 1e-6f));
  }
  const float gi = g[i] * scale;
  float pi = p[i] * (1.0f - lr * wd);
  const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
  const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
  m[i] = mi; v[i] = vi;
  const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
  p[i] = pi - (lr / bc1) * mi / denom;
}

// ---- host launchers ------------------------------------------------------------------------
static inline dim3 grid1d(long count, int block) { return dim3(nq_cdiv(count, block)); }

// measured (round 5, B = 2048): four channels per thread upd_rev 5.16 / upd_b 1.85 / upd_a 1.40 ms per step vs 4.98 / 1.79 / 1.36 with one channel per thread -- the
// scalar form keeps four times the wavefronts in flight; only k_silu_rev (two tensors in, two out) gains from 16-byte accesses (0.57 -> 0.50 ms).  -DNQ_NODE_VEC4 selects the vector form.
#ifdef NQ_NODE_VEC4
#define NQ_NODE_V nq_f4
#else
#define NQ_NODE_V float
#endif
int nq_upd_a(hipStream_t st, const UpdArgs& q, bool tan) {
  NQ_PROF(st, "upd_a");
  if (q.N <= 0) return NQ_OK;
  if (tan) hipLaunchKernelGGL((k_upd_a<true, NQ_NODE_V>), grid1d((long)q.N * q.F / NqVW<NQ_NODE_V>::w, 256), dim3(256), 0, st, q);
  else hipLaunchKernelGGL((k_upd_a<false, NQ_NODE_V>), grid1d((long)q.N * q.F / NqVW<NQ_NODE_V>::w, 256), dim3(256), 0, st, q);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_upd_b(hipStream_t st, const UpdArgs& q, bool tan) {
  NQ_PROF(st, "upd_b");
  if (q.N <= 0) return NQ_OK;
  if (tan) hipLaunchKernelGGL((k_upd_b<true, NQ_NODE_V>), grid1d((long)q.N * q.F / NqVW<NQ_NODE_V>::w, 256), dim3(256), 0, st, q);
  else hipLaunchKernelGGL((k_upd_b<false, NQ_NODE_V>), grid1d((long)q.N * q.F / NqVW<NQ_NODE_V>::w, 256), dim3(256), 0, st, q);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_silu_tan(hipStream_t st, const float* Z, const float* TZ, float* TH, long count) {
  NQ_PROF(st, "silu_tan");
  if (count <= 0) return NQ_OK;
  hipLaunchKernelGGL(k_silu_tan, grid1d(count, 256), dim3(256), 0, st, Z, TZ, TH, count);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_silu_rev(hipStream_t st, const float* Z, const float* TZ, float* G, float* GT, long count, bool dual, bool lite) {
  NQ_PROF(st, "silu_rev");
  if (count <= 0) return NQ_OK;
  if (count & 3) return nq_fail(NQ_ERR_ARG, "silu_rev: element count %ld is not a multiple of 4", count);
  if (dual && lite) hipLaunchKernelGGL((k_silu_rev<true, true>), grid1d(count / 4, 256), dim3(256), 0, st, Z, TZ, G, GT, count);
  else if (dual) hipLaunchKernelGGL((k_silu_rev<true>), grid1d(count / 4, 256), dim3(256), 0, st, Z, TZ, G, GT, count);
  else hipLaunchKernelGGL((k_silu_rev<false>), grid1d(count / 4, 256), dim3(256), 0, st, Z, TZ, G, GT, count);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_upd_rev(hipStream_t st, const UpdRevArgs& q, int stage, bool dual) {
  NQ_PROF(st, "upd_rev");
  if (q.N <= 0) return NQ_OK;
  dim3 g = grid1d((long)q.N * q.F / NqVW<NQ_NODE_V>::w, 256), b(256);
  if (stage == 1) {
    if (dual) hipLaunchKernelGGL((k_upd_rev1<true, NQ_NODE_V>), g, b, 0, st, q);
    else hipLaunchKernelGGL((k_upd_rev1<false, NQ_NODE_V>), g, b, 0, st, q);
  } else {
    if (dual) hipLaunchKernelGGL((k_upd_rev2<true, NQ_NODE_V>), g, b, 0, st, q);
    else hipLaunchKernelGGL((k_upd_rev2<false, NQ_NODE_V>), g, b, 0, st, q);
  }
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_embed(hipStream_t st, const int* z, const float* emb, int N, int F, float* X0) {
  NQ_PROF(st, "embed");
  hipLaunchKernelGGL(k_embed, grid1d((long)N * F, 256), dim3(256), 0, st, z, emb, N, F, X0);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
#define EMB_CHUNK 512
// atoms per workgroup: 512 for large batches; small batches get more, shorter chunks (the per-atom LDS update chain is serial)
static int emb_chunk(int N) { const int c = nq_cdiv(N, 256); return c < 32 ? 32 : (c > EMB_CHUNK ? EMB_CHUNK : c); }
size_t nq_embed_grad_scratch_floats(int N, int F, int T) { return (size_t)nq_cdiv(N, emb_chunk(N)) * T * F; }
int nq_embed_grad(hipStream_t st, const int* z, const float* GX, int N, int F, int T, float* out, float* scratch) {
  NQ_PROF(st, "embed_grad");
  const int chunk = emb_chunk(N), chunks = nq_cdiv(N, chunk);
  const size_t lds = (size_t)T * F * sizeof(float);
  if (lds > 160 * 1024) return nq_fail(NQ_ERR_ARG, "embed_grad: num_elements*F too large for LDS (%zu B)", lds);
  NQ_DYN_LDS(k_embed_grad_partial, lds);
  hipLaunchKernelGGL(k_embed_grad_partial, dim3(chunks), dim3(F), lds, st, z, GX, N, F, T, chunk, scratch);
  NQ_LAUNCH_CHECK();
  return nq_reduce_partials(st, scratch, chunks, (long)T * F, (long)T * F, out);
}
int nq_readout(hipStream_t st, const ReadoutArgs& q, int mode) {
  NQ_PROF(st, "readout");
  if (q.N <= 0) return NQ_OK;
  if (mode == 0) hipLaunchKernelGGL((k_readout<0>), dim3(nq_cdiv(q.N, 4)), dim3(256), 0, st, q);
  else hipLaunchKernelGGL((k_readout<1>), dim3(nq_cdiv(q.N, 4)), dim3(256), 0, st, q);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_readout_rev(hipStream_t st, const ReadoutArgs& q, bool dual) {
  NQ_PROF(st, "readout_rev");
  if (q.N <= 0) return NQ_OK;
  if (dual) hipLaunchKernelGGL((k_readout_rev<true>), grid1d((long)q.N * q.H, 256), dim3(256), 0, st, q);
  else hipLaunchKernelGGL((k_readout_rev<false>), grid1d((long)q.N * q.H, 256), dim3(256), 0, st, q);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_mol_sum(hipStream_t st, const float* e_atom, const int* mol_ptr, int B, float* out) {
  hipLaunchKernelGGL(k_mol_sum, grid1d(B, 128), dim3(128), 0, st, e_atom, mol_ptr, B, out);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_atom_seeds(hipStream_t st, const float* gE, const int* atom_mol, int N, float* ge, float* gte) {
  hipLaunchKernelGGL(k_atom_seeds, grid1d(N, 256), dim3(256), 0, st, gE, atom_mol, N, ge, gte);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
__global__ void k_axpy(const float* __restrict__ x, float* __restrict__ y, long count) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) y[i] += x[i];
}
int nq_axpy(hipStream_t st, const float* x, float* y, long count) {
  hipLaunchKernelGGL(k_axpy, grid1d(count, 256), dim3(256), 0, st, x, y, count);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_negate(hipStream_t st, const float* in, float* out, long count) {
  hipLaunchKernelGGL(k_scale_neg, grid1d(count, 256), dim3(256), 0, st, in, out, count);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_loss_impl(hipStream_t st, const float* E, const float* y, int B, const float* Fc, const float* Ft, int N, float ce, float cf,
                 float* loss, float* gE, float* gF, bool mse) {
  NQ_PROF(st, "loss");
  if (mse) hipLaunchKernelGGL(k_loss<true>, dim3(1), dim3(1024), 0, st, E, y, B, Fc, Ft, N, ce, cf, loss, gE, gF);
  else hipLaunchKernelGGL(k_loss<false>, dim3(1), dim3(1024), 0, st, E, y, B, Fc, Ft, N, ce, cf, loss, gE, gF);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
#define SQN_BLOCKS 256
int nq_adamw_impl(hipStream_t st, float* p, const float* g, float* m, float* v, long count, float max_norm, float lr, float beta1,
                  float beta2, float eps, float wd, int step, float* scratch /* >= SQN_BLOCKS+1 floats */) {
  NQ_PROF(st, "adamw");
  hipLaunchKernelGGL(k_sqnorm_partial, dim3(SQN_BLOCKS), dim3(256), 0, st, g, count, scratch);
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sqnorm_final, dim3(1), dim3(64), 0, st, scratch, SQN_BLOCKS, scratch + SQN_BLOCKS);
  NQ_LAUNCH_CHECK();
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  hipLaunchKernelGGL(k_adamw, grid1d(count, 256), dim3(256), 0, st, p, g, m, v, count, scratch + SQN_BLOCKS, max_norm, lr, beta1,
                     beta2, eps, wd, bc1, bc2);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

// ---- GatedEquivariantBlock pieces of the direct-force head (painn.py:583-620); GEMMs in between via nq_linear_* ---------------------------
// ScaledSiLU(x) = silu(x) / 0.6 (layers.py:188-195)
__global__ void k_scaled_silu(const float* __restrict__ z, const float* __restrict__ gy, float* __restrict__ out, long count, int bwd) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float s = 1.0f / 0.6f;
  out[i] = bwd ? gy[i] * nq_dsilu(z[i]) * s : nq_silu(z[i]) * s;
}
// cat[n] = [x[n] | ||v1[n]||_xyz]     v1: [N][3][h]
__global__ void k_geb_cat(const float* __restrict__ x, const float* __restrict__ v1, long N, int h, float* __restrict__ cat) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * h) return;
  const long n = i / h; const int c = (int)(i % h);
  const float a = v1[(n * 3) * h + c], b = v1[(n * 3 + 1) * h + c], d = v1[(n * 3 + 2) * h + c];
  cat[n * 2 * h + c] = x[i];
  cat[n * 2 * h + h + c] = sqrtf(a * a + b * b + d * d);
}
__global__ void k_geb_cat_rev(const float* __restrict__ gcat, const float* __restrict__ v1, long N, int h, float* __restrict__ gx, float* __restrict__ gv1) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * h) return;
  const long n = i / h; const int c = (int)(i % h);
  gx[i] = gcat[n * 2 * h + c];
  const float a = v1[(n * 3) * h + c], b = v1[(n * 3 + 1) * h + c], d = v1[(n * 3 + 2) * h + c];
  const float nr = sqrtf(a * a + b * b + d * d);
  const float sc = nr > 0.f ? gcat[n * 2 * h + h + c] / nr : 0.f;
  gv1[(n * 3) * h + c] = sc * a; gv1[(n * 3 + 1) * h + c] = sc * b; gv1[(n * 3 + 2) * h + c] = sc * d;
}
// (xo | gate) = split(o2);  xout = ScaledSiLU(xo);  vout[n][xyz][j] = gate[n][j] * v2[n][xyz][j]
__global__ void k_geb_gate(const float* __restrict__ o2, const float* __restrict__ v2, long N, int o, float* __restrict__ xout, float* __restrict__ vout) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * o) return;
  const long n = i / o; const int j = (int)(i % o);
  xout[i] = nq_silu(o2[n * 2 * o + j]) * (1.0f / 0.6f);
  const float gate = o2[n * 2 * o + o + j];
  for (int c = 0; c < 3; ++c) vout[(n * 3 + c) * o + j] = gate * v2[(n * 3 + c) * o + j];
}
__global__ void k_geb_gate_rev(const float* __restrict__ o2, const float* __restrict__ v2, const float* __restrict__ gxout, const float* __restrict__ gvout,
                               long N, int o, float* __restrict__ go2, float* __restrict__ gv2) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * o) return;
  const long n = i / o; const int j = (int)(i % o);
  go2[n * 2 * o + j] = gxout[i] * nq_dsilu(o2[n * 2 * o + j]) * (1.0f / 0.6f);
  const float gate = o2[n * 2 * o + o + j];
  float gg = 0.f;
  for (int c = 0; c < 3; ++c) {
    const float gv = gvout[(n * 3 + c) * o + j];
    gg += gv * v2[(n * 3 + c) * o + j];
    gv2[(n * 3 + c) * o + j] = gv * gate;
  }
  go2[n * 2 * o + o + j] = gg;
}

extern "C" {
int nq_scaled_silu(const float* z, const float* grad_y, float* out, int64_t count, void* stream) {
  if (!z || !out || count < 0) return nq_fail(NQ_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (count > 0) hipLaunchKernelGGL(k_scaled_silu, grid1d(count, 256), dim3(256), 0, st, z, grad_y, out, (long)count, grad_y ? 1 : 0);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_geb_cat(const float* x, const float* v1, int64_t N, int32_t h, float* cat, void* stream) {
  if (!x || !v1 || !cat || h <= 0 || N < 0) return nq_fail(NQ_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (N > 0) hipLaunchKernelGGL(k_geb_cat, grid1d(N * h, 256), dim3(256), 0, st, x, v1, (long)N, h, cat);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_geb_cat_backward(const float* grad_cat, const float* v1, int64_t N, int32_t h, float* grad_x, float* grad_v1, void* stream) {
  if (!grad_cat || !v1 || !grad_x || !grad_v1 || h <= 0 || N < 0) return nq_fail(NQ_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (N > 0) hipLaunchKernelGGL(k_geb_cat_rev, grid1d(N * h, 256), dim3(256), 0, st, grad_cat, v1, (long)N, h, grad_x, grad_v1);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_geb_gate(const float* o2, const float* v2, int64_t N, int32_t o, float* xout, float* vout, void* stream) {
  if (!o2 || !v2 || !xout || !vout || o <= 0 || N < 0) return nq_fail(NQ_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (N > 0) hipLaunchKernelGGL(k_geb_gate, grid1d(N * o, 256), dim3(256), 0, st, o2, v2, (long)N, o, xout, vout);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_geb_gate_backward(const float* o2, const float* v2, const float* grad_xout, const float* grad_vout, int64_t N, int32_t o, float* grad_o2,
                         float* grad_v2, void* stream) {
  if (!o2 || !v2 || !grad_xout || !grad_vout || !grad_o2 || !grad_v2 || o <= 0 || N < 0) return nq_fail(NQ_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (N > 0) hipLaunchKernelGGL(k_geb_gate_rev, grid1d(N * o, 256), dim3(256), 0, st, o2, v2, grad_xout, grad_vout, (long)N, o, grad_o2, grad_v2);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
}  // extern "C"
