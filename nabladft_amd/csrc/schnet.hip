// SchNet (schnetpack-style, config/model/schnet.yaml) energy/force training step on the shared engine pieces:
// neighbour list (graph.hip), 13-tap Gaussian window record + k0-sorted first-layer weight gradient (edge.hip), MFMA GEMMs
// (gemm.hip), readout / loss / AdamW (node.hip).  This file adds the continuous-filter convolution kernels and the four sweeps
// (forward, force adjoint, tangent, dual reverse) stated in oracle/spk_schnet_ref.py:SchNetSweeps -- same buffer names.
//
//   filter network   z1 = W1 g(d) + b1   (windowed, LDS-resident W1^T)   a1 = ssp(z1)   h2 = W2 a1 + b2   (MFMA GEMM over edges)
//   cfconv           m_i = sum_{e=(i<-j)} y_j * h2_e * fcut(d_e),  y = in2f(x)
//   f2out            x += W_o2 ssp(W_o1 m + b_o1) + b_o2
//
// Every per-edge filter quantity depends on d_e only, so it is identical on an edge and on its reverse edge:
//   * the filter network (window layer, GEMMs, weight gradients) runs once per undirected PAIR p (P = E/2 rows): pair (i <- j, j < i) is
//     numbered lowptr[i] + position in row i (the lower neighbours are the first entries of a row), PAIR_OF[slot] maps both directed
//     slots to it;
//   * all reverse-mode scatters over the source atom are evaluated as gathers over the atom's own CSR row -- no atomics, bitwise
//     reproducible.
// PARITY UNPINNED (schnetpack is not part of the reference tree): checked against this repo's restatement only.
#include "common.h"
#include "lanes.h"
#include "../../include/nablaq.h"

#define LN2F 0.69314718055994530942f
__device__ __forceinline__ float sn_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float sn_ssp(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))) - LN2F; }   // softplus(x) - ln 2

static inline dim3 sn_grid1d(long count, int block) { return dim3((unsigned)((count + block - 1) / block)); }

// ---- elementwise ------------------------------------------------------------------------------------------------------------
__global__ void k_sn_ssp(const float* __restrict__ Z, float* __restrict__ U, long count) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) U[i] = sn_ssp(Z[i]);
}
__global__ void k_sn_ssp_tan(const float* __restrict__ Z, const float* __restrict__ TZ, float* __restrict__ TU, long count) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) TU[i] = sn_sigmoid(Z[i]) * TZ[i];
}
// in place on the adjoints:  G <- G sig(Z) (+ GT sig'(Z) TZ);  GT <- GT sig(Z)
template <bool DUAL>
__global__ void k_sn_ssp_rev(const float* __restrict__ Z, const float* __restrict__ TZ, float* __restrict__ G, float* __restrict__ GT, long count) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float s = sn_sigmoid(Z[i]);
  float g = G[i] * s;
  if (DUAL) {
    const float gt = GT[i];
    g += gt * s * (1.0f - s) * TZ[i];
    GT[i] = gt * s;
  }
  G[i] = g;
}
__global__ void k_sn_add(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ O, long count) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) O[i] = A[i] + B[i];
}

// ---- undirected pairs ---------------------------------------------------------------------------------------------------------
// one thread per CSR slot: PAIR_OF[sp]; the canonical (lower) slot also writes PAIR_SLOT[p] and the pair's geometry record
__global__ void k_sn_pairs(NqGraphView g, const int* __restrict__ dst, int* __restrict__ PAIR_OF, int* __restrict__ PAIR_SLOT, float4* __restrict__ GEOMP) {
  const int sp = blockIdx.x * blockDim.x + threadIdx.x;
  if (sp >= g.E) return;
  const int i = dst[sp], k = g.col[sp];
  if (k < i) {
    const int p = g.lowptr[i] + (sp - g.row_ptr[i]);
    PAIR_OF[sp] = p; PAIR_SLOT[p] = sp; GEOMP[p] = g.geom[sp];
  } else {
    const int r = g.rev[sp];                        // slot of (k <- i), a lower entry of row k
    PAIR_OF[sp] = g.lowptr[k] + (r - g.row_ptr[k]);
  }
}
__global__ void k_sn_gather_f(const float* __restrict__ in, const int* __restrict__ idx, int n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[idx[i]];
}

// ---- first filter layer from the window record ------------------------------------------------------------------------------
// One wavefront walks a contiguous chunk of edges; lane l owns channels [l*CH, (l+1)*CH).  W1^T [R][F] lives in LDS.
//   TAN = false:  A1[e]  = ssp(z1_e),                 z1_e = b1 + sum_t rho_t W1T[k0+t]
//   TAN = true :  TA1[e] = sig(z1_e) * td_e * psi_e,  psi_e = sum_t rho'_t W1T[k0+t]
template <bool PSI, int CH>
__device__ __forceinline__ void sn_filter1(const WinRegs<PSI>& w, const float* wt, int F, int fb, const float (&b1)[CH], float (&z)[CH], float (&psi)[CH]) {
  typedef VOps<CH> O;
  typename O::V vz = O::from(b1), vp = O::splat(0.f);
  const int k0 = __builtin_amdgcn_readfirstlane(__float_as_int(w.rr[13]));
  const float* wk = wt + k0 * F + fb;
#pragma unroll
  for (int t = 0; t < FWIN; ++t) {
    const typename O::V wv = lds_tap<CH>(wk + t * F);
    vz = O::fma(wv, O::splat(w.rr[t]), vz);
    if (PSI) vp = O::fma(wv, O::splat(w.dd[t]), vp);
  }
  O::to(z, vz); O::to(psi, vp);
}

// 16 wavefronts share one LDS copy of W1^T: the per-edge chain (scalar window load -> LDS taps -> exp/log -> store) is latency-bound,
// so occupancy (2 workgroups x 16 waves per CU at <= 64 VGPRs) is what hides it.  4 waves per workgroup measured 2.3 ms/step per flavour.
#define SN_F1_THREADS 1024
#define SN_F1_PROLOGUE                                                                          \
  extern __shared__ __attribute__((aligned(16))) float wt[];                                   \
  {                                                                                            \
    const int total4 = (R * F) >> 2, padded4 = ((R < FWIN ? FWIN : R) * F) >> 2;                \
    const float4* src = reinterpret_cast<const float4*>(W1T);                                  \
    float4* dst4 = reinterpret_cast<float4*>(wt);                                              \
    for (int i = threadIdx.x; i < padded4; i += blockDim.x)                                    \
      dst4[i] = i < total4 ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);                         \
  }                                                                                            \
  __syncthreads();                                                                             \
  const int lane = threadIdx.x & 63, fb = lane * CH;                                           \
  const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));   \
  const int nwaves = gridDim.x * (blockDim.x >> 6);                                            \
  const int chunk = (E + nwaves - 1) / nwaves;                                                 \
  const int e0 = wave * chunk, e1 = min(E, e0 + chunk);                                        \
  float b1r[CH];                                                                               \
  ldv<CH>(b1r, b1 + fb);                                                                       \
  if (e0 >= e1) return;

template <bool TAN, int CH>
__global__ __launch_bounds__(SN_F1_THREADS) void k_sn_filter1(const float* __restrict__ RW, const float* __restrict__ W1T, const float* __restrict__ b1,
                                                    const float* __restrict__ TD, float* __restrict__ OUT, int E, int F, int R) {
  SN_F1_PROLOGUE
  WinRegs<TAN> win;
  load_win<TAN>(win, RW, __builtin_amdgcn_readfirstlane(e0));
  for (int e = e0; e < e1; ++e) {
    float z[CH], psi[CH], o[CH];
    sn_filter1<TAN, CH>(win, wt, F, fb, b1r, z, psi);
    __builtin_amdgcn_sched_barrier(0);
    load_win<TAN>(win, RW, __builtin_amdgcn_readfirstlane(min(e + 1, e1 - 1)));     // scalar loads after the LDS reads (shared lgkmcnt), branch-free
    if (TAN) {
      const float td = TD[e];
#pragma unroll
      for (int c = 0; c < CH; ++c) o[c] = sn_sigmoid(z[c]) * td * psi[c];
    } else {
#pragma unroll
      for (int c = 0; c < CH; ++c) o[c] = sn_ssp(z[c]);
    }
    stv<CH>(OUT + (long)e * F + fb, o);
  }
}

// reverse of the first filter layer (z1, psi recomputed from the window):
//   DUAL = false (force adjoint):  gd_e += sum_c ga1 sig(z1) psi
//   DUAL = true :  GA1[e] <- gz1 = ga1 sig(z1) + gta1 sig'(z1) td psi ;  GTA1[e] <- gtz1 * td = gta1 sig(z1) td   (in place; these are the
//                  (gphi, gpsi) operands of the k0-sorted weight-gradient kernel)
template <bool DUAL, int CH>
__global__ __launch_bounds__(SN_F1_THREADS) void k_sn_filter1_rev(const float* __restrict__ RW, const float* __restrict__ W1T, const float* __restrict__ b1,
                                                        const float* __restrict__ TD, float* __restrict__ GA1, float* __restrict__ GTA1,
                                                        float* __restrict__ GD, int E, int F, int R) {
  SN_F1_PROLOGUE
  WinRegs<true> win;
  load_win<true>(win, RW, __builtin_amdgcn_readfirstlane(e0));
  float ga[CH], gta[CH];
  ldv<CH>(ga, GA1 + (long)e0 * F + fb);
  if (DUAL) ldv<CH>(gta, GTA1 + (long)e0 * F + fb);
  for (int e = e0; e < e1; ++e) {
    const int en = min(e + 1, e1 - 1);
    float gan[CH], gtan[CH];
    ldv<CH>(gan, GA1 + (long)en * F + fb);
    if (DUAL) ldv<CH>(gtan, GTA1 + (long)en * F + fb);
    float z[CH], psi[CH];
    sn_filter1<true, CH>(win, wt, F, fb, b1r, z, psi);
    __builtin_amdgcn_sched_barrier(0);
    load_win<true>(win, RW, __builtin_amdgcn_readfirstlane(en));
    if (DUAL) {
      const float td = TD[e];
      float gz[CH], gt[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const float s = sn_sigmoid(z[c]);
        gz[c] = ga[c] * s + gta[c] * s * (1.0f - s) * td * psi[c];
        gt[c] = gta[c] * s * td;
      }
      stv<CH>(GA1 + (long)e * F + fb, gz);
      stv<CH>(GTA1 + (long)e * F + fb, gt);
    } else {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < CH; ++c) acc += ga[c] * sn_sigmoid(z[c]) * psi[c];
      acc = nq_wave_sum(acc);
      if (lane == 0) GD[e] += acc;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) { ga[c] = gan[c]; if (DUAL) gta[c] = gtan[c]; }
  }
}

// ---- continuous-filter convolution: one wavefront per atom ------------------------------------------------------------------
// XCD-aware block -> atom mapping: blocks are dealt round-robin to the 8 XCDs; XCD x takes the x-th contiguous eighth of the atoms.
__device__ __forceinline__ int sn_atom_of_wave(int N) {
  const int wpb = blockDim.x >> 6;
  const int nb = gridDim.x, per = nb >> 3;                     // grid is a multiple of 8
  const int blk = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  return __builtin_amdgcn_readfirstlane(blk * wpb + (int)(threadIdx.x >> 6));
}

struct SnRow { int kk, pp; float rc, dt; };   // per lane: neighbour, pair id, fcut, fcut' * td of edge (c0 + lane)
template <bool DUAL>
__device__ __forceinline__ void sn_load_row(SnRow& r, const int* __restrict__ col, const int* __restrict__ PAIR_OF, const float* __restrict__ RW,
                                            const float* __restrict__ TD, int c0, int cnt, int lane) {
  r.kk = 0; r.pp = 0; r.rc = 0.f; r.dt = 0.f;
  if (lane < cnt) {
    const int sp = c0 + lane;
    r.kk = col[sp];
    r.pp = PAIR_OF[sp];
    r.rc = RW[(long)r.pp * RW_STRIDE + 14];
    if (DUAL) r.dt = RW[(long)r.pp * RW_STRIDE + 30] * TD[r.pp];
  }
}

// SINGLE (DUAL=false):  O1_i = sum_e A_k w_e                               w  = h2 fcut
// DUAL              :  O1_i = sum_e A_k w_e + B_k wt_e,  O2_i = sum_e B_k w_e   (O2 optional)      wt = th2 fcut + h2 fcut' td
template <bool DUAL, int CH>
__global__ __launch_bounds__(256) void k_sn_conv(NqGraphView g, int F, const int* __restrict__ PAIR_OF, const float* __restrict__ RW, const float* __restrict__ TD,
                                                 const float* __restrict__ A, const float* __restrict__ Bv, const float* __restrict__ H2,
                                                 const float* __restrict__ TH2, float* __restrict__ O1, float* __restrict__ O2) {
  const int n = sn_atom_of_wave(g.N);
  if (n >= g.N) return;
  const int lane = threadIdx.x & 63, fb = lane * CH;
  const int beg = __builtin_amdgcn_readfirstlane(g.row_ptr[n]), end = __builtin_amdgcn_readfirstlane(g.row_ptr[n + 1]);
  float o1[CH], o2[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) o1[c] = o2[c] = 0.f;
  for (int c0 = beg; c0 < end; c0 += 64) {
    const int cnt = min(64, end - c0);
    SnRow row;
    sn_load_row<DUAL>(row, g.col, PAIR_OF, RW, TD, c0, cnt, lane);
    float a[CH], b[CH], h[CH], th[CH];
    {
      const int k = bl_i(row.kk, 0), p = bl_i(row.pp, 0);
      ldv<CH>(a, A + (long)k * F + fb); ldv<CH>(h, H2 + (long)p * F + fb);
      if (DUAL) { ldv<CH>(b, Bv + (long)k * F + fb); ldv<CH>(th, TH2 + (long)p * F + fb); }
    }
    for (int j = 0; j < cnt; ++j) {
      const int jn = min(j + 1, cnt - 1);
      const int kn = bl_i(row.kk, jn), pn = bl_i(row.pp, jn);
      float an[CH], bn[CH], hn[CH], thn[CH];
      ldv<CH>(an, A + (long)kn * F + fb); ldv<CH>(hn, H2 + (long)pn * F + fb);
      if (DUAL) { ldv<CH>(bn, Bv + (long)kn * F + fb); ldv<CH>(thn, TH2 + (long)pn * F + fb); }
      const float rc = bl_f(row.rc, j);
      const float dt = DUAL ? bl_f(row.dt, j) : 0.f;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const float w = h[c] * rc;
        o1[c] += a[c] * w;
        if (DUAL) {
          const float wtc = th[c] * rc + h[c] * dt;
          o1[c] += b[c] * wtc;
          o2[c] += b[c] * w;
        }
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) { a[c] = an[c]; h[c] = hn[c]; if (DUAL) { b[c] = bn[c]; th[c] = thn[c]; } }
    }
  }
  stv<CH>(O1 + (long)n * F + fb, o1);
  if (DUAL && O2) stv<CH>(O2 + (long)n * F + fb, o2);
}

// adjoint of the filter per PAIR: the wavefront of atom i owns the pairs (i, k), k < i (contiguous pair rows lowptr[i]..lowptr[i+1]);
// its own rows stay in registers, the neighbour's rows are gathered.  Both directed edges of the pair contribute:
//   DUAL = false:  gW = gm_i y_k + gm_k y_i ;  GH2[p] = gW fcut ;  GD[p] = fcut' sum_c gW h2_p
//   DUAL = true :  GH2[p] = fcut (gm_i y_k + gm_k y_i + gtm_i ty_k + gtm_k ty_i) + fcut' td (gtm_i y_k + gtm_k y_i)
//                  GTH2[p] = fcut (gtm_i y_k + gtm_k y_i)
template <bool DUAL, int CH>
__global__ __launch_bounds__(256) void k_sn_pair_rev(NqGraphView g, int F, const float* __restrict__ RW, const float* __restrict__ TD,
                                                     const float* __restrict__ GM, const float* __restrict__ GTM, const float* __restrict__ Y,
                                                     const float* __restrict__ TY, const float* __restrict__ H2, float* __restrict__ GH2,
                                                     float* __restrict__ GTH2, float* __restrict__ GD) {
  const int n = sn_atom_of_wave(g.N);
  if (n >= g.N) return;
  const int lane = threadIdx.x & 63, fb = lane * CH;
  const int p_beg = __builtin_amdgcn_readfirstlane(g.lowptr[n]), p_end = __builtin_amdgcn_readfirstlane(g.lowptr[n + 1]);
  const int s_beg = __builtin_amdgcn_readfirstlane(g.row_ptr[n]);          // the lower neighbours are the first p_end - p_beg slots of the row
  float gmi[CH], gtmi[CH], yi[CH], tyi[CH];
  ldv<CH>(gmi, GM + (long)n * F + fb); ldv<CH>(yi, Y + (long)n * F + fb);
  if (DUAL) { ldv<CH>(gtmi, GTM + (long)n * F + fb); ldv<CH>(tyi, TY + (long)n * F + fb); }
  for (int q0 = p_beg; q0 < p_end; q0 += 64) {
    const int cnt = min(64, p_end - q0);
    int kk = 0; float rcl = 0.f, dtl = 0.f;
    if (lane < cnt) {
      const int p = q0 + lane;
      kk = g.col[s_beg + (p - p_beg)];
      rcl = RW[(long)p * RW_STRIDE + 14];
      dtl = RW[(long)p * RW_STRIDE + 30] * (DUAL ? TD[p] : 1.0f);      // force mode: fcut' itself
    }
    float y[CH], gm[CH], ty[CH], gtm[CH], h[CH];
    {
      const int k = bl_i(kk, 0);
      ldv<CH>(y, Y + (long)k * F + fb); ldv<CH>(gm, GM + (long)k * F + fb);
      if (DUAL) { ldv<CH>(ty, TY + (long)k * F + fb); ldv<CH>(gtm, GTM + (long)k * F + fb); } else ldv<CH>(h, H2 + (long)q0 * F + fb);
    }
    float gd_lane = 0.f;
    for (int j = 0; j < cnt; ++j) {
      const int jn = min(j + 1, cnt - 1);
      const int kn = bl_i(kk, jn);
      float yn[CH], gmn[CH], tyn[CH], gtmn[CH], hn[CH];
      ldv<CH>(yn, Y + (long)kn * F + fb); ldv<CH>(gmn, GM + (long)kn * F + fb);
      if (DUAL) { ldv<CH>(tyn, TY + (long)kn * F + fb); ldv<CH>(gtmn, GTM + (long)kn * F + fb); } else ldv<CH>(hn, H2 + (long)(q0 + jn) * F + fb);
      const float rc = bl_f(rcl, j), dt = bl_f(dtl, j);
      float o[CH], ot[CH];
      if (DUAL) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const float tw = gtmi[c] * y[c] + gtm[c] * yi[c];
          o[c] = rc * (gmi[c] * y[c] + gm[c] * yi[c] + gtmi[c] * ty[c] + gtm[c] * tyi[c]) + dt * tw;
          ot[c] = rc * tw;
        }
        stv_stream<CH>(GH2 + (long)(q0 + j) * F + fb, o);
        stv_stream<CH>(GTH2 + (long)(q0 + j) * F + fb, ot);
      } else {
        float sacc = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const float gw = gmi[c] * y[c] + gm[c] * yi[c];
          o[c] = gw * rc;
          sacc += gw * h[c];
        }
        stv_stream<CH>(GH2 + (long)(q0 + j) * F + fb, o);
        sacc = nq_wave_sum(sacc);
        gd_lane = (lane == j) ? sacc * dt : gd_lane;
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) { y[c] = yn[c]; gm[c] = gmn[c]; if (DUAL) { ty[c] = tyn[c]; gtm[c] = gtmn[c]; } else h[c] = hn[c]; }
    }
    if (!DUAL && lane < cnt) GD[q0 + lane] = gd_lane;
  }
}

// F_i = -dE/dr_i = sum_{e in row i} u_e gd_pair(e)      (gd_pair = dE/dd of the pair = both directed edges; dd/dr_i = -u_e on the own row)
__global__ void k_sn_forces(NqGraphView g, const int* __restrict__ PAIR_OF, const float* __restrict__ GD, float* __restrict__ forces) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= g.N) return;
  float fx = 0.f, fy = 0.f, fz = 0.f;
  for (int e = g.row_ptr[n]; e < g.row_ptr[n + 1]; ++e) {
    const float4 u = g.geom[e];
    const float s = GD[PAIR_OF[e]];
    fx += u.x * s; fy += u.y * s; fz += u.z * s;
  }
  forces[3 * (long)n] = fx; forces[3 * (long)n + 1] = fy; forces[3 * (long)n + 2] = fz;
}

// ---- launchers --------------------------------------------------------------------------------------------------------------
static int sn_ssp(hipStream_t st, const float* Z, float* U, long count) {
  NQ_PROF(st, "sn_ssp");
  hipLaunchKernelGGL(k_sn_ssp, sn_grid1d(count, 256), dim3(256), 0, st, Z, U, count);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
static int sn_ssp_tan(hipStream_t st, const float* Z, const float* TZ, float* TU, long count) {
  NQ_PROF(st, "sn_ssp");
  hipLaunchKernelGGL(k_sn_ssp_tan, sn_grid1d(count, 256), dim3(256), 0, st, Z, TZ, TU, count);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
static int sn_ssp_rev(hipStream_t st, const float* Z, const float* TZ, float* G, float* GT, long count, bool dual) {
  NQ_PROF(st, "sn_ssp");
  if (dual) hipLaunchKernelGGL((k_sn_ssp_rev<true>), sn_grid1d(count, 256), dim3(256), 0, st, Z, TZ, G, GT, count);
  else hipLaunchKernelGGL((k_sn_ssp_rev<false>), sn_grid1d(count, 256), dim3(256), 0, st, Z, TZ, G, GT, count);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
static int sn_add(hipStream_t st, const float* A, const float* B, float* O, long count) {
  NQ_PROF(st, "sn_add");
  hipLaunchKernelGGL(k_sn_add, sn_grid1d(count, 256), dim3(256), 0, st, A, B, O, count);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

#define SN_CH_SWITCH(F, BODY)                                                             \
  switch ((F) / 64) {                                                                     \
    case 1: { constexpr int CHV = 1; BODY; } break;                                       \
    case 2: { constexpr int CHV = 2; BODY; } break;                                       \
    case 4: { constexpr int CHV = 4; BODY; } break;                                       \
    default: return nq_fail(NQ_ERR_ARG, "SchNet kernels need n_atom_basis in {64,128,256}"); \
  }

static int sn_f1_grid(int E, size_t lds) {
  const int per_cu = (int)((156 * 1024) / (lds ? lds : 1));
  const int wgs = 256 * (per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu));
  const int need = nq_cdiv(E, (SN_F1_THREADS / 64) * 16);   // at least 16 edges per wavefront
  return need < wgs ? (need < 1 ? 1 : need) : wgs;
}
#define SN_F1_LAUNCH(KERN, ...)                                                                                     \
  do {                                                                                                             \
    NQ_DYN_LDS(KERN, lds);                                                                                         \
    hipLaunchKernelGGL(KERN, dim3(grid), dim3(SN_F1_THREADS), lds, st, __VA_ARGS__);                                         \
  } while (0)

static int sn_filter1(hipStream_t st, const float* RW, const float* W1T, const float* b1, const float* TD, float* OUT, int E, int F, int R, bool tan) {
  NQ_PROF(st, tan ? "sn_filter1_tan" : "sn_filter1");
  if (E <= 0) return NQ_OK;
  const size_t lds = (size_t)(R < FWIN ? FWIN : R) * F * sizeof(float);
  const int grid = sn_f1_grid(E, lds);
  SN_CH_SWITCH(F, {
    if (tan) SN_F1_LAUNCH((k_sn_filter1<true, CHV>), RW, W1T, b1, TD, OUT, E, F, R);
    else SN_F1_LAUNCH((k_sn_filter1<false, CHV>), RW, W1T, b1, TD, OUT, E, F, R);
  })
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
static int sn_filter1_rev(hipStream_t st, const float* RW, const float* W1T, const float* b1, const float* TD, float* GA1, float* GTA1, float* GD,
                          int E, int F, int R, bool dual) {
  NQ_PROF(st, dual ? "sn_filter1_rev_dual" : "sn_filter1_rev_force");
  if (E <= 0) return NQ_OK;
  const size_t lds = (size_t)(R < FWIN ? FWIN : R) * F * sizeof(float);
  const int grid = sn_f1_grid(E, lds);
  SN_CH_SWITCH(F, {
    if (dual) SN_F1_LAUNCH((k_sn_filter1_rev<true, CHV>), RW, W1T, b1, TD, GA1, GTA1, GD, E, F, R);
    else SN_F1_LAUNCH((k_sn_filter1_rev<false, CHV>), RW, W1T, b1, TD, GA1, GTA1, GD, E, F, R);
  })
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
static int sn_atom_grid(int N) { return ((nq_cdiv(N, 4) + 7) / 8) * 8; }
static int sn_conv(hipStream_t st, const NqGraphView& g, int F, const int* PAIR_OF, const float* RW, const float* TD, const float* A, const float* B,
                   const float* H2, const float* TH2, float* O1, float* O2, bool dual, const char* name) {
  NQ_PROF(st, name);
  if (g.N <= 0) return NQ_OK;
  const int grid = sn_atom_grid(g.N);
  SN_CH_SWITCH(F, {
    if (dual) hipLaunchKernelGGL((k_sn_conv<true, CHV>), dim3(grid), dim3(256), 0, st, g, F, PAIR_OF, RW, TD, A, B, H2, TH2, O1, O2);
    else hipLaunchKernelGGL((k_sn_conv<false, CHV>), dim3(grid), dim3(256), 0, st, g, F, PAIR_OF, RW, TD, A, B, H2, TH2, O1, O2);
  })
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
static int sn_pair_rev(hipStream_t st, const NqGraphView& g, int F, const float* RW, const float* TD, const float* GM, const float* GTM,
                       const float* Y, const float* TY, const float* H2, float* GH2, float* GTH2, float* GD, bool dual) {
  NQ_PROF(st, dual ? "sn_pair_rev_dual" : "sn_pair_rev_force");
  if (g.N <= 0) return NQ_OK;
  const int grid = sn_atom_grid(g.N);
  SN_CH_SWITCH(F, {
    if (dual) hipLaunchKernelGGL((k_sn_pair_rev<true, CHV>), dim3(grid), dim3(256), 0, st, g, F, RW, TD, GM, GTM, Y, TY, H2, GH2, GTH2, GD);
    else hipLaunchKernelGGL((k_sn_pair_rev<false, CHV>), dim3(grid), dim3(256), 0, st, g, F, RW, TD, GM, GTM, Y, TY, H2, GH2, GTH2, GD);
  })
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

// ---- parameter and workspace layout -----------------------------------------------------------------------------------------
struct SnLayerP { size_t Win, W1, b1, W2, b2, Wo1, bo1, Wo2, bo2; };
struct SnParams { size_t emb; SnLayerP lay[NQ_MAX_LAYERS]; size_t O1, o1, w2, o2, total; };
static void sn_param_layout(const nq_schnet_cfg* c, SnParams* P) {
  const size_t F = c->n_atom_basis, R = c->n_rbf, H = F / 2;
  size_t o = 0;
  P->emb = o; o += (size_t)c->max_z * F;
  for (int l = 0; l < c->n_interactions; ++l) {
    SnLayerP& y = P->lay[l];
    y.Win = o; o += F * F;
    y.W1 = o; o += F * R; y.b1 = o; o += F;
    y.W2 = o; o += F * F; y.b2 = o; o += F;
    y.Wo1 = o; o += F * F; y.bo1 = o; o += F;
    y.Wo2 = o; o += F * F; y.bo2 = o; o += F;
  }
  P->O1 = o; o += H * F; P->o1 = o; o += H; P->w2 = o; o += H; P->o2 = o; o += 1;
  P->total = o;
}
struct SnLayerW { size_t W1T, A1, H2, Y, M, T1, U; };   // A1, H2: [2][P][F] (primal, tangent) per undirected pair; Y, M, T1, U: [2][N][F]
struct SnWs {
  size_t X[NQ_MAX_LAYERS + 1];                          // [2][N][F]
  SnLayerW lay[NQ_MAX_LAYERS];
  size_t RW, ORDER, TD, TDP, GD, GDT, PAIR_OF, PAIR_SLOT, GEOMP, pos_dot, ZO, e_atom, te_atom, ge, gte, GZO, TMPW, GX, GU, GM, GY, V, GH2, GA1, scratch, total;
};
static size_t sn_a4(size_t x) { return (x + 3) & ~(size_t)3; }
static size_t sn_max(size_t a, size_t b) { return a > b ? a : b; }
static void sn_ws_layout(const nq_schnet_cfg* c, size_t N, size_t E, size_t B, SnWs* W) {
  const size_t F = c->n_atom_basis, R = c->n_rbf, H = F / 2, L = c->n_interactions;
  const size_t Pn = E / 2;          // undirected pairs (the edge list is symmetric)
  size_t o = 0;
  auto take = [&](size_t n) { const size_t at = o; o += sn_a4(n); return at; };
  for (size_t l = 0; l <= L; ++l) W->X[l] = take(2 * N * F);
  for (size_t l = 0; l < L; ++l) {
    SnLayerW& y = W->lay[l];
    y.W1T = take(R * F); y.A1 = take(2 * Pn * F); y.H2 = take(2 * Pn * F);
    y.Y = take(2 * N * F); y.M = take(2 * N * F); y.T1 = take(2 * N * F); y.U = take(2 * N * F);
  }
  W->RW = take(Pn * RW_STRIDE); W->ORDER = take(Pn); W->TD = take(E); W->TDP = take(Pn); W->GD = take(Pn); W->GDT = take(Pn);
  W->PAIR_OF = take(E); W->PAIR_SLOT = take(Pn); W->GEOMP = take(4 * Pn); W->pos_dot = take(3 * N);
  W->ZO = take(2 * N * H); W->e_atom = take(N); W->te_atom = take(N); W->ge = take(N); W->gte = take(N);
  W->GZO = take(2 * N * H); W->TMPW = take(N * H);
  W->GX = take(2 * N * F); W->GU = take(2 * N * F); W->GM = take(2 * N * F); W->GY = take(2 * N * F); W->V = take(2 * N * F);
  W->GH2 = take(2 * Pn * F); W->GA1 = take(2 * Pn * F);
  size_t s = nq_gemm_tn_scratch_floats(2 * (long)Pn, (int)F, (int)F);
  s = sn_max(s, nq_gemm_tn_scratch_floats(2 * (long)N, (int)F, (int)F));
  s = sn_max(s, nq_gwr_scratch_floats((int)Pn, (int)F, (int)R, 1));
  s = sn_max(s, nq_colsum_scratch_floats((long)Pn, (int)F));
  s = sn_max(s, nq_k0_sort_scratch_ints((int)Pn, (int)R));
  s = sn_max(s, nq_embed_grad_scratch_floats((int)N, (int)F, c->max_z - 1));
  W->scratch = take(s + 64);
  W->total = o;
  (void)B;
}
static NqGraphView sn_view(const nq_graph* g) {
  NqGraphView v;
  v.N = g->N; v.B = g->B; v.E = g->E; v.mol_ptr = g->mol_ptr; v.row_ptr = g->row_ptr; v.col = g->col; v.rev = g->rev;
  v.geom = reinterpret_cast<const float4*>(g->geom); v.z = g->z; v.atom_mol = g->atom_mol; v.lowptr = g->lowptr;
  return v;
}
static int sn_check(const nq_schnet_cfg* c, const nq_graph* g, const void* ws, size_t ws_bytes, SnWs* W, SnParams* P) {
  if (!c || !g || !ws) return nq_fail(NQ_ERR_ARG, "null argument");
  const int F = c->n_atom_basis;
  if (!(F == 64 || F == 128 || F == 256)) return nq_fail(NQ_ERR_ARG, "n_atom_basis must be 64, 128 or 256");
  if (c->n_interactions < 1 || c->n_interactions > NQ_MAX_LAYERS) return nq_fail(NQ_ERR_ARG, "n_interactions out of range");
  if (c->n_rbf < 2 || (size_t)(c->n_rbf < FWIN ? FWIN : c->n_rbf) * F * sizeof(float) > 156 * 1024) return nq_fail(NQ_ERR_ARG, "n_rbf * n_atom_basis does not fit LDS");
  if (c->max_z < 2) return nq_fail(NQ_ERR_ARG, "max_z");
  if (g->N <= 0 || g->E <= 0 || g->B <= 0) return nq_fail(NQ_ERR_ARG, "empty graph");
  if ((g->E & 1) || !g->lowptr || !g->dst) return nq_fail(NQ_ERR_ARG, "SchNet needs the symmetric edge list with lowptr (nq_graph_count / nq_graph_fill)");
  sn_param_layout(c, P);
  sn_ws_layout(c, g->N, g->E, g->B, W);
  if (ws_bytes < W->total * sizeof(float)) return nq_fail(NQ_ERR_ARG, "workspace too small: %zu < %zu bytes", ws_bytes, W->total * sizeof(float));
  return NQ_OK;
}

extern "C" {

size_t nq_schnet_num_params(const nq_schnet_cfg* cfg) {
  if (!cfg || cfg->n_interactions < 1 || cfg->n_interactions > NQ_MAX_LAYERS) return 0;
  SnParams P; sn_param_layout(cfg, &P);
  return P.total;
}
size_t nq_schnet_workspace_bytes(const nq_schnet_cfg* cfg, int32_t N, int32_t E, int32_t B) {
  if (!cfg || cfg->n_interactions < 1 || cfg->n_interactions > NQ_MAX_LAYERS || N < 0 || E < 0) return 0;
  SnWs W; sn_ws_layout(cfg, N, E, B, &W);
  return W.total * sizeof(float);
}

int nq_schnet_forward(const nq_schnet_cfg* cfg, const float* params, const float* rbf_offsets, const nq_graph* graph, void* workspace,
                      size_t workspace_bytes, float* energy, float* forces, void* stream) {
  SnWs W; SnParams P;
  NQ_TRY(sn_check(cfg, graph, workspace, workspace_bytes, &W, &P));
  if (!params || !rbf_offsets || !energy) return nq_fail(NQ_ERR_ARG, "null argument");
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const NqGraphView g = sn_view(graph);
  const int N = g.N, E = g.E, Pn = E / 2, F = cfg->n_atom_basis, R = cfg->n_rbf, H = F / 2, L = cfg->n_interactions;
  const size_t NF = (size_t)N * F;
  const float* RW = ws + W.RW;
  const int* PAIR_OF = reinterpret_cast<const int*>(ws + W.PAIR_OF);

  NQ_TRY(nq_embed(st, g.z, params + P.emb + F, N, F, ws + W.X[0]));       // table indexed by Z (row 0 = padding): shift by one row
  {
    NQ_PROF(st, "sn_pairs");
    hipLaunchKernelGGL(k_sn_pairs, sn_grid1d(E, 256), dim3(256), 0, st, g, graph->dst, reinterpret_cast<int*>(ws + W.PAIR_OF),
                       reinterpret_cast<int*>(ws + W.PAIR_SLOT), reinterpret_cast<float4*>(ws + W.GEOMP));
    NQ_LAUNCH_CHECK();
  }
  FilterArgs fa;
  nq_make_filter_args(&fa, nullptr, nullptr, rbf_offsets, RW, R, cfg->cutoff, 5, cfg->rbf_coeff, 2);
  NQ_TRY(nq_rbf_window(st, reinterpret_cast<const float4*>(ws + W.GEOMP), Pn, fa, ws + W.RW));
  NQ_TRY(nq_k0_sort(st, RW, Pn, R, reinterpret_cast<int*>(ws + W.ORDER), reinterpret_cast<int*>(ws + W.scratch)));
  for (int l = 0; l < L; ++l) {
    const SnLayerW& y = W.lay[l]; const SnLayerP& p = P.lay[l];
    NQ_TRY(nq_transpose(st, params + p.W1, F, R, ws + y.W1T));
    NQ_TRY(sn_filter1(st, RW, ws + y.W1T, params + p.b1, nullptr, ws + y.A1, Pn, F, R, false));
    NQ_TRY(nq_gemm_nt(st, ws + y.A1, params + p.W2, ws + y.H2, params + p.b2, nullptr, Pn, F, F, F, F, F, "sn:W2"));
    NQ_TRY(nq_gemm_nt(st, ws + W.X[l], params + p.Win, ws + y.Y, nullptr, nullptr, N, F, F, F, F, F, "sn:in2f"));
    NQ_TRY(sn_conv(st, g, F, PAIR_OF, RW, nullptr, ws + y.Y, nullptr, ws + y.H2, nullptr, ws + y.M, nullptr, false, "sn_conv"));
    NQ_TRY(nq_gemm_nt(st, ws + y.M, params + p.Wo1, ws + y.T1, params + p.bo1, nullptr, N, F, F, F, F, F, "sn:f2out0"));
    NQ_TRY(sn_ssp(st, ws + y.T1, ws + y.U, (long)NF));
    NQ_TRY(nq_gemm_nt(st, ws + y.U, params + p.Wo2, ws + W.V, params + p.bo2, nullptr, N, F, F, F, F, F, "sn:f2out1"));
    NQ_TRY(sn_add(st, ws + W.X[l], ws + W.V, ws + W.X[l + 1], (long)NF));
  }
  NQ_TRY(nq_gemm_nt(st, ws + W.X[L], params + P.O1, ws + W.ZO, params + P.o1, nullptr, N, H, F, F, F, H, "O1"));
  ReadoutArgs r{};
  r.N = N; r.H = H; r.ZO = ws + W.ZO; r.w2 = params + P.w2; r.o2 = params + P.o2; r.e_atom = ws + W.e_atom;
  NQ_TRY(nq_readout(st, r, 0));
  NQ_TRY(nq_mol_sum(st, ws + W.e_atom, g.mol_ptr, g.B, energy));
  if (!forces) return NQ_OK;

  // ---- force adjoint: seeds dE_tot/d eps_i = 1 -> gd[pair] -> forces --------------------------------------------------
  NQ_TRY(nq_atom_seeds(st, nullptr, g.atom_mol, N, ws + W.ge, nullptr));
  r.ge = ws + W.ge; r.GZO = ws + W.GZO;
  NQ_TRY(nq_readout_rev(st, r, false));
  NQ_TRY(nq_gemm_nn(st, ws + W.GZO, params + P.O1, ws + W.GX, N, H, F, H, F, F, 0, "O1"));
  float* gd_total = ws + W.GDT;
  NQ_HIP(hipMemsetAsync(gd_total, 0, (size_t)Pn * sizeof(float), st));
  for (int l = L - 1; l >= 0; --l) {
    const SnLayerW& y = W.lay[l]; const SnLayerP& p = P.lay[l];
    NQ_TRY(nq_gemm_nn(st, ws + W.GX, params + p.Wo2, ws + W.GU, N, F, F, F, F, F, 0, "sn:f2out1"));
    NQ_TRY(sn_ssp_rev(st, ws + y.T1, nullptr, ws + W.GU, nullptr, (long)NF, false));
    NQ_TRY(nq_gemm_nn(st, ws + W.GU, params + p.Wo1, ws + W.GM, N, F, F, F, F, F, 0, "sn:f2out0"));
    NQ_TRY(sn_pair_rev(st, g, F, RW, nullptr, ws + W.GM, nullptr, ws + y.Y, nullptr, ws + y.H2, ws + W.GH2, nullptr, ws + W.GD, false));
    NQ_TRY(nq_gemm_nn(st, ws + W.GH2, params + p.W2, ws + W.GA1, Pn, F, F, F, F, F, 0, "sn:W2"));
    NQ_TRY(sn_filter1_rev(st, RW, ws + y.W1T, params + p.b1, nullptr, ws + W.GA1, nullptr, ws + W.GD, Pn, F, R, false));
    NQ_TRY(sn_add(st, gd_total, ws + W.GD, gd_total, (long)Pn));
    NQ_TRY(sn_conv(st, g, F, PAIR_OF, RW, nullptr, ws + W.GM, nullptr, ws + y.H2, nullptr, ws + W.GY, nullptr, false, "sn_conv"));
    NQ_TRY(nq_gemm_nn(st, ws + W.GY, params + p.Win, ws + W.GX, N, F, F, F, F, F, 1, "sn:in2f"));
  }
  {
    NQ_PROF(st, "sn_forces");
    hipLaunchKernelGGL(k_sn_forces, sn_grid1d(N, 128), dim3(128), 0, st, g, PAIR_OF, gd_total, forces);
    NQ_LAUNCH_CHECK();
  }
  return NQ_OK;
}

int nq_schnet_backward(const nq_schnet_cfg* cfg, const float* params, const float* rbf_offsets, const nq_graph* graph, void* workspace,
                       size_t workspace_bytes, const float* grad_energy, const float* grad_forces, float* grad_params, void* stream) {
  SnWs W; SnParams P;
  NQ_TRY(sn_check(cfg, graph, workspace, workspace_bytes, &W, &P));
  if (!params || !grad_params || !rbf_offsets) return nq_fail(NQ_ERR_ARG, "null argument");
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  float* gp = grad_params;
  float* scr = ws + W.scratch;
  const NqGraphView g = sn_view(graph);
  const int N = g.N, E = g.E, Pn = E / 2, F = cfg->n_atom_basis, R = cfg->n_rbf, H = F / 2, L = cfg->n_interactions;
  const size_t NF = (size_t)N * F, EF = (size_t)Pn * F, NH = (size_t)N * H;   // EF: stride between the primal and tangent halves of a pair array
  const float* RW = ws + W.RW;
  const float* TD = ws + W.TDP;                                              // tangent distances per pair
  const int* PAIR_OF = reinterpret_cast<const int*>(ws + W.PAIR_OF);

  // ---- tangent forward along pos_dot = -dL/dF ---------------------------------------------------------------------------
  if (grad_forces) NQ_TRY(nq_negate(st, grad_forces, ws + W.pos_dot, 3L * N));
  else NQ_HIP(hipMemsetAsync(ws + W.pos_dot, 0, 3 * (size_t)N * sizeof(float), st));
  NQ_TRY(nq_geom_tan(st, g, graph->dst, ws + W.pos_dot, ws + W.TD, nullptr));
  hipLaunchKernelGGL(k_sn_gather_f, sn_grid1d(Pn, 256), dim3(256), 0, st, ws + W.TD, reinterpret_cast<const int*>(ws + W.PAIR_SLOT), Pn, ws + W.TDP);
  NQ_LAUNCH_CHECK();
  NQ_HIP(hipMemsetAsync(ws + W.X[0] + NF, 0, NF * sizeof(float), st));
  for (int l = 0; l < L; ++l) {
    const SnLayerW& y = W.lay[l]; const SnLayerP& p = P.lay[l];
    NQ_TRY(sn_filter1(st, RW, ws + y.W1T, params + p.b1, TD, ws + y.A1 + EF, Pn, F, R, true));
    NQ_TRY(nq_gemm_nt(st, ws + y.A1 + EF, params + p.W2, ws + y.H2 + EF, nullptr, nullptr, Pn, F, F, F, F, F, "sn:W2"));
    NQ_TRY(nq_gemm_nt(st, ws + W.X[l] + NF, params + p.Win, ws + y.Y + NF, nullptr, nullptr, N, F, F, F, F, F, "sn:in2f"));
    NQ_TRY(sn_conv(st, g, F, PAIR_OF, RW, TD, ws + y.Y + NF, ws + y.Y, ws + y.H2, ws + y.H2 + EF, ws + y.M + NF, nullptr, true, "sn_conv_tan"));
    NQ_TRY(nq_gemm_nt(st, ws + y.M + NF, params + p.Wo1, ws + y.T1 + NF, nullptr, nullptr, N, F, F, F, F, F, "sn:f2out0"));
    NQ_TRY(sn_ssp_tan(st, ws + y.T1, ws + y.T1 + NF, ws + y.U + NF, (long)NF));
    NQ_TRY(nq_gemm_nt(st, ws + y.U + NF, params + p.Wo2, ws + W.V, nullptr, nullptr, N, F, F, F, F, F, "sn:f2out1"));
    NQ_TRY(sn_add(st, ws + W.X[l] + NF, ws + W.V, ws + W.X[l + 1] + NF, (long)NF));
  }
  NQ_TRY(nq_gemm_nt(st, ws + W.X[L] + NF, params + P.O1, ws + W.ZO + NH, nullptr, nullptr, N, H, F, F, F, H, "O1"));
  ReadoutArgs r{};
  r.N = N; r.H = H; r.ZO = ws + W.ZO; r.TZO = ws + W.ZO + NH; r.w2 = params + P.w2; r.o2 = params + P.o2;
  r.e_atom = ws + W.e_atom; r.te_atom = ws + W.te_atom;
  NQ_TRY(nq_readout(st, r, 1));

  // ---- dual reverse: seeds (dL/dE_b, 1) on (E_b, Edot) --------------------------------------------------------------------
  if (grad_energy) NQ_TRY(nq_atom_seeds(st, grad_energy, g.atom_mol, N, ws + W.ge, ws + W.gte));
  else {
    NQ_HIP(hipMemsetAsync(ws + W.ge, 0, (size_t)N * sizeof(float), st));
    NQ_TRY(nq_atom_seeds(st, nullptr, g.atom_mol, N, ws + W.gte, nullptr));
  }
  r.ge = ws + W.ge; r.gte = ws + W.gte; r.GZO = ws + W.GZO; r.GTZO = ws + W.GZO + NH; r.TMPW = ws + W.TMPW;
  NQ_TRY(nq_readout_rev(st, r, true));
  NQ_TRY(nq_colsum(st, ws + W.TMPW, N, H, H, gp + P.w2, scr));
  NQ_TRY(nq_colsum(st, ws + W.ge, N, 1, 1, gp + P.o2, scr));
  NQ_TRY(nq_gemm_tn(st, ws + W.GZO, ws + W.X[L], gp + P.O1, 2L * N, H, F, H, F, scr, "O1", gp + P.o1, N));
  NQ_TRY(nq_gemm_nn(st, ws + W.GZO, params + P.O1, ws + W.GX, 2 * N, H, F, H, F, F, 0, "O1"));
  for (int l = L - 1; l >= 0; --l) {
    const SnLayerW& y = W.lay[l]; const SnLayerP& p = P.lay[l];
    // f2out
    NQ_TRY(nq_gemm_tn(st, ws + W.GX, ws + y.U, gp + p.Wo2, 2L * N, F, F, F, F, scr, "sn:f2out1", gp + p.bo2, N));
    NQ_TRY(nq_gemm_nn(st, ws + W.GX, params + p.Wo2, ws + W.GU, 2 * N, F, F, F, F, F, 0, "sn:f2out1"));
    NQ_TRY(sn_ssp_rev(st, ws + y.T1, ws + y.T1 + NF, ws + W.GU, ws + W.GU + NF, (long)NF, true));
    NQ_TRY(nq_gemm_tn(st, ws + W.GU, ws + y.M, gp + p.Wo1, 2L * N, F, F, F, F, scr, "sn:f2out0", gp + p.bo1, N));
    NQ_TRY(nq_gemm_nn(st, ws + W.GU, params + p.Wo1, ws + W.GM, 2 * N, F, F, F, F, F, 0, "sn:f2out0"));
    // continuous-filter convolution
    NQ_TRY(sn_pair_rev(st, g, F, RW, TD, ws + W.GM, ws + W.GM + NF, ws + y.Y, ws + y.Y + NF, nullptr, ws + W.GH2, ws + W.GH2 + EF, nullptr, true));
    NQ_TRY(sn_conv(st, g, F, PAIR_OF, RW, TD, ws + W.GM, ws + W.GM + NF, ws + y.H2, ws + y.H2 + EF, ws + W.GY, ws + W.GY + NF, true, "sn_conv_dual"));
    // filter network
    NQ_TRY(nq_gemm_tn(st, ws + W.GH2, ws + y.A1, gp + p.W2, 2L * Pn, F, F, F, F, scr, "sn:W2", gp + p.b2, Pn));
    NQ_TRY(nq_gemm_nn(st, ws + W.GH2, params + p.W2, ws + W.GA1, 2 * Pn, F, F, F, F, F, 0, "sn:W2"));
    NQ_TRY(sn_filter1_rev(st, RW, ws + y.W1T, params + p.b1, TD, ws + W.GA1, ws + W.GA1 + EF, nullptr, Pn, F, R, true));
    NQ_TRY(nq_gwr_sorted(st, ws + W.GA1, ws + W.GA1 + EF, RW, reinterpret_cast<const int*>(ws + W.ORDER), Pn, F, R, gp + p.W1, scr, 1));
    NQ_TRY(nq_colsum(st, ws + W.GA1, Pn, F, F, gp + p.b1, scr));
    // in2f and the residual stream
    NQ_TRY(nq_gemm_tn(st, ws + W.GY, ws + W.X[l], gp + p.Win, 2L * N, F, F, F, F, scr, "sn:in2f"));
    NQ_TRY(nq_gemm_nn(st, ws + W.GY, params + p.Win, ws + W.GX, 2 * N, F, F, F, F, F, 1, "sn:in2f"));
  }
  NQ_HIP(hipMemsetAsync(gp + P.emb, 0, (size_t)F * sizeof(float), st));                                  // padding row
  NQ_TRY(nq_embed_grad(st, g.z, ws + W.GX, N, F, cfg->max_z - 1, gp + P.emb + F, scr));
  return NQ_OK;
}

}  // extern "C"
