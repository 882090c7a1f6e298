// Edge-level kernels of the PaiNN interaction block (reference: painn.py:475-509 PaiNNMessage,
// layers.py:14-33,129-185 RadialBasis; math restated in oracle/painn_sweeps.py).
//
// All message kernels are node-centric over the symmetric CSR built by graph.hip: the row of one atom is owned by
// one workgroup (materialised-filter kernels k_msg_*: one thread per feature channel) or one wavefront (fused-filter
// kernels k_msgf_*: CH = F/64 channels per lane), which loops over the atom's CSR row in ascending neighbour order.
// The segment reduction therefore needs no atomics and its summation order equals the reference's sequential
// scatter_add order.  In the reverse sweeps the same row is read as the atom's OUT-edges (n -> k): phi/psi/d are
// symmetric, r and t_r change sign.  Which wavefront computes a row (rows are claimed from counters on large batches)
// never changes the row's result.
#include "common.h"
#include <type_traits>
#include "lanes.h"

struct RbfArgs {
  const float4* geom; int E; int R; float inv_cutoff; float p, a, b, c; float coeff; const float* mu;
  float* rho; float* drho;
  int type;                 // 0 GaussianSmearing(0,1,R), 1 SphericalBesselBasis (layers.py:51-80), 2 BernsteinBasis (layers.py:83-126)
  const float* theta;       // learnable basis parameters: frequencies [R] (1) / pregamma [1] (2)
  float norm_const;         // Bessel: sqrt(2 / cutoff^3)
  const float* grho;        // parameter-gradient mode: adjoints of rho and drho, [2][E][R]
  float* contrib;           // parameter-gradient mode: per-(edge, k) contribution to dL/dtheta, [E][R]
};

__device__ __forceinline__ void rbf_envelope(const RbfArgs& q, float ds, float& env, float& denv) {
  env = 0.f; denv = 0.f;
  if (ds < 1.0f) {
    if (q.p > 0.f) {   // PolynomialEnvelope(exponent p)  (layers.py:14-33)
      const float pm1 = powf(ds, q.p - 1.0f);
      const float p0 = pm1 * ds, p1 = p0 * ds, p2 = p1 * ds;
      env = 1.0f + q.a * p0 + q.b * p1 + q.c * p2;
      denv = q.a * q.p * pm1 + q.b * (q.p + 1.0f) * p0 + q.c * (q.p + 2.0f) * p1;
    } else {           // ExponentialEnvelope: exp(-ds^2 / ((1 - ds)(1 + ds)))  (layers.py:36-48), encoded as exponent 0
      const float om = (1.0f - ds) * (1.0f + ds);
      env = expf(-(ds * ds) / om);
      denv = -env * 2.0f * ds / (om * om);
    }
  }
}
__device__ __forceinline__ float ipow(float x, int e) {   // x^e for integer e >= 0 (torch.pow with an integer exponent), 0^0 = 1
  float r = 1.f, b = x;
  for (int n = e; n > 0; n >>= 1) { if (n & 1) r *= b; b *= b; }
  return r;
}

// rho[e,k] = env(d/rc) * basis_k(d/rc) and d rho/d d  (oracle: painn_ref.radial_basis, painn_sweeps.rbf_and_derivative);
// GRAD = true: contrib[e,k] = grho * d rho/d theta + gdrho * d (d rho/d d)/d theta for the learnable basis parameter(s)
template <bool GRAD>
__global__ void k_rbf(RbfArgs q) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)q.E * q.R) return;
  const int e = (int)(idx / q.R), k = (int)(idx % q.R);
  const float d = q.geom[e].w;
  const float ds = d * q.inv_cutoff;
  float env, denv;
  rbf_envelope(q, ds, env, denv);
  if (q.type == 0) {
    // same rounding order as GaussianSmearing: exp(coeff * (ds - mu_k)^2), mu = the module's fp32 offset buffer
    const float diff = ds - q.mu[k];
    const float g = expf(q.coeff * (diff * diff));
    if (!GRAD) { q.rho[idx] = env * g; q.drho[idx] = q.inv_cutoff * g * (denv + env * (2.0f * q.coeff) * diff); }
  } else if (q.type == 1) {
    const float f = q.theta[k], nc = q.norm_const;
    float sn, cs;
    sincosf(f * ds, &sn, &cs);
    if (!GRAD) {
      q.rho[idx] = env * (nc / ds * sn);
      q.drho[idx] = q.inv_cutoff * (denv * nc / ds * sn + env * nc * (f * cs / ds - sn / (ds * ds)));
    } else {
      const long ER = (long)q.E * q.R;
      q.contrib[idx] = q.grho[idx] * env * nc * cs + q.grho[ER + idx] * q.inv_cutoff * nc * (denv * cs - env * f * sn);
    }
  } else {
    const float pg = q.theta[0];
    const float gam = fmaxf(pg, 0.f) + log1pf(expf(-fabsf(pg)));       // softplus
    const float x = expf(-gam * ds), y = 1.0f - x;
    const int m = q.R - 1, a = k, b = m - k;
    const float C = q.mu[k];                                            // binom(R-1, k): the module's `prefactor` buffer
    const float B = C * ipow(x, a) * ipow(y, b);
    const float dB = C * ((a > 0 ? a * ipow(x, a - 1) * ipow(y, b) : 0.f) - (b > 0 ? b * ipow(x, a) * ipow(y, b - 1) : 0.f));
    if (!GRAD) {
      q.rho[idx] = env * B;
      q.drho[idx] = q.inv_cutoff * (denv * B + env * dB * (-gam * x));
    } else {
      const float d2B = C * ((a > 1 ? a * (a - 1) * ipow(x, a - 2) * ipow(y, b) : 0.f) - (a > 0 && b > 0 ? 2.f * a * b * ipow(x, a - 1) * ipow(y, b - 1) : 0.f)
                             + (b > 1 ? b * (b - 1) * ipow(x, a) * ipow(y, b - 2) : 0.f));
      const float dsx = ds * x, sig = 1.0f / (1.0f + expf(-pg));
      const float drho_dg = env * dB * (-dsx);
      const float ddrho_dg = q.inv_cutoff * (denv * dB * (-dsx) + env * (-x * dB + gam * dsx * (dB + x * d2B)));
      const long ER = (long)q.E * q.R;
      q.contrib[idx] = (q.grho[idx] * drho_dg + q.grho[ER + idx] * ddrho_dg) * sig;
    }
  }
}

// ---------------------------------------------------------------------------------------------

// x_msg = x + sum_j xh_a[j] phi_a ;  vec_msg[c] = vec[c] + sum_j vec[j][c] (xh_b[j] phi_b) + (xh_c[j] phi_c) r[c]
__global__ void k_msg_fwd(MsgArgs q) {
  const int n = blockIdx.x, f = threadIdx.x, F = q.F, F3 = 3 * q.F;
  const int beg = q.g.row_ptr[n], end = q.g.row_ptr[n + 1];
  float dx = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f;
  for (int sp = beg; sp < end; ++sp) {
    const int k = q.g.col[sp];
    const float4 gm = q.g.geom[sp];
    const float* ph = q.PHI + (long)sp * F3;
    const float* xh = q.XH + (long)k * F3;
    const float* vk = q.V + (long)k * F3;
    const float ma = xh[f] * ph[f], mb = xh[F + f] * ph[F + f], mc = xh[2 * F + f] * ph[2 * F + f];
    dx += ma;
    d0 += vk[f] * mb + mc * gm.x;
    d1 += vk[F + f] * mb + mc * gm.y;
    d2 += vk[2 * F + f] * mb + mc * gm.z;
  }
  const long o = (long)n * F, o3 = (long)n * F3;
  q.XM[o + f] = q.X[o + f] + dx;
  q.VM[o3 + f] = q.V[o3 + f] + d0;
  q.VM[o3 + F + f] = q.V[o3 + F + f] + d1;
  q.VM[o3 + 2 * F + f] = q.V[o3 + 2 * F + f] + d2;
}

// JVP of k_msg_fwd along (TXH, TV, TD, TR); phi_dot = psi * t_d
__global__ void k_msg_tan(MsgArgs q) {
  const int n = blockIdx.x, f = threadIdx.x, F = q.F, F3 = 3 * q.F;
  const int beg = q.g.row_ptr[n], end = q.g.row_ptr[n + 1];
  float dx = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f;
  for (int sp = beg; sp < end; ++sp) {
    const int k = q.g.col[sp];
    const float4 gm = q.g.geom[sp];
    const float td = q.TD[sp];
    const float tr0 = q.TR[3 * (long)sp], tr1 = q.TR[3 * (long)sp + 1], tr2 = q.TR[3 * (long)sp + 2];
    const float* ph = q.PHI + (long)sp * F3;
    const float* ps = q.PSI + (long)sp * F3;
    const float* xh = q.XH + (long)k * F3;
    const float* txh = q.TXH + (long)k * F3;
    const float* vk = q.V + (long)k * F3;
    const float* tvk = q.TV + (long)k * F3;
    const float xa = xh[f], xb = xh[F + f], xc = xh[2 * F + f];
    const float pa = ph[f], pb = ph[F + f], pc = ph[2 * F + f];
    const float mb = xb * pb, mc = xc * pc;
    const float tma = txh[f] * pa + xa * (ps[f] * td);
    const float tmb = txh[F + f] * pb + xb * (ps[F + f] * td);
    const float tmc = txh[2 * F + f] * pc + xc * (ps[2 * F + f] * td);
    dx += tma;
    d0 += tvk[f] * mb + vk[f] * tmb + tmc * gm.x + mc * tr0;
    d1 += tvk[F + f] * mb + vk[F + f] * tmb + tmc * gm.y + mc * tr1;
    d2 += tvk[2 * F + f] * mb + vk[2 * F + f] * tmb + tmc * gm.z + mc * tr2;
  }
  const long o = (long)n * F, o3 = (long)n * F3;
  q.TXM[o + f] = q.TX[o + f] + dx;
  q.TVM[o3 + f] = q.TV[o3 + f] + d0;
  q.TVM[o3 + F + f] = q.TV[o3 + F + f] + d1;
  q.TVM[o3 + 2 * F + f] = q.TV[o3 + 2 * F + f] + d2;
}

// ---------------------------------------------------------------------------------------------

// Reverse of the message block at node n in its SOURCE role: loops over out-edges (n -> k).
//   DUAL=false  force adjoint: accumulates d(E)/d(d_e), d(E)/d(r_e) per out-edge into GEDGE
//   DUAL=true   second-order sweep: also adjoints of the tangents, and gphi/gpsi for the rbf_proj gradient
template <bool DUAL>
__global__ void k_msg_rev(MsgRevArgs q) {
  const int n = blockIdx.x, f = threadIdx.x, F = q.F, F3 = 3 * q.F;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int beg = q.g.row_ptr[n], end = q.g.row_ptr[n + 1];
  const long o3 = (long)n * F3;
  const float xa = q.XH[o3 + f], xb = q.XH[o3 + F + f], xc = q.XH[o3 + 2 * F + f];
  const float v0 = q.V[o3 + f], v1 = q.V[o3 + F + f], v2 = q.V[o3 + 2 * F + f];
  float txa = 0.f, txb = 0.f, txc = 0.f, tv0 = 0.f, tv1 = 0.f, tv2 = 0.f;
  if (DUAL) {
    txa = q.TXH[o3 + f]; txb = q.TXH[o3 + F + f]; txc = q.TXH[o3 + 2 * F + f];
    tv0 = q.TV[o3 + f]; tv1 = q.TV[o3 + F + f]; tv2 = q.TV[o3 + 2 * F + f];
  }
  float gxa = 0.f, gxb = 0.f, gxc = 0.f, gv0 = 0.f, gv1 = 0.f, gv2 = 0.f;        // adjoints of xh[n], vec[n]
  float gtxa = 0.f, gtxb = 0.f, gtxc = 0.f, gtv0 = 0.f, gtv1 = 0.f, gtv2 = 0.f;  // adjoints of t_xh[n], t_vec[n]
  float sba = 0.f, sbb = 0.f, sbc = 0.f;                                            // sum over this atom's edges of gphi
  for (int sp = beg; sp < end; ++sp) {
    const int k = q.g.col[sp];                       // target of the out-edge (n -> k)
    const float4 gm = q.g.geom[sp];
    const float r0 = -gm.x, r1 = -gm.y, r2 = -gm.z;  // unit vector of (n -> k)
    const float* ph = q.PHI + (long)sp * F3;
    const float pa = ph[f], pb = ph[F + f], pc = ph[2 * F + f];
    const float* A = q.GV + (long)k * F3;
    const float A0 = A[f], A1 = A[F + f], A2 = A[2 * F + f];
    const float gma = q.GX[(long)k * F + f];
    const float mb = xb * pb, mc = xc * pc;
    float gmb = A0 * v0 + A1 * v1 + A2 * v2;
    float gmc = A0 * r0 + A1 * r1 + A2 * r2;
    gv0 += A0 * mb; gv1 += A1 * mb; gv2 += A2 * mb;
    if (DUAL) {
      const float td = q.TD[sp];
      const float tr0 = -q.TR[3 * (long)sp], tr1 = -q.TR[3 * (long)sp + 1], tr2 = -q.TR[3 * (long)sp + 2];
      const float* ps = q.PSI + (long)sp * F3;
      const float tpa = ps[f] * td, tpb = ps[F + f] * td, tpc = ps[2 * F + f] * td;  // phi_dot
      const float* T = q.GTV + (long)k * F3;
      const float T0 = T[f], T1 = T[F + f], T2 = T[2 * F + f];
      const float gtma = q.GTX[(long)k * F + f];
      const float tmb = txb * pb + xb * tpb;
      gmb += T0 * tv0 + T1 * tv1 + T2 * tv2;
      const float gtmb = T0 * v0 + T1 * v1 + T2 * v2;
      gmc += T0 * tr0 + T1 * tr1 + T2 * tr2;
      const float gtmc = T0 * r0 + T1 * r1 + T2 * r2;
      gv0 += T0 * tmb; gv1 += T1 * tmb; gv2 += T2 * tmb;
      gtv0 += T0 * mb; gtv1 += T1 * mb; gtv2 += T2 * mb;
      gxa += gma * pa + gtma * tpa; gxb += gmb * pb + gtmb * tpb; gxc += gmc * pc + gtmc * tpc;
      gtxa += gtma * pa; gtxb += gtmb * pb; gtxc += gtmc * pc;
      float* gp = q.GPHI + (long)sp * F3;
      float* gs = q.GPSI + (long)sp * F3;
      const float ga = gma * xa + gtma * txa, gb = gmb * xb + gtmb * txb, gc = gmc * xc + gtmc * txc;
      gp[f] = ga; gp[F + f] = gb; gp[2 * F + f] = gc;
      sba += ga; sbb += gb; sbc += gc;
      gs[f] = gtma * xa * td; gs[F + f] = gtmb * xb * td; gs[2 * F + f] = gtmc * xc * td;
    } else {
      gxa += gma * pa; gxb += gmb * pb; gxc += gmc * pc;
      const float* ps = q.PSI + (long)sp * F3;
      // per-edge scalars reduced over the channels of this wavefront
      float gd = gma * xa * ps[f] + gmb * xb * ps[F + f] + gmc * xc * ps[2 * F + f];
      float e0 = A0 * mc, e1 = A1 * mc, e2 = A2 * mc;
      gd = nq_wave_sum(gd); e0 = nq_wave_sum(e0); e1 = nq_wave_sum(e1); e2 = nq_wave_sum(e2);
      if (lane == 0) {
        float4* dstp = q.GEDGE + (long)wave * q.g.E + sp;
        float4 acc = *dstp;
        acc.x += gd; acc.y += e0; acc.z += e1; acc.w += e2;
        *dstp = acc;
      }
    }
  }
  q.GXH[o3 + f] = gxa; q.GXH[o3 + F + f] = gxb; q.GXH[o3 + 2 * F + f] = gxc;
  q.GV_out[o3 + f] = q.GV[o3 + f] + gv0;
  q.GV_out[o3 + F + f] = q.GV[o3 + F + f] + gv1;
  q.GV_out[o3 + 2 * F + f] = q.GV[o3 + 2 * F + f] + gv2;
  if (DUAL) {
    q.GTXH[o3 + f] = gtxa; q.GTXH[o3 + F + f] = gtxb; q.GTXH[o3 + 2 * F + f] = gtxc;
    q.GBR[o3 + f] = sba; q.GBR[o3 + F + f] = sbb; q.GBR[o3 + 2 * F + f] = sbc;
    q.GTV_out[o3 + f] = q.GTV[o3 + f] + gtv0;
    q.GTV_out[o3 + F + f] = q.GTV[o3 + F + f] + gtv1;
    q.GTV_out[o3 + 2 * F + f] = q.GTV[o3 + 2 * F + f] + gtv2;
  }
}


// =============================================================================================
// Fused message kernels: the radial filter is evaluated inside the kernel.
//
// GaussianSmearing(0,1,R) has sigma = Delta = 1/(R-1): a basis function k contributes
// exp(-0.5 ((d/rc - mu_k)/Delta)^2) < 7e-10 once it is more than 6.5 Delta away, i.e. below fp32
// resolution of the sum.  So per edge only the WIN = 13 Gaussians around the nearest centre are
// evaluated: phi_e[f] = br[f] + sum_{t<13} WrT[k0+t][f] rho_{k0+t}(d_e)  (13 FMAs per output instead of a
// K=100 GEMM, 7.7x fewer flops) and phi/psi never touch HBM.
//
// Mapping: one persistent 1024-thread workgroup per CU keeps WrT (R x 3F fp32 = 153.6 kB at F=128,R=100)
// resident in the 160-kB LDS; the workgroup's 1024/F node slots (2 wavefronts each at F=128) walk CSR
// rows.  Per edge, lane t < 13 of every wavefront evaluates rho/drho for basis k0+t (one expf per wave
// and edge), v_readlane broadcasts them as scalars, and each thread reads its WrT column entries with
// conflict-free ds_read_b32 (consecutive channels -> consecutive banks).
// =============================================================================================
#define FUSED_THREADS 1024
// measured (profiles/r01_fused_tuning.txt): any VGPR spill in these loops costs 1.3-2x, so the two register-hungry flavours trade waves for registers
#ifndef NQ_DUAL2_THREADS
#define NQ_DUAL2_THREADS 512    // dual reverse, 2 channels/lane: at the 256-VGPR limit (768 threads: 31 spilled VGPRs, 8.3 ms vs 5.5-6.4 ms per step; 384: +15 %)
#endif
#ifndef NQ_TAN2_THREADS
#define NQ_TAN2_THREADS 768     // tangent, 2 channels/lane: 148 VGPRs (1024 threads: 26 spilled, 7.3 ms vs 3.35 ms per step)
#endif
// workgroup size per kernel flavour and channels-per-lane (register budget: 1024 thr -> 128 VGPRs, 768 -> 168, 512 -> 256)
#ifndef NQ_DUALNG2_THREADS
#define NQ_DUALNG2_THREADS 768  // dual reverse without the pair rows, 2 channels/lane: 165 VGPRs, no spill
#endif
// kind: 0 forward, 1 tangent, 2 force adjoint, 3 dual reverse, 4 dual reverse without pair rows.  Workgroup size = VGPR budget (one workgroup per CU: LDS holds WrT)
__host__ __device__ constexpr int fused_threads(int kind, int ch) {
  return ch >= 4 ? 512
                 : (ch == 2 ? (kind == 3 ? NQ_DUAL2_THREADS : (kind == 4 ? NQ_DUALNG2_THREADS : (kind == 1 ? NQ_TAN2_THREADS : 1024)))
                            : (kind >= 3 ? 768 : 1024));
}
#define FUSED_THREADS_DUAL 1024  // window records live in SGPRs (scalar loads), so the dual reverse also fits 16 waves per CU

__device__ __forceinline__ float bcast_lane(float v, int t) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), t));
}

// Per-edge window record, computed ONCE per step (the distances do not change between layers / sweeps):
// RW[e][0..12] = rho_{k0+t}(d_e), RW[e][13] = k0 (int bits), RW[e][16..28] = d rho_{k0+t} / d d.  128 B per edge.
// One thread per edge (the envelope, its powf and the window start are per-edge work: with 16 threads per edge every wavefront repeated them for 4 edges
// only), 13 taps in a loop, the 32-float record transposed through LDS so that the stores are whole contiguous rows.
#define RBFW_THREADS 256
__global__ __launch_bounds__(RBFW_THREADS) void k_rbf_window(const float4* __restrict__ geom, int E, FilterArgs fa, float* __restrict__ RW) {
  __shared__ float tile[RBFW_THREADS][RW_STRIDE + 1];
  const int e0 = blockIdx.x * RBFW_THREADS, e = e0 + threadIdx.x;
  if (e < E) {
    const int R = fa.R;
    const float dist = geom[e].w;
    const float ds = dist * fa.inv_cutoff;
    const int nwin = R < FWIN ? R : FWIN;
    int kc = (int)rintf(ds * (float)(R - 1));
    kc = min(max(kc, 0), R - 1);
    const int k0 = min(max(kc - FWIN / 2, 0), R - nwin);
    float beta = 1.f, dbeta = 0.f, env = 0.f, denv = 0.f;
    if (fa.mode == 0) {
      if (ds < 1.0f) {
        if (fa.p > 0.f) {
          const float pm1 = powf(ds, fa.p - 1.0f);
          const float p0 = pm1 * ds, p1 = p0 * ds, p2 = p1 * ds;
          env = 1.0f + fa.a * p0 + fa.b * p1 + fa.c * p2;
          denv = fa.a * fa.p * pm1 + fa.b * (fa.p + 1.0f) * p0 + fa.c * (fa.p + 2.0f) * p1;
        } else {   // ExponentialEnvelope (exponent 0, see k_rbf)
          const float om = (1.0f - ds) * (1.0f + ds);
          env = expf(-(ds * ds) / om);
          denv = -env * 2.0f * ds / (om * om);
        }
      }
    } else {
      // schnetpack: W_ij = fcut(d) * (filter_net(gauss(d)) + b)  ->  rho = fcut * gauss (unscaled d), bias multiplier beta = fcut
      const float arg = dist * (3.14159265358979323846f / fa.cutoff);
      const bool inside = dist < fa.cutoff;
      beta = inside ? 0.5f * (cosf(arg) + 1.0f) : 0.f;
      dbeta = inside ? -0.5f * (3.14159265358979323846f / fa.cutoff) * sinf(arg) : 0.f;
    }
    float* rw = tile[threadIdx.x];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      float rl = 0.f, drl = 0.f;
      if (t < nwin) {
        if (fa.mode == 0) {
          const float diff = ds - fa.mu[k0 + t];
          const float g = expf(fa.coeff * (diff * diff));
          rl = env * g;
          drl = fa.inv_cutoff * g * (denv + env * (2.0f * fa.coeff) * diff);
        } else {
          const float diff = dist - fa.mu[k0 + t];
          const float g = expf(fa.coeff * (diff * diff));
          if (fa.mode == 2) {   // SchNet: the filter network sees the bare Gaussians; fcut multiplies its OUTPUT (slots 14 / 30 carry fcut, fcut')
            rl = g;
            drl = g * (2.0f * fa.coeff) * diff;
          } else {
            rl = beta * g;
            drl = dbeta * g + beta * g * (2.0f * fa.coeff) * diff;
          }
        }
      }
      rw[t] = (t == 13) ? __int_as_float(k0) : (t == 14 ? beta : rl);    // [14] = bias multiplier, [30] = its derivative
      rw[16 + t] = (t == 14) ? dbeta : drl;
    }
  }
  __syncthreads();
  const long base = (long)e0 * RW_STRIDE;
  const long lim = (long)min(RBFW_THREADS, E - e0) * RW_STRIDE;
#pragma unroll 4
  for (int i = threadIdx.x; i < RBFW_THREADS * RW_STRIDE; i += RBFW_THREADS)
    if (i < lim) RW[base + i] = tile[i / RW_STRIDE][i % RW_STRIDE];
}

// ---- row preload: lane L of the wavefront holds index / geometry (/ tangents) of the row's edge L ------------
struct RowRegs { int kk; float gx, gy, gz, td, t0, t1, t2, pf; };
// The window records are streamed once per launch (128 B per edge: an HBM miss for every edge's scalar loads, issued only half an edge ahead because SMEM and
// LDS share a counter).  Lane L of the row preload therefore TOUCHES edge L's record with a vector load: the line is in L2 by the time the scalar loads of
// edges 1.. ask for it.  row_touch_done() keeps the load alive (its value is not used) and is placed after the row's last edge, where the wait is free.
// Measured (profiles/r06_message_kernel_prefetch_ab.txt): -0.8 ms per step over the four message kernels.  Touching the NEXT row's records one row early, touching
// every edge's record three edges ahead, and preloading the next row's index / geometry registers were all measured slower than this (a 4-MB L2 turns over
// within one row: a touch must be young; the row-start bubble is covered by the other wavefronts) and removed.
#ifndef NQ_RW_TOUCH
#define NQ_RW_TOUCH 1
#endif
__device__ __forceinline__ void row_touch_done(const RowRegs& r) {
#if NQ_RW_TOUCH
  asm volatile("" ::"v"(r.pf));
#endif
}

template <bool NEED_T>
__device__ __forceinline__ void load_row(RowRegs& r, const NqGraphView& g, const float* __restrict__ TD, const float* __restrict__ TR,
                                         const float* __restrict__ RW, int sp0, int cnt, int lane) {
  r.kk = 0; r.gx = r.gy = r.gz = 0.f; r.td = r.t0 = r.t1 = r.t2 = 0.f; r.pf = 0.f;
  if (lane < cnt) {
    const int sp = sp0 + lane;
#if NQ_RW_TOUCH
    r.pf = RW[(long)sp * RW_STRIDE + 15];
#endif
    r.kk = g.col[sp];
    const float4 gm = g.geom[sp];
    r.gx = gm.x; r.gy = gm.y; r.gz = gm.z;
    if (NEED_T) { r.td = TD[sp]; r.t0 = TR[3 * (long)sp]; r.t1 = TR[3 * (long)sp + 1]; r.t2 = TR[3 * (long)sp + 2]; }
  }
}
// phi (and psi) for this lane's CH channels of each of the three parts, from the LDS-resident WrT.
// Written on CH-wide vectors so that every FMA pair becomes one v_pk_fma_f32 (the scalar form left the psi half unpacked).
template <bool PSI, int CH>
__device__ __forceinline__ void filter_eval(const WinRegs<PSI>& w, const float* wrt, int F, int F3, int fb, const float (&bra)[CH],
                                            const float (&brb)[CH], const float (&brc)[CH], float (&pa)[CH], float (&pb)[CH], float (&pc)[CH],
                                            float (&qa)[CH], float (&qb)[CH], float (&qc)[CH]) {
  typedef VOps<CH> O;
  typedef typename O::V V;
  // bias enters as beta*b (phi) and beta'*b (psi); beta = 1, beta' = 0 in painn_pyg mode (exact)
  const V ba = O::from(bra), bb = O::from(brb), bc = O::from(brc);
  V va = ba * O::splat(w.rr[14]), vb = bb * O::splat(w.rr[14]), vc = bc * O::splat(w.rr[14]);
  V ua = O::splat(0.f), ub = ua, uc = ua;
  if (PSI) { ua = ba * O::splat(w.dd[14]); ub = bb * O::splat(w.dd[14]); uc = bc * O::splat(w.dd[14]); }
  const int k0 = __builtin_amdgcn_readfirstlane(__float_as_int(w.rr[13]));
  const float* wk = wrt + k0 * F3 + fb;
#pragma unroll
  for (int t = 0; t < FWIN; ++t) {  // always 13 taps: LDS rows >= R and window taps >= R are zero
    const V wa = lds_tap<CH>(wk + t * F3), wb = lds_tap<CH>(wk + t * F3 + F), wc = lds_tap<CH>(wk + t * F3 + 2 * F);
    const V r = O::splat(w.rr[t]);
    va = O::fma(wa, r, va); vb = O::fma(wb, r, vb); vc = O::fma(wc, r, vc);
    if (PSI) {
      const V d = O::splat(w.dd[t]);
      ua = O::fma(wa, d, ua); ub = O::fma(wb, d, ub); uc = O::fma(wc, d, uc);
    }
  }
  O::to(pa, va); O::to(pb, vb); O::to(pc, vc);
  O::to(qa, ua); O::to(qb, ub); O::to(qc, uc);
}

// Row scheduling of the fused message kernels.  Static: wavefront w of the XCD's sweep takes rows w, w + stride, ...  Claimed (kinds in
// NQ_CLAIM_KINDS, bit = kind 0 forward / 1 tangent / 2 force adjoint / 3 dual reverse): a wavefront takes the next NQ_CLAIM_ROWS
// unprocessed atoms of its XCD's range from a zeroed counter (FilterArgs::row_ctr), so all wavefronts of the XCD stay on one moving
// front of consecutive atoms; the claim for the following chunk is issued before the current one is processed.  Which wavefront
// computes a row does not change its result.  Measured (profiles/r01_fused_tuning.txt section 4): only the dual reverse gains (-8 %, one row per
// claim; larger chunks lose the balance at the tail), the short-row kernels pay for the same-address atomics -> default kinds = 8, rows = 1.
#ifndef NQ_CLAIM_KINDS
#define NQ_CLAIM_KINDS 15
#endif
#ifndef NQ_CLAIM_ROWS
#define NQ_CLAIM_ROWS 1
#endif
#ifndef NQ_CLAIM_GROUPS_DUAL
#define NQ_CLAIM_GROUPS_DUAL 1
#endif
#ifndef NQ_CLAIM_GROUPS
#define NQ_CLAIM_GROUPS 2   // counters (= sub-ranges, = separate fronts) per XCD: spreads the same-address atomics of the short-row kernels
#endif
#define FUSED_ROWS(KIND)                                                                        \
  constexpr bool claim__ = ((NQ_CLAIM_KINDS >> (KIND)) & 1) != 0;                               \
  const bool dyn__ = claim__ && fa.row_ctr != nullptr;   /* null counter pointer (small batches): static striding */ \
  constexpr int crows__ = claim__ ? NQ_CLAIM_ROWS : 1;                                          \
  constexpr int cgmax__ = ((KIND) == 3) ? NQ_CLAIM_GROUPS_DUAL : NQ_CLAIM_GROUPS;               \
  const int cgrp__ = max(1, min(cgmax__, (int)(gridDim.x / nxcd / nslices)));   /* every group needs a workgroup of its own on the XCD */ \
  const int grp__ = wg % cgrp__;                                                                \
  const int sub__ = (n_hi - x_lo + cgrp__ - 1) / cgrp__;                                        \
  const int s_lo__ = dyn__ ? x_lo + grp__ * sub__ : x_lo;                                       \
  const int s_hi__ = dyn__ ? min(n_hi, s_lo__ + sub__) : n_hi;                                  \
  int* const ctr__ = dyn__ ? fa.row_ctr + (((int)(blockIdx.x % nxcd) * nslices + slice) * cgmax__ + grp__) * NQ_ROWCTR_PAD : nullptr; \
  int n_static__ = x_lo + wg * nslots + slot;                                                   \
  auto claim_rows = [&]() __attribute__((always_inline)) -> int {                               \
    if (dyn__) {                                                                                \
      int v__ = 0;                                                                              \
      if (lane == 0) v__ = __hip_atomic_fetch_add(ctr__, crows__, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
      return s_lo__ + __builtin_amdgcn_readfirstlane(v__);                                      \
    }                                                                                           \
    const int v__ = n_static__; n_static__ += n_step; return v__;                               \
  };                                                                                            \
  for (int blk__ = claim_rows(), nxt__ = claim_rows(); blk__ < s_hi__; blk__ = nxt__, nxt__ = claim_rows()) \
    for (int n = blk__; n < min(blk__ + (dyn__ ? crows__ : 1), s_hi__); ++n)

// One wavefront per atom; lane l owns channels [l*CH, (l+1)*CH) of each part (F = 64*CH).
#define FUSED_PROLOGUE                                                                          \
  extern __shared__ __attribute__((aligned(16))) float wrt[];                                  \
  const int F = q.F, F3 = 3 * q.F;                                                             \
  constexpr int FL = 64 * CH;                  /* channels of one slice = one wavefront */      \
  const int nslices = F / FL;                                                                  \
  /* blocks are dealt round-robin to the 8 XCDs (blockIdx % 8); inside an XCD consecutive blocks take the slices of one atom group */ \
  const int nxcd = (gridDim.x % (8 * nslices)) == 0 ? 8 : 1;                                   \
  const int bx = (int)(blockIdx.x / nxcd);                                                     \
  const int slice = bx % nslices, wg = bx / nslices;                                           \
  {   /* LDS copy of this slice's columns of WrT: [Rpad][3][FL] */                              \
    const int Rp = fa.R < FWIN ? FWIN : fa.R;                                                  \
    const int per_row4 = (3 * FL) >> 2, fl4 = FL >> 2;                                         \
    float4* dst4 = reinterpret_cast<float4*>(wrt);                                             \
    for (int i = threadIdx.x; i < Rp * per_row4; i += blockDim.x) {                            \
      const int r = i / per_row4, c4 = i % per_row4, part = c4 / fl4, cc = c4 % fl4;           \
      dst4[i] = r < fa.R ? reinterpret_cast<const float4*>(fa.WRT + (long)r * F3 + part * F + slice * FL)[cc] : make_float4(0.f, 0.f, 0.f, 0.f); \
    }                                                                                          \
  }                                                                                            \
  __syncthreads();                                                                             \
  const int nslots = blockDim.x >> 6;                                                          \
  const int slot = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                           \
  const int lane = threadIdx.x & 63;                                                           \
  const int lfb = lane * CH;                   /* channel offset inside the LDS slice */        \
  const int fb = slice * FL + lfb;             /* channel offset in the global rows */           \
  float bra[CH], brb[CH], brc[CH];                                                             \
  ldv<CH>(bra, fa.br + fb); ldv<CH>(brb, fa.br + F + fb); ldv<CH>(brc, fa.br + 2 * F + fb);    \
  /* XCD-aware sweep: the workgroups of one XCD walk ONE contiguous eighth of the atoms together, wavefront by wavefront, so at     \
     any time an XCD works on ~6 molecules and their rows are fetched into that XCD's 4 MB L2 once. */                              \
  const int per_x = (q.g.N + nxcd - 1) / nxcd;                                                 \
  const int x_lo = (int)(blockIdx.x % nxcd) * per_x, n_hi = min(q.g.N, x_lo + per_x);          \
  const int n_step = (int)(gridDim.x / nxcd / nslices) * nslots;                               \
  /* Rows are CLAIMED, not strided: every wavefront of an XCD takes the next unprocessed atom of that XCD's range from a counter, so  \
     the wavefronts stay on one moving front of consecutive atoms (with static striding a wavefront that met short rows ran ahead by   \
     whole strides of ~400 atoms and the live set outgrew the L2).  The claim for the next row is issued before the current row is     \
     processed; which wavefront computes a row does not change its result.  */                                                        \
  /* rows: see FUSED_ROWS */

// ---- forward / tangent -------------------------------------------------------------------------------------
template <bool TAN, int CH>
struct FwdOps { float xa[CH], xb[CH], xc[CH], va[CH], vb[CH], vc[CH], txa[CH], txb[CH], txc[CH], tva[CH], tvb[CH], tvc[CH]; };

// Neighbour rows through buffer descriptors (lanes.h: ldv_buf): the row byte offsets are scalars, the lane offset one constant VGPR.
template <bool TAN>
struct FwdSrc {
  __amdgpu_buffer_rsrc_t xh, v, txh, tv;
  __device__ __forceinline__ FwdSrc(const MsgArgs& q) : xh(row_rsrc(q.XH)), v(row_rsrc(q.V)), txh(row_rsrc(TAN ? q.TXH : q.XH)), tv(row_rsrc(TAN ? q.TV : q.V)) {}
};
template <bool TAN, int CH>
__device__ __forceinline__ void load_fwd(FwdOps<TAN, CH>& o, const FwdSrc<TAN>& src, int k, int F, int F3, int fb) {
  const int ob = fb * 4, r0 = k * F3 * 4, r1 = r0 + F * 4, r2 = r1 + F * 4;
  ldv_buf<CH>(o.xa, src.xh, ob, r0); ldv_buf<CH>(o.xb, src.xh, ob, r1); ldv_buf<CH>(o.xc, src.xh, ob, r2);
  ldv_buf<CH>(o.va, src.v, ob, r0); ldv_buf<CH>(o.vb, src.v, ob, r1); ldv_buf<CH>(o.vc, src.v, ob, r2);
  if (TAN) {
    ldv_buf<CH>(o.txa, src.txh, ob, r0); ldv_buf<CH>(o.txb, src.txh, ob, r1); ldv_buf<CH>(o.txc, src.txh, ob, r2);
    ldv_buf<CH>(o.tva, src.tv, ob, r0); ldv_buf<CH>(o.tvb, src.tv, ob, r1); ldv_buf<CH>(o.tvc, src.tv, ob, r2);
  }
}

template <bool TAN, int CH>
__global__ __launch_bounds__(fused_threads(TAN ? 1 : 0, CH)) void k_msgf_fwd(MsgArgs q, FilterArgs fa, const float* __restrict__ RW) {
  FUSED_PROLOGUE
  const FwdSrc<TAN> src(q);
  FUSED_ROWS(TAN ? 1 : 0) {
    const int beg = __builtin_amdgcn_readfirstlane(q.g.row_ptr[n]), end = __builtin_amdgcn_readfirstlane(q.g.row_ptr[n + 1]);  // scalar
    float dx[CH], d0[CH], d1[CH], d2[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) dx[c] = d0[c] = d1[c] = d2[c] = 0.f;
    for (int c0 = beg; c0 < end; c0 += 64) {
      const int cnt = min(64, end - c0);
      RowRegs row;
      load_row<TAN>(row, q.g, q.TD, q.TR, RW, c0, cnt, lane);
      // Per edge: issue the NEXT edge's gathers, evaluate the filter of the current one from LDS, only then issue the next
      // window's scalar loads (SMEM and LDS share lgkmcnt: an outstanding s_load would turn every LDS wait of the filter
      // into a wait for the scalar cache), then the arithmetic.
      // Two operand sets in ping-pong (no register copies, so the wait for a gather sits at its first use one edge later).
      FwdOps<TAN, CH> opA, opB;
      load_fwd<TAN, CH>(opA, src, bl_i(row.kk, 0), F, F3, fb);
      WinRegs<TAN> win;               // ONE window register set: it is dead once the filter is evaluated
      load_win<TAN>(win, RW, c0);
      auto step = [&](FwdOps<TAN, CH>& cur, FwdOps<TAN, CH>& nxt, int j, auto prefetch) __attribute__((always_inline)) {
        const int jn = min(j + 1, cnt - 1);   // branch-free: past the end the last edge is re-loaded (keeps the loop one basic block)
        if (decltype(prefetch)::value) load_fwd<TAN, CH>(nxt, src, bl_i(row.kk, jn), F, F3, fb);
        float pa[CH], pb[CH], pc[CH], qa[CH], qb[CH], qc[CH];
        filter_eval<TAN, CH>(win, wrt, FL, 3 * FL, lfb, bra, brb, brc, pa, pb, pc, qa, qb, qc);
        __builtin_amdgcn_sched_barrier(0);
        if (decltype(prefetch)::value) load_win<TAN>(win, RW, c0 + jn);
        const float gx = bl_f(row.gx, j), gy = bl_f(row.gy, j), gz = bl_f(row.gz, j);
        float td = 0.f, tr0 = 0.f, tr1 = 0.f, tr2 = 0.f;
        if (TAN) { td = bl_f(row.td, j); tr0 = bl_f(row.t0, j); tr1 = bl_f(row.t1, j); tr2 = bl_f(row.t2, j); }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const float mb = cur.xb[c] * pb[c], mc = cur.xc[c] * pc[c];
          if (!TAN) {
            dx[c] += cur.xa[c] * pa[c];
            d0[c] += cur.va[c] * mb + mc * gx;
            d1[c] += cur.vb[c] * mb + mc * gy;
            d2[c] += cur.vc[c] * mb + mc * gz;
          } else {
            const float tma = cur.txa[c] * pa[c] + cur.xa[c] * (qa[c] * td);
            const float tmb = cur.txb[c] * pb[c] + cur.xb[c] * (qb[c] * td);
            const float tmc = cur.txc[c] * pc[c] + cur.xc[c] * (qc[c] * td);
            dx[c] += tma;
            d0[c] += cur.tva[c] * mb + cur.va[c] * tmb + tmc * gx + mc * tr0;
            d1[c] += cur.tvb[c] * mb + cur.vb[c] * tmb + tmc * gy + mc * tr1;
            d2[c] += cur.tvc[c] * mb + cur.vc[c] * tmb + tmc * gz + mc * tr2;
          }
        }
      };
      const int pairs = cnt & ~1;
#pragma nounroll
      for (int j = 0; j < pairs; j += 2) {
        step(opA, opB, j, std::true_type());
        step(opB, opA, j + 1, std::true_type());
      }
      if (cnt & 1) step(opA, opB, cnt - 1, std::false_type());
      row_touch_done(row);
    }
    const long o = (long)n * F + fb, o3 = (long)n * F3 + fb;
    float x0[CH], w0[CH], w1[CH], w2[CH];
    if (!TAN) {
      ldv<CH>(x0, q.X + o); ldv<CH>(w0, q.V + o3); ldv<CH>(w1, q.V + o3 + F); ldv<CH>(w2, q.V + o3 + 2 * F);
    } else {
      ldv<CH>(x0, q.TX + o); ldv<CH>(w0, q.TV + o3); ldv<CH>(w1, q.TV + o3 + F); ldv<CH>(w2, q.TV + o3 + 2 * F);
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) { x0[c] += dx[c]; w0[c] += d0[c]; w1[c] += d1[c]; w2[c] += d2[c]; }
    if (!TAN) {
      stv<CH>(q.XM + o, x0); stv<CH>(q.VM + o3, w0); stv<CH>(q.VM + o3 + F, w1); stv<CH>(q.VM + o3 + 2 * F, w2);
    } else {
      stv<CH>(q.TXM + o, x0); stv<CH>(q.TVM + o3, w0); stv<CH>(q.TVM + o3 + F, w1); stv<CH>(q.TVM + o3 + 2 * F, w2);
    }
  }
}

// ---- reverse (force adjoint / dual) ----------------------------------------------------------------------------
#ifndef NQ_ABLATE
#define NQ_ABLATE 0   // development only (scripts/ablate.sh): 1 no gphi/gpsi stores, 2 no filter evaluation, 3 gathers hit the own row, 4 fixed window
#endif
#if NQ_ABLATE == 3
#define ABL_K(j) n
#else
#define ABL_K(j) bl_i(row.kk, j)
#endif
#if NQ_ABLATE == 4
#define ABL_SP(x) 0
#else
#define ABL_SP(x) (x)
#endif
template <bool DUAL, int CH>
struct RevOps { float A0[CH], A1[CH], A2[CH], gma[CH], T0[CH], T1[CH], T2[CH], gtma[CH]; };

template <bool DUAL>
struct RevSrc {
  __amdgpu_buffer_rsrc_t gv, gx, gtv, gtx;
  __device__ __forceinline__ RevSrc(const MsgRevArgs& q) : gv(row_rsrc(q.GV)), gx(row_rsrc(q.GX)), gtv(row_rsrc(DUAL ? q.GTV : q.GV)), gtx(row_rsrc(DUAL ? q.GTX : q.GX)) {}
};
template <bool DUAL, int CH>
__device__ __forceinline__ void load_rev(RevOps<DUAL, CH>& o, const RevSrc<DUAL>& src, int k, int F, int F3, int fb) {
  const int ob = fb * 4, r0 = k * F3 * 4, r1 = r0 + F * 4, r2 = r1 + F * 4, rx = k * F * 4;   // (see load_fwd)
  ldv_buf<CH>(o.A0, src.gv, ob, r0); ldv_buf<CH>(o.A1, src.gv, ob, r1); ldv_buf<CH>(o.A2, src.gv, ob, r2);
  ldv_buf<CH>(o.gma, src.gx, ob, rx);
  if (DUAL) {
    ldv_buf<CH>(o.T0, src.gtv, ob, r0); ldv_buf<CH>(o.T1, src.gtv, ob, r1); ldv_buf<CH>(o.T2, src.gtv, ob, r2);
    ldv_buf<CH>(o.gtma, src.gtx, ob, rx);
  }
}

// Dual flavour and the rbf_proj gradient: gphi / gpsi of the two directions of a pair (n -> k and k -> n) multiply the SAME window
// (d is symmetric), so only their sum is needed.  The row of atom n therefore stores, for its LOWER neighbours k < n (the first
// lowptr[n+1] - lowptr[n] edges of the ascending row), gphi_nk + gphi_kn at the slot of edge (n -> k), and nothing for k > n: half the
// stores here and half the reads in k_gwr_sorted (which visits lower slots only).  The reverse direction needs the primal / tangent rows of k
// (gathered: L2 hits, the adjoint rows of k are being fetched anyway) and the adjoint rows of n (row-resident).
#ifndef NQ_DUAL_NA_RESIDENT
#define NQ_DUAL_NA_RESIDENT 1   // adjoint rows of n kept in registers for the whole row (1) or re-read from L1 per lower edge (0)
#endif
#ifndef NQ_DUAL_MIRROR
#define NQ_DUAL_MIRROR 1        // odd lower-edge count: mirrored copy of the upper loop (1) or one operand-set copy per row chunk (0)
#endif
#ifndef NQ_DUAL_KX_EARLY
#define NQ_DUAL_KX_EARLY 1      // neighbour primal / tangent rows requested at the start of the step (1) or after the filter evaluation (0)
#endif
// GW (dual only): this kernel also produces the pair rows gphi / gpsi and the per-atom bias sums for the rbf_proj gradient (rounds 1-4, still the path
// for molecules that do not fit the LDS of molpair.hip); GW = false: the gradient is recomputed from node rows by k_gwr_mol, nothing is stored per pair.
// LITE (dual, compile time): the adjoints of t_xh / t_vec are not accumulated at all (MsgRevArgs::lite at run time only skips their stores)
template <bool DUAL, int CH, bool GW, bool LITE = false>
__device__ __forceinline__ void msgf_rev_body(const MsgRevArgs& q, const FilterArgs& fa, const float* __restrict__ RW) {
  FUSED_PROLOGUE
  const RevSrc<DUAL> src(q);
  FUSED_ROWS(DUAL ? 3 : 2) {
    if (DUAL && q.row_filter) {   // mixed batches: the pair-row flavour takes the rows of the molecules that do not fit the LDS of k_gwr_mol, the other flavour the rest
      const int mm = __builtin_amdgcn_readfirstlane(q.g.atom_mol[n]);
      const bool big = __builtin_amdgcn_readfirstlane(q.g.mol_ptr[mm + 1]) - __builtin_amdgcn_readfirstlane(q.g.mol_ptr[mm]) > q.mol_cap;
      if (big != (q.row_filter == 2)) continue;
    }
    const int beg = __builtin_amdgcn_readfirstlane(q.g.row_ptr[n]), end = __builtin_amdgcn_readfirstlane(q.g.row_ptr[n + 1]);  // scalar
    const int nlow = (DUAL && GW) ? __builtin_amdgcn_readfirstlane(q.g.lowptr[n + 1]) - __builtin_amdgcn_readfirstlane(q.g.lowptr[n]) : 0;
    const long o3 = (long)n * F3 + fb;
    float xa[CH], xb[CH], xc[CH], v0[CH], v1[CH], v2[CH], txa[CH], txb[CH], txc[CH], tv0[CH], tv1[CH], tv2[CH];
    ldv<CH>(xa, q.XH + o3); ldv<CH>(xb, q.XH + o3 + F); ldv<CH>(xc, q.XH + o3 + 2 * F);
    ldv<CH>(v0, q.V + o3); ldv<CH>(v1, q.V + o3 + F); ldv<CH>(v2, q.V + o3 + 2 * F);
    float nA0[CH], nA1[CH], nA2[CH], ngma[CH], nT0[CH], nT1[CH], nT2[CH], ngtma[CH];   // adjoint rows of n itself (dual: reverse-direction gphi)
    if (DUAL) {
      ldv<CH>(txa, q.TXH + o3); ldv<CH>(txb, q.TXH + o3 + F); ldv<CH>(txc, q.TXH + o3 + 2 * F);
      ldv<CH>(tv0, q.TV + o3); ldv<CH>(tv1, q.TV + o3 + F); ldv<CH>(tv2, q.TV + o3 + 2 * F);
#if NQ_DUAL_NA_RESIDENT
      if (GW) {
      ldv<CH>(nA0, q.GV + o3); ldv<CH>(nA1, q.GV + o3 + F); ldv<CH>(nA2, q.GV + o3 + 2 * F);
      ldv<CH>(nT0, q.GTV + o3); ldv<CH>(nT1, q.GTV + o3 + F); ldv<CH>(nT2, q.GTV + o3 + 2 * F);
      ldv<CH>(ngma, q.GX + (long)n * F + fb); ldv<CH>(ngtma, q.GTX + (long)n * F + fb);
      }
#endif
    } else {
#pragma unroll
      for (int c = 0; c < CH; ++c) txa[c] = txb[c] = txc[c] = tv0[c] = tv1[c] = tv2[c] = 0.f;
    }
    float gxa[CH], gxb[CH], gxc[CH], gv0[CH], gv1[CH], gv2[CH], gtxa[CH], gtxb[CH], gtxc[CH], gtv0[CH], gtv1[CH], gtv2[CH], sba[CH], sbb[CH], sbc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      gxa[c] = gxb[c] = gxc[c] = gv0[c] = gv1[c] = gv2[c] = 0.f;
      gtxa[c] = gtxb[c] = gtxc[c] = gtv0[c] = gtv1[c] = gtv2[c] = 0.f;
      sba[c] = sbb[c] = sbc[c] = 0.f;
    }
    for (int c0 = beg; c0 < end; c0 += 64) {
      const int cnt = min(64, end - c0);
      const int nlo = (DUAL && GW) ? max(0, min(cnt, nlow - (c0 - beg))) : 0;   // lower edges of this chunk come first
      RowRegs row;
      load_row<DUAL>(row, q.g, q.TD, q.TR, RW, c0, cnt, lane);
      RevOps<DUAL, CH> opA, opB;   // ping-pong operands; scalar window loads are issued after the filter's LDS reads (see k_msgf_fwd)
      load_rev<DUAL, CH>(opA, src, ABL_K(0), F, F3, fb);
      WinRegs<true> win;
      load_win<true>(win, RW, ABL_SP(c0));
      float4 eacc = make_float4(0.f, 0.f, 0.f, 0.f);
      // LOW (dual only): this edge stores gphi / gpsi of both directions; otherwise the dual flavour stores nothing for the edge
      auto step = [&](RevOps<DUAL, CH>& cur, RevOps<DUAL, CH>& nxt, int j, auto low_tag) __attribute__((always_inline)) {
        constexpr bool LOW = decltype(low_tag)::value;
        const int sp = c0 + j;
        const int jn = min(j + 1, cnt - 1);   // branch-free: past the end the last edge is re-loaded
        load_rev<DUAL, CH>(nxt, src, ABL_K(jn), F, F3, fb);
        float kxa[CH], kxb[CH], kxc[CH], kv0[CH], kv1[CH], kv2[CH], ktxa[CH], ktxb[CH], ktxc[CH], ktv0[CH], ktv1[CH], ktv2[CH];
        if (DUAL && GW && LOW && NQ_DUAL_KX_EARLY) {
          const long k3 = (long)ABL_K(j) * F3 + fb;
          ldv<CH>(kxa, q.XH + k3); ldv<CH>(kxb, q.XH + k3 + F); ldv<CH>(kxc, q.XH + k3 + 2 * F);
          ldv<CH>(kv0, q.V + k3); ldv<CH>(kv1, q.V + k3 + F); ldv<CH>(kv2, q.V + k3 + 2 * F);
          ldv<CH>(ktxa, q.TXH + k3); ldv<CH>(ktxb, q.TXH + k3 + F); ldv<CH>(ktxc, q.TXH + k3 + 2 * F);
          ldv<CH>(ktv0, q.TV + k3); ldv<CH>(ktv1, q.TV + k3 + F); ldv<CH>(ktv2, q.TV + k3 + 2 * F);
        }
        float pa[CH], pb[CH], pc[CH], qa[CH], qb[CH], qc[CH];
#if NQ_ABLATE == 2
#pragma unroll
        for (int c = 0; c < CH; ++c) { pa[c] = bra[c] * win.rr[0]; pb[c] = brb[c] * win.rr[1]; pc[c] = brc[c] * win.rr[2]; qa[c] = bra[c] * win.dd[0]; qb[c] = brb[c] * win.dd[1]; qc[c] = brc[c] * win.dd[2]; }
#else
        filter_eval<true, CH>(win, wrt, FL, 3 * FL, lfb, bra, brb, brc, pa, pb, pc, qa, qb, qc);
#endif
        const float beta = win.rr[14], dbeta = win.dd[14];
        __builtin_amdgcn_sched_barrier(0);
        load_win<true>(win, RW, ABL_SP(c0 + jn));
        if (DUAL && GW && LOW && !NQ_DUAL_KX_EARLY) {   // primal / tangent rows of the neighbour: issued here, consumed after the main block of the step
          const long k3 = (long)ABL_K(j) * F3 + fb;
          ldv<CH>(kxa, q.XH + k3); ldv<CH>(kxb, q.XH + k3 + F); ldv<CH>(kxc, q.XH + k3 + 2 * F);
          ldv<CH>(kv0, q.V + k3); ldv<CH>(kv1, q.V + k3 + F); ldv<CH>(kv2, q.V + k3 + 2 * F);
          ldv<CH>(ktxa, q.TXH + k3); ldv<CH>(ktxb, q.TXH + k3 + F); ldv<CH>(ktxc, q.TXH + k3 + 2 * F);
          ldv<CH>(ktv0, q.TV + k3); ldv<CH>(ktv1, q.TV + k3 + F); ldv<CH>(ktv2, q.TV + k3 + 2 * F);
        }
        const float r0 = -bl_f(row.gx, j), r1 = -bl_f(row.gy, j), r2 = -bl_f(row.gz, j);  // unit vector of the out-edge (n -> k)
        float td = 0.f, tr0 = 0.f, tr1 = 0.f, tr2 = 0.f;
        if (DUAL) { td = bl_f(row.td, j); tr0 = -bl_f(row.t0, j); tr1 = -bl_f(row.t1, j); tr2 = -bl_f(row.t2, j); }
        float gd = 0.f, e0 = 0.f, e1 = 0.f, e2 = 0.f;
        float ga[CH], gb[CH], gc[CH], ha[CH], hb[CH], hc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const float A0 = cur.A0[c], A1 = cur.A1[c], A2 = cur.A2[c], gma = cur.gma[c];
          const float mb = xb[c] * pb[c], mc = xc[c] * pc[c];
          float gmb = A0 * v0[c] + A1 * v1[c] + A2 * v2[c];
          float gmc = A0 * r0 + A1 * r1 + A2 * r2;
          gv0[c] += A0 * mb; gv1[c] += A1 * mb; gv2[c] += A2 * mb;
          if (DUAL) {
            const float tpa = qa[c] * td, tpb = qb[c] * td, tpc = qc[c] * td;
            const float T0 = cur.T0[c], T1 = cur.T1[c], T2 = cur.T2[c], gtma = cur.gtma[c];
            const float tmb = txb[c] * pb[c] + xb[c] * tpb;
            gmb += T0 * tv0[c] + T1 * tv1[c] + T2 * tv2[c];
            const float gtmb = T0 * v0[c] + T1 * v1[c] + T2 * v2[c];
            gmc += T0 * tr0 + T1 * tr1 + T2 * tr2;
            const float gtmc = T0 * r0 + T1 * r1 + T2 * r2;
            gv0[c] += T0 * tmb; gv1[c] += T1 * tmb; gv2[c] += T2 * tmb;
            if (!LITE) { gtv0[c] += T0 * mb; gtv1[c] += T1 * mb; gtv2[c] += T2 * mb; }
            gxa[c] += gma * pa[c] + gtma * tpa; gxb[c] += gmb * pb[c] + gtmb * tpb; gxc[c] += gmc * pc[c] + gtmc * tpc;
            if (!LITE) { gtxa[c] += gtma * pa[c]; gtxb[c] += gtmb * pb[c]; gtxc[c] += gtmc * pc[c]; }
            if (GW) {
              ga[c] = gma * xa[c] + gtma * txa[c]; gb[c] = gmb * xb[c] + gtmb * txb[c]; gc[c] = gmc * xc[c] + gtmc * txc[c];
              ha[c] = gtma * xa[c] * td; hb[c] = gtmb * xb[c] * td; hc[c] = gtmc * xc[c] * td;
              sba[c] += ga[c] * beta + ha[c] * dbeta;
              sbb[c] += gb[c] * beta + hb[c] * dbeta;
              sbc[c] += gc[c] * beta + hc[c] * dbeta;
            }
          } else {
            gxa[c] += gma * pa[c]; gxb[c] += gmb * pb[c]; gxc[c] += gmc * pc[c];
            gd += gma * xa[c] * qa[c] + gmb * xb[c] * qb[c] + gmc * xc[c] * qc[c];
            e0 += A0 * mc; e1 += A1 * mc; e2 += A2 * mc;
          }
        }
        if (DUAL && GW && LOW && NQ_ABLATE != 1) {
          // direction k -> n: roles swapped, unit vector and its tangent negated, t_d unchanged.  The adjoint rows of n are the same
          // addresses for every edge of the row (L1 hits): loaded only now, so that they do not occupy registers during the block above
#if !NQ_DUAL_NA_RESIDENT
          __builtin_amdgcn_sched_barrier(0);
          ldv<CH>(nA0, q.GV + o3); ldv<CH>(nA1, q.GV + o3 + F); ldv<CH>(nA2, q.GV + o3 + 2 * F);
          ldv<CH>(nT0, q.GTV + o3); ldv<CH>(nT1, q.GTV + o3 + F); ldv<CH>(nT2, q.GTV + o3 + 2 * F);
          ldv<CH>(ngma, q.GX + (long)n * F + fb); ldv<CH>(ngtma, q.GTX + (long)n * F + fb);
#endif
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const float rgmb = nA0[c] * kv0[c] + nA1[c] * kv1[c] + nA2[c] * kv2[c] + (nT0[c] * ktv0[c] + nT1[c] * ktv1[c] + nT2[c] * ktv2[c]);
            const float rgtmb = nT0[c] * kv0[c] + nT1[c] * kv1[c] + nT2[c] * kv2[c];
            const float rgmc = -(nA0[c] * r0 + nA1[c] * r1 + nA2[c] * r2) - (nT0[c] * tr0 + nT1[c] * tr1 + nT2[c] * tr2);
            const float rgtmc = -(nT0[c] * r0 + nT1[c] * r1 + nT2[c] * r2);
            ga[c] += ngma[c] * kxa[c] + ngtma[c] * ktxa[c]; gb[c] += rgmb * kxb[c] + rgtmb * ktxb[c]; gc[c] += rgmc * kxc[c] + rgtmc * ktxc[c];
            ha[c] += ngtma[c] * kxa[c] * td; hb[c] += rgtmb * kxb[c] * td; hc[c] += rgtmc * kxc[c] * td;
          }
          float* gp = q.GPHI + (long)sp * F3 + fb;
          float* gs = q.GPSI + (long)sp * F3 + fb;
          stv_stream<CH>(gp, ga); stv_stream<CH>(gp + F, gb); stv_stream<CH>(gp + 2 * F, gc);
          stv_stream<CH>(gs, ha); stv_stream<CH>(gs + F, hb); stv_stream<CH>(gs + 2 * F, hc);
        } else if (!DUAL) {
          gd = nq_wave_sum(gd); e0 = nq_wave_sum(e0); e1 = nq_wave_sum(e1); e2 = nq_wave_sum(e2);
          // lane j keeps the four reduced scalars of edge j; one coalesced float4 update per CSR row chunk below
          const bool mine = lane == j;
          eacc.x = mine ? gd : eacc.x; eacc.y = mine ? e0 : eacc.y; eacc.z = mine ? e1 : eacc.z; eacc.w = mine ? e2 : eacc.w;
        }
      };
      // edges [0, nlo) with the LOW flavour, [nlo, cnt) without; the ping-pong parity carries over the boundary
      int j = 0;
#pragma nounroll
      for (; j + 1 < nlo; j += 2) {
        step(opA, opB, j, std::true_type());
        step(opB, opA, j + 1, std::true_type());
      }
#if NQ_DUAL_MIRROR
      if (j < nlo) {   // odd number of lower edges: the next edge's operands sit in opB -> mirrored copy of the loop
        step(opA, opB, j, std::true_type());
        ++j;
#pragma nounroll
        for (; j + 1 < cnt; j += 2) {
          step(opB, opA, j, std::false_type());
          step(opA, opB, j + 1, std::false_type());
        }
        if (j < cnt) step(opB, opA, j, std::false_type());
      } else {
#pragma nounroll
        for (; j + 1 < cnt; j += 2) {
          step(opA, opB, j, std::false_type());
          step(opB, opA, j + 1, std::false_type());
        }
        if (j < cnt) step(opA, opB, j, std::false_type());
      }
#else
      if (j < nlo) {   // odd number of lower edges: the next edge's operands land in opB -> move them over (one copy per row chunk)
        step(opA, opB, j, std::true_type());
        ++j;
        opA = opB;
      }
#pragma nounroll
      for (; j + 1 < cnt; j += 2) {
        step(opA, opB, j, std::false_type());
        step(opB, opA, j + 1, std::false_type());
      }
      if (j < cnt) step(opA, opB, j, std::false_type());
#endif
      row_touch_done(row);
      if (!DUAL && lane < cnt) {
        float4* dstp = q.GEDGE + (long)slice * q.g.E + c0 + lane;   // one accumulator plane per channel slice (summed by k_geom_rev)
        float4 acc = *dstp;
        acc.x += eacc.x; acc.y += eacc.y; acc.z += eacc.z; acc.w += eacc.w;
        *dstp = acc;
      }
    }
    float g0[CH], g1[CH], g2[CH];
    stv<CH>(q.GXH + o3, gxa); stv<CH>(q.GXH + o3 + F, gxb); stv<CH>(q.GXH + o3 + 2 * F, gxc);
    ldv<CH>(g0, q.GV + o3); ldv<CH>(g1, q.GV + o3 + F); ldv<CH>(g2, q.GV + o3 + 2 * F);
#pragma unroll
    for (int c = 0; c < CH; ++c) { g0[c] += gv0[c]; g1[c] += gv1[c]; g2[c] += gv2[c]; }
    stv<CH>(q.GV_out + o3, g0); stv<CH>(q.GV_out + o3 + F, g1); stv<CH>(q.GV_out + o3 + 2 * F, g2);
    if (DUAL) {
      if (GW) { stv<CH>(q.GBR + o3, sba); stv<CH>(q.GBR + o3 + F, sbb); stv<CH>(q.GBR + o3 + 2 * F, sbc); }
      if (!LITE && !q.lite) {   // (lite: the adjoints of t_xh / t_vec are the force sweep's gxh / gvec_out of this layer, already stored)
        stv<CH>(q.GTXH + o3, gtxa); stv<CH>(q.GTXH + o3 + F, gtxb); stv<CH>(q.GTXH + o3 + 2 * F, gtxc);
        ldv<CH>(g0, q.GTV + o3); ldv<CH>(g1, q.GTV + o3 + F); ldv<CH>(g2, q.GTV + o3 + 2 * F);
#pragma unroll
        for (int c = 0; c < CH; ++c) { g0[c] += gtv0[c]; g1[c] += gtv1[c]; g2[c] += gtv2[c]; }
        stv<CH>(q.GTV_out + o3, g0); stv<CH>(q.GTV_out + o3 + F, g1); stv<CH>(q.GTV_out + o3 + 2 * F, g2);
      }
    }
  }
}


template <bool DUAL, int CH>
__global__ __launch_bounds__(fused_threads(DUAL ? 3 : 2, CH)) void k_msgf_rev(MsgRevArgs q, FilterArgs fa, const float* __restrict__ RW) {
  msgf_rev_body<DUAL, CH, DUAL>(q, fa, RW);
}
template <bool LITE, int CH>   // always the dual flavour; LITE = the stored-tangent-adjoint form of engine.hip (the default), false = the full dual sweep
__global__ __launch_bounds__(fused_threads(4, CH)) void k_msgf_rev_nopair(MsgRevArgs q, FilterArgs fa, const float* __restrict__ RW) {
  msgf_rev_body<true, CH, false, LITE>(q, fa, RW);
}

// =============================================================================================
// Gradient of rbf_proj.weight with the windowed filter, register-resident and deterministic.
//   gWr[c][k] = sum_e gphi[e][c] rho_k(d_e) + gpsi[e][c] drho_k(d_e),  k in [k0_e, k0_e + 13)
// Edges are visited in k0-sorted order (stable counting sort, once per step: the windows depend only on the
// geometry), so the 13-tap window of a wavefront's edge stream only ever slides upwards: every lane keeps 16
// accumulators per owned column in registers, flushes the row that leaves the window, and never touches LDS or
// atomics.  26 FMAs per (edge, column) instead of a K = 2E GEMM against all R basis functions (7.7x fewer flops).
// One wavefront = (one contiguous chunk of the sorted edge list) x (one 64*CH-column slice = one filter part);
// per-chunk partial rows are combined in chunk order -> bitwise reproducible.
// =============================================================================================
#define SORT_CHUNK 256
// pass A: per 256-edge chunk, histogram over k0 and the stable local rank of every edge inside its bin
// row_of / col (optional): only LOWER slots (col[e] < row_of[e]) take part -- the PaiNN dual sweep stores one gphi / gpsi row per pair
struct K0Filter { const int* row_of; const int* col; const int* mol_ptr; const int* atom_mol; int cap; };   // cap > 0: only molecules of more than cap atoms
__device__ __forceinline__ int k0_key(const float* __restrict__ RW, int e, int E, const K0Filter& f) {
  if (e >= E) return -1;
  if (f.row_of && f.col[e] >= f.row_of[e]) return -1;
  if (f.cap > 0) { const int m = f.atom_mol[f.row_of[e]]; if (f.mol_ptr[m + 1] - f.mol_ptr[m] <= f.cap) return -1; }
  return __float_as_int(RW[(long)e * RW_STRIDE + 13]);
}
__global__ __launch_bounds__(SORT_CHUNK) void k_k0_hist(const float* __restrict__ RW, int E, int nbins, int* __restrict__ chunk_hist,
                                                        int* __restrict__ local_rank, K0Filter flt) {
  __shared__ int keys[SORT_CHUNK];
  __shared__ int hist[256];
  const int e = blockIdx.x * SORT_CHUNK + threadIdx.x;
  const int key = k0_key(RW, e, E, flt);
  keys[threadIdx.x] = key;
  if (threadIdx.x < nbins) hist[threadIdx.x] = 0;
  __syncthreads();
  if (key >= 0) {
    int r = 0;
    for (int i = 0; i < (int)threadIdx.x; ++i) r += (keys[i] == key);
    local_rank[e] = r;
    atomicAdd(&hist[key], 1);   // integer atomics: order-independent result
  }
  __syncthreads();
  if (threadIdx.x < nbins) chunk_hist[(long)threadIdx.x * gridDim.x + blockIdx.x] = hist[threadIdx.x];   // [bin][chunk]
}
// pass B: one wavefront per bin scans that bin's chunk counts (shuffle scan, 64 chunks per step) in place -> exclusive
// offsets inside the bin; bin totals go to bin_total[].
__global__ __launch_bounds__(256) void k_k0_scan(int* __restrict__ hist, int nchunks, int nbins, int* __restrict__ bin_total) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= nbins) return;
  int* row = hist + (long)b * nchunks;
  int carry = 0;
  for (int c0 = 0; c0 < nchunks; c0 += 64) {
    const int c = c0 + lane;
    const int v = c < nchunks ? row[c] : 0;
    int sc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(sc, off, 64);
      if (lane >= off) sc += t;
    }
    if (c < nchunks) row[c] = carry + sc - v;
    carry += __shfl(sc, 63, 64);
  }
  if (lane == 0) bin_total[b] = carry;
}
// pass C: order[bin_base + offset_in_bin(chunk) + local_rank] = edge slot
__global__ __launch_bounds__(SORT_CHUNK) void k_k0_scatter(const float* __restrict__ RW, int E, int nbins, const int* __restrict__ hist,
                                                           const int* __restrict__ bin_total, const int* __restrict__ local_rank,
                                                           int* __restrict__ order, K0Filter flt, int* __restrict__ count_out) {
  __shared__ int base[256];
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < nbins; ++i) { base[i] = run; run += bin_total[i]; }
    if (count_out && blockIdx.x == 0) *count_out = run;
  }
  __syncthreads();
  const int e = blockIdx.x * SORT_CHUNK + threadIdx.x;
  const int key = k0_key(RW, e, E, flt);
  if (key < 0) return;
  order[base[key] + hist[(long)key * gridDim.x + blockIdx.x] + local_rank[e]] = e;
}

// Latency: the k0-sorted stream visits the edge rows of GPHI / GPSI in a permuted order, so every edge is a fresh HBM access (~1 us).  A
// wavefront therefore keeps GWR_DEPTH edges in flight (5 VGPRs each: CH floats of gphi and gpsi per lane plus the edge's 32-float
// window record spread over the lanes, broadcast with v_readlane when it is consumed); with one edge in flight the kernel ran at
// edges x latency / waves (1.29 ms per launch at 1.6 M edges), i.e. 3.1 TB/s.
#ifndef GWR_WAVES
#define GWR_WAVES 4
#endif
#ifndef GWR_DEPTH
#define GWR_DEPTH 6
#endif
#ifndef GWR_CHUNKS
#define GWR_CHUNKS 1360
#endif
template <int CH>
__global__ __launch_bounds__(GWR_WAVES * 64) void k_gwr_sorted(const float* __restrict__ GPHI, const float* __restrict__ GPSI, const float* __restrict__ RW,
                                                               const int* __restrict__ order, int E_, int F, int F3, int R, int chunk_len,
                                                               float* __restrict__ part, int* __restrict__ chunk_range, const int* __restrict__ count_dev) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int chunk = blockIdx.x * GWR_WAVES + wave;
  const int E = count_dev ? min(E_, __builtin_amdgcn_readfirstlane(*count_dev)) : E_;   // mixed batches: only the pairs of the large molecules are listed
  const int r0 = chunk * chunk_len, r1 = min(E, r0 + chunk_len);
  if (r0 >= E) {   // wave-uniform; an empty chunk covers no row (k_gwr_reduce bisects on non-decreasing ranges)
    if (count_dev && lane == 0 && blockIdx.y == 0 && r0 < E_) { chunk_range[2 * chunk] = 0x7fffffff; chunk_range[2 * chunk + 1] = 0x7fffffff; }
    return;
  }
  const int col = blockIdx.y * F + lane * CH;   // F3 = row length of GPHI/GPSI (3F for the PaiNN filter, F for SchNet's first filter layer)
  typedef VOps<CH> VO;
  typedef typename VO::V V;
  V acc[16];   // CH-wide accumulators: every tap is two v_pk_fma_f32 at CH = 2
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = VO::splat(0.f);
  float* out = part + (long)chunk * R * F3 + col;
  V g[GWR_DEPTH], h[GWR_DEPTH];
  float rw[GWR_DEPTH];
  auto issue = [&](int slot, int r) __attribute__((always_inline)) {
    const int e = __builtin_amdgcn_readfirstlane(order[r]);
    g[slot] = VO::load(GPHI + (long)e * F3 + col);
    h[slot] = VO::load(GPSI + (long)e * F3 + col);
    rw[slot] = RW[(long)e * RW_STRIDE + (lane & 31)];
  };
#pragma unroll
  for (int s = 0; s < GWR_DEPTH; ++s) issue(s, min(r0 + s, r1 - 1));
  int base = __builtin_amdgcn_readlane(__float_as_int(rw[0]), 13);
  const int lo = base;
  for (int rb = r0; rb < r1; rb += GWR_DEPTH) {
#pragma unroll
    for (int s = 0; s < GWR_DEPTH; ++s) {
      const int r = rb + s;
      if (r < r1) {   // wave-uniform
        const int rwi = __float_as_int(rw[s]);
        const int k0 = __builtin_amdgcn_readlane(rwi, 13);
        while (base < k0) {   // the window slides up: flush the row that leaves it, shift the accumulators
          *reinterpret_cast<V*>(out + (long)base * F3) = acc[0];
#pragma unroll
          for (int t = 0; t < 15; ++t) acc[t] = acc[t + 1];
          acc[15] = VO::splat(0.f);
          ++base;
        }
#pragma unroll
        for (int t = 0; t < FWIN; ++t) {
          const float rr = __int_as_float(__builtin_amdgcn_readlane(rwi, t)), dd = __int_as_float(__builtin_amdgcn_readlane(rwi, 16 + t));
          acc[t] = VO::fma(g[s], VO::splat(rr), VO::fma(h[s], VO::splat(dd), acc[t]));
        }
        if (r + GWR_DEPTH < r1) issue(s, r + GWR_DEPTH);
      }
    }
  }
  int hi = base;
#pragma unroll
  for (int t = 0; t < FWIN; ++t) {
    if (base + t < R) {
      *reinterpret_cast<V*>(out + (long)(base + t) * F3) = acc[t];
      hi = base + t + 1;
    }
  }
  if (lane == 0 && blockIdx.y == 0) { chunk_range[2 * chunk] = lo; chunk_range[2 * chunk + 1] = hi; }
}

// gWr[c][k] = sum over chunks (in chunk order) whose row range [lo, hi) covers k.  The edge stream is sorted by k0, so
// lo and hi are non-decreasing in the chunk index: the covering chunks form one contiguous run found by bisection.
__global__ void k_gwr_reduce(const float* __restrict__ part, const int* __restrict__ chunk_range, int nchunks, int R, int F3, float* __restrict__ gWr) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * F3) return;
  const int k = idx / F3, c = idx % F3;
  int a = 0, b = nchunks;          // first chunk with hi > k
  while (a < b) { const int m = (a + b) >> 1; if (chunk_range[2 * m + 1] > k) b = m; else a = m + 1; }
  const int first = a;
  a = first; b = nchunks;          // first chunk with lo > k
  while (a < b) { const int m = (a + b) >> 1; if (chunk_range[2 * m] > k) b = m; else a = m + 1; }
  float s = 0.f;
  for (int ch = first; ch < a; ++ch) s += part[((long)ch * R + k) * F3 + c];
  gWr[(long)c * R + k] = s;
}

// out[c][r] = in[r][c]  (rbf_proj.weight [3F][R] -> WrT [R][3F]); 32x32 LDS tile, coalesced both ways
__global__ void k_transpose(const float* __restrict__ in, int rows, int cols, float* __restrict__ out) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    tile[i][threadIdx.x] = (r < rows && c < cols) ? in[(long)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (c < cols && r < rows) out[(long)c * rows + r] = tile[threadIdx.x][i];
  }
}

// ---------------------------------------------------------------------------------------------
// geometry tangent: t_d = r . (pd[src] - pd[dst]);  t_r = ((pd[src]-pd[dst]) - r t_d) / d      (per CSR slot)
__global__ void k_geom_tan(NqGraphView g, const int* __restrict__ dst, const float* __restrict__ pd, float* __restrict__ TD,
                           float* __restrict__ TR) {
  const int sp = blockIdx.x * blockDim.x + threadIdx.x;
  if (sp >= g.E) return;
  const int k = g.col[sp], n = dst[sp];
  const float4 gm = g.geom[sp];
  const float wx = pd[3 * (long)k] - pd[3 * (long)n], wy = pd[3 * (long)k + 1] - pd[3 * (long)n + 1],
              wz = pd[3 * (long)k + 2] - pd[3 * (long)n + 2];
  const float td = gm.x * wx + gm.y * wy + gm.z * wz;
  const float inv = 1.0f / gm.w;
  TD[sp] = td;
  if (!TR) return;   // SchNet: only distances enter the model
  TR[3 * (long)sp] = (wx - gm.x * td) * inv;
  TR[3 * (long)sp + 1] = (wy - gm.y * td) * inv;
  TR[3 * (long)sp + 2] = (wz - gm.z * td) * inv;
}

// geometry reverse: per out-edge (n->k) stored at slot sp of row n: r_out = -geom.xyz,
//   gw(sp) = gd r_out + (gr - (gr.r_out) r_out)/d ;  dE/dpos[n] = sum_sp gw(sp) - gw(rev[sp]) ;  F = -dE/dpos
__device__ __forceinline__ void edge_gw(const NqGraphView& g, const float4* __restrict__ GEDGE, int nwaves, int sp, float& x, float& y, float& z) {
  float4 a = GEDGE[sp];
  for (int w = 1; w < nwaves; ++w) {
    const float4 b = GEDGE[(long)w * g.E + sp];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  const float4 gm = g.geom[sp];
  const float r0 = -gm.x, r1 = -gm.y, r2 = -gm.z;
  const float dot = a.y * r0 + a.z * r1 + a.w * r2;
  const float inv = 1.0f / gm.w;
  x = a.x * r0 + (a.y - dot * r0) * inv;
  y = a.x * r1 + (a.z - dot * r1) * inv;
  z = a.x * r2 + (a.w - dot * r2) * inv;
}

__global__ void k_geom_rev(NqGraphView g, const float4* __restrict__ GEDGE, int nwaves, float* __restrict__ forces) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= g.N) return;
  float fx = 0.f, fy = 0.f, fz = 0.f;
  for (int sp = g.row_ptr[n]; sp < g.row_ptr[n + 1]; ++sp) {
    float ax, ay, az, bx, by, bz;
    edge_gw(g, GEDGE, nwaves, sp, ax, ay, az);
    edge_gw(g, GEDGE, nwaves, g.rev[sp], bx, by, bz);
    fx += ax - bx; fy += ay - by; fz += az - bz;
  }
  forces[3 * (long)n] = -fx; forces[3 * (long)n + 1] = -fy; forces[3 * (long)n + 2] = -fz;
}

// ---- host launchers ------------------------------------------------------------------------
static void rbf_fill(RbfArgs* q, const float4* geom, int E, int R, double cutoff, int env_p, float coeff, const float* offsets, int type,
                     const float* theta) {
  q->geom = geom; q->E = E; q->R = R; q->inv_cutoff = (float)(1.0 / cutoff);
  const double p = env_p;
  q->p = (float)p; q->a = (float)(-(p + 1) * (p + 2) / 2); q->b = (float)(p * (p + 2)); q->c = (float)(-p * (p + 1) / 2);
  q->mu = offsets;   // type 0: GaussianSmearing(0, 1, R).offset (buffer radial_basis.rbf.offset); type 2: BernsteinBasis.prefactor
  q->coeff = coeff;  // -0.5 / (offset[1]-offset[0])^2, computed by the host exactly as PyG does
  q->type = type; q->theta = theta; q->norm_const = (float)sqrt(2.0 / (cutoff * cutoff * cutoff));
  q->rho = nullptr; q->drho = nullptr; q->grho = nullptr; q->contrib = nullptr;
}
int nq_rbf(hipStream_t st, const float4* geom, int E, int R, double cutoff, int env_p, float coeff, const float* offsets,
           float* rho, float* drho, int type, const float* theta) {
  NQ_PROF(st, "rbf");
  if (E <= 0) return NQ_OK;
  RbfArgs q;
  rbf_fill(&q, geom, E, R, cutoff, env_p, coeff, offsets, type, theta);
  q.rho = rho; q.drho = drho;
  hipLaunchKernelGGL((k_rbf<false>), dim3(nq_cdiv((long)E * R, 256)), dim3(256), 0, st, q);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
// contrib [E][R] (see k_rbf<true>); the caller reduces it over edges (per k for the Bessel frequencies, over everything for pregamma)
int nq_rbf_param_grad(hipStream_t st, const float4* geom, int E, int R, double cutoff, int env_p, const float* offsets, int type, const float* theta,
                      const float* grho, float* contrib) {
  NQ_PROF(st, "rbf_param_grad");
  if (E <= 0 || type == 0) return NQ_OK;
  RbfArgs q;
  rbf_fill(&q, geom, E, R, cutoff, env_p, 0.f, offsets, type, theta);
  q.grho = grho; q.contrib = contrib;
  hipLaunchKernelGGL((k_rbf<true>), dim3(nq_cdiv((long)E * R, 256)), dim3(256), 0, st, q);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_msg_fwd(hipStream_t st, const MsgArgs& q, bool tangent) {
  NQ_PROF(st, "msg_fwd");
  if (q.g.N <= 0) return NQ_OK;
  if (tangent) hipLaunchKernelGGL(k_msg_tan, dim3(q.g.N), dim3(q.F), 0, st, q);
  else hipLaunchKernelGGL(k_msg_fwd, dim3(q.g.N), dim3(q.F), 0, st, q);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_msg_rev(hipStream_t st, const MsgRevArgs& q, bool dual) {
  NQ_PROF(st, "msg_rev");
  if (q.g.N <= 0) return NQ_OK;
  if (dual) hipLaunchKernelGGL((k_msg_rev<true>), dim3(q.g.N), dim3(q.F), 0, st, q);
  else hipLaunchKernelGGL((k_msg_rev<false>), dim3(q.g.N), dim3(q.F), 0, st, q);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_geom_tan(hipStream_t st, const NqGraphView& g, const int* dst, const float* pos_dot, float* TD, float* TR) {
  NQ_PROF(st, "geom_tan");
  if (g.E <= 0) return NQ_OK;
  hipLaunchKernelGGL(k_geom_tan, dim3(nq_cdiv(g.E, 256)), dim3(256), 0, st, g, dst, pos_dot, TD, TR);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_geom_rev(hipStream_t st, const NqGraphView& g, const float4* GEDGE, int nwaves, float* forces) {
  NQ_PROF(st, "geom_rev");
  hipLaunchKernelGGL(k_geom_rev, dim3(nq_cdiv(g.N, 128)), dim3(128), 0, st, g, GEDGE, nwaves, forces);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

// ---- fused-filter launchers ------------------------------------------------------------------
bool nq_filter_fits_lds(int F, int R) {
  return (size_t)(R < FWIN ? FWIN : R) * 3 * F * sizeof(float) <= 156 * 1024 && (F == 64 || F == 128 || F == 256);
}

void nq_make_filter_args(FilterArgs* fa, const float* WRT, const float* br, const float* mu, const float* RW, int R, double cutoff, int env_p,
                         float coeff, int mode) {
  const double p = env_p;
  fa->WRT = WRT; fa->br = br; fa->mu = mu; fa->RW = RW; fa->R = R; fa->inv_cutoff = (float)(1.0 / cutoff);
  fa->p = (float)p; fa->a = (float)(-(p + 1) * (p + 2) / 2); fa->b = (float)(p * (p + 2)); fa->c = (float)(-p * (p + 1) / 2);
  fa->coeff = coeff; fa->mode = mode; fa->cutoff = (float)cutoff; fa->row_ctr = nullptr;
}

int nq_rbf_window(hipStream_t st, const float4* geom, int E, const FilterArgs& fa, float* RW) {
  NQ_PROF(st, "rbf_window");
  if (E <= 0) return NQ_OK;
  hipLaunchKernelGGL(k_rbf_window, dim3(nq_cdiv(E, RBFW_THREADS)), dim3(RBFW_THREADS), 0, st, geom, E, fa, RW);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_transpose(hipStream_t st, const float* in, int rows, int cols, float* out) {
  NQ_PROF(st, "transpose");
  hipLaunchKernelGGL(k_transpose, dim3(nq_cdiv(cols, 32), nq_cdiv(rows, 32)), dim3(32, 8), 0, st, in, rows, cols, out);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

// Channels per lane of each kernel flavour (kind: 0 forward, 1 tangent, 2 force adjoint, 3 dual reverse).  A wavefront covers a SLICE of
// 64*CH channels; F/(64*CH) slices of one atom run as separate wavefronts in separate workgroups, each with only its slice of WrT in
// LDS (R*3*64*CH*4 bytes) -- fewer channels per lane = fewer VGPRs and a smaller LDS copy = more wavefronts per CU to hide the gathers,
// at the price of repeating the per-edge scalar work per slice.  Defaults from profiles/r01_fused_tuning.txt; -DNQ_CH_<kind>=n overrides.
#ifndef NQ_CH_FWD
#define NQ_CH_FWD 0
#endif
#ifndef NQ_CH_TAN
#define NQ_CH_TAN 0
#endif
#ifndef NQ_CH_FORCE
#define NQ_CH_FORCE 0
#endif
#ifndef NQ_CH_DUAL
#define NQ_CH_DUAL 0
#endif
#ifndef NQ_SMALL_SLICE_ATOMS
#define NQ_SMALL_SLICE_ATOMS 2048   // below this many atoms per launch a row is split into two channel slices (two wavefronts per atom); measured at 32 / 64 / 96 / 128 / 192 conformers: profiles/r05_slice_threshold_32_to_192.txt
#endif
static int fused_ch(int kind, int F, int N) {
  const int forced = kind == 0 ? NQ_CH_FWD : (kind == 1 ? NQ_CH_TAN : (kind == 2 ? NQ_CH_FORCE : NQ_CH_DUAL));
  const int full = F / 64;                       // one slice: the whole row in one wavefront
  if (forced > 0 && forced <= full && full % forced == 0) return forced;
  // Small batches (the reference's 32 conformers = 1.3 k atoms): fewer rows than wavefront slots, so a launch is one row walk per wavefront at single-wave issue
  // rates; two half-width wavefronts per atom halve the per-edge VALU chain and the WrT fill of each workgroup.  At throughput sizes the slices only add
  // per-edge scalar work (profiles/r01_fused_tuning.txt section 3).
  if (N < NQ_SMALL_SLICE_ATOMS && full >= 2) return full / 2;
  return full;
}
static int fused_grid(int N, int F, int ch, int* threads, size_t* lds, int R, int max_threads = FUSED_THREADS) {
  const int nslots = max_threads / 64;   // one wavefront per (atom, slice)
  const int nslices = F / (64 * ch);
  *threads = nslots * 64;
  *lds = (size_t)(R < FWIN ? FWIN : R) * 3 * 64 * ch * sizeof(float);
  int per_cu = (int)((156 * 1024) / *lds);
  per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
  int groups = 256 * per_cu / nslices;           // atom groups (one workgroup per slice each)
  groups = groups < 8 ? 8 : (groups / 8) * 8;    // multiple of 8 -> the XCD-aware sweep applies
  const int need = nq_cdiv(N, nslots);
  if (need < groups) groups = need;              // small batches: plain interleave
  return groups * nslices;
}

// the >64 KB dynamic-LDS opt-in is sticky per (device, kernel): nq_dyn_lds sets it when the requested size grows, not on every launch
#define FUSED_LAUNCH(KERN, FLAG, CHV, Q)                                                                          \
  do {                                                                                                            \
    NQ_DYN_LDS((KERN<FLAG, CHV>), lds);                                                                           \
    hipLaunchKernelGGL((KERN<FLAG, CHV>), dim3(grid), dim3(threads), lds, st, Q, fa, fa.RW);                      \
  } while (0)
#define FUSED_DISPATCH(KERN, FLAG, Q)                                   \
  do {                                                                  \
    switch (ch) {                                                       \
      case 1: FUSED_LAUNCH(KERN, FLAG, 1, Q); break;                    \
      case 2: FUSED_LAUNCH(KERN, FLAG, 2, Q); break;                    \
      case 4: FUSED_LAUNCH(KERN, FLAG, 4, Q); break;                    \
      default: return nq_fail(NQ_ERR_ARG, "fused message kernels need hidden_channels in {64,128,256}"); \
    }                                                                   \
  } while (0)

int nq_msgf_fwd(hipStream_t st, const MsgArgs& q, const FilterArgs& fa, bool tangent) {
  NQ_PROF(st, tangent ? "msgf_tan" : "msgf_fwd");
  if (q.g.N <= 0) return NQ_OK;
  if ((size_t)q.g.N * 3 * q.F * 4 >= 0xfffff000ull) return nq_fail(NQ_ERR_ARG, "fused message kernels: a node array of %d atoms x %d channels exceeds the 4 GB a buffer descriptor's scalar offset reaches", q.g.N, 3 * q.F);
  int threads; size_t lds;
  const int ch = fused_ch(tangent ? 1 : 0, q.F, q.g.N);
  const int grid = fused_grid(q.g.N, q.F, ch, &threads, &lds, fa.R, fused_threads(tangent ? 1 : 0, ch));
  if (tangent) FUSED_DISPATCH(k_msgf_fwd, true, q);
  else FUSED_DISPATCH(k_msgf_fwd, false, q);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_msgf_rev(hipStream_t st, const MsgRevArgs& q, const FilterArgs& fa, bool dual, bool pair_rows) {
  NQ_PROF(st, dual ? (pair_rows ? "msgf_rev_dual" : "msgf_rev_dual_ng") : "msgf_rev_force");
  if (q.g.N <= 0) return NQ_OK;
  if ((size_t)q.g.N * 3 * q.F * 4 >= 0xfffff000ull) return nq_fail(NQ_ERR_ARG, "fused message kernels: a node array of %d atoms x %d channels exceeds the 4 GB a buffer descriptor's scalar offset reaches", q.g.N, 3 * q.F);
  int threads; size_t lds;
  const int kind = dual ? (pair_rows ? 3 : 4) : 2;
  const int ch = fused_ch(dual ? 3 : 2, q.F, q.g.N);
  const int grid = fused_grid(q.g.N, q.F, ch, &threads, &lds, fa.R, fused_threads(kind, ch));
  if (dual && !pair_rows && q.lite) FUSED_DISPATCH(k_msgf_rev_nopair, true, q);
  else if (dual && !pair_rows) FUSED_DISPATCH(k_msgf_rev_nopair, false, q);
  else if (dual) FUSED_DISPATCH(k_msgf_rev, true, q);
  else FUSED_DISPATCH(k_msgf_rev, false, q);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

// ---- k0-sorted edge order + register-resident rbf_proj weight gradient ---------------------------------------------
static int gwr_nbins(int R) { return R - (R < FWIN ? R : FWIN) + 1; }
static int gwr_chunks(int E) {   // ~4096 wavefronts (16 per CU) over 3 column slices
  int chunk_len = nq_cdiv(E, GWR_CHUNKS);
  if (chunk_len < 64) chunk_len = 64;
  return nq_cdiv(E, chunk_len);
}
size_t nq_k0_sort_scratch_ints(int E, int R) { return (size_t)nq_cdiv(E, SORT_CHUNK) * gwr_nbins(R) + (size_t)E + 256; }
// order[] <- CSR slots sorted (stably) by window start k0 (all E slots, or the E/2 lower slots when row_of / col are given); scratch: nq_k0_sort_scratch_ints() ints
int nq_k0_sort(hipStream_t st, const float* RW, int E, int R, int* order, int* scratch, const int* row_of, const int* col, const int* mol_ptr,
               const int* atom_mol, int cap, int* count_out) {
  NQ_PROF(st, "k0_sort");
  K0Filter flt{row_of, col, mol_ptr, atom_mol, (row_of && mol_ptr && atom_mol) ? cap : 0};
  const int nbins = gwr_nbins(R), nchunks = nq_cdiv(E, SORT_CHUNK);
  if (nbins > 256) return nq_fail(NQ_ERR_ARG, "k0 sort supports at most 256 bins (num_rbf <= 268)");
  int* chunk_hist = scratch; int* local_rank = scratch + (size_t)nchunks * nbins; int* bin_total = local_rank + E;
  hipLaunchKernelGGL(k_k0_hist, dim3(nchunks), dim3(SORT_CHUNK), 0, st, RW, E, nbins, chunk_hist, local_rank, flt);
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_k0_scan, dim3(nq_cdiv(nbins, 4)), dim3(256), 0, st, chunk_hist, nchunks, nbins, bin_total);
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_k0_scatter, dim3(nchunks), dim3(SORT_CHUNK), 0, st, RW, E, nbins, chunk_hist, bin_total, local_rank, order, flt, count_out);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

size_t nq_gwr_scratch_floats(int E, int F, int R, int parts) { const size_t nc = gwr_chunks(E); return nc * R * parts * F + 2 * nc + 16; }

int nq_gwr_sorted(hipStream_t st, const float* GPHI, const float* GPSI, const float* RW, const int* order, int E, int F, int R, float* gWr,
                  float* scratch, int parts, const int* count_dev) {
  NQ_PROF(st, "gwr_sorted");
  const int nchunks = gwr_chunks(E), chunk_len = nq_cdiv(E, nchunks);
  float* part = scratch;
  const int F3 = parts * F;
  int* chunk_range = reinterpret_cast<int*>(scratch + (size_t)nchunks * R * F3);
  dim3 grid(nq_cdiv(nchunks, GWR_WAVES), parts);
  switch (F / 64) {
    case 1: hipLaunchKernelGGL((k_gwr_sorted<1>), grid, dim3(GWR_WAVES * 64), 0, st, GPHI, GPSI, RW, order, E, F, F3, R, chunk_len, part, chunk_range, count_dev); break;
    case 2: hipLaunchKernelGGL((k_gwr_sorted<2>), grid, dim3(GWR_WAVES * 64), 0, st, GPHI, GPSI, RW, order, E, F, F3, R, chunk_len, part, chunk_range, count_dev); break;
    case 4: hipLaunchKernelGGL((k_gwr_sorted<4>), grid, dim3(GWR_WAVES * 64), 0, st, GPHI, GPSI, RW, order, E, F, F3, R, chunk_len, part, chunk_range, count_dev); break;
    default: return nq_fail(NQ_ERR_ARG, "gwr_sorted needs hidden_channels in {64,128,256}");
  }
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_gwr_reduce, dim3(nq_cdiv((long)R * F3, 256)), dim3(256), 0, st, part, chunk_range, nq_cdiv(E, chunk_len), R, F3, gWr);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
