// Geometry bases of the Hamiltonian models (SURVEY.md section 8, row a25 and the radial part of a13):
//   real spherical harmonics Y_0..Y_4 of unit vectors  -- closed forms of phisnet/nn/spherical_harmonics/spherical_harmonics_any_order.py:11-100
//     (no 1/sqrt(4 pi), Condon-Shortley phase, m = -l..l, Y_1 = sqrt(3) (y, z, x))
//   exponential Bernstein radial basis with the smooth cutoff  -- phisnet/nn/modules/exponential_bernstein_radial_basis_functions.py:13-41,
//     qhnet/layers.py:86-120:  rbf_k(r) = fc(r) exp(logC_k + n_k x + v_k log(1 - e^x)),  x = -softplus(_alpha) r,  fc = exp(-r^2 / ((c - r)(c + r)))
// One thread per output element; both are bandwidth-trivial next to the tensor products they feed.
#include "common.h"
#include "../../include/nablaq.h"

__global__ void k_sph_harm(const float* __restrict__ u, long P, int L, float* __restrict__ out) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int nc = (L + 1) * (L + 1);
  float* o = out + p * nc;
  const float x = u[3 * p], y = u[3 * p + 1], z = u[3 * p + 2];
  o[0] = 1.0f;
  if (L < 1) return;
  const float s3 = 1.7320508075688772f;
  o[1] = s3 * y; o[2] = s3 * z; o[3] = s3 * x;
  if (L < 2) return;
  const float x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, yz = y * z, xz = x * z;
  const float x2my2 = x2 - y2, t3z2m1 = 3.0f * z2 - 1.0f;
  const float s15 = 3.872983346207417f, s5o2 = 1.118033988749895f, s15o2 = 1.9364916731037085f;
  o[4] = s15 * xy; o[5] = s15 * yz; o[6] = s5o2 * t3z2m1; o[7] = s15 * xz; o[8] = s15o2 * x2my2;
  if (L < 3) return;
  const float xyz = xy * z, t3x2my2 = 3.0f * x2 - y2, x2m3y2 = x2 - 3.0f * y2;
  const float s70o4 = 2.091650066335189f, s105 = 10.246950765959598f, s42o4 = 1.620185174601965f, s7o2 = 1.3228756555322954f, s105o2 = 5.123475382979799f;
  const float t5z2 = 5.0f * z2, t5z2m1 = t5z2 - 1.0f;
  o[9] = s70o4 * y * t3x2my2; o[10] = s105 * xyz; o[11] = s42o4 * y * t5z2m1; o[12] = s7o2 * z * (t5z2 - 3.0f);
  o[13] = s42o4 * x * t5z2m1; o[14] = s105o2 * z * x2my2; o[15] = s70o4 * x * x2m3y2;
  if (L < 4) return;
  const float x4 = x2 * x2, y4 = y2 * y2, x2y2 = x2 * y2;
  const float a = 8.874119674649425f /* sqrt(35)*3/2 */, b = 18.824850597016705f /* sqrt(70)*9/4 */, c = 3.3541019662496847f /* sqrt(45)/2 */,
              d = 2.3717082451262845f /* sqrt(10)*3/4 */, e = 1.6770509831248424f /* sqrt(45)/4 */, f = 6.274950199005566f /* sqrt(70)*3/4 */,
              g = 2.2185299186623562f /* sqrt(35)*3/8 */;
  const float t7z2 = 7.0f * z2, t7z2m1 = t7z2 - 1.0f, t7z2m3 = t7z2 - 3.0f;
  o[16] = a * xy * x2my2; o[17] = b * yz * (x2 - y2 / 3.0f); o[18] = c * xy * t7z2m1; o[19] = d * yz * t7z2m3;
  o[20] = 0.125f * (z2 * (105.0f * z2 - 90.0f) + 9.0f);
  o[21] = d * xz * t7z2m3; o[22] = e * t7z2m1 * x2my2; o[23] = f * xz * x2m3y2; o[24] = g * (x4 - 6.0f * x2y2 + y4);
}

// gu[p] = sum_c g[p][c] dY_c/d(x, y, z): the closed forms above differentiated as polynomials of free (x, y, z) -- what autograd does with the reference's
// expressions (spherical_harmonics_any_order.py) before the chain rule through u = r / |r| (phisnet forces, neural_network.py:981-984)
__global__ void k_sph_harm_bwd(const float* __restrict__ u, const float* __restrict__ gy, long P, int L, float* __restrict__ gu) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int nc = (L + 1) * (L + 1);
  const float* g = gy + p * nc;
  const float x = u[3 * p], y = u[3 * p + 1], z = u[3 * p + 2];
  float gx = 0.f, gyv = 0.f, gz = 0.f;
  if (L >= 1) {
    const float s3 = 1.7320508075688772f;
    gyv += s3 * g[1]; gz += s3 * g[2]; gx += s3 * g[3];
  }
  const float x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, yz = y * z, xz = x * z;
  if (L >= 2) {
    const float s15 = 3.872983346207417f, s5o2 = 1.118033988749895f, s15o2 = 1.9364916731037085f;
    gx += s15 * y * g[4];  gyv += s15 * x * g[4];                       // s15 xy
    gyv += s15 * z * g[5]; gz += s15 * y * g[5];                        // s15 yz
    gz += s5o2 * 6.0f * z * g[6];                                       // s5o2 (3 z^2 - 1)
    gx += s15 * z * g[7];  gz += s15 * x * g[7];                        // s15 xz
    gx += s15o2 * 2.0f * x * g[8]; gyv -= s15o2 * 2.0f * y * g[8];      // s15o2 (x^2 - y^2)
  }
  if (L >= 3) {
    const float s70o4 = 2.091650066335189f, s105 = 10.246950765959598f, s42o4 = 1.620185174601965f, s7o2 = 1.3228756555322954f, s105o2 = 5.123475382979799f;
    const float t5z2m1 = 5.0f * z2 - 1.0f;
    // o9 = s70o4 y (3 x^2 - y^2)
    gx += s70o4 * 6.0f * xy * g[9];  gyv += s70o4 * (3.0f * x2 - 3.0f * y2) * g[9];
    // o10 = s105 x y z
    gx += s105 * yz * g[10]; gyv += s105 * xz * g[10]; gz += s105 * xy * g[10];
    // o11 = s42o4 y (5 z^2 - 1)
    gyv += s42o4 * t5z2m1 * g[11]; gz += s42o4 * 10.0f * yz * g[11];
    // o12 = s7o2 z (5 z^2 - 3)
    gz += s7o2 * (15.0f * z2 - 3.0f) * g[12];
    // o13 = s42o4 x (5 z^2 - 1)
    gx += s42o4 * t5z2m1 * g[13]; gz += s42o4 * 10.0f * xz * g[13];
    // o14 = s105o2 z (x^2 - y^2)
    gx += s105o2 * 2.0f * xz * g[14]; gyv -= s105o2 * 2.0f * yz * g[14]; gz += s105o2 * (x2 - y2) * g[14];
    // o15 = s70o4 x (x^2 - 3 y^2)
    gx += s70o4 * (3.0f * x2 - 3.0f * y2) * g[15]; gyv -= s70o4 * 6.0f * xy * g[15];
  }
  if (L >= 4) {
    const float a = 8.874119674649425f, b = 18.824850597016705f, c = 3.3541019662496847f, d = 2.3717082451262845f, e = 1.6770509831248424f,
                f = 6.274950199005566f, gg = 2.2185299186623562f;
    const float x2my2 = x2 - y2, t7z2m1 = 7.0f * z2 - 1.0f, t7z2m3 = 7.0f * z2 - 3.0f, x2m3y2 = x2 - 3.0f * y2;
    // o16 = a xy (x^2 - y^2)
    gx += a * (3.0f * x2 * y - y2 * y) * g[16]; gyv += a * (x2 * x - 3.0f * x * y2) * g[16];
    // o17 = b yz (x^2 - y^2 / 3)
    gx += b * 2.0f * x * yz * g[17]; gyv += b * z * (x2 - y2) * g[17]; gz += b * y * (x2 - y2 / 3.0f) * g[17];
    // o18 = c xy (7 z^2 - 1)
    gx += c * y * t7z2m1 * g[18]; gyv += c * x * t7z2m1 * g[18]; gz += c * 14.0f * xy * z * g[18];
    // o19 = d yz (7 z^2 - 3)
    gyv += d * z * t7z2m3 * g[19]; gz += d * y * (21.0f * z2 - 3.0f) * g[19];
    // o20 = (105 z^4 - 90 z^2 + 9) / 8
    gz += 0.125f * (420.0f * z2 * z - 180.0f * z) * g[20];
    // o21 = d xz (7 z^2 - 3)
    gx += d * z * t7z2m3 * g[21]; gz += d * x * (21.0f * z2 - 3.0f) * g[21];
    // o22 = e (7 z^2 - 1)(x^2 - y^2)
    gx += e * t7z2m1 * 2.0f * x * g[22]; gyv -= e * t7z2m1 * 2.0f * y * g[22]; gz += e * 14.0f * z * x2my2 * g[22];
    // o23 = f xz (x^2 - 3 y^2)
    gx += f * z * (3.0f * x2 - 3.0f * y2) * g[23]; gyv -= f * 6.0f * xz * y * g[23]; gz += f * x * x2m3y2 * g[23];
    // o24 = gg (x^4 - 6 x^2 y^2 + y^4)
    gx += gg * (4.0f * x2 * x - 12.0f * x * y2) * g[24]; gyv += gg * (4.0f * y2 * y - 12.0f * x2 * y) * g[24];
  }
  gu[3 * p] = gx; gu[3 * p + 1] = gyv; gu[3 * p + 2] = gz;
}

// rbf[p][k] and (optionally) the integrand of d/d alpha:  drbf/dalpha = rbf * (-r) * (n_k - v_k e^x / (1 - e^x))
template <int GRAD>   // 0: values; 1: sum_k g dRBF/dalpha per row; 2: sum_k g dRBF/dr per row
__global__ void k_bernstein_rbf(const float* __restrict__ r, long P, int K, float alpha_host, float cutoff, const float* __restrict__ logc,
                                const float* __restrict__ nk, const float* __restrict__ vk, float* __restrict__ out,
                                const float* __restrict__ gout, float* __restrict__ galpha_rows, const float* __restrict__ alpha_dev) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P * (GRAD ? 1 : K)) return;
  const float alpha = alpha_dev ? *alpha_dev : alpha_host;     // device scalar: no host read of the learnable parameter (stream capture)
  if (!GRAD) {
    const long p = idx / K; const int k = (int)(idx % K);
    const float rr = r[p];
    float val = 0.f;
    if (rr < cutoff) {
      const float x = -alpha * rr;
      const float fc = expf(-(rr * rr) / ((cutoff - rr) * (cutoff + rr)));
      val = fc * expf(logc[k] + nk[k] * x + vk[k] * logf(-expm1f(x)));
    }
    out[idx] = val;
  } else {
    const long p = idx;
    const float rr = r[p];
    float acc = 0.f;
    if (rr < cutoff) {
      const float x = -alpha * rr;
      const float fc = expf(-(rr * rr) / ((cutoff - rr) * (cutoff + rr)));
      const float om = -expm1f(x);                 // 1 - e^x
      const float ratio = (1.0f - om) / om;        // e^x / (1 - e^x)
      const float lg = logf(om);
      const float c2mr2 = (cutoff - rr) * (cutoff + rr);
      const float dlogfc = -2.0f * rr * cutoff * cutoff / (c2mr2 * c2mr2);      // d/dr of -r^2 / (c^2 - r^2)
      for (int k = 0; k < K; ++k) {
        const float val = fc * expf(logc[k] + nk[k] * x + vk[k] * lg);
        const float dx = nk[k] - vk[k] * ratio;                                  // d/dx of the exponent
        acc = fmaf(gout[p * K + k] * val, GRAD == 1 ? -rr * dx : dlogfc - alpha * dx, acc);
      }
    }
    galpha_rows[p] = acc;
  }
}

extern "C" {

int nq_sph_harm(const float* unit_vectors, int64_t P, int32_t order, float* out, void* stream) {
  if (!unit_vectors || !out || order < 0 || order > 4 || P < 0) return nq_fail(NQ_ERR_ARG, "bad argument (orders 0..4)");
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "sph_harm");
  if (P > 0) hipLaunchKernelGGL(k_sph_harm, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, unit_vectors, (long)P, order, out);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_bernstein_rbf(const float* r, int64_t P, int32_t K, float alpha, float cutoff, const float* logc, const float* n, const float* v, float* out,
                     void* stream) {
  if (!r || !logc || !n || !v || !out || K <= 0 || P < 0) return nq_fail(NQ_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "bernstein_rbf");
  if (P > 0) hipLaunchKernelGGL((k_bernstein_rbf<0>), dim3((unsigned)((P * K + 255) / 256)), dim3(256), 0, st, r, (long)P, K, alpha, cutoff, logc, n, v,
                                out, nullptr, nullptr, nullptr);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

/* the same with alpha = softplus(_alpha) read from a device scalar (no host synchronisation: usable inside a captured HIP graph) */
int nq_bernstein_rbf_dev(const float* r, int64_t P, int32_t K, const float* alpha_dev, float cutoff, const float* logc, const float* n, const float* v, float* out,
                         void* stream) {
  if (!r || !alpha_dev || !logc || !n || !v || !out || K <= 0 || P < 0) return nq_fail(NQ_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "bernstein_rbf");
  if (P > 0) hipLaunchKernelGGL((k_bernstein_rbf<0>), dim3((unsigned)((P * K + 255) / 256)), dim3(256), 0, st, r, (long)P, K, 0.f, cutoff, logc, n, v,
                                out, nullptr, nullptr, alpha_dev);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_bernstein_rbf_grad_alpha_dev(const float* r, const float* grad_out, int64_t P, int32_t K, const float* alpha_dev, float cutoff, const float* logc,
                                    const float* n, const float* v, float* galpha_rows, void* stream) {
  if (!r || !grad_out || !alpha_dev || !logc || !n || !v || !galpha_rows || K <= 0 || P < 0) return nq_fail(NQ_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "bernstein_rbf_grad");
  if (P > 0) hipLaunchKernelGGL((k_bernstein_rbf<1>), dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, r, (long)P, K, 0.f, cutoff, logc, n, v,
                                nullptr, grad_out, galpha_rows, alpha_dev);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

/* gr[p] = sum_k grad_out[p][k] * d rbf[p][k] / d r (smooth cutoff included): the radial half of -dE/dR (phisnet/nn/neural_network.py:981-984) */
int nq_bernstein_rbf_grad_r_dev(const float* r, const float* grad_out, int64_t P, int32_t K, const float* alpha_dev, float cutoff, const float* logc,
                                const float* n, const float* v, float* gr, void* stream) {
  if (!r || !grad_out || !alpha_dev || !logc || !n || !v || !gr || K <= 0 || P < 0) return nq_fail(NQ_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "bernstein_rbf_grad_r");
  if (P > 0) hipLaunchKernelGGL((k_bernstein_rbf<2>), dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, r, (long)P, K, 0.f, cutoff, logc, n, v,
                                nullptr, grad_out, gr, alpha_dev);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

/* gu [P][3] = sum_c grad_out[P][(order+1)^2] dY_c / d(x, y, z) (the harmonics as polynomials of a free vector; the caller chains through u = r / |r|) */
int nq_sph_harm_backward(const float* unit_vectors, const float* grad_out, int64_t P, int32_t order, float* gu, void* stream) {
  if (!unit_vectors || !grad_out || !gu || order < 0 || order > 4 || P < 0) return nq_fail(NQ_ERR_ARG, "bad argument (orders 0..4)");
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "sph_harm_bwd");
  if (P > 0) hipLaunchKernelGGL(k_sph_harm_bwd, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, unit_vectors, grad_out, (long)P, order, gu);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

/* galpha_rows[p] = sum_k grad_out[p][k] * d rbf[p][k] / d alpha  (alpha = softplus(_alpha); the caller sums over p and applies sigmoid(_alpha)) */
int nq_bernstein_rbf_grad_alpha(const float* r, const float* grad_out, int64_t P, int32_t K, float alpha, float cutoff, const float* logc, const float* n,
                                const float* v, float* galpha_rows, void* stream) {
  if (!r || !grad_out || !logc || !n || !v || !galpha_rows || K <= 0 || P < 0) return nq_fail(NQ_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "bernstein_rbf_grad");
  if (P > 0) hipLaunchKernelGGL((k_bernstein_rbf<1>), dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, r, (long)P, K, alpha, cutoff, logc, n, v,
                                nullptr, grad_out, galpha_rows, nullptr);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

}  // extern "C"

// ---- learnable feature-wise activations of PhiSNet (phisnet/nn/modules/swish.py:10-24, shifted_softplus.py:14-32) --------------------------
//   kind 0: swish  y = alpha_f x sigmoid(beta_f x)          kind 1: ssp  y = alpha_f (softplus(beta_f x) - ln 2) / beta_f   (0.5 alpha x if beta = 0)
// backward: gx in place of a fresh buffer, and per-row partials of dL/dalpha, dL/dbeta (summed over rows by the caller).
__device__ __forceinline__ float act_sig(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float act_softplus(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
template <bool BWD>
__global__ void k_feature_act(const float* __restrict__ x, const float* __restrict__ alpha, const float* __restrict__ beta, long rows, int F, int kind,
                              float* __restrict__ y, const float* __restrict__ gy, float* __restrict__ gx, float* __restrict__ ga_rows,
                              float* __restrict__ gb_rows) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * F) return;
  const int f = (int)(idx % F);
  const float xv = x[idx], a = alpha[f], b = beta[f];
  const float ln2 = 0.69314718055994530942f;
  if (kind == 0) {
    const float s = act_sig(b * xv);
    if (!BWD) { y[idx] = a * xv * s; return; }
    const float g = gy[idx];
    gx[idx] = g * a * (s + xv * b * s * (1.0f - s));
    ga_rows[idx] = g * xv * s;
    gb_rows[idx] = g * a * xv * xv * s * (1.0f - s);
  } else {
    if (b != 0.f) {
      const float sp = act_softplus(b * xv) - ln2;
      if (!BWD) { y[idx] = a * sp / b; return; }
      const float g = gy[idx], s = act_sig(b * xv);
      gx[idx] = g * a * s;
      ga_rows[idx] = g * sp / b;
      gb_rows[idx] = g * a * (xv * s / b - sp / (b * b));
    } else {
      if (!BWD) { y[idx] = 0.5f * a * xv; return; }
      const float g = gy[idx];
      gx[idx] = g * 0.5f * a; ga_rows[idx] = g * 0.5f * xv; gb_rows[idx] = g * a * xv * xv * 0.125f;   // limit beta -> 0 of the derivative
    }
  }
}

extern "C" {
int nq_feature_act(const float* x, const float* alpha, const float* beta, int64_t rows, int32_t F, int32_t kind, float* y, void* stream) {
  if (!x || !alpha || !beta || !y || F <= 0 || rows < 0 || kind < 0 || kind > 1) return nq_fail(NQ_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "feature_act");
  if (rows > 0) hipLaunchKernelGGL((k_feature_act<false>), dim3((unsigned)((rows * F + 255) / 256)), dim3(256), 0, st, x, alpha, beta, (long)rows, F, kind, y,
                                   nullptr, nullptr, nullptr, nullptr);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_feature_act_backward(const float* x, const float* alpha, const float* beta, const float* grad_y, int64_t rows, int32_t F, int32_t kind,
                            float* grad_x, float* grad_alpha_rows, float* grad_beta_rows, void* stream) {
  if (!x || !alpha || !beta || !grad_y || !grad_x || !grad_alpha_rows || !grad_beta_rows || F <= 0 || rows < 0 || kind < 0 || kind > 1)
    return nq_fail(NQ_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "feature_act_bwd");
  if (rows > 0) hipLaunchKernelGGL((k_feature_act<true>), dim3((unsigned)((rows * F + 255) / 256)), dim3(256), 0, st, x, alpha, beta, (long)rows, F, kind, nullptr,
                                   grad_y, grad_x, grad_alpha_rows, grad_beta_rows);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
}  // extern "C"

// The same activation on the scalar component of a PACKED irreps tensor [rows][ncomp][F]; every other component is copied (residual_block.py:58-64
// activates xs[0] only).  One launch instead of slice + activation + concatenation.
template <bool BWD>
__global__ void k_packed_act0(const float* __restrict__ x, const float* __restrict__ alpha, const float* __restrict__ beta, long rows, int ncomp, int F, int kind,
                              float* __restrict__ y, const float* __restrict__ gy, float* __restrict__ gx, float* __restrict__ ga_rows,
                              float* __restrict__ gb_rows) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * ncomp * F) return;
  const int f = (int)(idx % F);
  const long rc = idx / F;
  if (rc % ncomp != 0) {
    if (BWD) gx[idx] = gy[idx]; else y[idx] = x[idx];
    return;
  }
  const long r = rc / ncomp;
  const float xv = x[idx], a = alpha[f], b = beta[f];
  const float ln2 = 0.69314718055994530942f;
  float yv, dx, da, db;
  if (kind == 0) {
    const float s = act_sig(b * xv);
    yv = a * xv * s; dx = a * (s + xv * b * s * (1.0f - s)); da = xv * s; db = a * xv * xv * s * (1.0f - s);
  } else if (b != 0.f) {
    const float sp = act_softplus(b * xv) - ln2, s = act_sig(b * xv);
    yv = a * sp / b; dx = a * s; da = sp / b; db = a * (xv * s / b - sp / (b * b));
  } else {
    yv = 0.5f * a * xv; dx = 0.5f * a; da = 0.5f * xv; db = a * xv * xv * 0.125f;
  }
  if (!BWD) { y[idx] = yv; return; }
  const float g = gy[idx];
  gx[idx] = g * dx;
  ga_rows[r * F + f] = g * da;
  gb_rows[r * F + f] = g * db;
}

extern "C" {
int nq_packed_act0(const float* x, const float* alpha, const float* beta, int64_t rows, int32_t ncomp, int32_t F, int32_t kind, float* y, void* stream) {
  if (!x || !alpha || !beta || !y || F <= 0 || ncomp <= 0 || rows < 0 || kind < 0 || kind > 1) return nq_fail(NQ_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "packed_act0");
  const long n = (long)rows * ncomp * F;
  if (n > 0) hipLaunchKernelGGL((k_packed_act0<false>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, alpha, beta, (long)rows, ncomp, F, kind, y,
                                nullptr, nullptr, nullptr, nullptr);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_packed_act0_backward(const float* x, const float* alpha, const float* beta, const float* grad_y, int64_t rows, int32_t ncomp, int32_t F, int32_t kind,
                            float* grad_x, float* grad_alpha_rows, float* grad_beta_rows, void* stream) {
  if (!x || !alpha || !beta || !grad_y || !grad_x || !grad_alpha_rows || !grad_beta_rows || F <= 0 || ncomp <= 0 || rows < 0 || kind < 0 || kind > 1)
    return nq_fail(NQ_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "packed_act0_bwd");
  const long n = (long)rows * ncomp * F;
  if (n > 0) hipLaunchKernelGGL((k_packed_act0<true>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, alpha, beta, (long)rows, ncomp, F, kind, nullptr,
                                grad_y, grad_x, grad_alpha_rows, grad_beta_rows);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
}  // extern "C"

// ---- pair <-> atom data movement for the interaction blocks (interaction_block.py:135-142: torch.gather over idx_j, index_add over idx_i) -----
// rows are [C] floats (C = (2l+1) * F); seg_ptr [N+1] delimits the contiguous pair rows of every atom (pairs sorted by centre atom), so the
// sum is a fixed-order loop per output element: deterministic, no atomics.
__global__ void k_gather_rows(const float* __restrict__ x, const long long* __restrict__ idx, long P, int C, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P * C) return;
  const long p = i / C; const int c = (int)(i % C);
  out[i] = x[idx[p] * C + c];
}
__global__ void k_segment_sum(const float* __restrict__ rows, const long long* __restrict__ seg_ptr, const float* __restrict__ base, long N, int C,
                              float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const long n = i / C; const int c = (int)(i % C);
  float s = base ? base[i] : 0.f;
  for (long long p = seg_ptr[n]; p < seg_ptr[n + 1]; ++p) s += rows[p * C + c];
  out[i] = s;
}
// reverse of the gather with an arbitrary index: out[n] = sum over the rows p with idx[p] == n, rows listed per n in (order, ptr)
__global__ void k_segment_sum_perm(const float* __restrict__ rows, const long long* __restrict__ order, const long long* __restrict__ seg_ptr, long N, int C,
                                   float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const long n = i / C; const int c = (int)(i % C);
  float s = 0.f;
  for (long long q = seg_ptr[n]; q < seg_ptr[n + 1]; ++q) s += rows[order[q] * C + c];
  out[i] = s;
}

extern "C" {
int nq_gather_rows(const float* x, const int64_t* idx, int64_t P, int32_t C, float* out, void* stream) {
  if (!x || !idx || !out || C <= 0 || P < 0) return nq_fail(NQ_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gather_rows");
  if (P > 0) hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((P * C + 255) / 256)), dim3(256), 0, st, x, (const long long*)idx, (long)P, C, out);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
/* out[n] = base[n] (nullable) + sum of rows[seg_ptr[n] .. seg_ptr[n+1]) ; with `order` != NULL the rows are taken through it (rows[order[q]]) */
int nq_segment_sum(const float* rows, const int64_t* order, const int64_t* seg_ptr, const float* base, int64_t N, int32_t C, float* out, void* stream) {
  if (!rows || !seg_ptr || !out || C <= 0 || N < 0) return nq_fail(NQ_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "segment_sum");
  if (N > 0) {
    if (order) {
      if (base) return nq_fail(NQ_ERR_ARG, "base is not supported together with order");
      hipLaunchKernelGGL(k_segment_sum_perm, dim3((unsigned)((N * C + 255) / 256)), dim3(256), 0, st, rows, (const long long*)order, (const long long*)seg_ptr, (long)N, C, out);
    } else {
      hipLaunchKernelGGL(k_segment_sum, dim3((unsigned)((N * C + 255) / 256)), dim3(256), 0, st, rows, (const long long*)seg_ptr, base, (long)N, C, out);
    }
  }
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
}  // extern "C"

// ---- the other radial bases PhiSNet's NeuralNetwork can be built with (neural_network.py:210-221) -----------------------------------------------
//   kind 1  gaussian              fc * exp(-width (r - center_k)^2)                          gaussian_radial_basis_functions.py:10-33
//   kind 2  exp-gaussian          fc * exp(-width (e^{-alpha r} - center_k)^2)               exponential_gaussian_radial_basis_functions.py:10-32
//   kind 3  overlap-bernstein     x = log1p(alpha r) - alpha r; fc * exp(logc + n x + v log(1 - e^x))   overlap_bernstein_radial_basis_functions.py:10-41
//   kind 4  bernstein             x = log(r / cutoff);          fc * exp(logc + n x + v log(1 - e^x))   bernstein_radial_basis_functions.py:10-37
// t0 = center_k (1, 2) or logc_k (3, 4); t1 = n_k, t2 = v_k (3, 4).  GRAD: per-row dL/dalpha (kinds 2, 3; the others have no learnable parameter).
template <bool GRAD>
__global__ void k_radial_basis(int kind, const float* __restrict__ r, long P, int K, float alpha, float cutoff, float width, const float* __restrict__ t0,
                               const float* __restrict__ t1, const float* __restrict__ t2, float* __restrict__ out, const float* __restrict__ gout,
                               float* __restrict__ galpha_rows) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P * (GRAD ? 1 : K)) return;
  const long p = GRAD ? idx : idx / K;
  const float rr = r[p];
  const bool in = rr < cutoff;
  const float fc = in ? expf(-(rr * rr) / ((cutoff - rr) * (cutoff + rr))) : 0.f;
  float acc = 0.f;
  for (int k = GRAD ? 0 : (int)(idx % K); k < (GRAD ? K : (int)(idx % K) + 1); ++k) {
    float val = 0.f, dval = 0.f;                       // value and d(value)/d(alpha)
    if (in) {
      if (kind == 1) {
        const float d = rr - t0[k];
        val = fc * expf(-width * d * d);
      } else if (kind == 2) {
        const float e = expf(-alpha * rr), d = e - t0[k];
        val = fc * expf(-width * d * d);
        dval = val * (-2.0f * width * d) * (-rr * e);
      } else {
        float x, dx = 0.f;
        if (kind == 3) { const float ar = alpha * rr; x = log1pf(ar) - ar; dx = -rr * ar / (1.0f + ar); }
        else x = logf(rr / cutoff);
        const float om = -expm1f(x);                   // 1 - e^x
        val = fc * expf(t0[k] + t1[k] * x + t2[k] * logf(om));
        dval = val * (t1[k] - t2[k] * (1.0f - om) / om) * dx;
      }
    }
    if (GRAD) acc = fmaf(gout[p * K + k], dval, acc);
    else out[idx] = val;
  }
  if (GRAD) galpha_rows[p] = acc;
}

extern "C" {
int nq_radial_basis(int32_t kind, const float* r, int64_t P, int32_t K, float alpha, float cutoff, float width, const float* t0, const float* t1, const float* t2,
                    float* out, void* stream) {
  if (!r || !t0 || !out || kind < 1 || kind > 4 || (kind >= 3 && (!t1 || !t2))) return nq_fail(NQ_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "radial_basis");
  if (P > 0) hipLaunchKernelGGL((k_radial_basis<false>), dim3((unsigned)((P * K + 255) / 256)), dim3(256), 0, st, kind, r, (long)P, K, alpha, cutoff, width, t0, t1, t2, out,
                                nullptr, nullptr);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_radial_basis_grad_alpha(int32_t kind, const float* r, const float* grad_out, int64_t P, int32_t K, float alpha, float cutoff, float width, const float* t0,
                               const float* t1, const float* t2, float* galpha_rows, void* stream) {
  if (!r || !t0 || !grad_out || !galpha_rows || (kind != 2 && kind != 3) || (kind == 3 && (!t1 || !t2))) return nq_fail(NQ_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "radial_basis_grad");
  if (P > 0) hipLaunchKernelGGL((k_radial_basis<true>), dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, kind, r, (long)P, K, alpha, cutoff, width, t0, t1, t2, nullptr,
                                grad_out, galpha_rows);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
}
