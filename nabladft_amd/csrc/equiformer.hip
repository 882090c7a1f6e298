// EquiformerV2 building blocks (second half; reference /root/reference/nablaDFT/equiformer_v2/):
//   layer_norm.py:117-215          EquivariantLayerNormArraySphericalHarmonics (LayerNorm on the scalars, one balanced RMS over all l > 0)   k_eq_normsh_*
//   radial_function.py:5-28, transformer_block.py:150-151   torch.nn.LayerNorm over the last axis of a row                                 k_eq_ln_*
//   transformer_block.py:343-350, activation.py:52-61       SmoothLeakyReLU + dot with alpha_dot -> attention logits                      k_eq_logits_*
//   transformer_block.py:352       torch_geometric.utils.softmax over the in-edges of every target atom (edges are sorted by target: CSR)  k_eq_softmax_*
//   transformer_block.py:357-370   messages [E][coefficients][heads][value channels] times the attention weights [E][heads]             k_eq_headscale_*
//   so3.py:121-136 (rotate_inv rescale), drop.py:57-71 (GraphDropPath)   out[n][i][c] = x[n][i][c] * row_scale[n] * coef_scale[i]        k_eq_scale
// The SO(2) convolutions, the S2 grids, the rotations and the radius graph reuse escn.hip / gemm.hip.  Sums are in a fixed order; nothing here uses atomics.
#include "common.h"

namespace {

constexpr int EQ_MAXL = 6;
constexpr int EQ_MAXSEG = 8;

// sum over the workgroup (blockDim.x a multiple of 64, <= 1024), result to every thread; red: 17 floats of LDS
__device__ __forceinline__ float eq_block_sum(float v, float* red) {
  v = nq_wave_sum(v);
  const int nw = blockDim.x >> 6;
  if (nw == 1) return v;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) { float s = 0.f; for (int w = 0; w < nw; ++w) s += red[w]; red[16] = s; }
  __syncthreads();
  return red[16];
}

// ---- torch.nn.LayerNorm over rows of width W (one wavefront per row, lanes stride the row) -----------------------------------------------------------------
__global__ __launch_bounds__(256) void k_eq_ln_fwd(const float* __restrict__ x, long x_stride, const float* __restrict__ w, const float* __restrict__ b, long rows,
                                                   int W, float eps, float* __restrict__ y, long y_stride, float2* __restrict__ stats) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float* xr = x + r * x_stride;
  float s = 0.f;
  for (int c = lane; c < W; c += 64) s += xr[c];
  const float mean = nq_wave_sum(s) / W;
  float q = 0.f;
  for (int c = lane; c < W; c += 64) { const float d = xr[c] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(nq_wave_sum(q) / W + eps);
  float* yr = y + r * y_stride;
  for (int c = lane; c < W; c += 64) yr[c] = (xr[c] - mean) * rstd * w[c] + b[c];
  if (lane == 0) stats[r] = make_float2(mean, rstd);
}
// gx = rstd (gh - mean(gh) - xhat mean(gh xhat)), gh = g w
__global__ __launch_bounds__(256) void k_eq_ln_bwd(const float* __restrict__ x, long x_stride, const float* __restrict__ w, const float* __restrict__ g, long g_stride,
                                                   const float2* __restrict__ stats, long rows, int W, float* __restrict__ gx, long gx_stride) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float* xr = x + r * x_stride; const float* gr = g + r * g_stride;
  const float2 st = stats[r];
  float a = 0.f, bsum = 0.f;
  for (int c = lane; c < W; c += 64) { const float gh = gr[c] * w[c], xh = (xr[c] - st.x) * st.y; a += gh; bsum += gh * xh; }
  a = nq_wave_sum(a) / W; bsum = nq_wave_sum(bsum) / W;
  float* o = gx + r * gx_stride;
  for (int c = lane; c < W; c += 64) { const float gh = gr[c] * w[c], xh = (xr[c] - st.x) * st.y; o[c] = st.y * (gh - a - xh * bsum); }
}
// parameter gradients: part[chunk][0][c] = sum_r g xhat, part[chunk][1][c] = sum_r g over the chunk's rows; 64 columns x 16 row lanes per workgroup (lane j takes
// the rows r0 + j, r0 + j + 16, ...; the partial sums are added in lane order)
__global__ __launch_bounds__(1024) void k_eq_ln_wgrad(const float* __restrict__ x, long x_stride, const float* __restrict__ g, long g_stride,
                                                      const float2* __restrict__ stats, long rows, int W, int rows_per_chunk, float* __restrict__ part) {
  __shared__ float red[2][16][64];
  const int lane = threadIdx.x & 63, j = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + lane;
  const long r0 = (long)blockIdx.x * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  float gw = 0.f, gb = 0.f;
  if (c < W)
    for (long r = r0 + j; r < r1; r += 16) { const float2 st = stats[r]; const float gv = g[r * g_stride + c]; gw += gv * (x[r * x_stride + c] - st.x) * st.y; gb += gv; }
  red[0][j][lane] = gw; red[1][j][lane] = gb;
  __syncthreads();
  if (j < 2 && c < W) {
    float a = 0.f;
    for (int k = 0; k < 16; ++k) a += red[j][k][lane];
    part[((long)blockIdx.x * 2 + j) * W + c] = a;
  }
}
// out[i] = sum_k part[k * stride + i] in k order, i < cnt
__global__ void k_eq_reduce(const float* __restrict__ part, int nparts, long stride, long cnt, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;                               // four independent chains (fixed association), loads in flight together
  int k = 0;
  for (; k + 3 < nparts; k += 4) {
    a0 += part[(long)k * stride + i]; a1 += part[(long)(k + 1) * stride + i]; a2 += part[(long)(k + 2) * stride + i]; a3 += part[(long)(k + 3) * stride + i];
  }
  for (; k < nparts; ++k) a0 += part[(long)k * stride + i];
  out[i] = (a0 + a1) + (a2 + a3);
}

// ---- EquivariantLayerNormArraySphericalHarmonics: x [N][I][C], I = (lmax+1)^2; one workgroup per atom, thread = channel (blockDim = C rounded up to 64) ----
struct NormSh { const float* x; const float* w0; const float* b0; const float* aw; const float* bw; float* y; float* stats; long N; int I, C, lmax; float eps; };
__global__ __launch_bounds__(1024) void k_eq_normsh_fwd(NormSh p) {
  __shared__ float red[17];
  const long n = blockIdx.x;
  const int c = threadIdx.x;
  const bool on = c < p.C;
  const float* x = p.x + n * (long)p.I * p.C;
  float* y = p.y + n * (long)p.I * p.C;
  const float x0 = on ? x[c] : 0.f;
  const float mean = eq_block_sum(x0, red) / p.C;
  const float d0 = on ? x0 - mean : 0.f;
  const float rstd = 1.0f / sqrtf(eq_block_sum(d0 * d0, red) / p.C + p.eps);
  float q = 0.f;
  if (on) for (int i = 1; i < p.I; ++i) { const float v = x[(long)i * p.C + c]; q += p.bw[i - 1] * v * v; }
  const float s = 1.0f / sqrtf(eq_block_sum(q, red) / p.C + p.eps);          // (feature_norm + eps).pow(-0.5), layer_norm.py:199-200
  if (on) {
    y[c] = d0 * rstd * p.w0[c] + p.b0[c];
    int i = 1;
    for (int l = 1; l <= p.lmax; ++l) { const float a = p.aw[(l - 1) * p.C + c] * s; for (int m = 0; m < 2 * l + 1; ++m, ++i) y[(long)i * p.C + c] = x[(long)i * p.C + c] * a; }
  }
  if (c == 0) { p.stats[3 * n] = mean; p.stats[3 * n + 1] = rstd; p.stats[3 * n + 2] = s; }
}
struct NormShBwd { const float* x; const float* w0; const float* aw; const float* bw; const float* g; const float* stats; float* gx; float* part; long N;
                   int I, C, lmax, atoms_per_block; };
// part[block][0][c] = gw0, [1][c] = gb0, [1 + l][c] = g(affine_weight[l - 1]) summed over the block's atoms in order
__global__ __launch_bounds__(1024) void k_eq_normsh_bwd(NormShBwd p) {
  __shared__ float red[17];
  const int c = threadIdx.x;
  const bool on = c < p.C;
  float pw0 = 0.f, pb0 = 0.f, pa[EQ_MAXL];
  for (int l = 0; l < EQ_MAXL; ++l) pa[l] = 0.f;
  const long n0 = (long)blockIdx.x * p.atoms_per_block, n1 = min(p.N, n0 + p.atoms_per_block);
  for (long n = n0; n < n1; ++n) {
    const float* x = p.x + n * (long)p.I * p.C; const float* g = p.g + n * (long)p.I * p.C;
    float* gx = p.gx + n * (long)p.I * p.C;
    const float mean = p.stats[3 * n], rstd = p.stats[3 * n + 1], s = p.stats[3 * n + 2];
    // scalars: LayerNorm
    const float xh = on ? (x[c] - mean) * rstd : 0.f, g0 = on ? g[c] : 0.f, gh = on ? g0 * p.w0[c] : 0.f;
    const float a = eq_block_sum(gh, red) / p.C, b = eq_block_sum(gh * xh, red) / p.C;
    if (on) { gx[c] = rstd * (gh - a - xh * b); pw0 += g0 * xh; pb0 += g0; }
    // l > 0: y = x s aw_l,  s = (mean_c sum_i bw_i x_i^2 + eps)^-1/2
    float G = 0.f;
    if (on) {
      int i = 1;
      for (int l = 1; l <= p.lmax; ++l) {
        const float aw = p.aw[(l - 1) * p.C + c];
        float t = 0.f;
        for (int m = 0; m < 2 * l + 1; ++m, ++i) t += g[(long)i * p.C + c] * x[(long)i * p.C + c];
        pa[l - 1] += t * s;
        G += t * aw;
      }
    }
    G = eq_block_sum(G, red);
    const float k = G * s * s * s / p.C;
    if (on) {
      int i = 1;
      for (int l = 1; l <= p.lmax; ++l) {
        const float as = p.aw[(l - 1) * p.C + c] * s;
        for (int m = 0; m < 2 * l + 1; ++m, ++i) gx[(long)i * p.C + c] = g[(long)i * p.C + c] * as - k * p.bw[i - 1] * x[(long)i * p.C + c];
      }
    }
  }
  if (on) {
    float* o = p.part + (long)blockIdx.x * (p.lmax + 2) * p.C;
    o[c] = pw0; o[p.C + c] = pb0;
    for (int l = 0; l < p.lmax; ++l) o[(long)(2 + l) * p.C + c] = pa[l];
  }
}

// ---- attention logits: z[e][h] = sum_a alpha_dot[h][a] act(x[e][h][a]),  act(x) = (1+s)/2 x + (1-s)/2 x (2 sigmoid(x) - 1), s = 0.2 ----------------------
__device__ __forceinline__ float eq_sleaky(float x) { return 0.6f * x + 0.4f * x * (2.0f * nq_sigmoid(x) - 1.0f); }
__device__ __forceinline__ float eq_dsleaky(float x) { const float sg = nq_sigmoid(x); return 0.6f + 0.4f * ((2.0f * sg - 1.0f) + 2.0f * x * sg * (1.0f - sg)); }
// one wavefront per (e, h) row
__global__ __launch_bounds__(256) void k_eq_logits_fwd(const float* __restrict__ x, const float* __restrict__ ad, long rows, int H, int A, float* __restrict__ z) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  const int h = (int)(r % H);
  float s = 0.f;
  for (int a = lane; a < A; a += 64) s += ad[h * A + a] * eq_sleaky(x[r * A + a]);
  s = nq_wave_sum(s);
  if (lane == 0) z[r] = s;
}
__global__ void k_eq_logits_bwd_x(const float* __restrict__ x, const float* __restrict__ ad, const float* __restrict__ gz, long rows, int H, int A,
                                  float* __restrict__ gx) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * A) return;
  const long r = t / A; const int a = (int)(t - r * A), h = (int)(r % H);
  gx[t] = gz[r] * ad[h * A + a] * eq_dsleaky(x[t]);
}
// part[chunk][h][a] = sum over the chunk's edges of gz[e][h] act(x[e][h][a]); thread = (h, a)
__global__ __launch_bounds__(256) void k_eq_logits_bwd_w(const float* __restrict__ x, const float* __restrict__ gz, long E, int H, int A, int edges_per_chunk,
                                                         float* __restrict__ part) {
  const int t = blockIdx.y * 256 + threadIdx.x;
  if (t >= H * A) return;
  const int h = t / A;
  const long e0 = (long)blockIdx.x * edges_per_chunk, e1 = min(E, e0 + edges_per_chunk);
  float s = 0.f;
  for (long e = e0; e < e1; ++e) s += gz[e * H + h] * eq_sleaky(x[e * H * A + t]);
  part[(long)blockIdx.x * H * A + t] = s;
}

// ---- softmax over the in-edges of a target atom (CSR ptr), per head; thread = (atom, head) ---------------------------------------------------------------------
__global__ void k_eq_softmax_fwd(const float* __restrict__ z, const int* __restrict__ ptr, long N, int H, float* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * H) return;
  const long n = t / H; const int h = (int)(t - n * H);
  const int a = ptr[n], b = ptr[n + 1];
  if (a >= b) return;
  float mx = -INFINITY;
  for (int e = a; e < b; ++e) mx = fmaxf(mx, z[(long)e * H + h]);
  float den = 0.f;
  for (int e = a; e < b; ++e) den += expf(z[(long)e * H + h] - mx);
  const float inv = 1.0f / (den + 1e-16f);                                  // torch_geometric.utils.softmax: out / (sum + 1e-16)
  for (int e = a; e < b; ++e) out[(long)e * H + h] = expf(z[(long)e * H + h] - mx) * inv;
}
__global__ void k_eq_softmax_bwd(const float* __restrict__ y, const float* __restrict__ gy, const int* __restrict__ ptr, long N, int H, float* __restrict__ gz) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * H) return;
  const long n = t / H; const int h = (int)(t - n * H);
  const int a = ptr[n], b = ptr[n + 1];
  float dot = 0.f;
  for (int e = a; e < b; ++e) dot += y[(long)e * H + h] * gy[(long)e * H + h];
  for (int e = a; e < b; ++e) gz[(long)e * H + h] = y[(long)e * H + h] * (gy[(long)e * H + h] - dot);
}

// ---- messages times attention weights: per-block tensors x_b [E][rows_b][H V], alpha [E][H] ---------------------------------------------------------------------
struct HeadSeg { int n; int rows[EQ_MAXSEG]; const float* x[EQ_MAXSEG]; const float* g[EQ_MAXSEG]; float* out[EQ_MAXSEG]; };
// forward (g == null): out_b = x_b alpha.  backward: out_b = g_b alpha (the gradient of x_b)
__global__ __launch_bounds__(256) void k_eq_headscale(HeadSeg s, const float* __restrict__ alpha, long E, int H, int V, bool backward) {
  const long e = blockIdx.x;
  const int HV = H * V;
  for (int k = 0; k < s.n; ++k) {
    const long base = e * (long)s.rows[k] * HV;
    const float* src = backward ? s.g[k] : s.x[k];
    for (int t = threadIdx.x; t < s.rows[k] * HV; t += 256) s.out[k][base + t] = src[base + t] * alpha[e * H + (t % HV) / V];
  }
}
// galpha[e][h] = sum_b sum_rows sum_v g_b x_b; one workgroup per edge, thread = channel (h, v) loops over the rows, then V partial sums are added in order
__global__ __launch_bounds__(1024) void k_eq_headscale_bwd_alpha(HeadSeg s, long E, int H, int V, float* __restrict__ galpha) {
  extern __shared__ float hs_lds[];
  const long e = blockIdx.x;
  const int HV = H * V, t = threadIdx.x;
  if (t < HV) {
    float acc = 0.f;
    for (int k = 0; k < s.n; ++k) {
      const long base = e * (long)s.rows[k] * HV;
      for (int r = 0; r < s.rows[k]; ++r) acc += s.g[k][base + (long)r * HV + t] * s.x[k][base + (long)r * HV + t];
    }
    hs_lds[t] = acc;
  }
  __syncthreads();
  if (t < H) { float a = 0.f; for (int v = 0; v < V; ++v) a += hs_lds[t * V + v]; galpha[e * H + t] = a; }
}

// ---- out[n][i][c] = x[n][i][c] * (row_scale ? row_scale[row_index ? row_index[n] : n] : 1) * (coef_scale ? coef_scale[i] : 1) ------------------------------------
__global__ void k_eq_scale(const float* __restrict__ x, const float* __restrict__ row_scale, const int* __restrict__ row_index, const float* __restrict__ coef_scale,
                           long N, int I, int C, float* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * I * C) return;
  const long n = t / ((long)I * C); const int i = (int)((t - n * (long)I * C) / C);
  float f = 1.0f;
  if (row_scale) f = row_scale[row_index ? row_index[n] : n];
  if (coef_scale) f *= coef_scale[i];
  out[t] = x[t] * f;
}

}  // namespace

#define EQ_GRID(total) dim3((unsigned)(((total) + 255) / 256)), dim3(256), 0, st

extern "C" {

int nq_eq_layernorm_forward(const float* x, int64_t x_stride, const float* weight, const float* bias, int64_t rows, int32_t W, float eps, float* y, int64_t y_stride,
                            float* stats, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "eq_ln_fwd");
  if (rows <= 0) return NQ_OK;
  if (!x || !weight || !bias || !y || !stats || W <= 0) return nq_fail(NQ_ERR_ARG, "layernorm: bad argument");
  hipLaunchKernelGGL(k_eq_ln_fwd, dim3(nq_cdiv(rows, 4)), dim3(256), 0, st, x, (long)x_stride, weight, bias, (long)rows, W, eps, y, (long)y_stride, (float2*)stats);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
static int ln_rows_per_chunk(long rows) { const int per = nq_cdiv(rows, 128); return per < 64 ? 64 : per; }        // <= 128 chunks of >= 64 rows
size_t nq_eq_layernorm_scratch_floats(int64_t rows, int32_t W) { return (size_t)nq_cdiv(rows, ln_rows_per_chunk(rows)) * 2 * W + 64; }
/* grad_x (strided like x), grad_weight [W], grad_bias [W]; scratch: nq_eq_layernorm_scratch_floats(rows, W) floats. */
int nq_eq_layernorm_backward(const float* x, int64_t x_stride, const float* weight, const float* grad_y, int64_t g_stride, const float* stats, int64_t rows, int32_t W,
                             float* grad_x, int64_t gx_stride, float* grad_weight, float* grad_bias, float* scratch, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "eq_ln_bwd");
  if (!x || !weight || !grad_y || !stats || !grad_x || !grad_weight || !grad_bias || !scratch || W <= 0) return nq_fail(NQ_ERR_ARG, "layernorm: bad argument");
  if (rows <= 0) {
    NQ_HIP(hipMemsetAsync(grad_weight, 0, sizeof(float) * W, st));
    NQ_HIP(hipMemsetAsync(grad_bias, 0, sizeof(float) * W, st));
    return NQ_OK;
  }
  hipLaunchKernelGGL(k_eq_ln_bwd, dim3(nq_cdiv(rows, 4)), dim3(256), 0, st, x, (long)x_stride, weight, grad_y, (long)g_stride, (const float2*)stats, (long)rows, W,
                     grad_x, (long)gx_stride);
  NQ_LAUNCH_CHECK();
  const int per = ln_rows_per_chunk(rows);
  const int chunks = nq_cdiv(rows, per);
  hipLaunchKernelGGL(k_eq_ln_wgrad, dim3(chunks, nq_cdiv(W, 64)), dim3(1024), 0, st, x, (long)x_stride, grad_y, (long)g_stride, (const float2*)stats, (long)rows, W,
                     per, scratch);
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_eq_reduce, EQ_GRID((long)W), scratch, chunks, (long)2 * W, (long)W, grad_weight);        // part[k][0][c] at k * 2W + c
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_eq_reduce, EQ_GRID((long)W), scratch + W, chunks, (long)2 * W, (long)W, grad_bias);      // part[k][1][c] at k * 2W + W + c
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

/* x, y [N][(lmax+1)^2][C]; norm_l0 weight / bias [C]; affine_weight [lmax][C]; balance_degree_weight [(lmax+1)^2 - 1]; stats [N][3] (out). */
int nq_eq_norm_sh_forward(const float* x, const float* w0, const float* b0, const float* affine_weight, const float* balance_weight, int64_t N, int32_t lmax,
                          int32_t C, float eps, float* y, float* stats, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "eq_normsh_fwd");
  if (N <= 0) return NQ_OK;
  if (!x || !w0 || !b0 || !affine_weight || !balance_weight || !y || !stats || lmax < 1 || lmax > EQ_MAXL || C < 1 || C > 1024)
    return nq_fail(NQ_ERR_ARG, "norm_sh: bad argument (1 <= lmax <= %d, C <= 1024)", EQ_MAXL);
  NormSh p{x, w0, b0, affine_weight, balance_weight, y, stats, (long)N, (lmax + 1) * (lmax + 1), C, lmax, eps};
  hipLaunchKernelGGL(k_eq_normsh_fwd, dim3((unsigned)N), dim3((C + 63) / 64 * 64), 0, st, p);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
size_t nq_eq_norm_sh_scratch_floats(int64_t N, int32_t lmax, int32_t C) { return (size_t)nq_cdiv(N, 4) * (lmax + 2) * C + 64; }
/* grad_x [N][I][C]; grad_w0 [C], grad_b0 [C], grad_affine_weight [lmax][C]. */
int nq_eq_norm_sh_backward(const float* x, const float* w0, const float* affine_weight, const float* balance_weight, const float* grad_y, const float* stats,
                           int64_t N, int32_t lmax, int32_t C, float* grad_x, float* grad_w0, float* grad_b0, float* grad_affine_weight, float* scratch,
                           void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "eq_normsh_bwd");
  if (!x || !w0 || !affine_weight || !balance_weight || !grad_y || !stats || !grad_x || !grad_w0 || !grad_b0 || !grad_affine_weight || !scratch || lmax < 1 ||
      lmax > EQ_MAXL || C < 1 || C > 1024)
    return nq_fail(NQ_ERR_ARG, "norm_sh: bad argument");
  if (N <= 0) {
    NQ_HIP(hipMemsetAsync(grad_w0, 0, sizeof(float) * C, st));
    NQ_HIP(hipMemsetAsync(grad_b0, 0, sizeof(float) * C, st));
    NQ_HIP(hipMemsetAsync(grad_affine_weight, 0, sizeof(float) * lmax * C, st));
    return NQ_OK;
  }
  const int per = 4, blocks = nq_cdiv(N, per);
  NormShBwd p{x, w0, affine_weight, balance_weight, grad_y, stats, grad_x, scratch, (long)N, (lmax + 1) * (lmax + 1), C, lmax, per};
  hipLaunchKernelGGL(k_eq_normsh_bwd, dim3(blocks), dim3((C + 63) / 64 * 64), 0, st, p);
  NQ_LAUNCH_CHECK();
  const long cnt = (long)(lmax + 2) * C;
  hipLaunchKernelGGL(k_eq_reduce, EQ_GRID((long)C), scratch, blocks, cnt, (long)C, grad_w0);
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_eq_reduce, EQ_GRID((long)C), scratch + C, blocks, cnt, (long)C, grad_b0);
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_eq_reduce, EQ_GRID((long)lmax * C), scratch + 2 * C, blocks, cnt, (long)lmax * C, grad_affine_weight);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

/* x [E][H][A] (after alpha_norm), alpha_dot [H][A] -> z [E][H]. */
int nq_eq_logits_forward(const float* x, const float* alpha_dot, int64_t E, int32_t H, int32_t A, float* z, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "eq_logits_fwd");
  if (E <= 0) return NQ_OK;
  if (!x || !alpha_dot || !z || H < 1 || A < 1) return nq_fail(NQ_ERR_ARG, "logits: bad argument");
  hipLaunchKernelGGL(k_eq_logits_fwd, dim3(nq_cdiv((long)E * H, 4)), dim3(256), 0, st, x, alpha_dot, (long)E * H, H, A, z);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
size_t nq_eq_logits_scratch_floats(int64_t E, int32_t H, int32_t A) { return (size_t)nq_cdiv(E, nq_cdiv(E, 256) < 64 ? 64 : nq_cdiv(E, 256)) * H * A + 64; }
int nq_eq_logits_backward(const float* x, const float* alpha_dot, const float* grad_z, int64_t E, int32_t H, int32_t A, float* grad_x, float* grad_alpha_dot,
                          float* scratch, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "eq_logits_bwd");
  if (!x || !alpha_dot || !grad_z || !grad_x || !grad_alpha_dot || !scratch || H < 1 || A < 1) return nq_fail(NQ_ERR_ARG, "logits: bad argument");
  if (E <= 0) { NQ_HIP(hipMemsetAsync(grad_alpha_dot, 0, sizeof(float) * H * A, st)); return NQ_OK; }
  hipLaunchKernelGGL(k_eq_logits_bwd_x, EQ_GRID((long)E * H * A), x, alpha_dot, grad_z, (long)E * H, H, A, grad_x);
  NQ_LAUNCH_CHECK();
  const int per = nq_cdiv(E, 256) < 64 ? 64 : nq_cdiv(E, 256);
  const int chunks = nq_cdiv(E, per);
  hipLaunchKernelGGL(k_eq_logits_bwd_w, dim3(chunks, nq_cdiv(H * A, 256)), dim3(256), 0, st, x, grad_z, (long)E, H, A, per, scratch);
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_eq_reduce, EQ_GRID((long)H * A), scratch, chunks, (long)H * A, (long)H * A, grad_alpha_dot);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

/* z, out [E][H]; ptr [N + 1]: the in-edges of atom n are [ptr[n], ptr[n+1]). */
int nq_eq_softmax_forward(const float* z, const int32_t* ptr, int64_t N, int32_t H, float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "eq_softmax_fwd");
  if (N <= 0) return NQ_OK;
  if (!z || !ptr || !out) return nq_fail(NQ_ERR_ARG, "softmax: null argument");
  hipLaunchKernelGGL(k_eq_softmax_fwd, EQ_GRID((long)N * H), z, ptr, (long)N, H, out);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_eq_softmax_backward(const float* y, const float* grad_y, const int32_t* ptr, int64_t N, int32_t H, float* grad_z, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "eq_softmax_bwd");
  if (N <= 0) return NQ_OK;
  if (!y || !grad_y || !ptr || !grad_z) return nq_fail(NQ_ERR_ARG, "softmax: null argument");
  hipLaunchKernelGGL(k_eq_softmax_bwd, EQ_GRID((long)N * H), y, grad_y, ptr, (long)N, H, grad_z);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

/* nseg per-block tensors x_b [E][rows_b][H V] (HOST arrays of device pointers); alpha [E][H].
 * grad == NULL: out_b = x_b alpha.  grad != NULL: out_b = grad_b alpha (= d/dx_b) and, if grad_alpha != NULL, grad_alpha [E][H] = sum g_b x_b. */
int nq_eq_head_scale(int32_t nseg, const int32_t* rows, const float* const* x, const float* const* grad, const float* alpha, int64_t E, int32_t H, int32_t V,
                     float* const* out, float* grad_alpha, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, grad ? "eq_headscale_bwd" : "eq_headscale_fwd");
  if (E <= 0) return NQ_OK;
  if (nseg < 1 || nseg > EQ_MAXSEG || !rows || !x || !alpha || !out || H < 1 || V < 1 || H * V > 1024) return nq_fail(NQ_ERR_ARG, "head_scale: bad argument");
  HeadSeg s{};
  s.n = nseg;
  for (int k = 0; k < nseg; ++k) {
    s.rows[k] = rows[k]; s.x[k] = x[k]; s.g[k] = grad ? grad[k] : nullptr; s.out[k] = out[k];
    if (!x[k] || !out[k] || (grad && !grad[k])) return nq_fail(NQ_ERR_ARG, "head_scale: null block");
  }
  hipLaunchKernelGGL(k_eq_headscale, dim3((unsigned)E), dim3(256), 0, st, s, alpha, (long)E, H, V, grad != nullptr);
  NQ_LAUNCH_CHECK();
  if (grad && grad_alpha) {
    hipLaunchKernelGGL(k_eq_headscale_bwd_alpha, dim3((unsigned)E), dim3((H * V + 63) / 64 * 64), sizeof(float) * H * V, st, s, (long)E, H, V, grad_alpha);
    NQ_LAUNCH_CHECK();
  }
  return NQ_OK;
}

/* out[n][i][c] = x[n][i][c] * row_scale[row_index ? row_index[n] : n] * coef_scale[i]; row_scale / row_index / coef_scale nullable. */
int nq_eq_scale(const float* x, const float* row_scale, const int32_t* row_index, const float* coef_scale, int64_t N, int32_t I, int32_t C, float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "eq_scale");
  if (N <= 0) return NQ_OK;
  if (!x || !out) return nq_fail(NQ_ERR_ARG, "scale: null argument");
  hipLaunchKernelGGL(k_eq_scale, EQ_GRID((long)N * I * C), x, row_scale, row_index, coef_scale, (long)N, I, C, out);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

}  // extern "C"
