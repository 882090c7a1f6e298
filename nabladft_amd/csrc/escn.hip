// eSCN building blocks (reference /root/reference/nablaDFT/escn/escn.py, so3.py):
//   escn.py:253-255,435-487   directed radius graph (first K sources per target in index order), edge vectors / distances, edge rotation matrices
//   so3.py:377-425            Wigner-D matrices from the rotation matrices: D^l = Z(alpha) J_l Z(beta) J_l Z(gamma), written directly as the rows the SO(2)
//                             convolution keeps (|m| <= mmax, m-primary order of CoefficientMapping, so3.py:23-118)
//   so3.py:265-300,361-375    SO3_Embedding._rotate / _rotate_inv / to_grid / from_grid and the point sampling of escn.py:399-407: all are "one small
//                             matrix per row times the row's [coefficients x channels] block" -> k_rowop_*: per-row or shared matrix, optional gather of
//                             the row from node storage, strided outputs, optional accumulation
//   smearing.py:14-31         GaussianSmearing
// The dense layers in between run on the fp32 MFMA GEMMs (gemm.hip).  Sums are in a fixed order; nothing here uses atomics.
#include "common.h"

typedef unsigned long long es_u64;
__device__ __forceinline__ es_u64 es_below(int lane) { return lane ? (~0ull >> (64 - lane)) : 0ull; }
__device__ __forceinline__ float es_sqrt_rn(float x) {
  if (!(x > 0.0f)) return 0.0f;
  const float s = __builtin_sqrtf(x);
  const float r = fmaf(-s, s, x);
  return fmaf(r, 0.5f / s, s);
}

// ---- directed radius graph: one wavefront per target atom, candidates = the atoms of its molecule in index order, strict d^2 < r^2, first K kept ---------
__global__ __launch_bounds__(64) void k_es_graph(const float* __restrict__ pos, const int* __restrict__ mol_ptr, const int* __restrict__ atom_mol, float r2, int K,
                                                 const int* __restrict__ ptr, int* __restrict__ deg, int* __restrict__ src, int* __restrict__ dst,
                                                 float4* __restrict__ geom) {
  const int i = blockIdx.x, lane = threadIdx.x;
  const int g = atom_mol[i], a0 = mol_ptr[g], a1 = mol_ptr[g + 1];
  const float xi = pos[3 * (long)i], yi = pos[3 * (long)i + 1], zi = pos[3 * (long)i + 2];
  int kept = 0;
  const int base = ptr ? ptr[i] : 0;
  for (int j0 = a0; j0 < a1 && kept < K; j0 += 64) {
    const int j = j0 + lane;
    bool in = false;
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (j < a1 && j != i) {
      dx = pos[3 * (long)j] - xi; dy = pos[3 * (long)j + 1] - yi; dz = pos[3 * (long)j + 2] - zi;     // edge_distance_vec = pos[j] - pos[i] (escn.py:282)
      float s = __fmul_rn(dx, dx);
      s = __fadd_rn(s, __fmul_rn(dy, dy));
      s = __fadd_rn(s, __fmul_rn(dz, dz));
      in = s < r2;
    }
    const es_u64 m = __ballot(in);
    const int rank = kept + __popcll(m & es_below(lane));
    if (in && rank < K && src) {
      float s = __fmul_rn(dx, dx);                    // distance as torch.norm evaluates it on the CPU (see gemnet_graph.hip: k_gn_geom)
      s = fmaf(dy, dy, s);
      s = fmaf(dz, dz, s);
      const int slot = base + rank;
      src[slot] = j; dst[slot] = i;
      geom[slot] = make_float4(dx, dy, dz, es_sqrt_rn(s));
    }
    kept = min(K, kept + __popcll(m));
  }
  if (lane == 0 && deg) deg[i] = kept;
}
__global__ __launch_bounds__(1024) void k_es_scan(const int* __restrict__ in, int n, int* __restrict__ out) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < n ? in[i] : 0;
    int s = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(s, off, 64); if (lane >= off) s += t; }
    if (lane == 63) wsum[wave] = s;
    __syncthreads();
    int p = carry;
    for (int w = 0; w < wave; ++w) p += wsum[w];
    if (i < n) out[i] = p + s - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = p + s;
    __syncthreads();
  }
  if (threadIdx.x == 0) out[n] = carry;
}

// ---- edge frames (escn.py:435-487).  The reference draws a random helper vector per edge; any helper not parallel to the edge gives the same model output
// (the SO(2) convolution commutes with rotations about the edge), so the helper here is the coordinate axis least aligned with the edge: deterministic. --------
__global__ void k_es_frames(const float4* __restrict__ geom, int E, float* __restrict__ rot) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const float4 v = geom[e];
  const float inv = 1.0f / v.w;
  const float nx0 = v.x * inv, nx1 = v.y * inv, nx2 = v.z * inv;
  float h0 = 0.f, h1 = 0.f, h2 = 0.f;
  const float a0 = fabsf(nx0), a1 = fabsf(nx1), a2 = fabsf(nx2);
  if (a0 <= a1 && a0 <= a2) h0 = 1.f; else if (a1 <= a2) h1 = 1.f; else h2 = 1.f;
  float z0 = nx1 * h2 - nx2 * h1, z1 = nx2 * h0 - nx0 * h2, z2 = nx0 * h1 - nx1 * h0;       // norm_z = normalize(norm_x x helper)
  float n = rsqrtf(z0 * z0 + z1 * z1 + z2 * z2);
  z0 *= n; z1 *= n; z2 *= n;
  float y0 = nx1 * z2 - nx2 * z1, y1 = nx2 * z0 - nx0 * z2, y2 = nx0 * z1 - nx1 * z0;       // norm_y = normalize(norm_x x norm_z), then negated
  n = rsqrtf(y0 * y0 + y1 * y1 + y2 * y2);
  y0 *= -n; y1 *= -n; y2 *= -n;
  float* r = rot + 9 * (long)e;      // edge_rot_mat = transpose([norm_z | norm_x | norm_y] as columns): rows are norm_z, norm_x, norm_y
  r[0] = z0; r[1] = z1; r[2] = z2; r[3] = nx0; r[4] = nx1; r[5] = nx2; r[6] = y0; r[7] = y1; r[8] = y2;
}

// ---- Euler angles of the frames (so3.py:378-383 with e3nn's y-polar conventions) ----------------------------------------------------------------------------
// Evaluated in float64: beta = acos(x_y) loses digits like 1 / sqrt(1 - x_y^2) for edges close to the polar axis, and every Wigner entry is a product of
// cos / sin of up to l = 6 times these angles.  In float32 (what the reference's CPU run does) the rows carry errors of several 1e-6, which the network
// passes on to the forces at the 1e-5 level; in float64 the rows are exact to float32 rounding at the cost of ~30 double-precision sincos per row, once per batch.
__global__ void k_es_angles(const float* __restrict__ rot, int E, double* __restrict__ ang) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const float* R = rot + 9 * (long)e;
  double x0 = R[1], x1 = R[4], x2 = R[7];                          // R @ (0, 1, 0)
  const double n = 1.0 / sqrt(x0 * x0 + x1 * x1 + x2 * x2);
  x0 *= n; x1 *= n; x2 *= n;
  x1 = fmin(1.0, fmax(-1.0, x1));
  const double beta = acos(x1), alpha = atan2(x0, x2);
  // first row of (Ry(alpha) Rx(beta))^T R: Ry Rx has first column (cos a, 0, -sin a)
  const double ca = cos(alpha), sa = sin(alpha);
  const double m00 = ca * R[0] - sa * R[6], m02 = ca * R[2] - sa * R[8];
  ang[3 * (long)e] = alpha; ang[3 * (long)e + 1] = beta; ang[3 * (long)e + 2] = atan2(m02, m00);
}

// ---- Wigner rows: W[e][b][:] = row red_row[b] of block l = red_l[b] of D(e), placed at the columns of that block (zero elsewhere) ----------------------------
#define ES_MAXL 6
__global__ void k_es_wigner(const double* __restrict__ ang, int E, const float* __restrict__ Jall, const int* __restrict__ Joff, const int* __restrict__ red_l,
                            const int* __restrict__ red_row, int n_red, int n_full, float* __restrict__ W) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)E * n_red) return;
  const int e = (int)(t / n_red), b = (int)(t - (long)e * n_red);
  const int l = red_l[b], i = red_row[b], w = 2 * l + 1;
  const double alpha = ang[3 * (long)e], beta = ang[3 * (long)e + 1], gamma = ang[3 * (long)e + 2];
  const float* J = Jall + Joff[l];
  double A[2 * ES_MAXL + 1], B[2 * ES_MAXL + 1], Cc[2 * ES_MAXL + 1];
  const double fi = (double)(l - i);
  const double ci = cos(fi * alpha), si = sin(fi * alpha);
  // row i of Z(alpha) J: Z[i][i] = cos(f_i a), Z[i][2l - i] = sin(f_i a) (the diagonal wins at i = l)
  for (int p = 0; p < w; ++p) A[p] = (i == l ? 1.0 : ci) * (double)J[i * w + p] + (i == l ? 0.0 : si * (double)J[(2 * l - i) * w + p]);
  // times Z(beta): (A Z)[q] = A[q] cos(f_q b) + A[2l - q] sin(f_{2l-q} b)
  for (int q = 0; q < w; ++q) {
    const double fq = (double)(l - q);
    B[q] = q == l ? A[q] : A[q] * cos(fq * beta) + A[2 * l - q] * sin(-fq * beta);
  }
  for (int q = 0; q < w; ++q) {
    double s = 0.0;
    for (int p = 0; p < w; ++p) s += B[p] * (double)J[p * w + q];
    Cc[q] = s;
  }
  float* out = W + ((long)e * n_red + b) * n_full;
  for (int q = 0; q < n_full; ++q) out[q] = 0.f;
  for (int q = 0; q < w; ++q) {
    const double fq = (double)(l - q);
    out[l * l + q] = (float)(q == l ? Cc[q] : Cc[q] * cos(fq * gamma) + Cc[2 * l - q] * sin(-fq * gamma));
  }
}

// ---- GaussianSmearing: out[e][k] = exp(coeff (d_e - offset_k)^2) -------------------------------------------------------------------------------------------
__global__ void k_es_smear(const float4* __restrict__ geom, long E, int K, const float* __restrict__ offset, float coeff, float* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= E * K) return;
  const long e = t / K; const int k = (int)(t - e * K);
  const float x = geom[e].w - offset[k];
  out[t] = expf(coeff * x * x);
}

// ---- row operators ---------------------------------------------------------------------------------------------------------------------------------------------
// forward:   out[o][i][c] (+)= sum_s R_o[i][s] X_row(o)[s][c]          R_o = R + o * r_stride (r_stride 0: one shared matrix), row(o) = index ? index[o] : o
// transpose: out[o][s][c] (+)= sum_i R_o[i][s] Y[o][i][c]
// One workgroup per row; the matrix and the row block are staged in LDS.
#define ROWOP_MAXSEG 8
// A "segmented" side: the rows [seg_start[k], seg_start[k+1]) of that side live in their own contiguous tensor seg_ptr[k] ([n][rows_k][C]) -- the m-blocks of
// an SO(2) layer -- instead of one [n][rows][C] tensor.
struct RowSeg { int n; int start[ROWOP_MAXSEG + 1]; float* ptr[ROWOP_MAXSEG]; };
struct RowOp { const float* R; long r_stride; const float* X; long x_stride; const int* index; float* out; long out_stride; int I, NSS, C, accumulate;
               RowSeg segI, segS; };

__device__ __forceinline__ float* rowseg_addr(const RowSeg& sg, long o, int row, int C) {
  int k = 0;
  while (k + 1 < sg.n && row >= sg.start[k + 1]) ++k;
  return sg.ptr[k] + (o * (sg.start[k + 1] - sg.start[k]) + (row - sg.start[k])) * C;
}

__global__ __launch_bounds__(256) void k_rowop_fwd(RowOp p) {
  extern __shared__ __attribute__((aligned(16))) float es_lds[];
  float* sR = es_lds; float* sX = es_lds + p.I * p.NSS;
  const long o = blockIdx.x;
  const float* R = p.R + o * p.r_stride;
  for (int t = threadIdx.x; t < p.I * p.NSS; t += 256) sR[t] = R[t];
  if (p.segS.n) {                                   // input rows (the s side) come from the per-block tensors
    for (int t = threadIdx.x; t < p.NSS * p.C; t += 256) { const int k = t / p.C; sX[t] = rowseg_addr(p.segS, o, k, p.C)[t - k * p.C]; }
  } else {
    const float* X = p.X + (p.index ? (long)p.index[o] : o) * p.x_stride;
    for (int t = threadIdx.x; t < p.NSS * p.C; t += 256) sX[t] = X[t];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < p.I * p.C; t += 256) {
    const int i = t / p.C, ch = t - i * p.C;
    float acc = 0.f;
    for (int k = 0; k < p.NSS; ++k) acc += sR[i * p.NSS + k] * sX[k * p.C + ch];
    float* dst = p.segI.n ? rowseg_addr(p.segI, o, i, p.C) + ch : p.out + o * p.out_stride + t;
    *dst = p.accumulate ? *dst + acc : acc;
  }
}
__global__ __launch_bounds__(256) void k_rowop_tr(RowOp p) {
  extern __shared__ __attribute__((aligned(16))) float es_lds[];
  float* sR = es_lds; float* sY = es_lds + p.I * p.NSS;
  const long o = blockIdx.x;
  const float* R = p.R + o * p.r_stride;
  for (int t = threadIdx.x; t < p.I * p.NSS; t += 256) sR[t] = R[t];
  if (p.segI.n) {                                   // input rows (the i side) come from the per-block tensors
    for (int t = threadIdx.x; t < p.I * p.C; t += 256) { const int i = t / p.C; sY[t] = rowseg_addr(p.segI, o, i, p.C)[t - i * p.C]; }
  } else {
    const float* Y = p.X + (p.index ? (long)p.index[o] : o) * p.x_stride;
    for (int t = threadIdx.x; t < p.I * p.C; t += 256) sY[t] = Y[t];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < p.NSS * p.C; t += 256) {
    const int k = t / p.C, ch = t - k * p.C;
    float acc = 0.f;
    for (int i = 0; i < p.I; ++i) acc += sR[i * p.NSS + k] * sY[i * p.C + ch];
    float* dst = p.segS.n ? rowseg_addr(p.segS, o, k, p.C) + ch : p.out + o * p.out_stride + t;
    *dst = p.accumulate ? *dst + acc : acc;
  }
}

// ---- shared-matrix row operator on the matrix cores -----------------------------------------------------------------------------------------------------------
// The S2-grid transforms (to_grid / from_grid, so3.py:301-375) and the sphere-point sampling apply ONE constant matrix to every edge / node row: per row o an
// [M x K] x [K x C] product with M, K = (grid points, coefficients) or the reverse.  k_rowop_* does it with two LDS reads per FMA (measured 12 TFLOP/s on the
// per-edge grid activations: 38 ms of the 177-ms eSCN step); here the constant matrix sits in LDS as the MFMA A operand for the lifetime of a persistent
// workgroup, the row's block X_o (B operand) is staged once per row, and v_mfma_f32_32x32x2_f32 does the arithmetic (exact f32, fixed k order).
//   forward   (TR = false): out[o][i][c] = sum_s R[i][s] X_o[s][c]      M = I,   K = NSS
//   transpose (TR = true):  out[o][s][c] = sum_i R[i][s] X_o[i][c]      M = NSS, K = I
// One work item = (row o, slice of CS <= 128 channels); the M x CS result is MT x CS/32 tiles of 32 x 32, dealt round-robin to the 4 wavefronts.
#define RMM_MAXT 6
typedef float es_f32x16 __attribute__((ext_vector_type(16)));
template <bool TR, int TW>   // TW = tiles per wavefront (a wavefront whose last tile does not exist recomputes the previous one and does not store it)
__global__ __launch_bounds__(256) void k_rowmm(RowOp p, int CS, long n_items) {
  extern __shared__ __attribute__((aligned(16))) float es_lds[];
  const int M = TR ? p.NSS : p.I, K = TR ? p.I : p.NSS;
  const int MT = (M + 31) >> 5, Kp = (K + 1) & ~1, KS = Kp + 1, NTc = CS >> 5, nsl = p.C / CS, CS4 = CS >> 2;
  float* sA = es_lds;                       // [MT*32][KS]: A(m, k), zero outside M x K; KS odd: the 32 rows of a fragment hit 32 banks
  float* sB = es_lds + ((MT * 32 * KS + 3) & ~3);   // [max(Kp, MT*32)][CS]: the row block X_o, then the result tile on its way out
  for (int t = threadIdx.x; t < MT * 32 * KS; t += 256) {
    const int m = t / KS, k = t - m * KS;
    sA[t] = (m < M && k < K) ? (TR ? p.R[(long)k * p.NSS + m] : p.R[(long)m * p.NSS + k]) : 0.f;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int ntiles = MT * NTc;
  int aoff[TW], boff[TW];
#pragma unroll
  for (int t = 0; t < TW; ++t) {
    const int tile = min(wave + 4 * t, ntiles - 1), mt = tile / NTc, nt = tile - mt * NTc;
    aoff[t] = (mt * 32 + r) * KS + h;
    boff[t] = h * CS + nt * 32 + r;
  }
  const RowSeg& in_seg = TR ? p.segI : p.segS;
  const RowSeg& out_seg = TR ? p.segS : p.segI;
  for (long item = blockIdx.x; item < n_items; item += gridDim.x) {
    const long o = item / nsl;
    const int c0 = (int)(item - o * nsl) * CS;
    __syncthreads();                        // sA staged / the previous item's fragments read before sB is overwritten
    const float* xrow = p.X ? p.X + (p.index ? (long)p.index[o] : o) * p.x_stride : nullptr;
    for (int t = threadIdx.x; t < Kp * CS4; t += 256) {
      const int k = t / CS4, q = t - k * CS4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < K) {
        const float* src = in_seg.n ? rowseg_addr(in_seg, o, k, p.C) : xrow + (long)k * p.C;
        v = *reinterpret_cast<const float4*>(src + c0 + 4 * q);
      }
      *reinterpret_cast<float4*>(sB + k * CS + 4 * q) = v;
    }
    __syncthreads();
    es_f32x16 acc[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
    for (int j = 0; j < Kp; j += 2) {
#pragma unroll
      for (int t = 0; t < TW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[aoff[t] + j], sB[boff[t] + j * CS], acc[t], 0, 0, 0);
    }
    // result: accumulators -> LDS (the B buffer is free once every wavefront has left the k loop) -> coalesced 16-byte stores, one row resolution per float4
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TW; ++t) {
      const int tile = wave + 4 * t;
      if (tile < ntiles) {                     // wave-uniform
        const int mt = tile / NTc, nt = tile - mt * NTc;
        float* so = sB + (mt * 32 + 4 * h) * CS + nt * 32 + r;
#pragma unroll
        for (int q = 0; q < 16; ++q) so[((q & 3) + 8 * (q >> 2)) * CS] = acc[t][q];
      }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < M * CS4; t += 256) {
      const int m = t / CS4, q = t - m * CS4;
      float* dst = out_seg.n ? rowseg_addr(out_seg, o, m, p.C) : p.out + o * p.out_stride + (long)m * p.C;
      *reinterpret_cast<float4*>(dst + c0 + 4 * q) = *reinterpret_cast<const float4*>(sB + m * CS + 4 * q);
    }
  }
}
// eligibility and launch of the matrix-core form (shared matrix, no accumulation into the output, whole 32-channel groups, LDS and register budget)
static bool rowmm_plan(const RowOp& p, long r_stride, int transpose, int* CS, size_t* lds) {
  if (r_stride != 0 || p.accumulate || (p.C & 31)) return false;
  const int M = transpose ? p.NSS : p.I, K = transpose ? p.I : p.NSS;
  const int MT = (M + 31) / 32, Kp = (K + 1) & ~1, KS = Kp + 1;
  int cs = p.C % 128 == 0 ? 128 : (p.C % 64 == 0 ? 64 : 32);
  while (cs > 32 && (MT * (cs / 32) + 3) / 4 > RMM_MAXT) cs >>= 1;
  if ((MT * (cs / 32) + 3) / 4 > RMM_MAXT) return false;
  const size_t need = sizeof(float) * ((size_t)((MT * 32 * KS + 3) & ~3) + (size_t)(Kp > MT * 32 ? Kp : MT * 32) * cs);
  if (need > 150 * 1024) return false;
  *CS = cs; *lds = need;
  return true;
}
static int rowmm_launch(hipStream_t st, const RowOp& p, long n, int transpose, int CS, size_t lds) {
  const long items = n * (p.C / CS);
  const int per_cu = (int)(lds > 75 * 1024 ? 1 : (160 * 1024 / lds > 4 ? 4 : 160 * 1024 / lds));
  const long grid = items < 256L * per_cu ? items : 256L * per_cu;
  const int M = transpose ? p.NSS : p.I;
  const int tw = (((M + 31) / 32) * (CS / 32) + 3) / 4;
#define RMM_GO(TRV, TWV)                                                                                                                          \
  do {                                                                                                                                            \
    if (lds > 64 * 1024) NQ_HIP(hipFuncSetAttribute((const void*)k_rowmm<TRV, TWV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));       \
    hipLaunchKernelGGL((k_rowmm<TRV, TWV>), dim3((unsigned)grid), dim3(256), lds, st, p, CS, items);                                              \
  } while (0)
#define RMM_TW(TRV)                                                                                                                               \
  switch (tw) { case 1: RMM_GO(TRV, 1); break; case 2: RMM_GO(TRV, 2); break; case 3: RMM_GO(TRV, 3); break; case 4: RMM_GO(TRV, 4); break;       \
                case 5: RMM_GO(TRV, 5); break; default: RMM_GO(TRV, 6); break; }
  if (transpose) { RMM_TW(true) } else { RMM_TW(false) }
#undef RMM_TW
#undef RMM_GO
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

// ---- fused S2 activation on the matrix cores: to_grid -> SiLU -> from_grid in one kernel (so3.py:301-318, activation.py:155-176) --------------------------------
// y[o][s'][c] = sum_i F[i][s'] silu( sum_s T[i][s] x[o][s][c] ): per edge the [G x C] grid tensor (5x the coefficient tensor: 0.97 GB per eSCN layer at 16
// conformers, written, read, written and read again by the three separate kernels) never leaves the registers.  One work item = (edge o, 128 channels); wavefront
// w owns channels 32 w .. 32 w + 31 and ALL MT grid-row tiles of them:
//   product 1  Gr = T x        MT accumulators of 32 x 32 (A = T from LDS, B = x_o from LDS)
//   SiLU on the accumulators
//   product 2  y = F^T silu(Gr): the accumulators ARE the B operands -- MFMA(q) of tile mt contracts the two grid rows its lanes hold in register q
//              (rows 32 mt + rho(q, h), rho = (q&3) + 8 (q>>2) + 4 h), the A operand F^T[s'][that row] comes from LDS.  No data movement between the products.
// Backward (BWD): dGr = (F dy) * silu'(T x) with both products in the same register layout, dx = T^T dGr by the same accumulator-as-operand product.
// T, F: [G][S] row-major (to_grid / from_grid matrices, columns in the order of the blocks); x, y (and dy, dx): per-block tensors (RowSeg over the S side).
template <int MT, bool BWD>
__global__ __launch_bounds__(256) void k_s2act(const float* __restrict__ T, const float* __restrict__ F, RowSeg xin, RowSeg gyin, RowSeg out, int G, int S, int C,
                                              int CS, long n_items) {
  extern __shared__ __attribute__((aligned(16))) float es_lds[];
  // CS = 128: one item (edge, 128 channels) per pass, wavefront w owns channels 32 w..; CS = 64: TWO items per pass, wavefronts {0,1} / {2,3} own one each
  const int CS4 = CS >> 2, wpi = CS >> 5, nsub = 4 / wpi;            // float4 per row, wavefronts per item, items per pass
  const int Sp = (S + 1) & ~1, KS1 = Sp + 1, KS2 = MT * 32 + 1, nsl = C / CS;
  float* sT = es_lds;                                   // [MT*32][KS1]  A(m = i, k = s) = T[i][s]
  float* sF = sT + MT * 32 * KS1;                       // BWD only: same image of F
  float* sM2 = sF + (BWD ? MT * 32 * KS1 : 0);          // [32][KS2]     A(m = s, k = i) = (BWD ? T : F)[i][s]
  float* sX = sM2 + ((32 * KS2 + 3) & ~3);              // [nsub][32][CS] x rows (zero beyond S); the result tiles on their way out
  float* sY = sX + 32 * 128;                            // BWD only: dy rows
  for (int t = threadIdx.x; t < MT * 32 * KS1; t += 256) {
    const int i = t / KS1, k = t - i * KS1;
    const bool in = i < G && k < S;
    sT[t] = in ? T[(long)i * S + k] : 0.f;
    if (BWD) sF[t] = in ? F[(long)i * S + k] : 0.f;
  }
  for (int t = threadIdx.x; t < 32 * KS2; t += 256) {
    const int sr = t / KS2, i = t - sr * KS2;
    sM2[t] = (sr < S && i < G) ? (BWD ? T : F)[(long)i * S + sr] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int sub = wave / wpi, wcol = (wave - sub * wpi) * 32;
  // the A operands of the SECOND product are the same for every item: 16 MT registers per lane for the lifetime of the workgroup (no LDS read per MFMA)
  // (not in the 3-tile backward: its two accumulator sets leave no room -- 282 registers would halve the resident wavefronts)
  constexpr bool HOIST = false;   // see the note at the pass loop
  float a2[HOIST ? MT : 1][16];
  if (HOIST) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int q = 0; q < 16; ++q) a2[m][q] = sM2[r * KS2 + m * 32 + (q & 3) + 8 * (q >> 2) + 4 * h];
  }
  const long n_pass = (n_items + nsub - 1) / nsub;
  // (tried and measured slower: A operands of the second product resident in registers, and the next pass's rows prefetched into registers under the
  // MFMAs -- both cost resident workgroups (186-282 registers), and this kernel lives on the overlap BETWEEN workgroups: 0.63-0.74 / 1.0-1.3 ms per launch vs
  // < 0.6 / 0.93 ms forward / backward for 27 k edges x 128 channels.  It runs at ~43 % of its MFMA floor: 93 / 138 MFMAs per item of which 27 % are padding.)
  for (long pass = blockIdx.x; pass < n_pass; pass += gridDim.x) {
    __syncthreads();
    for (int t = threadIdx.x; t < 32 * 32; t += 256) {               // 32 rows x 128 floats of staging: [sub][k][CS]
      const int k = t >> 5, c4 = t & 31, sb = c4 / CS4, q = c4 - sb * CS4;
      const long item = pass * nsub + sb;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f), w = v;
      if (k < S && item < n_items) {
        const long o = item / nsl;
        const int c0 = (int)(item - o * nsl) * CS;
        v = *reinterpret_cast<const float4*>(rowseg_addr(xin, o, k, C) + c0 + 4 * q);
        if (BWD) w = *reinterpret_cast<const float4*>(rowseg_addr(gyin, o, k, C) + c0 + 4 * q);
      }
      *reinterpret_cast<float4*>(sX + (sb * 32 + k) * CS + 4 * q) = v;
      if (BWD) *reinterpret_cast<float4*>(sY + (sb * 32 + k) * CS + 4 * q) = w;
    }
    __syncthreads();
    es_f32x16 gr[MT], da[BWD ? MT : 1];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int q = 0; q < 16; ++q) { gr[m][q] = 0.f; if (BWD) da[m][q] = 0.f; }
    const int bo = (sub * 32 + h) * CS + wcol + r;
#pragma unroll 3
    for (int j = 0; j < Sp; j += 2) {
      const float bx = sX[bo + j * CS];
      float by = 0.f;
      if (BWD) by = sY[bo + j * CS];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        gr[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(sT[(m * 32 + r) * KS1 + h + j], bx, gr[m], 0, 0, 0);
        if (BWD) da[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(sF[(m * 32 + r) * KS1 + h + j], by, da[m], 0, 0, 0);
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int q = 0; q < 16; ++q) gr[m][q] = BWD ? da[m][q] * nq_dsilu(gr[m][q]) : nq_silu(gr[m][q]);
    es_f32x16 res;
#pragma unroll
    for (int q = 0; q < 16; ++q) res[q] = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int q = 0; q < 16; ++q)
        res = __builtin_amdgcn_mfma_f32_32x32x2f32(HOIST ? a2[HOIST ? m : 0][q] : sM2[r * KS2 + m * 32 + (q & 3) + 8 * (q >> 2) + 4 * h], gr[m][q], res, 0, 0, 0);
    __syncthreads();                                    // every wavefront has left the x rows: the buffer takes the result tiles
#pragma unroll
    for (int q = 0; q < 16; ++q) sX[(sub * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * CS + wcol + r] = res[q];
    __syncthreads();
    for (int t = threadIdx.x; t < 32 * 32; t += 256) {
      const int k = t >> 5, c4 = t & 31, sb = c4 / CS4, q = c4 - sb * CS4;
      const long item = pass * nsub + sb;
      if (k < S && item < n_items) {
        const long o = item / nsl;
        const int c0 = (int)(item - o * nsl) * CS;
        *reinterpret_cast<float4*>(rowseg_addr(out, o, k, C) + c0 + 4 * q) = *reinterpret_cast<const float4*>(sX + (sb * 32 + k) * CS + 4 * q);
      }
    }
  }
}

// ---- rotations with the block structure of the Wigner matrices ------------------------------------------------------------------------------------------------
// Row i of an edge's Wigner block (degree l_i) touches only the 2 l_i + 1 coefficients of that degree: 235 of the 29 x 49 entries at lmax 6 / mmax 2.  One
// workgroup per edge (or per node when the transposed form also sums over a node's edges), thread = channel: the coefficients of one degree sit in registers,
// the matrix entries are wavefront-uniform loads; no LDS.  The edge-frame side lives in per-m-block tensors [n][rows_k][c_stride] at channel offset c_off
// (two rotations can fill the two halves of one concatenated block).
#define ROT_MAXROWS 13
struct RotArgs {
  const float* W; long w_stride;                 // per-edge rows [.][n_full]
  const float* X; long x_stride; const int* index;   // forward input: [.][n_full][C], row of edge o = index ? index[o] : o
  float* out;                                    // transposed output [n_out][n_full][C]
  const int* ptr; const int* order;              // transposed: out row n sums the edges order[q] (or q), q in [ptr[n], ptr[n+1]); ptr null: n is the edge
  const float* coef_scale;                       // transposed: optional factor per output coefficient
  int n_full, lmax, C, c_stride, c_off;
  int nrows[ES_MAXL + 1];
  unsigned char rows[ES_MAXL + 1][ROT_MAXROWS];  // reduced rows of every degree
  unsigned char seg_of[64], row_in_seg[64];
  int seg_rows[ROWOP_MAXSEG];
  float* seg_ptr[ROWOP_MAXSEG];
};
__device__ __forceinline__ float* rot_addr(const RotArgs& p, long o, int i) {
  const int sg = p.seg_of[i];
  return p.seg_ptr[sg] + (o * p.seg_rows[sg] + p.row_in_seg[i]) * (long)p.c_stride + p.c_off;
}
__global__ __launch_bounds__(256) void k_es_rot_fwd(RotArgs p) {
  const long o = blockIdx.x;
  const float* __restrict__ R = p.W + o * p.w_stride;
  const float* __restrict__ X = p.X + (p.index ? (long)p.index[o] : o) * p.x_stride;
  for (int ch = threadIdx.x; ch < p.C; ch += blockDim.x) {
#pragma unroll
    for (int l = 0; l <= ES_MAXL; ++l) {
      if (l <= p.lmax) {
        float x[2 * ES_MAXL + 1];
#pragma unroll
        for (int k = 0; k < 2 * l + 1; ++k) x[k] = X[(long)(l * l + k) * p.C + ch];
        for (int r = 0; r < p.nrows[l]; ++r) {
          const int i = p.rows[l][r];
          const float* __restrict__ Rr = R + i * p.n_full + l * l;
          float acc = 0.f;
#pragma unroll
          for (int k = 0; k < 2 * l + 1; ++k) acc = fmaf(Rr[k], x[k], acc);
          rot_addr(p, o, i)[ch] = acc;
        }
      }
    }
  }
}
// grid (output rows, degrees): the workgroup (n, l) owns the 2l + 1 output coefficients of degree l (7x the parallelism of one workgroup per output row; the
// degrees touch disjoint matrix entries and block rows, so nothing is read twice)
template <int L>
__device__ __forceinline__ void rot_tr_degree(const RotArgs& p, long n, long q0, long q1) {
  for (int ch = threadIdx.x; ch < p.C; ch += blockDim.x) {
    float acc[2 * L + 1];
#pragma unroll
    for (int k = 0; k < 2 * L + 1; ++k) acc[k] = 0.f;
    for (long q = q0; q < q1; ++q) {
      const long o = p.order ? (long)p.order[q] : q;
      const float* __restrict__ R = p.W + o * p.w_stride + L * L;
      for (int r = 0; r < p.nrows[L]; ++r) {
        const int i = p.rows[L][r];
        const float y = rot_addr(p, o, i)[ch];
        const float* __restrict__ Rr = R + i * p.n_full;
#pragma unroll
        for (int k = 0; k < 2 * L + 1; ++k) acc[k] = fmaf(Rr[k], y, acc[k]);
      }
    }
    float* out = p.out + (n * (long)p.n_full + L * L) * p.C + ch;
#pragma unroll
    for (int k = 0; k < 2 * L + 1; ++k) out[(long)k * p.C] = p.coef_scale ? acc[k] * p.coef_scale[L * L + k] : acc[k];
  }
}
__global__ __launch_bounds__(256) void k_es_rot_tr(RotArgs p) {
  const long n = blockIdx.x;
  long q0 = n, q1 = n + 1;
  if (p.ptr) { q0 = p.ptr[n]; q1 = p.ptr[n + 1]; }
  switch (blockIdx.y) {
    case 0: rot_tr_degree<0>(p, n, q0, q1); break;
    case 1: rot_tr_degree<1>(p, n, q0, q1); break;
    case 2: rot_tr_degree<2>(p, n, q0, q1); break;
    case 3: rot_tr_degree<3>(p, n, q0, q1); break;
    case 4: rot_tr_degree<4>(p, n, q0, q1); break;
    case 5: rot_tr_degree<5>(p, n, q0, q1); break;
    default: rot_tr_degree<6>(p, n, q0, q1); break;
  }
}
static int rot_setup(RotArgs& p, const float* W, int64_t w_stride, int32_t nseg, const int32_t* seg_rows, float* const* seg_ptrs, int32_t c_stride, int32_t c_off,
                     const int32_t* red_l, int32_t n_red, int32_t lmax, int32_t C) {
  if (!W || !seg_rows || !seg_ptrs || !red_l || nseg < 1 || nseg > ROWOP_MAXSEG || n_red < 1 || n_red > 64 || lmax < 0 || lmax > ES_MAXL || C < 1 ||
      c_off < 0 || c_off + C > c_stride)
    return nq_fail(NQ_ERR_ARG, "rotate: bad argument");
  p = RotArgs{};
  p.W = W; p.w_stride = (long)w_stride; p.n_full = (lmax + 1) * (lmax + 1); p.lmax = lmax; p.C = C; p.c_stride = c_stride; p.c_off = c_off;
  int tot = 0;
  for (int k = 0; k < nseg; ++k) {
    if (!seg_ptrs[k] || seg_rows[k] < 1) return nq_fail(NQ_ERR_ARG, "rotate: bad block %d", k);
    p.seg_rows[k] = seg_rows[k]; p.seg_ptr[k] = seg_ptrs[k];
    for (int r = 0; r < seg_rows[k] && tot < 64; ++r, ++tot) { p.seg_of[tot] = (unsigned char)k; p.row_in_seg[tot] = (unsigned char)r; }
  }
  if (tot != n_red) return nq_fail(NQ_ERR_ARG, "rotate: block rows sum to %d, expected %d", tot, n_red);
  for (int i = 0; i < n_red; ++i) {
    const int l = red_l[i];
    if (l < 0 || l > lmax || p.nrows[l] >= ROT_MAXROWS) return nq_fail(NQ_ERR_ARG, "rotate: bad degree list");
    p.rows[l][p.nrows[l]++] = (unsigned char)i;
  }
  return NQ_OK;
}

// =========================================================================================================================================================
#define ES_GRID(total) dim3((unsigned)(((total) + 255) / 256)), dim3(256), 0, st

extern "C" {

/* Pass 1 (src == NULL): degrees -> ptr (exclusive scan), returns E in *E_host (synchronises).  Pass 2: fills src / dst / geom [E][4] = {pos[j] - pos[i], |.|}. */
int nq_es_graph_count(const float* pos, const int32_t* mol_ptr, const int32_t* atom_mol, int32_t N, double cutoff, int32_t K, int32_t* deg, int32_t* ptr,
                      int32_t* E_host, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "es_graph_count");
  if (!pos || !mol_ptr || !atom_mol || !deg || !ptr || !E_host || N <= 0) return nq_fail(NQ_ERR_ARG, "bad argument");
  const float r2 = (float)(cutoff * cutoff);
  hipLaunchKernelGGL(k_es_graph, dim3(N), dim3(64), 0, st, pos, mol_ptr, atom_mol, r2, K, nullptr, deg, nullptr, nullptr, nullptr);
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_es_scan, dim3(1), dim3(1024), 0, st, deg, N, ptr);
  NQ_LAUNCH_CHECK();
  NQ_HIP(hipMemcpyAsync(E_host, ptr + N, sizeof(int), hipMemcpyDeviceToHost, st));
  NQ_HIP(hipStreamSynchronize(st));
  return NQ_OK;
}
int nq_es_graph_fill(const float* pos, const int32_t* mol_ptr, const int32_t* atom_mol, int32_t N, double cutoff, int32_t K, const int32_t* ptr, int32_t* src,
                     int32_t* dst, float* geom, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "es_graph_fill");
  if (!pos || !ptr || !src || !dst || !geom) return nq_fail(NQ_ERR_ARG, "null argument");
  hipLaunchKernelGGL(k_es_graph, dim3(N), dim3(64), 0, st, pos, mol_ptr, atom_mol, (float)(cutoff * cutoff), K, ptr, nullptr, src, dst, (float4*)geom);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_es_frames(const float* geom, int32_t E, float* rot, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "es_frames");
  if (E <= 0) return NQ_OK;
  hipLaunchKernelGGL(k_es_frames, ES_GRID((long)E), (const float4*)geom, E, rot);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
/* rot [E][3][3] -> W [E][n_red][n_full]; J: the (2l+1)^2 matrices of l = 0..lmax back to back, J_offset[l] their starts; red_l / red_row [n_red]: degree and
 * row inside the degree's block of every kept coefficient (in the order the caller wants them, e.g. m-primary); scratch: f32[3 E]. */
int nq_es_wigner(const float* rot, int32_t E, const float* J, const int32_t* J_offset, const int32_t* red_l, const int32_t* red_row, int32_t n_red, int32_t n_full,
                 int32_t lmax, float* scratch, float* W, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "es_wigner");
  if (lmax > ES_MAXL) return nq_fail(NQ_ERR_ARG, "lmax %d > %d", lmax, ES_MAXL);
  if (E <= 0) return NQ_OK;
  if (reinterpret_cast<uintptr_t>(scratch) & 7) return nq_fail(NQ_ERR_ARG, "es_wigner: scratch must be 8-byte aligned (3 E doubles)");
  double* ang = reinterpret_cast<double*>(scratch);
  hipLaunchKernelGGL(k_es_angles, ES_GRID((long)E), rot, E, ang);
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_es_wigner, ES_GRID((long)E * n_red), ang, E, J, J_offset, red_l, red_row, n_red, n_full, W);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_es_smearing(const float* geom, int64_t E, int32_t K, const float* offset, float coeff, float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "es_smear");
  if (E <= 0) return NQ_OK;
  hipLaunchKernelGGL(k_es_smear, ES_GRID(E * K), (const float4*)geom, (long)E, K, offset, coeff, out);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
/* transpose == 0: out[o][i][c] (+)= sum_s R_o[i][s] X_row(o)[s][c];  transpose != 0: out[o][s][c] (+)= sum_i R_o[i][s] X_row(o)[i][c].  Strides in floats. */
int nq_rowop(const float* R, int64_t r_stride, const float* X, int64_t x_stride, const int32_t* index, float* out, int64_t out_stride, int64_t n, int32_t I,
             int32_t NSS, int32_t C, int32_t transpose, int32_t accumulate, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, transpose ? "rowop_tr" : "rowop_fwd");
  if (n <= 0) return NQ_OK;
  if (!R || !X || !out) return nq_fail(NQ_ERR_ARG, "null argument");
  RowOp p{R, (long)r_stride, X, (long)x_stride, index, out, (long)out_stride, I, NSS, C, accumulate, {}, {}};
  p.segI.n = p.segS.n = 0;
  {
    int CS; size_t l2;
    if (rowmm_plan(p, (long)r_stride, transpose, &CS, &l2)) return rowmm_launch(st, p, (long)n, transpose, CS, l2);
  }
  const size_t lds = sizeof(float) * ((size_t)I * NSS + (size_t)(transpose ? I : NSS) * C);
  if (lds > 64 * 1024) return nq_fail(NQ_ERR_ARG, "rowop: I=%d NSS=%d C=%d needs %zu bytes of LDS (> 64 kB): split the matrix rows", I, NSS, C, lds);
  if (transpose) hipLaunchKernelGGL(k_rowop_tr, dim3((unsigned)n), dim3(256), lds, st, p);
  else hipLaunchKernelGGL(k_rowop_fwd, dim3((unsigned)n), dim3(256), lds, st, p);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
/* nq_rowop with one side split into per-block tensors (the m-blocks of an SO(2) layer): seg_side 0 = the I side (rows of R), 1 = the S side (columns of R);
 * seg_rows[nseg] rows per block, seg_ptrs[nseg] device pointers of the contiguous [n][rows_k][C] tensors (HOST arrays).  The other side is x_or_out:
 * transpose == 0: S side is the input, I side the output; transpose != 0: I side is the input, S side the output. */
int nq_rowop_blocks(const float* R, int64_t r_stride, float* x_or_out, int64_t stride, const int32_t* index, int32_t seg_side, int32_t nseg,
                    const int32_t* seg_rows, float* const* seg_ptrs, int64_t n, int32_t I, int32_t NSS, int32_t C, int32_t transpose, int32_t accumulate,
                    void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, transpose ? "rowop_tr" : "rowop_fwd");
  if (n <= 0) return NQ_OK;
  if (!R || !x_or_out || !seg_rows || !seg_ptrs || nseg < 1 || nseg > ROWOP_MAXSEG) return nq_fail(NQ_ERR_ARG, "bad argument");
  const size_t lds = sizeof(float) * ((size_t)I * NSS + (size_t)(transpose ? I : NSS) * C);
  RowOp p{R, (long)r_stride, nullptr, 0, index, nullptr, 0, I, NSS, C, accumulate, {}, {}};
  p.segI.n = p.segS.n = 0;
  RowSeg& sg = seg_side == 0 ? p.segI : p.segS;
  sg.n = nseg;
  int tot = 0;
  for (int k = 0; k < nseg; ++k) { sg.start[k] = tot; tot += seg_rows[k]; sg.ptr[k] = seg_ptrs[k]; }
  sg.start[nseg] = tot;
  if (tot != (seg_side == 0 ? I : NSS)) return nq_fail(NQ_ERR_ARG, "rowop: block rows sum to %d, expected %d", tot, seg_side == 0 ? I : NSS);
  const bool seg_is_input = (transpose != 0) == (seg_side == 0);          // fwd: input = S side; tr: input = I side
  if (seg_is_input) { p.out = x_or_out; p.out_stride = stride; }
  else { p.X = x_or_out; p.x_stride = stride; }
  {
    int CS; size_t l2;
    if (rowmm_plan(p, (long)r_stride, transpose, &CS, &l2)) return rowmm_launch(st, p, (long)n, transpose, CS, l2);
  }
  if (lds > 64 * 1024) return nq_fail(NQ_ERR_ARG, "rowop: I=%d NSS=%d C=%d needs %zu bytes of LDS (> 64 kB)", I, NSS, C, lds);
  if (transpose) hipLaunchKernelGGL(k_rowop_tr, dim3((unsigned)n), dim3(256), lds, st, p);
  else hipLaunchKernelGGL(k_rowop_fwd, dim3((unsigned)n), dim3(256), lds, st, p);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

/* Fused S2 activation (to_grid -> SiLU -> from_grid; backward != 0: its adjoint with the grid recomputed) on per-block tensors: nseg blocks of seg_rows[k] rows
 * ([n][rows_k][C] each, HOST arrays of device pointers); T, F: [G][S] with S = sum of seg_rows.  Requires C % 64 == 0, S <= 32, G <= 96 (else NQ_ERR_ARG: the
 * caller falls back to nq_rowop_blocks + the activation kernel). */
int nq_s2_activation_blocks(const float* T, const float* F, int32_t G, int32_t S, int32_t C, int64_t n, int32_t nseg, const int32_t* seg_rows, float* const* x_ptrs,
                            float* const* gy_ptrs, float* const* out_ptrs, int32_t backward, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, backward ? "s2act_bwd" : "s2act_fwd");
  if (n <= 0) return NQ_OK;
  if (!T || !F || !seg_rows || !x_ptrs || !out_ptrs || (backward && !gy_ptrs) || nseg < 1 || nseg > ROWOP_MAXSEG) return nq_fail(NQ_ERR_ARG, "s2 activation: bad argument");
  if ((C & 63) || S > 32 || G > 96 || S < 1 || G < 1) return nq_fail(NQ_ERR_ARG, "s2 activation: needs C %% 64 == 0, S <= 32, G <= 96 (C=%d S=%d G=%d)", C, S, G);
  const int CS = (C & 127) ? 64 : 128;
  RowSeg sx{}, sg{}, so{};
  int tot = 0;
  for (int k = 0; k < nseg; ++k) {
    sx.start[k] = sg.start[k] = so.start[k] = tot; tot += seg_rows[k];
    sx.ptr[k] = x_ptrs[k]; sg.ptr[k] = backward ? gy_ptrs[k] : nullptr; so.ptr[k] = out_ptrs[k];
  }
  sx.start[nseg] = sg.start[nseg] = so.start[nseg] = tot;
  sx.n = sg.n = so.n = nseg;
  if (tot != S) return nq_fail(NQ_ERR_ARG, "s2 activation: block rows sum to %d, expected %d", tot, S);
  const int MT = (G + 31) / 32, Sp = (S + 1) & ~1, KS1 = Sp + 1, KS2 = MT * 32 + 1;
  const size_t lds = sizeof(float) * ((size_t)MT * 32 * KS1 * (backward ? 2 : 1) + (size_t)((32 * KS2 + 3) & ~3) + (size_t)32 * 128 * (backward ? 2 : 1));
  const long items = n * (C / CS), passes = (items + (128 / CS) - 1) / (128 / CS);
  const int per_cu = (int)(160 * 1024 / lds > 4 ? 4 : (160 * 1024 / lds < 1 ? 1 : 160 * 1024 / lds));
  const long grid = passes < 256L * per_cu ? passes : 256L * per_cu;
#define S2_GO(MTV, BW)                                                                                                                             \
  do {                                                                                                                                             \
    if (lds > 64 * 1024) NQ_HIP(hipFuncSetAttribute((const void*)k_s2act<MTV, BW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));         \
    hipLaunchKernelGGL((k_s2act<MTV, BW>), dim3((unsigned)grid), dim3(256), lds, st, T, F, sx, sg, so, G, S, C, CS, items);                            \
  } while (0)
  if (backward) { if (MT == 1) S2_GO(1, true); else if (MT == 2) S2_GO(2, true); else S2_GO(3, true); }
  else { if (MT == 1) S2_GO(1, false); else if (MT == 2) S2_GO(2, false); else S2_GO(3, false); }
#undef S2_GO
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

/* Rotation into the edge frame with the degree-block structure of W (SO3_Embedding._rotate, so3.py:265-283): for every edge o and kept row i (degree
 * red_l[i], HOST array, rows in the order of W): block(i)[o][row][c_off + c] = sum_k W_o[i][l^2 + k] x[index ? index[o] : o][l^2 + k][c], k < 2l + 1.
 * Blocks: nseg contiguous tensors [E][seg_rows_k][c_stride] (HOST arrays).  W_o = W + o * w_stride, rows of n_full = (lmax+1)^2 floats. */
int nq_es_rotate(const float* W, int64_t w_stride, const float* x, int64_t x_stride, const int32_t* index, int32_t nseg, const int32_t* seg_rows,
                 float* const* seg_ptrs, int32_t c_stride, int32_t c_off, int64_t E, const int32_t* red_l, int32_t n_red, int32_t lmax, int32_t C, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "es_rotate");
  if (E <= 0) return NQ_OK;
  if (!x) return nq_fail(NQ_ERR_ARG, "rotate: null input");
  RotArgs p;
  NQ_TRY(rot_setup(p, W, w_stride, nseg, seg_rows, seg_ptrs, c_stride, c_off, red_l, n_red, lmax, C));
  p.X = x; p.x_stride = (long)x_stride; p.index = index;
  hipLaunchKernelGGL(k_es_rot_fwd, dim3((unsigned)E), dim3(C >= 256 ? 256 : (C + 63) / 64 * 64), 0, st, p);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
/* The transposed rotation (rotate_inv, so3.py:285-300) fused with the sum over edges: out[n][s][c] = coef_scale[s] * sum_{q in [ptr[n], ptr[n+1])}
 * sum_i W_o[i][s] block(i)[o][row][c_off + c], o = order ? order[q] : q (ptr == NULL: n_out edges, no sum).  coef_scale nullable. */
int nq_es_rotate_back(const float* W, int64_t w_stride, int32_t nseg, const int32_t* seg_rows, float* const* seg_ptrs, int32_t c_stride, int32_t c_off,
                      const int32_t* ptr, const int32_t* order, const float* coef_scale, float* out, int64_t n_out, const int32_t* red_l, int32_t n_red,
                      int32_t lmax, int32_t C, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "es_rotate_back");
  if (n_out <= 0) return NQ_OK;
  if (!out) return nq_fail(NQ_ERR_ARG, "rotate_back: null output");
  RotArgs p;
  NQ_TRY(rot_setup(p, W, w_stride, nseg, seg_rows, seg_ptrs, c_stride, c_off, red_l, n_red, lmax, C));
  p.out = out; p.ptr = ptr; p.order = order; p.coef_scale = coef_scale;
  hipLaunchKernelGGL(k_es_rot_tr, dim3((unsigned)n_out, lmax + 1), dim3(C >= 256 ? 256 : (C + 63) / 64 * 64), 0, st, p);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

}  // extern "C"
