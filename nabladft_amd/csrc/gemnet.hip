// GemNet-OC interaction kernels (fp32).  Every graph is a CSR by target atom (gemnet_graph.hip), so "triplets" and "quadruplets" are loops over the other
// in-edges of an atom: no index lists, no zero-padded [E, Kmax, C] tiles, the angular basis is evaluated in registers from the two unit vectors.
//
// Replaces (reference, /root/reference/nablaDFT/gemnet_oc/):
//   layers/radial_basis.py:21-37,60-77,196-220   PolynomialEnvelope x GaussianBasis on d / cutoff                                  -> k_gn_rbf
//   layers/basis.py:215-295 + spherical_basis.py  Y_l0(cos) (zero_m_only real spherical harmonics), "legendre_outer" Y_l0 x Y_l'0  -> gn_zonal (in registers)
//   layers/efficient.py:152-253                   EfficientInteractionBilinear: sum_k sph[e,s,k] m[e,k,c] on padded tiles           -> k_gn_tri / k_gn_quad
//                                                 then rad_W1[e,i,s] @ that                                                          -> k_gn_rowmm
//   layers/efficient.py:60-149                    BasisEmbedding without inner index (the cbf weights of the quadruplet path)        -> k_gn_cir
//   gemnet_oc.py:597-655                          calculate_quad_angles                                                              -> k_gn_quad (in registers)
//   layers/interaction_block.py:689-739           PairInteraction: rad_basis @ padded x                                              -> k_gn_pair
//   layers/embedding_block.py:58-92               EdgeEmbedding's cat[h_s, h_t, m]                                                   -> k_gn_cat
//   layers/atom_update_block.py:73-172            m * rbf embedding, scatter to the target atom                                      -> k_gn_mulsum
//   gemnet_oc.py:1216-1243                        direct forces: F_st averaged with the counter-edge, projected on the edge vector, summed per target
// Sums run over CSR rows in a fixed order: results are bitwise reproducible (the reference's scatter / index_put on a GPU are not).
// The bases depend on positions only and the model predicts forces directly (direct_forces, config/model/gemnet-oc.yaml:28), so no gradient flows to them.
#include "common.h"

#define GN_MAXNS 8
#define GN_MAXR 16

struct GnSet { int n; const int* ptr; const int* src; const int* dst; const float4* geom; };   // one edge set, CSR by target atom

// Y_l0(z) = sqrt((2l+1)/(4 pi)) P_l(z), l < NS (basis.py:243: sph_harm_prefactor(l, 0) * P_l^0), times the fitted ScaleFactor of the basis layer
__device__ __forceinline__ void gn_zonal(float z, int NS, float scale, float* Y) {
  float pm = 1.f, p = z;
  const float inv4pi = 0.07957747154594767f;
#pragma unroll
  for (int l = 0; l < GN_MAXNS; ++l) {
    if (l < NS) {
      const float pl = l == 0 ? 1.f : (l == 1 ? z : ((2 * l - 1) * z * p - (l - 1) * pm) / l);
      if (l >= 2) { pm = p; p = pl; }
      Y[l] = sqrtf((2 * l + 1) * inv4pi) * pl * scale;
    } else {
      Y[l] = 0.f;
    }
  }
}
__device__ __forceinline__ float gn_dot(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float gn_clamp1(float x) { return fminf(1.f, fmaxf(-1.f, x)); }
__device__ __forceinline__ float4 gn_cross(float4 a, float4 b) { return make_float4(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x, 0.f); }

// ---- radial basis ---------------------------------------------------------------------------------------------------------------------------------
__global__ void k_gn_rbf(const float4* __restrict__ geom, long n, int R, const float* __restrict__ offset, float inv_cutoff, float coeff, float pexp,
                         float scale, float* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * R) return;
  const long e = t / R; const int k = (int)(t - e * R);
  const float ds = geom[e].w * inv_cutoff;
  const float a = -(pexp + 1.f) * (pexp + 2.f) / 2.f, b = pexp * (pexp + 2.f), c = -pexp * (pexp + 1.f) / 2.f;
  const float dp = powf(ds, pexp);
  const float env = ds < 1.f ? 1.f + a * dp + b * dp * ds + c * dp * ds * ds : 0.f;
  const float x = ds - offset[k];
  out[t] = env * expf(coeff * x * x) * scale;
}

// ---- triplets: S[o][s][c] = sum_{p in row_G(target(o)), source(p) != source(o)} Y_s(v_o . v_p) X[p][c] ----------------------------------------------
__global__ void k_gn_tri_fwd(GnSet O, GnSet G, const float* __restrict__ X, int C, int NS, float scale, float* __restrict__ S) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)O.n * C) return;
  const int o = (int)(t / C), ch = (int)(t - (long)o * C);
  const int a = O.dst[o], c = O.src[o];
  const float4 v = O.geom[o];
  float acc[GN_MAXNS];
#pragma unroll
  for (int s = 0; s < GN_MAXNS; ++s) acc[s] = 0.f;
  for (int p = G.ptr[a]; p < G.ptr[a + 1]; ++p) {
    if (G.src[p] == c) continue;
    float Y[GN_MAXNS];
    gn_zonal(gn_clamp1(gn_dot(v, G.geom[p])), NS, scale, Y);
    const float x = X[(long)p * C + ch];
#pragma unroll
    for (int s = 0; s < GN_MAXNS; ++s) acc[s] += Y[s] * x;
  }
  for (int s = 0; s < NS; ++s) S[((long)o * NS + s) * C + ch] = acc[s];
}
// adjoint: dX[p][c] = sum_{o in row_O(target(p)), source(o) != source(p)} sum_s Y_s(v_o . v_p) dS[o][s][c]
__global__ void k_gn_tri_bwd(GnSet O, GnSet G, const float* __restrict__ dS, int C, int NS, float scale, float* __restrict__ dX) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)G.n * C) return;
  const int p = (int)(t / C), ch = (int)(t - (long)p * C);
  const int a = G.dst[p], b = G.src[p];
  const float4 w = G.geom[p];
  float acc = 0.f;
  for (int o = O.ptr[a]; o < O.ptr[a + 1]; ++o) {
    if (O.src[o] == b) continue;
    float Y[GN_MAXNS];
    gn_zonal(gn_clamp1(gn_dot(O.geom[o], w)), NS, scale, Y);
    const float* g = dS + (long)o * NS * C + ch;
#pragma unroll
    for (int s = 0; s < GN_MAXNS; ++s) if (s < NS) acc += Y[s] * g[(long)s * C];
  }
  dX[t] = acc;
}

// ---- quadruplets c -> a <- b <- d: out edge o = (c -> a) of the main graph, q = (b -> a) of the qint graph, p = (d -> b) of the main graph, d not in {a, c},
// b != c.  X rows are indexed (q, position of p in row b): tin_ptr[q] + j.  basis[l][l'] = Y_l(cos cab) Y_l'(cos of the dihedral half angle). -------------
__device__ __forceinline__ float gn_cos_dihedral(float4 crossA, float4 vdb, float4 vba) {
  const float4 crossD = gn_cross(vdb, vba);
  const float x = gn_dot(crossA, crossD);
  const float4 cc = gn_cross(crossA, crossD);
  const float y = fmaxf(sqrtf(gn_dot(cc, cc)), 1e-9f);
  return x * rsqrtf(x * x + y * y);                 // = cos(atan2(y, x)) (gemnet_oc.py:650 + spherical_basis.py:107: torch.cos of the atan2 angle)
}
__global__ void k_gn_quad_fwd(GnSet M, GnSet Q, const int* __restrict__ tin_ptr, const float* __restrict__ X, int C, int NS, float scale,
                              float* __restrict__ S) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)M.n * C) return;
  const int o = (int)(t / C), ch = (int)(t - (long)o * C);
  const int a = M.dst[o], c = M.src[o];
  const float4 vca = M.geom[o];
  float acc[GN_MAXNS][GN_MAXNS];
#pragma unroll
  for (int l = 0; l < GN_MAXNS; ++l)
#pragma unroll
    for (int k = 0; k < GN_MAXNS; ++k) acc[l][k] = 0.f;
  for (int q = Q.ptr[a]; q < Q.ptr[a + 1]; ++q) {
    const int b = Q.src[q];
    if (b == c) continue;
    const float4 vba = Q.geom[q];
    float Yl[GN_MAXNS], tl[GN_MAXNS];
    gn_zonal(gn_clamp1(gn_dot(vca, vba)), NS, scale, Yl);
#pragma unroll
    for (int k = 0; k < GN_MAXNS; ++k) tl[k] = 0.f;
    const float4 crossA = gn_cross(vca, vba);
    const long base = tin_ptr[q];
    const int pb = M.ptr[b], pe = M.ptr[b + 1];
    for (int p = pb; p < pe; ++p) {
      const int d = M.src[p];
      if (d == a || d == c) continue;
      float Yk[GN_MAXNS];
      gn_zonal(gn_cos_dihedral(crossA, M.geom[p], vba), NS, 1.f, Yk);
      const float x = X[(base + (p - pb)) * C + ch];
#pragma unroll
      for (int k = 0; k < GN_MAXNS; ++k) tl[k] += Yk[k] * x;
    }
#pragma unroll
    for (int l = 0; l < GN_MAXNS; ++l)
#pragma unroll
      for (int k = 0; k < GN_MAXNS; ++k) acc[l][k] += Yl[l] * tl[k];
  }
  for (int l = 0; l < NS; ++l)
    for (int k = 0; k < NS; ++k) S[((long)o * NS * NS + l * NS + k) * C + ch] = acc[l][k];
}
// Per-atom form of the two kernels above/below (the ones launched): one workgroup per centre atom a.  All out edges of a share the qint row of a and the main
// rows of its sources, so the feature rows (q, .) are staged once in LDS, the 7-vector Y_l'(dihedral) of every (o, q, p) is evaluated ONCE by one thread
// (the per-(edge, channel) kernels evaluate it in every channel lane: ~100 instructions per inner iteration) and the inner loop becomes 8 LDS reads + 7 FMAs.
#define GQ_PC 64      // main in-edges per staged chunk
#define GQ_OC 8       // out edges per pass
__global__ __launch_bounds__(256) void k_gn_quad_fwd_atom(GnSet M, GnSet Q, const int* __restrict__ tin_ptr, const float* __restrict__ X, int C, int NS, float scale,
                                                            float* __restrict__ S) {
  __shared__ float sX[GQ_PC * 64];
  __shared__ float sY[GQ_OC * GQ_PC * GN_MAXNS];
  __shared__ float4 sPg[GQ_PC], sOg[GQ_OC];
  __shared__ int sPs[GQ_PC], sOs[GQ_OC];
  const int a = blockIdx.x, tid = threadIdx.x;
  const int ob = M.ptr[a], on = M.ptr[a + 1] - ob, qb = Q.ptr[a], qe = Q.ptr[a + 1];
  const int OC = min(GQ_OC, 256 / C);
  const int ol = tid / C, ch = tid - ol * C;
  const bool active = ol < OC;
  for (int o0 = 0; o0 < on; o0 += OC) {
    const int o = ob + o0 + ol;
    const bool ovalid = active && (o0 + ol) < on;
    const int c = ovalid ? M.src[o] : -1;
    const float4 vca = ovalid ? M.geom[o] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (tid < OC && o0 + tid < on) { sOg[tid] = M.geom[ob + o0 + tid]; sOs[tid] = M.src[ob + o0 + tid]; }
    float acc[GN_MAXNS][GN_MAXNS];
#pragma unroll
    for (int l = 0; l < GN_MAXNS; ++l)
#pragma unroll
      for (int k = 0; k < GN_MAXNS; ++k) acc[l][k] = 0.f;
    for (int q = qb; q < qe; ++q) {
      const int b = Q.src[q];
      const float4 vba = Q.geom[q];
      const int pb = M.ptr[b], pn = M.ptr[b + 1] - pb;
      const long base = tin_ptr[q];
      float tl[GN_MAXNS];
#pragma unroll
      for (int k = 0; k < GN_MAXNS; ++k) tl[k] = 0.f;
      for (int p0 = 0; p0 < pn; p0 += GQ_PC) {
        const int np = min(GQ_PC, pn - p0);
        for (int e = tid; e < np * C; e += 256) sX[e] = X[(base + p0) * C + e];
        if (tid < np) { sPg[tid] = M.geom[pb + p0 + tid]; sPs[tid] = M.src[pb + p0 + tid]; }
        __syncthreads();
        for (int e = tid; e < OC * np; e += 256) {
          const int oi = e / np, pj = e - oi * np;
          float Yk[GN_MAXNS];
#pragma unroll
          for (int k = 0; k < GN_MAXNS; ++k) Yk[k] = 0.f;
          if (o0 + oi < on) {
            const int cc = sOs[oi], d = sPs[pj];
            if (cc != b && d != a && d != cc) gn_zonal(gn_cos_dihedral(gn_cross(sOg[oi], vba), sPg[pj], vba), NS, 1.f, Yk);
          }
#pragma unroll
          for (int k = 0; k < GN_MAXNS; ++k) sY[(oi * GQ_PC + pj) * GN_MAXNS + k] = Yk[k];
        }
        __syncthreads();
        if (ovalid) {
          for (int pj = 0; pj < np; ++pj) {
            const float x = sX[pj * C + ch];
            const float* y = &sY[(ol * GQ_PC + pj) * GN_MAXNS];
#pragma unroll
            for (int k = 0; k < GN_MAXNS; ++k) tl[k] += y[k] * x;
          }
        }
        __syncthreads();
      }
      if (ovalid && b != c) {
        float Yl[GN_MAXNS];
        gn_zonal(gn_clamp1(gn_dot(vca, vba)), NS, scale, Yl);
#pragma unroll
        for (int l = 0; l < GN_MAXNS; ++l)
#pragma unroll
          for (int k = 0; k < GN_MAXNS; ++k) acc[l][k] += Yl[l] * tl[k];
      }
    }
    if (ovalid)
      for (int l = 0; l < NS; ++l)
        for (int k = 0; k < NS; ++k) S[((long)o * NS * NS + l * NS + k) * C + ch] = acc[l][k];
  }
}
// dX[(q, j)][c] for all qint in-edges q of atom a.  Per (q, block of 32 main in-edges p of source(q), chunk of 16 out edges o): edge geometry, the U rows of
// the chunk and the whole 32 x 16 table of Y_l'(dihedral(o, q, p)) are staged in LDS (the table is computed from LDS-resident geometry: no dependent global
// loads in the loop), then every thread accumulates its rows of the block.  Dynamic LDS: (16 * 8 * C + 32 * 16 * 8 + 48 * 5) floats (C = 32: 33 kB).
#define GQ_UC 16
#define GQ_PB 32
__global__ __launch_bounds__(256) void k_gn_quad_bwd_x_atom(GnSet M, GnSet Q, const int* __restrict__ tin_ptr, const float* __restrict__ U, int C, int NS, int KQ,
                                                              float* __restrict__ dX) {
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  float* sU = dyn_lds;                                   // [GQ_UC][8][C]
  float* sY = sU + GQ_UC * GN_MAXNS * C;                 // [GQ_PB][GQ_UC][8]
  float4* sPg = reinterpret_cast<float4*>(sY + GQ_PB * GQ_UC * GN_MAXNS);   // [GQ_PB] geometry of the p edges
  float4* sOg = sPg + GQ_PB;                             // [GQ_UC] geometry of the o edges
  int* sPs = reinterpret_cast<int*>(sOg + GQ_UC);        // [GQ_PB] sources d
  int* sOs = sPs + GQ_PB;                                // [GQ_UC] sources c
  const int a = blockIdx.x, tid = threadIdx.x;
  const int ob = M.ptr[a], on = M.ptr[a + 1] - ob, qb = Q.ptr[a], qe = Q.ptr[a + 1];
  const int PC = min(8, 256 / C);
  const int NPASS = GQ_PB / PC;                          // <= 8
  const int pl = tid / C, ch = tid - pl * C;
  const bool active = pl < PC;
  for (int q = qb; q < qe; ++q) {
    const int b = Q.src[q], jq = q - qb;
    const float4 vba = Q.geom[q];
    const int pb = M.ptr[b], pn = M.ptr[b + 1] - pb;
    const long base = tin_ptr[q];
    for (int pblk = 0; pblk < pn; pblk += GQ_PB) {
      const int npb = min(GQ_PB, pn - pblk);
      float acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = 0.f;
      if (tid < npb) { sPg[tid] = M.geom[pb + pblk + tid]; sPs[tid] = M.src[pb + pblk + tid]; }
      for (int o0 = 0; o0 < on; o0 += GQ_UC) {
        const int no = min(GQ_UC, on - o0);
        if (tid >= 64 && tid < 64 + no) { sOg[tid - 64] = M.geom[ob + o0 + tid - 64]; sOs[tid - 64] = M.src[ob + o0 + tid - 64]; }
        for (int e = tid; e < no * NS * C; e += 256) {
          const int oi = e / (NS * C), r = e - oi * NS * C;
          sU[oi * GN_MAXNS * C + r] = U[((long)(ob + o0 + oi) * KQ + jq) * NS * C + r];
        }
        __syncthreads();
        for (int e = tid; e < npb * no; e += 256) {
          const int pi = e / no, oi = e - pi * no;
          float Yk[GN_MAXNS];
#pragma unroll
          for (int k = 0; k < GN_MAXNS; ++k) Yk[k] = 0.f;
          const int d = sPs[pi], cc = sOs[oi];
          if (d != a && cc != b && cc != d) gn_zonal(gn_cos_dihedral(gn_cross(sOg[oi], vba), sPg[pi], vba), NS, 1.f, Yk);
#pragma unroll
          for (int k = 0; k < GN_MAXNS; ++k) sY[(pi * GQ_UC + oi) * GN_MAXNS + k] = Yk[k];
        }
        __syncthreads();
        if (active) {
#pragma unroll
          for (int ps = 0; ps < 8; ++ps) {
            const int pi = ps * PC + pl;
            if (ps < NPASS && pi < npb) {
              float s0 = 0.f;
              for (int oi = 0; oi < no; ++oi) {
                const float* y = &sY[(pi * GQ_UC + oi) * GN_MAXNS];
                const float* u = &sU[oi * GN_MAXNS * C + ch];
#pragma unroll
                for (int k = 0; k < GN_MAXNS; ++k) if (k < NS) s0 += y[k] * u[k * C];
              }
              acc[ps] += s0;
            }
          }
        }
        __syncthreads();
      }
      if (active) {
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
          const int pi = ps * PC + pl;
          if (ps < NPASS && pi < npb) dX[(base + pblk + pi) * C + ch] = acc[ps];
        }
      }
    }
  }
}

// adjoint, step 1: U[o][jq][l'][c] = sum_l Y_l(cos cab(o, q)) dS[o][l][l'][c]  for the jq-th qint in-edge q of target(o) (KQ = maximum qint in-degree)
__global__ void k_gn_quad_bwd_u(GnSet M, GnSet Q, const float* __restrict__ dS, int C, int NS, int KQ, float scale, float* __restrict__ U) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)M.n * KQ * C) return;
  const int ch = (int)(t % C); const long r = t / C; const int jq = (int)(r % KQ); const int o = (int)(r / KQ);
  const int a = M.dst[o];
  const int q = Q.ptr[a] + jq;
  float* u = U + ((long)o * KQ + jq) * NS * C + ch;
  if (q >= Q.ptr[a + 1] || Q.src[q] == M.src[o]) {
    for (int k = 0; k < NS; ++k) u[(long)k * C] = 0.f;
    return;
  }
  float Yl[GN_MAXNS];
  gn_zonal(gn_clamp1(gn_dot(M.geom[o], Q.geom[q])), NS, scale, Yl);
  const float* g = dS + (long)o * NS * NS * C + ch;
  for (int k = 0; k < NS; ++k) {
    float s = 0.f;
#pragma unroll
    for (int l = 0; l < GN_MAXNS; ++l) if (l < NS) s += Yl[l] * g[(long)(l * NS + k) * C];
    u[(long)k * C] = s;
  }
}
// adjoint, step 2: dX[(q, j)][c] = sum_{o in row_M(target(q)), source(o) not in {source(q), d}} sum_l' Y_l'(dihedral(o, q, p_j)) U[o][jq(q)][l'][c]
__global__ void k_gn_quad_bwd_x(GnSet M, GnSet Q, const int* __restrict__ tin_ptr, const float* __restrict__ U, int C, int NS, int KQ,
                                float* __restrict__ dX) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long T = tin_ptr[Q.n];
  if (t >= T * C) return;
  const int ch = (int)(t % C); const long row = t / C;
  int lo = 0, hi = Q.n - 1;                       // q = last edge with tin_ptr[q] <= row (rows of empty blocks never match)
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tin_ptr[mid] <= row) lo = mid; else hi = mid - 1; }
  const int q = lo;
  const int a = Q.dst[q], b = Q.src[q];
  const int p = M.ptr[b] + (int)(row - tin_ptr[q]);
  const int d = M.src[p];
  float acc = 0.f;
  if (d != a) {
    const float4 vba = Q.geom[q], vdb = M.geom[p];
    const int jq = q - Q.ptr[a];
    for (int o = M.ptr[a]; o < M.ptr[a + 1]; ++o) {
      const int c = M.src[o];
      if (c == b || c == d) continue;
      float Yk[GN_MAXNS];
      gn_zonal(gn_cos_dihedral(gn_cross(M.geom[o], vba), vdb, vba), NS, 1.f, Yk);
      const float* u = U + ((long)o * KQ + jq) * NS * C + ch;
#pragma unroll
      for (int k = 0; k < GN_MAXNS; ++k) if (k < NS) acc += Yk[k] * u[(long)k * C];
    }
  }
  dX[t] = acc;
}

// adjoint of x_tin[row] = x[tin_main[row]]: out[p][c] = sum_{q : source(q) = target(p)} g[tin_ptr[q] + (p - first slot of the row)][c]
__global__ void k_gn_tin_scatter(GnSet M, const int* __restrict__ a2a_ptr, const int* __restrict__ q_of_rev, const int* __restrict__ tin_ptr,
                                 const float* __restrict__ g, int C, float* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)M.n * C) return;
  const int p = (int)(t / C), ch = (int)(t - (long)p * C);
  const int b = M.dst[p], j = p - M.ptr[b];
  float acc = 0.f;
  for (int s = a2a_ptr[b]; s < a2a_ptr[b + 1]; ++s) {
    const int q = q_of_rev[s];
    if (q >= 0) acc += g[((long)tin_ptr[q] + j) * C + ch];
  }
  out[t] = acc;
}

// ---- circular basis of the quadruplet path: cir[(q, j)][i] = sum_s RW[q][i * NS + s] Y_s(v_q . v_p),  p = j-th main in-edge of source(q) ----------------
__global__ void k_gn_cir_fwd(GnSet M, GnSet Q, const int* __restrict__ tin_ptr, const float* __restrict__ RW, int I, int NS, float scale,
                             float* __restrict__ cir) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long T = tin_ptr[Q.n];
  if (t >= T * I) return;
  const int i = (int)(t % I); const long row = t / I;
  int lo = 0, hi = Q.n - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tin_ptr[mid] <= row) lo = mid; else hi = mid - 1; }
  const int q = lo;
  const int p = M.ptr[Q.src[q]] + (int)(row - tin_ptr[q]);
  float Y[GN_MAXNS];
  gn_zonal(gn_clamp1(gn_dot(Q.geom[q], M.geom[p])), NS, scale, Y);
  const float* w = RW + ((long)q * I + i) * NS;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < GN_MAXNS; ++k) if (k < NS) s += w[k] * Y[k];
  cir[t] = s;
}
__global__ void k_gn_cir_bwd(GnSet M, GnSet Q, const int* __restrict__ tin_ptr, const float* __restrict__ dcir, int I, int NS, float scale,
                             float* __restrict__ dRW) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)Q.n * I) return;
  const int i = (int)(t % I); const int q = (int)(t / I);
  const int pb = M.ptr[Q.src[q]], n = tin_ptr[q + 1] - tin_ptr[q];
  const float4 vq = Q.geom[q];
  float acc[GN_MAXNS];
#pragma unroll
  for (int k = 0; k < GN_MAXNS; ++k) acc[k] = 0.f;
  for (int j = 0; j < n; ++j) {
    float Y[GN_MAXNS];
    gn_zonal(gn_clamp1(gn_dot(vq, M.geom[pb + j])), NS, scale, Y);
    const float g = dcir[((long)tin_ptr[q] + j) * I + i];
#pragma unroll
    for (int k = 0; k < GN_MAXNS; ++k) acc[k] += Y[k] * g;
  }
  for (int k = 0; k < NS; ++k) dRW[((long)q * I + i) * NS + k] = acc[k];
}

// ---- per-row product out[o][i][c] = sum_s R[o][i * NSS + s] S[o][s][c] (rad_W1 @ sph_m, efficient.py:231-244) and its two adjoints --------------------
// One workgroup per row: R[o] (I x NSS), S[o] (NSS x C) (and dout[o], I x C) are staged in LDS once (S with a padded row stride so that the dR contraction,
// whose lanes differ in s, is conflict free); every global element is read exactly once.
__global__ __launch_bounds__(256) void k_gn_rowmm_fwd(const float* __restrict__ R, const float* __restrict__ S, long n, int I, int NSS, int C,
                                                        float* __restrict__ out) {
  extern __shared__ float lds[];
  float* sR = lds; float* sS = lds + I * NSS;
  const long o = blockIdx.x;
  for (int t = threadIdx.x; t < I * NSS; t += 256) sR[t] = R[o * I * NSS + t];
  for (int t = threadIdx.x; t < NSS * C; t += 256) sS[t] = S[o * NSS * C + t];
  __syncthreads();
  for (int t = threadIdx.x; t < I * C; t += 256) {
    const int i = t / C, ch = t - i * C;
    float acc = 0.f;
    for (int k = 0; k < NSS; ++k) acc += sR[i * NSS + k] * sS[k * C + ch];
    out[o * I * C + t] = acc;
  }
}
__global__ __launch_bounds__(256) void k_gn_rowmm_bwd(const float* __restrict__ R, const float* __restrict__ S, const float* __restrict__ dout, long n, int I,
                                                        int NSS, int C, float* __restrict__ dR, float* __restrict__ dS) {
  extern __shared__ float lds[];
  const int CP = C + 1;
  float* sR = lds; float* sS = sR + I * NSS; float* sG = sS + NSS * CP;
  const long o = blockIdx.x;
  for (int t = threadIdx.x; t < I * NSS; t += 256) sR[t] = R[o * I * NSS + t];
  for (int t = threadIdx.x; t < NSS * C; t += 256) { const int k = t / C; sS[k * CP + (t - k * C)] = S[o * NSS * C + t]; }
  for (int t = threadIdx.x; t < I * C; t += 256) sG[t] = dout[o * I * C + t];
  __syncthreads();
  if (dS)
    for (int t = threadIdx.x; t < NSS * C; t += 256) {
      const int k = t / C, ch = t - k * C;
      float acc = 0.f;
      for (int i = 0; i < I; ++i) acc += sR[i * NSS + k] * sG[i * C + ch];
      dS[o * NSS * C + t] = acc;
    }
  if (dR)
    for (int t = threadIdx.x; t < I * NSS; t += 256) {
      const int i = t / NSS, k = t - i * NSS;
      float acc = 0.f;
      for (int ch = 0; ch < C; ++ch) acc += sG[i * C + ch] * sS[k * CP + ch];
      dR[o * I * NSS + t] = acc;
    }
}

// ---- atom-atom pairs: out[a][r][c] = sum_{p in row(a)} RW[p][r] X[source(p)][c] (interaction_block.py:721-733) ---------------------------------------
__global__ void k_gn_pair_fwd(GnSet A, const float* __restrict__ RW, const float* __restrict__ X, int N, int Rr, int C, float* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)N * C) return;
  const int a = (int)(t / C), ch = (int)(t - (long)a * C);
  float acc[GN_MAXR];
#pragma unroll
  for (int r = 0; r < GN_MAXR; ++r) acc[r] = 0.f;
  for (int p = A.ptr[a]; p < A.ptr[a + 1]; ++p) {
    const float x = X[(long)A.src[p] * C + ch];
    const float* w = RW + (long)p * Rr;
#pragma unroll
    for (int r = 0; r < GN_MAXR; ++r) if (r < Rr) acc[r] += w[r] * x;
  }
  for (int r = 0; r < Rr; ++r) out[((long)a * Rr + r) * C + ch] = acc[r];
}
// dX[j][c] = sum over edges with source j = the reverses of row(j): sum_{p in row(j)} sum_r RW[rev p][r] dout[source(p)][r][c]
__global__ void k_gn_pair_bwd_x(GnSet A, const int* __restrict__ rev, const float* __restrict__ RW, const float* __restrict__ dout, int N, int Rr, int C,
                                float* __restrict__ dX) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)N * C) return;
  const int j = (int)(t / C), ch = (int)(t - (long)j * C);
  float acc = 0.f;
  for (int p = A.ptr[j]; p < A.ptr[j + 1]; ++p) {
    const float* w = RW + (long)rev[p] * Rr;
    const float* g = dout + (long)A.src[p] * Rr * C + ch;
    for (int r = 0; r < Rr; ++r) acc += w[r] * g[(long)r * C];
  }
  dX[t] = acc;
}
__global__ void k_gn_pair_bwd_w(GnSet A, const float* __restrict__ X, const float* __restrict__ dout, int Rr, int C, float* __restrict__ dRW) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)A.n * Rr) return;
  const int r = (int)(t % Rr); const int p = (int)(t / Rr);
  const float* x = X + (long)A.src[p] * C;
  const float* g = dout + ((long)A.dst[p] * Rr + r) * C;
  float acc = 0.f;
  for (int ch = 0; ch < C; ++ch) acc += x[ch] * g[ch];
  dRW[t] = acc;
}

// ---- edge embedding input cat[e] = [h[source] | h[target] | m[e]] and its adjoint ------------------------------------------------------------------------
__global__ void k_gn_cat_fwd(GnSet M, const float* __restrict__ h, const float* __restrict__ m, int A, int Em, float* __restrict__ cat) {
  const int W = 2 * A + Em;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)M.n * W) return;
  const int e = (int)(t / W), k = (int)(t - (long)e * W);
  cat[t] = k < A ? h[(long)M.src[e] * A + k] : (k < 2 * A ? h[(long)M.dst[e] * A + k - A] : m[(long)e * Em + k - 2 * A]);
}
__global__ void k_gn_cat_bwd_h(GnSet M, const int* __restrict__ rev, const float* __restrict__ dcat, int N, int A, int Em, float* __restrict__ dh) {
  const int W = 2 * A + Em;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)N * A) return;
  const int j = (int)(t / A), k = (int)(t - (long)j * A);
  float acc = 0.f;
  for (int p = M.ptr[j]; p < M.ptr[j + 1]; ++p) acc += dcat[(long)rev[p] * W + k] + dcat[(long)p * W + A + k];
  dh[t] = acc;
}

// ---- out[a][c] = sum_{p in row(a)} m[p][c] r[p][c]  (AtomUpdateBlock / OutputBlock: m * dense_rbf(basis), scatter to the target) --------------------------
__global__ void k_gn_mulsum_fwd(GnSet M, const float* __restrict__ m, const float* __restrict__ r, int N, int C, float* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)N * C) return;
  const int a = (int)(t / C), ch = (int)(t - (long)a * C);
  float acc = 0.f;
  for (int p = M.ptr[a]; p < M.ptr[a + 1]; ++p) acc += m[(long)p * C + ch] * r[(long)p * C + ch];
  out[t] = acc;
}
__global__ void k_gn_mulsum_bwd(GnSet M, const float* __restrict__ m, const float* __restrict__ r, const float* __restrict__ dout, int C,
                                float* __restrict__ dm, float* __restrict__ dr) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)M.n * C) return;
  const int p = (int)(t / C), ch = (int)(t - (long)p * C);
  const float g = dout[(long)M.dst[p] * C + ch];
  dm[t] = g * r[t];
  dr[t] = g * m[t];
}

// ---- direct forces (gemnet_oc.py:1216-1243): Fc[e] = (F[e] + F[rev e]) / 2 (forces_coupled) ; F_atom[a] = sum_{p in row(a)} Fc[p] v_p ----------------
__global__ void k_gn_forces_fwd(GnSet M, const int* __restrict__ rev, const float* __restrict__ F, int N, int coupled, float* __restrict__ out) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= N) return;
  float fx = 0.f, fy = 0.f, fz = 0.f;
  for (int p = M.ptr[a]; p < M.ptr[a + 1]; ++p) {
    const float f = coupled ? (F[p] + F[rev[p]]) / 2.f : F[p];
    const float4 v = M.geom[p];
    fx += f * v.x; fy += f * v.y; fz += f * v.z;
  }
  out[3 * (long)a] = fx; out[3 * (long)a + 1] = fy; out[3 * (long)a + 2] = fz;
}
__global__ void k_gn_forces_bwd(GnSet M, const int* __restrict__ rev, const float* __restrict__ dout, int coupled, float* __restrict__ dF) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M.n) return;
  const float4 v = M.geom[p];
  const float* g = dout + 3 * (long)M.dst[p];
  float s = g[0] * v.x + g[1] * v.y + g[2] * v.z;
  if (coupled) {
    const int r = rev[p];
    const float4 w = M.geom[r];
    const float* gr = dout + 3 * (long)M.dst[r];
    s = (s + gr[0] * w.x + gr[1] * w.y + gr[2] * w.z) / 2.f;
  }
  dF[p] = s;
}

// ---- small data movement ---------------------------------------------------------------------------------------------------------------------------------
// out[p] = x[idx[p]] (* y[p])
__global__ void k_gn_gather(const float* __restrict__ x, const int* __restrict__ idx, const float* __restrict__ y, long P, int C, float* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P * C) return;
  const long p = t / C; const int ch = (int)(t - p * C);
  const float v = x[(long)idx[p] * C + ch];
  out[t] = y ? v * y[t] : v;
}
// out[n] = sum_{q in [ptr[n], ptr[n+1])} rows[order ? order[q] : q] (entries with order[q] < 0 are skipped), optionally times y[row]
__global__ void k_gn_segsum(const float* __restrict__ rows, const float* __restrict__ y, const int* __restrict__ order, const int* __restrict__ ptr, long N,
                            int C, float* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * C) return;
  const long n = t / C; const int ch = (int)(t - n * C);
  float acc = 0.f;
  for (int q = ptr[n]; q < ptr[n + 1]; ++q) {
    const int r = order ? order[q] : q;
    if (r < 0) continue;
    const float v = rows[(long)r * C + ch];
    acc += y ? v * y[(long)r * C + ch] : v;
  }
  out[t] = acc;
}
__global__ void k_gn_mul(const float* __restrict__ a, const float* __restrict__ b, long n, float* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) out[t] = a[t] * b[t];
}
// out = alpha * a + beta * b (b nullable)
__global__ void k_gn_lincomb(const float* __restrict__ a, const float* __restrict__ b, float alpha, float beta, long n, float* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) out[t] = b ? alpha * a[t] + beta * b[t] : alpha * a[t];
}
__global__ void k_gn_ssilu_bwd(const float* __restrict__ z, const float* __restrict__ g, float scale, long n, float* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) out[t] = scale * g[t] * nq_dsilu(z[t]);
}
// dW[t][c] = sum_{n : z[n] == t + 1} g[n][c]  (adjoint of Embedding(z - 1), embedding_block.py:39-53), fixed order
__global__ void k_gn_embed_grad(const int* __restrict__ z, const float* __restrict__ g, int N, int T, int C, float* __restrict__ dW) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)T * C) return;
  const int el = (int)(t / C), ch = (int)(t - (long)el * C);
  float acc = 0.f;
  for (int n = 0; n < N; ++n) if (z[n] == el + 1) acc += g[(long)n * C + ch];
  dW[t] = acc;
}

// =========================================================================================================================================================
#define GN_GRID(total) dim3((unsigned)(((total) + 255) / 256)), dim3(256), 0, st
#define GN_CHECK_NS(NS) if ((NS) < 1 || (NS) > GN_MAXNS) return nq_fail(NQ_ERR_ARG, "num_spherical %d outside [1, %d]", (int)(NS), GN_MAXNS)

struct nq_gn_set_c { int32_t n, reserved; const int32_t* ptr; const int32_t* src; const int32_t* dst; const float* geom; };
static GnSet gn_view(const nq_gn_set_c* s) { GnSet v; v.n = s->n; v.ptr = s->ptr; v.src = s->src; v.dst = s->dst; v.geom = (const float4*)s->geom; return v; }

extern "C" {

int nq_gn_radial_basis(const float* geom, int64_t n, int32_t num_radial, const float* offset, double cutoff, double exponent, float scale, float* out,
                       void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_rbf");
  if (n <= 0) return NQ_OK;
  if (!geom || !offset || !out || num_radial < 2) return nq_fail(NQ_ERR_ARG, "bad argument");
  const double width = 1.0 / (num_radial - 1);
  hipLaunchKernelGGL(k_gn_rbf, GN_GRID(n * num_radial), (const float4*)geom, (long)n, num_radial, offset, (float)(1.0 / cutoff), (float)(-0.5 / (width * width)),
                     (float)exponent, scale, out);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_gn_triplet_forward(const void* out_set, const void* in_set, const float* x, int32_t C, int32_t NS, float scale, float* S, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_tri_fwd");
  GN_CHECK_NS(NS);
  const GnSet O = gn_view((const nq_gn_set_c*)out_set), G = gn_view((const nq_gn_set_c*)in_set);
  if (O.n <= 0) return NQ_OK;
  hipLaunchKernelGGL(k_gn_tri_fwd, GN_GRID((long)O.n * C), O, G, x, C, NS, scale, S);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_gn_triplet_backward(const void* out_set, const void* in_set, const float* dS, int32_t C, int32_t NS, float scale, float* dx, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_tri_bwd");
  GN_CHECK_NS(NS);
  const GnSet O = gn_view((const nq_gn_set_c*)out_set), G = gn_view((const nq_gn_set_c*)in_set);
  if (G.n <= 0) return NQ_OK;
  hipLaunchKernelGGL(k_gn_tri_bwd, GN_GRID((long)G.n * C), O, G, dS, C, NS, scale, dx);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

static int g_gn_quad_variant = 1;
extern "C" void nq_gn_set_quad_variant(int32_t v) { g_gn_quad_variant = v; }

int nq_gn_quad_forward(const void* main_set, const void* qint_set, const int32_t* tin_ptr, int32_t n_atoms, const float* x, int32_t C, int32_t NS, float scale,
                       float* S, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_quad_fwd");
  GN_CHECK_NS(NS);
  const GnSet M = gn_view((const nq_gn_set_c*)main_set), Q = gn_view((const nq_gn_set_c*)qint_set);
  if (M.n <= 0) return NQ_OK;
  if (C <= 64 && g_gn_quad_variant == 1) {
    hipLaunchKernelGGL(k_gn_quad_fwd_atom, dim3(n_atoms), dim3(256), 0, st, M, Q, tin_ptr, x, C, NS, scale, S);
  } else {
    hipLaunchKernelGGL(k_gn_quad_fwd, GN_GRID((long)M.n * C), M, Q, tin_ptr, x, C, NS, scale, S);
  }
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
/* scratch: f32[main edges * KQ * NS * C] */
int nq_gn_quad_backward(const void* main_set, const void* qint_set, const int32_t* tin_ptr, int32_t n_atoms, int64_t T, const float* dS, int32_t C, int32_t NS,
                        int32_t KQ, float scale, float* scratch, float* dx, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_quad_bwd");
  GN_CHECK_NS(NS);
  const GnSet M = gn_view((const nq_gn_set_c*)main_set), Q = gn_view((const nq_gn_set_c*)qint_set);
  if (M.n <= 0 || T <= 0 || KQ <= 0) return NQ_OK;
  hipLaunchKernelGGL(k_gn_quad_bwd_u, GN_GRID((long)M.n * KQ * C), M, Q, dS, C, NS, KQ, scale, scratch);
  NQ_LAUNCH_CHECK();
  if (C <= 64 && g_gn_quad_variant == 1) hipLaunchKernelGGL(k_gn_quad_bwd_x_atom, dim3(n_atoms), dim3(256), sizeof(float) * (GQ_UC * GN_MAXNS * C + GQ_PB * GQ_UC * GN_MAXNS + 5 * (GQ_PB + GQ_UC)),
                                                        st, M, Q, tin_ptr, scratch, C, NS, KQ, dx);
  else hipLaunchKernelGGL(k_gn_quad_bwd_x, GN_GRID(T * C), M, Q, tin_ptr, scratch, C, NS, KQ, dx);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_gn_tin_scatter(const void* main_set, const int32_t* a2a_row_ptr, const int32_t* q_of_rev, const int32_t* tin_ptr, const float* g, int32_t C,
                      float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_tin_scatter");
  const GnSet M = gn_view((const nq_gn_set_c*)main_set);
  if (M.n <= 0) return NQ_OK;
  hipLaunchKernelGGL(k_gn_tin_scatter, GN_GRID((long)M.n * C), M, a2a_row_ptr, q_of_rev, tin_ptr, g, C, out);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_gn_cir_forward(const void* main_set, const void* qint_set, const int32_t* tin_ptr, int64_t T, const float* rad_w1, int32_t I, int32_t NS, float scale,
                      float* cir, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_cir_fwd");
  GN_CHECK_NS(NS);
  const GnSet M = gn_view((const nq_gn_set_c*)main_set), Q = gn_view((const nq_gn_set_c*)qint_set);
  if (T <= 0) return NQ_OK;
  hipLaunchKernelGGL(k_gn_cir_fwd, GN_GRID(T * I), M, Q, tin_ptr, rad_w1, I, NS, scale, cir);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_gn_cir_backward(const void* main_set, const void* qint_set, const int32_t* tin_ptr, const float* dcir, int32_t I, int32_t NS, float scale,
                       float* d_rad_w1, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_cir_bwd");
  GN_CHECK_NS(NS);
  const GnSet M = gn_view((const nq_gn_set_c*)main_set), Q = gn_view((const nq_gn_set_c*)qint_set);
  if (Q.n <= 0) return NQ_OK;
  hipLaunchKernelGGL(k_gn_cir_bwd, GN_GRID((long)Q.n * I), M, Q, tin_ptr, dcir, I, NS, scale, d_rad_w1);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_gn_rowmm_forward(const float* R, const float* S, int64_t n, int32_t I, int32_t NSS, int32_t C, float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_rowmm_fwd");
  if (n <= 0) return NQ_OK;
  const size_t lds = sizeof(float) * ((size_t)I * NSS + (size_t)NSS * C);
  if (lds > 64 * 1024) return nq_fail(NQ_ERR_ARG, "rowmm: I=%d NSS=%d C=%d does not fit the LDS tile", I, NSS, C);
  hipLaunchKernelGGL(k_gn_rowmm_fwd, dim3((unsigned)n), dim3(256), lds, st, R, S, (long)n, I, NSS, C, out);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_gn_rowmm_backward(const float* R, const float* S, const float* dout, int64_t n, int32_t I, int32_t NSS, int32_t C, float* dR, float* dS, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_rowmm_bwd");
  if (n <= 0 || (!dR && !dS)) return NQ_OK;
  const size_t lds = sizeof(float) * ((size_t)I * NSS + (size_t)NSS * (C + 1) + (size_t)I * C);
  if (lds > 64 * 1024) return nq_fail(NQ_ERR_ARG, "rowmm: I=%d NSS=%d C=%d does not fit the LDS tile", I, NSS, C);
  hipLaunchKernelGGL(k_gn_rowmm_bwd, dim3((unsigned)n), dim3(256), lds, st, R, S, dout, (long)n, I, NSS, C, dR, dS);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_gn_pair_forward(const void* a2a_set, const float* rad_w, const float* x, int32_t N, int32_t Rr, int32_t C, float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_pair_fwd");
  if (Rr < 1 || Rr > GN_MAXR) return nq_fail(NQ_ERR_ARG, "emb_size_rbf %d outside [1, %d]", Rr, GN_MAXR);
  const GnSet A = gn_view((const nq_gn_set_c*)a2a_set);
  hipLaunchKernelGGL(k_gn_pair_fwd, GN_GRID((long)N * C), A, rad_w, x, N, Rr, C, out);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_gn_pair_backward(const void* a2a_set, const int32_t* rev, const float* rad_w, const float* x, const float* dout, int32_t N, int32_t Rr, int32_t C,
                        float* d_rad_w, float* dx, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_pair_bwd");
  const GnSet A = gn_view((const nq_gn_set_c*)a2a_set);
  if (dx) { hipLaunchKernelGGL(k_gn_pair_bwd_x, GN_GRID((long)N * C), A, rev, rad_w, dout, N, Rr, C, dx); NQ_LAUNCH_CHECK(); }
  if (d_rad_w && A.n > 0) { hipLaunchKernelGGL(k_gn_pair_bwd_w, GN_GRID((long)A.n * Rr), A, x, dout, Rr, C, d_rad_w); NQ_LAUNCH_CHECK(); }
  return NQ_OK;
}

int nq_gn_cat_forward(const void* main_set, const float* h, const float* m, int32_t A, int32_t Em, float* cat, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_cat_fwd");
  const GnSet M = gn_view((const nq_gn_set_c*)main_set);
  if (M.n <= 0) return NQ_OK;
  hipLaunchKernelGGL(k_gn_cat_fwd, GN_GRID((long)M.n * (2 * A + Em)), M, h, m, A, Em, cat);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_gn_cat_backward_h(const void* main_set, const int32_t* rev, const float* dcat, int32_t N, int32_t A, int32_t Em, float* dh, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_cat_bwd");
  const GnSet M = gn_view((const nq_gn_set_c*)main_set);
  hipLaunchKernelGGL(k_gn_cat_bwd_h, GN_GRID((long)N * A), M, rev, dcat, N, A, Em, dh);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_gn_mulsum_forward(const void* main_set, const float* m, const float* r, int32_t N, int32_t C, float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_mulsum_fwd");
  const GnSet M = gn_view((const nq_gn_set_c*)main_set);
  hipLaunchKernelGGL(k_gn_mulsum_fwd, GN_GRID((long)N * C), M, m, r, N, C, out);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_gn_mulsum_backward(const void* main_set, const float* m, const float* r, const float* dout, int32_t C, float* dm, float* dr, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_mulsum_bwd");
  const GnSet M = gn_view((const nq_gn_set_c*)main_set);
  if (M.n <= 0) return NQ_OK;
  hipLaunchKernelGGL(k_gn_mulsum_bwd, GN_GRID((long)M.n * C), M, m, r, dout, C, dm, dr);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_gn_forces_forward(const void* main_set, const int32_t* rev, const float* f_edge, int32_t N, int32_t coupled, float* forces, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_forces_fwd");
  const GnSet M = gn_view((const nq_gn_set_c*)main_set);
  hipLaunchKernelGGL(k_gn_forces_fwd, GN_GRID((long)N), M, rev, f_edge, N, coupled, forces);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_gn_forces_backward(const void* main_set, const int32_t* rev, const float* d_forces, int32_t coupled, float* d_f_edge, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_forces_bwd");
  const GnSet M = gn_view((const nq_gn_set_c*)main_set);
  if (M.n <= 0) return NQ_OK;
  hipLaunchKernelGGL(k_gn_forces_bwd, GN_GRID((long)M.n), M, rev, d_forces, coupled, d_f_edge);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_gn_gather(const float* x, const int32_t* idx, const float* y, int64_t P, int32_t C, float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_gather");
  if (P <= 0) return NQ_OK;
  hipLaunchKernelGGL(k_gn_gather, GN_GRID(P * C), x, idx, y, (long)P, C, out);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_gn_segment_sum(const float* rows, const float* y, const int32_t* order, const int32_t* ptr, int64_t N, int32_t C, float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_segsum");
  if (N <= 0) return NQ_OK;
  hipLaunchKernelGGL(k_gn_segsum, GN_GRID(N * C), rows, y, order, ptr, (long)N, C, out);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_gn_mul(const float* a, const float* b, int64_t n, float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_mul");
  if (n <= 0) return NQ_OK;
  hipLaunchKernelGGL(k_gn_mul, GN_GRID(n), a, b, (long)n, out);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_gn_lincomb(const float* a, const float* b, float alpha, float beta, int64_t n, float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_lincomb");
  if (n <= 0) return NQ_OK;
  hipLaunchKernelGGL(k_gn_lincomb, GN_GRID(n), a, b, alpha, beta, (long)n, out);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_gn_ssilu_backward(const float* z, const float* g, float scale, int64_t n, float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_ssilu_bwd");
  if (n <= 0) return NQ_OK;
  hipLaunchKernelGGL(k_gn_ssilu_bwd, GN_GRID(n), z, g, scale / 0.6f, (long)n, out);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
int nq_gn_embed_grad(const int32_t* z, const float* g, int32_t N, int32_t num_elements, int32_t C, float* dW, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NQ_PROF(st, "gn_embed_grad");
  hipLaunchKernelGGL(k_gn_embed_grad, GN_GRID((long)num_elements * C), z, g, N, num_elements, C, dW);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

}  // extern "C"
