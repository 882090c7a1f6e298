// Neighbour list for the PaiNN path, one workgroup per molecule.
//
// Replaces (reference, /root/reference/nablaDFT/painn_pyg/):
//   painn.py:411-416   torch_cluster.radius_graph(pos, r, batch, max_num_neighbors)
//   painn.py:418-423   distance_vec / edge_dist / compute_neighbors (utils.py:469-481)
//   painn.py:319-321   unit vectors
//   painn.py:233-295   symmetrize_edges (mask j<i, concat flips, per-graph reorder, id_swap)
//
// Output 1 (what the reference returns; bit-exact contract): canonical edge list, per graph
//   [kept edges (src j < dst i), dst ascending then src ascending] ++ [their flips].
// Output 2 (what the engine consumes): CSR by target atom with sources ascending, the slot of
//   the reverse edge, and {unit vector, distance} per slot.  Because the canonical list is
//   symmetric, the CSR row of atom n is both "in-edges of n" and "out-edges of n".
//
// Mapping to the hardware: positions of one molecule live in LDS; a wavefront owns one centre
// atom, its 64 lanes test 64 candidate neighbours at once, __ballot gives the adjacency word
// and popcount prefixes give compaction slots: no atomics, no sort, deterministic order.
#include "common.h"

#define NQ_MAX_MOL_ATOMS 512
#define GRAPH_THREADS 256

typedef unsigned long long u64;

__device__ __forceinline__ u64 bits_below(int b) {  // mask of bit positions < b (b may be <=0 or >=64)
  if (b <= 0) return 0ull;
  if (b >= 64) return ~0ull;
  return (1ull << b) - 1ull;
}

// squared distance exactly as torch evaluates (x_i - x_j).pow(2).sum(-1): ((dx*dx + dy*dy) + dz*dz), one rounding per
// operation, no FMA, no reassociation.  Written with explicit VALU instructions: with plain C the packed-math SLP pass was
// observed to emit (dx*dx + dz*dz) + dy*dy (15 % of distances then differ from the reference in the last bit).
__device__ __forceinline__ float dist2_nofma(const float* a, const float* b) {
  float dx, dy, dz, xx, yy, zz, s;
  asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dx) : "v"(a[0]), "v"(b[0]));
  asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dy) : "v"(a[1]), "v"(b[1]));
  asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dz) : "v"(a[2]), "v"(b[2]));
  asm volatile("v_mul_f32 %0, %1, %1" : "=v"(xx) : "v"(dx));
  asm volatile("v_mul_f32 %0, %1, %1" : "=v"(yy) : "v"(dy));
  asm volatile("v_mul_f32 %0, %1, %1" : "=v"(zz) : "v"(dz));
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(s) : "v"(xx), "v"(yy));
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(s) : "v"(s), "v"(zz));
  return s;
}

// Build in LDS: LK[i][w] = kept lower neighbours (j < i, first K in ascending j) of atom i,
//               S[i][w]  = symmetric adjacency = LK[i] | {j > i : i in LK[j]}.
__device__ void build_adjacency(const float* sp, int n, int W, float r2, int K, u64* LK, u64* S) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  for (int i = wave; i < n; i += nwaves) {
    int kept = 0;
    for (int w = 0; w < W; ++w) {
      u64 m = 0ull;
      if (64 * w < i) {  // only candidates j < i matter (wave-uniform branch)
        int j = 64 * w + lane;
        bool pred = (j < i) && (dist2_nofma(sp + 3 * i, sp + 3 * j) < r2);
        m = __ballot(pred);
        int c = __popcll(m);
        if (kept + c > K) {  // keep the first (K - kept) neighbours of this word
          int allow = K - kept;
          while (c > allow) {
            m &= ~(1ull << (63 - __clzll((long long)m)));
            --c;
          }
        }
        kept += c;
      }
      if (lane == 0) LK[i * W + w] = m;
    }
  }
  __syncthreads();
  for (int i = wave; i < n; i += nwaves) {
    const int wi = i >> 6, bi = i & 63;
    for (int w = 0; w < W; ++w) {
      int j = 64 * w + lane;
      bool up = (j > i) && (j < n) && ((LK[j * W + wi] >> bi) & 1ull);
      u64 m = __ballot(up) | LK[i * W + w];
      if (lane == 0) S[i * W + w] = m;
    }
  }
  __syncthreads();
}

// Correctly rounded sqrt.  hipcc's sqrtf is only faithful (measured on gfx950: 15 % of results differ from IEEE by 1 ulp);
// one FMA-residual Newton step s + (x - s*s) * (0.5/s) from a <=1-ulp start rounds to the exact result (verified exhaustively
// against numpy on 2e6 samples for starts of -1/0/+1 ulp).
__device__ __forceinline__ float sqrt_rn(float x) {
  if (!(x > 0.0f)) return x == 0.0f ? 0.0f : sqrtf(x);
  const float s = __builtin_sqrtf(x);
  const float r = fmaf(-s, s, x);
  return fmaf(r, 0.5f / s, s);
}

__device__ __forceinline__ int rank_in_row(const u64* row, int idx) {  // set bits of row at positions < idx
  int r = 0;
  const int wi = idx >> 6;
  for (int w = 0; w < wi; ++w) r += __popcll(row[w]);
  return r + __popcll(row[wi] & bits_below(idx & 63));
}

__global__ __launch_bounds__(GRAPH_THREADS) void k_graph_count(const float* __restrict__ pos, const int* __restrict__ mol_ptr,
                                                               float r2, int K, int* __restrict__ deg, int* __restrict__ lowdeg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int a = mol_ptr[blockIdx.x], n = mol_ptr[blockIdx.x + 1] - a;
  if (n <= 0) return;
  const int W = (n + 63) >> 6;
  u64* LK = (u64*)smem;
  u64* S = LK + n * W;
  float* sp = (float*)(S + n * W);
  for (int t = threadIdx.x; t < 3 * n; t += blockDim.x) sp[t] = pos[3 * (long)a + t];
  __syncthreads();
  build_adjacency(sp, n, W, r2, K, LK, S);
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int d = 0, l = 0;
    for (int w = 0; w < W; ++w) {
      d += __popcll(S[i * W + w]);
      l += __popcll(LK[i * W + w]);
    }
    deg[a + i] = d;
    lowdeg[a + i] = l;
  }
}

// exclusive scan of two int arrays by ONE workgroup (N is at most a few 1e5 atoms).  Every thread owns SCAN_PER consecutive elements (serial prefix in
// registers), the wavefront scans the thread totals with DPP shuffles, the 16 wavefront totals and the running carry go through LDS: 8192 elements per trip
// and three barriers per trip (was 1024 elements per trip: 84 trips = 0.13 ms at 86 k atoms, now 11).
#define SCAN_PER 8
__global__ __launch_bounds__(1024) void k_scan2(const int* __restrict__ in0, const int* __restrict__ in1, int n,
                                                int* __restrict__ out0, int* __restrict__ out1) {
  __shared__ int wsum0[16], wsum1[16];
  __shared__ int carry0, carry1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry0 = carry1 = 0;
  const bool vec = ((reinterpret_cast<uintptr_t>(in0) | reinterpret_cast<uintptr_t>(in1) | reinterpret_cast<uintptr_t>(out0) | reinterpret_cast<uintptr_t>(out1)) & 15) == 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024 * SCAN_PER) {
    const int i0 = base + threadIdx.x * SCAN_PER;
    int v0[SCAN_PER], v1[SCAN_PER];
    int t0 = 0, t1 = 0;
    if (vec && i0 + SCAN_PER <= n) {   // 16-byte loads (a thread's 8 elements are 32 contiguous bytes)
      const int4 a = reinterpret_cast<const int4*>(in0 + i0)[0], b = reinterpret_cast<const int4*>(in0 + i0)[1];
      const int4 c = reinterpret_cast<const int4*>(in1 + i0)[0], d = reinterpret_cast<const int4*>(in1 + i0)[1];
      v0[0] = a.x; v0[1] = a.y; v0[2] = a.z; v0[3] = a.w; v0[4] = b.x; v0[5] = b.y; v0[6] = b.z; v0[7] = b.w;
      v1[0] = c.x; v1[1] = c.y; v1[2] = c.z; v1[3] = c.w; v1[4] = d.x; v1[5] = d.y; v1[6] = d.z; v1[7] = d.w;
    } else {
#pragma unroll
      for (int e = 0; e < SCAN_PER; ++e) { v0[e] = i0 + e < n ? in0[i0 + e] : 0; v1[e] = i0 + e < n ? in1[i0 + e] : 0; }
    }
#pragma unroll
    for (int e = 0; e < SCAN_PER; ++e) { t0 += v0[e]; t1 += v1[e]; }
    int s0 = t0, s1 = t1;  // inclusive wave scan of the thread totals
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      int u0 = __shfl_up(s0, off, 64), u1 = __shfl_up(s1, off, 64);
      if (lane >= off) { s0 += u0; s1 += u1; }
    }
    if (lane == 63) { wsum0[wave] = s0; wsum1[wave] = s1; }
    __syncthreads();
    int p0 = carry0, p1 = carry1;
    for (int w = 0; w < wave; ++w) { p0 += wsum0[w]; p1 += wsum1[w]; }
    int r0 = p0 + s0 - t0, r1 = p1 + s1 - t1;   // exclusive prefix of this thread's first element
    int o0[SCAN_PER], o1[SCAN_PER];
#pragma unroll
    for (int e = 0; e < SCAN_PER; ++e) { o0[e] = r0; o1[e] = r1; r0 += v0[e]; r1 += v1[e]; }
    if (vec && i0 + SCAN_PER <= n) {
      reinterpret_cast<int4*>(out0 + i0)[0] = make_int4(o0[0], o0[1], o0[2], o0[3]); reinterpret_cast<int4*>(out0 + i0)[1] = make_int4(o0[4], o0[5], o0[6], o0[7]);
      reinterpret_cast<int4*>(out1 + i0)[0] = make_int4(o1[0], o1[1], o1[2], o1[3]); reinterpret_cast<int4*>(out1 + i0)[1] = make_int4(o1[4], o1[5], o1[6], o1[7]);
    } else {
#pragma unroll
      for (int e = 0; e < SCAN_PER; ++e)
        if (i0 + e < n) { out0[i0 + e] = o0[e]; out1[i0 + e] = o1[e]; }
    }
    __syncthreads();
    if (threadIdx.x == 1023) { carry0 = p0 + s0; carry1 = p1 + s1; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out0[n] = carry0; out1[n] = carry1; }
}


__global__ __launch_bounds__(GRAPH_THREADS) void k_graph_fill(GraphFillArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int g = blockIdx.x;
  const int a = p.mol_ptr[g], n = p.mol_ptr[g + 1] - a;
  if (n <= 0) {
    if (threadIdx.x == 0 && p.neighbors) p.neighbors[g] = 0;
    return;
  }
  const int W = (n + 63) >> 6;
  u64* LK = (u64*)smem;
  u64* S = LK + n * W;
  float* sp = (float*)(S + n * W);
  for (int t = threadIdx.x; t < 3 * n; t += blockDim.x) sp[t] = p.pos[3 * (long)a + t];
  for (int t = threadIdx.x; t < n; t += blockDim.x) p.atom_mol[a + t] = g;
  __syncthreads();
  build_adjacency(sp, n, W, p.r2, p.K, LK, S);
  const int low_a = p.lowptr[a];
  const int half = p.lowptr[a + n] - low_a;
  const int gbase = 2 * low_a;
  if (threadIdx.x == 0 && p.neighbors) p.neighbors[g] = 2 * (long long)half;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  for (int i = wave; i < n; i += nwaves) {
    int before = 0;
    const int row_base = p.row_ptr[a + i];
    for (int w = 0; w < W; ++w) {
      const u64 s = S[i * W + w];
      if ((s >> lane) & 1ull) {
        const int j = 64 * w + lane;
        const int slot = row_base + before + __popcll(s & bits_below(lane));
        const int rslot = p.row_ptr[a + j] + rank_in_row(S + j * W, i);
        int cid;
        if (j < i) cid = gbase + (p.lowptr[a + i] - low_a) + rank_in_row(LK + i * W, j);
        else       cid = gbase + half + (p.lowptr[a + j] - low_a) + rank_in_row(LK + j * W, i);
        // geometry exactly as painn.py:418-420,319-321: vec = pos[src]-pos[dst], d = sqrt(sum((pos[dst]-pos[src])^2))
        float d, rx, ry, rz;
        {
#pragma clang fp contract(off)
          const float* pi = sp + 3 * i; const float* pj = sp + 3 * j;
          float wx = pj[0] - pi[0], wy = pj[1] - pi[1], wz = pj[2] - pi[2];
          d = sqrt_rn(dist2_nofma(pi, pj));
          float c0 = (d <= 1e-6f) ? 1e-6f : 0.0f;
          float den = d + c0;
          rx = __fdiv_rn(wx, den); ry = __fdiv_rn(wy, den); rz = __fdiv_rn(wz, den);
        }
        p.col[slot] = a + j;
        p.dst[slot] = a + i;
        p.rev[slot] = rslot;
        p.geom[slot] = make_float4(rx, ry, rz, d);
        p.slot2canon[slot] = cid;
        if (p.c_src) {
          p.c_src[cid] = a + j;
          p.c_dst[cid] = a + i;
          p.c_dist[cid] = d;
          p.c_vec[3 * (long)cid + 0] = rx; p.c_vec[3 * (long)cid + 1] = ry; p.c_vec[3 * (long)cid + 2] = rz;
          p.id_swap[cid] = (j < i) ? cid + half : cid - half;
        }
      }
      before += __popcll(s);
    }
  }
}

// ---- host launchers ------------------------------------------------------------------------
static size_t graph_lds_bytes(int max_mol_atoms) {
  int W = (max_mol_atoms + 63) / 64;
  return (size_t)max_mol_atoms * W * 8 * 2 + (size_t)max_mol_atoms * 3 * 4;
}

int nq_graph_count_impl(const float* pos, const int* mol_ptr, int N, int B, int max_mol_atoms, float cutoff2, int K,
                        int* deg, int* lowdeg, int* row_ptr, int* lowptr, int* E_host, hipStream_t st) {
  NQ_PROF(st, "graph_count");
  if (max_mol_atoms > NQ_MAX_MOL_ATOMS)
    return nq_fail(NQ_ERR_MOL_TOO_LARGE, "molecule with %d atoms exceeds NQ_MAX_MOL_ATOMS=%d", max_mol_atoms, NQ_MAX_MOL_ATOMS);
  if (N <= 0 || B <= 0) return nq_fail(NQ_ERR_ARG, "empty batch (N=%d, B=%d)", N, B);
  size_t lds = graph_lds_bytes(max_mol_atoms);
  NQ_DYN_LDS(k_graph_count, lds);
  hipLaunchKernelGGL(k_graph_count, dim3(B), dim3(GRAPH_THREADS), lds, st, pos, mol_ptr, cutoff2, K, deg, lowdeg);
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_scan2, dim3(1), dim3(1024), 0, st, deg, lowdeg, N, row_ptr, lowptr);
  NQ_LAUNCH_CHECK();
  NQ_HIP(hipMemcpyAsync(E_host, row_ptr + N, sizeof(int), hipMemcpyDeviceToHost, st));
  NQ_HIP(hipStreamSynchronize(st));
  return NQ_OK;
}

int nq_graph_fill_impl(GraphFillArgs args, int B, int max_mol_atoms, hipStream_t st) {
  NQ_PROF(st, "graph_fill");
  size_t lds = graph_lds_bytes(max_mol_atoms);
  NQ_DYN_LDS(k_graph_fill, lds);
  hipLaunchKernelGGL(k_graph_fill, dim3(B), dim3(GRAPH_THREADS), lds, st, args);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
