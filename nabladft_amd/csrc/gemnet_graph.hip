// GemNet-OC graphs on the device, derived from ONE radius graph (the "a2a" graph: CSR by target atom, sources ascending, built by graph.hip).
//
// Replaces (reference, /root/reference/nablaDFT/gemnet_oc/):
//   gemnet_oc.py:892-958   get_graphs_and_indices: a2a graph -> main / a2ee2a / qint sub-graphs (subselect_graph :866-890, subselect_edges :777-826)
//   utils.py:408-500       get_max_neighbors_mask with enforce_max_strictly: per target atom, the K nearest of the edges inside the sub-graph's cutoff
//   gemnet_oc.py:694-775   symmetrize_edges of the main graph (keep source < target, add the flipped copies) and id_swap
//   utils.py:393-405       get_inner_idx (target_neighbor_idx = position inside the target's row: implicit in a CSR)
//   interaction_indices.py:13-282   triplet / mixed-triplet / quadruplet index lists -- NOT materialised here: with every graph stored as a CSR by target atom
//                          the interaction kernels (gemnet.hip) enumerate "the other in-edges of the same atom" directly; only the row offsets of the
//                          (qint edge, main in-edge of its source) pairs are produced (tin_ptr), because those rows carry features.
//
// Selection rule (utils.py:452-500, strict): inside one target's row, an edge is kept iff fewer than K eligible edges (distance <= cutoff) precede it in
// (distance, position) order.  torch.sort in the reference is not stable, so exact distance ties at the cap may resolve differently there; here the order
// is deterministic.  Everything is integer / compare work on <= 511 entries per row: one wavefront per atom, ballot + popcount compaction, no atomics.
#include "common.h"

#define GN_MAXDEG 512
typedef unsigned long long gn_u64;

struct GnGraphArgs {
  int N, E;
  const int* row_ptr; const int* col; const int* rev; float4* geom;         // a2a graph (geom is rewritten by k_gn_geom)
  const int* dst; const float* pos;
  float cut_m, cut_a, cut_q;
  int Km, Ka, Kq;
  unsigned char* flags;                                       // [E] bit0 main (directed, before symmetrisation), bit1 a2ee2a, bit2 qint
  int* degm; int* lowm; int* cnt_a; int* cnt_q; int* tin_atom;          // [N] each
  int* ptr_m; int* lowptr_m; int* ptr_a; int* ptr_q; int* tin_aptr;     // [N+1] each
  // fill outputs
  int* m_src; int* m_dst; int* m_rev; int* m_slot; float4* m_geom;      // main (symmetric): CSR by target, sources ascending
  int* a_src; int* a_dst; float4* a_geom; int* a_of_rev;                // a2ee2a; a_of_rev[a2a slot p] = a2ee2a edge (target(p) -> source(p)) or -1
  int* q_src; int* q_dst; float4* q_geom; int* tin_ptr;                 // qint; tin_ptr[q] = first row of the (q, main in-edges of source(q)) block
  int* mpos; int* apos;                                                 // [E] scratch: main / a2ee2a slot of an a2a slot, or -1
  int* tin_main; int* q_of_rev; int* qpos;                              // tin_main[row] = main slot of the row's in-edge; q_of_rev[a2a slot p] = qint edge (target(p) -> source(p)) or -1
};

__device__ __forceinline__ gn_u64 gn_below(int lane) { return lane ? (~0ull >> (64 - lane)) : 0ull; }

// Geometry as gemnet_oc.py:1324-1325,840 evaluates it on the CPU: d = torch.norm(pos[j] - pos[i]) -- ATen's float kernel accumulates x*x, fma(y, y, .),
// fma(z, z, .) and takes a correctly rounded square root (checked against torch 2.10 on 2056 edges: 0 mismatches; any other order: >= 144) -- and
// vector = -(pos[j] - pos[i]) / d.  Stored with graph.hip's sign (target -> source); k_gn_fill flips it.
__device__ __forceinline__ float gn_sqrt_rn(float x) {
  if (!(x > 0.0f)) return 0.0f;
  const float s = __builtin_sqrtf(x);
  const float r = fmaf(-s, s, x);
  return fmaf(r, 0.5f / s, s);
}
__global__ void k_gn_geom(GnGraphArgs g) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= g.E) return;
  const float* pj = g.pos + 3 * (long)g.col[p];
  const float* pi = g.pos + 3 * (long)g.dst[p];
  const float x = pj[0] - pi[0], y = pj[1] - pi[1], z = pj[2] - pi[2];
  float s = __fmul_rn(x, x);
  s = fmaf(y, y, s);
  s = fmaf(z, z, s);
  const float d = gn_sqrt_rn(s);
  g.geom[p] = make_float4(__fdiv_rn(x, d), __fdiv_rn(y, d), __fdiv_rn(z, d), d);
}

// one wavefront (= one workgroup) per target atom: the three keep bits of every edge of the row
__global__ __launch_bounds__(64) void k_gn_flags(GnGraphArgs g) {
  __shared__ float sd[GN_MAXDEG];
  const int i = blockIdx.x, lane = threadIdx.x;
  const int beg = g.row_ptr[i], deg = g.row_ptr[i + 1] - beg;
  for (int t = lane; t < deg; t += 64) sd[t] = g.geom[beg + t].w;
  __syncthreads();
  int na = 0, nq = 0;
  for (int t0 = 0; t0 < deg; t0 += 64) {
    const int t = t0 + lane;
    const bool valid = t < deg;
    const float d = valid ? sd[t] : 0.f;
    int rm = 0, ra = 0, rq = 0;
    for (int q = 0; q < deg; ++q) {
      const float dq = sd[q];
      const bool less = dq < d || (dq == d && q < t);
      rm += (less && dq <= g.cut_m); ra += (less && dq <= g.cut_a); rq += (less && dq <= g.cut_q);
    }
    const bool km = valid && d <= g.cut_m && rm < g.Km, ka = valid && d <= g.cut_a && ra < g.Ka, kq = valid && d <= g.cut_q && rq < g.Kq;
    if (valid) g.flags[beg + t] = (unsigned char)((km ? 1 : 0) | (ka ? 2 : 0) | (kq ? 4 : 0));
    na += __popcll(__ballot(ka));
    nq += __popcll(__ballot(kq));
  }
  if (lane == 0) { g.cnt_a[i] = na; g.cnt_q[i] = nq; }
}

// symmetrised main graph: row i = {j < i : (j -> i) kept} + {k > i : (i -> k) kept in row k}   (gemnet_oc.py:712-736: mask source < target, then the flips)
__device__ __forceinline__ bool gn_main_member(const GnGraphArgs& g, int i, int p, int src) {
  return src < i ? (g.flags[p] & 1) : (g.flags[g.rev[p]] & 1);
}

__global__ __launch_bounds__(64) void k_gn_maindeg(GnGraphArgs g) {
  const int i = blockIdx.x, lane = threadIdx.x;
  const int beg = g.row_ptr[i], end = g.row_ptr[i + 1];
  int n = 0, nl = 0;
  for (int p0 = beg; p0 < end; p0 += 64) {
    const int p = p0 + lane;
    bool mem = false, low = false;
    if (p < end) { const int src = g.col[p]; mem = gn_main_member(g, i, p, src); low = mem && src < i; }
    n += __popcll(__ballot(mem));
    nl += __popcll(__ballot(low));
  }
  if (lane == 0) { g.degm[i] = n; g.lowm[i] = nl; }
}

__global__ void k_gn_tin(GnGraphArgs g) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.N) return;
  int s = 0;
  for (int p = g.row_ptr[i]; p < g.row_ptr[i + 1]; ++p)
    if (g.flags[p] & 4) s += g.degm[g.col[p]];
  g.tin_atom[i] = s;
}

// exclusive scans of the five per-atom counters by one workgroup
struct GnScanArgs { const int* in[5]; int* out[5]; int n; };
__global__ __launch_bounds__(1024) void k_gn_scan(GnScanArgs a) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 0; k < 5; ++k) {
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < a.n; base += 1024) {
      const int i = base + threadIdx.x;
      const int v = i < a.n ? a.in[k][i] : 0;
      int s = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(s, off, 64); if (lane >= off) s += t; }
      if (lane == 63) wsum[wave] = s;
      __syncthreads();
      int p = carry;
      for (int w = 0; w < wave; ++w) p += wsum[w];
      if (i < a.n) a.out[k][i] = p + s - v;
      __syncthreads();
      if (threadIdx.x == 1023) carry = p + s;
      __syncthreads();
    }
    if (threadIdx.x == 0) a.out[k][a.n] = carry;
    __syncthreads();
  }
}

__global__ __launch_bounds__(64) void k_gn_fill(GnGraphArgs g) {
  const int i = blockIdx.x, lane = threadIdx.x;
  const int beg = g.row_ptr[i], end = g.row_ptr[i + 1];
  int om = g.ptr_m[i], oa = g.ptr_a[i], oq = g.ptr_q[i];
  for (int p0 = beg; p0 < end; p0 += 64) {
    const int p = p0 + lane;
    const bool valid = p < end;
    int src = 0; unsigned char f = 0; bool fm = false;
    float4 ge = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) {
      src = g.col[p]; f = g.flags[p]; fm = gn_main_member(g, i, p, src);
      const float4 r = g.geom[p];                 // r = (pos[src] - pos[i]) / d; GemNet's edge vector points from the source to the target (gemnet_oc.py:840)
      ge = make_float4(-r.x, -r.y, -r.z, r.w);
    }
    const gn_u64 bm = __ballot(fm), ba = __ballot(valid && (f & 2)), bq = __ballot(valid && (f & 4)), below = gn_below(lane);
    if (valid) {
      int pm = -1, pa = -1;
      if (fm) { pm = om + __popcll(bm & below); g.m_src[pm] = src; g.m_dst[pm] = i; g.m_slot[pm] = p; g.m_geom[pm] = ge; }
      if (f & 2) { pa = oa + __popcll(ba & below); g.a_src[pa] = src; g.a_dst[pa] = i; g.a_geom[pa] = ge; }
      int pq = -1;
      if (f & 4) { pq = oq + __popcll(bq & below); g.q_src[pq] = src; g.q_dst[pq] = i; g.q_geom[pq] = ge; }
      g.mpos[p] = pm; g.apos[p] = pa; g.qpos[p] = pq;
    }
    om += __popcll(bm); oa += __popcll(ba); oq += __popcll(bq);
  }
  int t = g.tin_aptr[i], k = g.ptr_q[i];
  for (int p = beg; p < end; ++p) {                  // wave-uniform walk over the row's qint edges
    if (!(g.flags[p] & 4)) continue;
    const int src = g.col[p], n = g.degm[src], base = g.ptr_m[src];
    if (lane == 0) g.tin_ptr[k] = t;
    for (int j = lane; j < n; j += 64) g.tin_main[t + j] = base + j;
    t += n; ++k;
  }
  if (lane == 0 && i == g.N - 1) g.tin_ptr[k] = t;
}

__global__ void k_gn_link(GnGraphArgs g) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= g.E) return;
  const int r = g.rev[p];
  const int pm = g.mpos[p];
  if (pm >= 0) g.m_rev[pm] = g.mpos[r];
  g.a_of_rev[p] = g.apos[r];
  g.q_of_rev[p] = g.qpos[r];
}

static int nq_gn_graph_count_impl(GnGraphArgs g, int* totals_host, hipStream_t st) {
  NQ_PROF(st, "gn_graph_count");
  if (g.N <= 0 || g.E <= 0) return nq_fail(NQ_ERR_NO_EDGES, "GemNet-OC graphs: empty a2a graph (N=%d, E=%d)", g.N, g.E);
  hipLaunchKernelGGL(k_gn_geom, dim3(nq_cdiv(g.E, 256)), dim3(256), 0, st, g);
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_gn_flags, dim3(g.N), dim3(64), 0, st, g);
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_gn_maindeg, dim3(g.N), dim3(64), 0, st, g);
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_gn_tin, dim3(nq_cdiv(g.N, 256)), dim3(256), 0, st, g);
  NQ_LAUNCH_CHECK();
  GnScanArgs s;
  s.n = g.N;
  s.in[0] = g.degm; s.in[1] = g.lowm; s.in[2] = g.cnt_a; s.in[3] = g.cnt_q; s.in[4] = g.tin_atom;
  s.out[0] = g.ptr_m; s.out[1] = g.lowptr_m; s.out[2] = g.ptr_a; s.out[3] = g.ptr_q; s.out[4] = g.tin_aptr;
  hipLaunchKernelGGL(k_gn_scan, dim3(1), dim3(1024), 0, st, s);
  NQ_LAUNCH_CHECK();
  int* src[4] = {g.ptr_m + g.N, g.ptr_a + g.N, g.ptr_q + g.N, g.tin_aptr + g.N};
  for (int k = 0; k < 4; ++k) NQ_HIP(hipMemcpyAsync(totals_host + k, src[k], sizeof(int), hipMemcpyDeviceToHost, st));
  NQ_HIP(hipStreamSynchronize(st));
  return NQ_OK;
}

static int nq_gn_graph_fill_impl(GnGraphArgs g, hipStream_t st) {
  NQ_PROF(st, "gn_graph_fill");
  hipLaunchKernelGGL(k_gn_fill, dim3(g.N), dim3(64), 0, st, g);
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_gn_link, dim3(nq_cdiv(g.E, 256)), dim3(256), 0, st, g);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

// public layout (include/nablaq.h: nq_gn_graphs)
struct nq_gn_graphs_c {
  int32_t N, E, k_main, k_aea, k_qint, reserved;
  double cutoff_main, cutoff_aea, cutoff_qint;
  const int32_t* row_ptr; const int32_t* col; const int32_t* rev; const int32_t* dst; const float* pos; float* geom;
  uint8_t* flags;
  int32_t* degm; int32_t* lowm; int32_t* cnt_a; int32_t* cnt_q; int32_t* tin_atom;
  int32_t* ptr_m; int32_t* lowptr_m; int32_t* ptr_a; int32_t* ptr_q; int32_t* tin_aptr;
  int32_t* m_src; int32_t* m_dst; int32_t* m_rev; int32_t* m_slot; float* m_geom;
  int32_t* a_src; int32_t* a_dst; float* a_geom; int32_t* a_of_rev;
  int32_t* q_src; int32_t* q_dst; float* q_geom; int32_t* tin_ptr;
  int32_t* mpos; int32_t* apos;
  int32_t* tin_main; int32_t* q_of_rev; int32_t* qpos;
};
static GnGraphArgs gn_args(const nq_gn_graphs_c* c) {
  GnGraphArgs g;
  g.N = c->N; g.E = c->E;
  g.row_ptr = c->row_ptr; g.col = c->col; g.rev = c->rev; g.geom = (float4*)c->geom; g.dst = c->dst; g.pos = c->pos;
  g.cut_m = (float)c->cutoff_main; g.cut_a = (float)c->cutoff_aea; g.cut_q = (float)c->cutoff_qint;
  g.Km = c->k_main; g.Ka = c->k_aea; g.Kq = c->k_qint;
  g.flags = c->flags;
  g.degm = c->degm; g.lowm = c->lowm; g.cnt_a = c->cnt_a; g.cnt_q = c->cnt_q; g.tin_atom = c->tin_atom;
  g.ptr_m = c->ptr_m; g.lowptr_m = c->lowptr_m; g.ptr_a = c->ptr_a; g.ptr_q = c->ptr_q; g.tin_aptr = c->tin_aptr;
  g.m_src = c->m_src; g.m_dst = c->m_dst; g.m_rev = c->m_rev; g.m_slot = c->m_slot; g.m_geom = (float4*)c->m_geom;
  g.a_src = c->a_src; g.a_dst = c->a_dst; g.a_geom = (float4*)c->a_geom; g.a_of_rev = c->a_of_rev;
  g.q_src = c->q_src; g.q_dst = c->q_dst; g.q_geom = (float4*)c->q_geom; g.tin_ptr = c->tin_ptr;
  g.mpos = c->mpos; g.apos = c->apos;
  g.tin_main = c->tin_main; g.q_of_rev = c->q_of_rev; g.qpos = c->qpos;
  return g;
}

extern "C" {
int nq_gn_graph_count(const void* graphs, int32_t max_degree, int32_t* totals_host, void* stream) {
  if (!graphs || !totals_host) return nq_fail(NQ_ERR_ARG, "null argument");
  if (max_degree > GN_MAXDEG) return nq_fail(NQ_ERR_MOL_TOO_LARGE, "a2a in-degree %d exceeds %d", max_degree, GN_MAXDEG);
  return nq_gn_graph_count_impl(gn_args((const nq_gn_graphs_c*)graphs), totals_host, (hipStream_t)stream);
}
int nq_gn_graph_fill(const void* graphs, void* stream) {
  if (!graphs) return nq_fail(NQ_ERR_ARG, "null argument");
  return nq_gn_graph_fill_impl(gn_args((const nq_gn_graphs_c*)graphs), (hipStream_t)stream);
}
}
