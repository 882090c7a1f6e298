// rbf_proj weight / bias gradient of the PaiNN dual-reverse sweep, molecule per workgroup (round 5).
// Reference semantics: painn.py:475-509 (rbfh = rbf_proj(edge_rbf); x = xh[j] * rbfh) differentiated twice -- the adjoints of
// phi = rbf_proj(rho(d)) and of its tangent psi * t_d, contracted with the 13-tap Gaussian window:
//     gWr[c][k0_p + t] += gphi_p[c] rho_t(d_p) + gpsi_p[c] drho_t(d_p),   p = undirected pair, both directions summed.
// Rounds 1-4 wrote gphi / gpsi per pair to HBM from the dual-reverse message kernel (2 x 1.5 kB per pair, 2.5 GB per layer at 2048
// conformers) and re-read them in k0 order (k_gwr_sorted).  gphi / gpsi do NOT depend on the filter, only on node rows:
//     direction s -> t:  gm_b = A_t.v_s + T_t.tv_s, gtm_b = T_t.v_s, gm_c = A_t.r + T_t.tr, gtm_c = T_t.r,
//                        gphi = (gma_t xa_s + gtma_t txa_s,  gm_b xb_s + gtm_b txb_s,  gm_c xc_s + gtm_c txc_s),  gpsi = t_d (gtma_t xa_s, gtm_b xb_s, gtm_c xc_s)
// (A, gma, T, gtma: adjoint rows of the target; x, tx, v, tv: primal / tangent rows of the source).  So here a workgroup stages the 20 rows
// of ONE molecule for a 32-channel slice in LDS with coalesced loads (2.5 kB per atom) and every pair is recomputed where it is consumed:
//   * wavefront w keeps GM_WMAX + 12 accumulator rows (window starts [wlo[w], wlo[w] + GM_WMAX), placed on the w-th quantile range of the batch's k0
//     histogram) for the WHOLE launch as the C/D registers of the matrix core -- no atomics, no LDS accumulator, no sliding; neighbouring windows
//     overlap, and every molecule's k0-sorted pair list is cut into GM_NW equal segments that respect them: all wavefronts reach the barrier together;
//   * one wavefront iteration = one pair: lanes 0-31 the direction n -> k, lanes 32-63 the direction k -> n, one channel per lane; the 20 operands are
//     five conflict-free ds_read_b128; one v_permlane32_swap + add per filter part leaves gphi (both directions summed) in lanes 0-31 and gpsi in
//     lanes 32-63 -- the B operand [K = {phi, psi}][N = 32 channels] of v_mfma_f32_32x32x2_f32.  Its A operand [M = 32 rows][K] is the pair's window
//     record shifted to the wavefront's rows: lane (j, half) loads rho / drho tap j - (k0 - wlo) (zero outside the 13 taps) with ONE coalesced
//     256-byte load (k_pair_arec prepares it once per step), requested four pairs ahead.  Row 31 carries the bias multiplier (beta, beta'), so the bias gradient rides in the same product.
//     The rank-2 update acc[j][c] += rho_j gphi[c] + drho_j gpsi[c] is then one matrix-core instruction per filter part (exact f32, fixed order);
//   * geometry / tangent scalars of a pair come as vector loads too (32 records per coalesced chunk load into a 1-kB LDS ring private to the wavefront,
//     broadcast ds_read_b128 + v_readfirstlane when consumed): no scalar-memory latency in the loop
//     (the first version of this kernel read the 32-dword record with s_load per pair and ran at HBM latency: 3.7 ms per launch);
//   * the pair lists per (molecule, wavefront) are built once per step (the windows depend on the geometry only), sorted by window start, slot-ascending,
//     so the summation order is fixed: results are bitwise reproducible;
//   * per-workgroup partial rows (one flush per launch) are summed in workgroup order by k_gwr_mol_reduce.
// HBM traffic per launch: 20 rows x N x F x 4 B read once (0.88 GB at 2048 conformers) instead of 2.5 GB written + 2.5 GB read.
#include "common.h"
#include "lanes.h"
#include <type_traits>

#ifndef GM_THREADS
#define GM_THREADS 1024   // 16 wavefronts = 4 per SIMD (8 / 12 / 16 measured within 3 %: profiles/r05_gwr_mol_wave_count_variants.txt); the staging below deals its blocks to 15 of them
#endif
#define GM_NW (GM_THREADS / 64)     // wavefronts per workgroup = owners of window-start ranges
#define GM_ROWS 32                  // rows of the matrix-core tile: 31 window rows + the bias row
#define GM_WMAX (GM_ROWS - 1 - (FWIN - 1))   // window starts per wavefront at most (19): row off + 12 <= 30
#define GM_CH 32                    // channels per slice
#define GM_ATOM_FLOATS (20 * GM_CH) // LDS floats per atom: 5 blocks [32 channels][4 rows]
#define GM_PART_FLOATS (GM_ROWS * 3 * GM_CH)   // per (workgroup, wavefront): [row][part][channel]
#define GM_MAX_BINS 128
#ifndef GM_ABLATE
#define GM_ABLATE 0   // development only (scripts/variants_r05.sh; results are wrong, timing only): 1 no pair loop, 2 no staging, 3 no matrix-core updates, 4 no record loads, 5 no reduce kernel; bit flags from 16: 16 no permlane swaps, 32 no LDS operand loads, 64 no per-pair arithmetic, 128 no barriers
#endif

typedef float f4 __attribute__((ext_vector_type(4)));
static_assert(GM_THREADS == 1024 || GM_ABLATE != 0, "the staging of k_gwr_mol assigns LDS blocks to wavefronts 0-14 of 16 (other counts: timing builds only)");

// ---- once per step: owners of the window starts and the per-(molecule, wavefront) pair lists -----------------------------------
__global__ __launch_bounds__(256) void k_pair_hist(const float* __restrict__ RW, const int* __restrict__ dst, const int* __restrict__ col, int E,
                                                    int* __restrict__ hist) {
  __shared__ int h[GM_MAX_BINS];
  if (threadIdx.x < GM_MAX_BINS) h[threadIdx.x] = 0;
  __syncthreads();
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x)
    if (col[e] < dst[e]) atomicAdd(&h[min(GM_MAX_BINS - 1, __float_as_int(RW[(long)e * RW_STRIDE + 13]))], 1);   // integer atomics: order-independent
  __syncthreads();
  if (threadIdx.x < GM_MAX_BINS && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

// wlo[w] = first accumulator row (window start) of wavefront w; its GM_WMAX rows are placed around the w-th 1/GM_NW quantile range of the batch's
// window-start histogram, neighbouring windows overlap or touch (wlo[w+1] <= wlo[w] + GM_WMAX) and together cover [bmin, bmax].  The windows overlap by
// design: which wavefront takes a pair is decided per molecule (k_pair_sched) so that every wavefront gets the same number of pairs of EVERY molecule --
// with one owner per window start (first build) the per-molecule barrier waited for the wavefront of the densest range: 55 % of the wave cycles.
__global__ void k_pair_windows(const int* __restrict__ hist, int nbins, int* __restrict__ wlo) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int bmin = nbins, bmax = -1;
  long total = 0;
  for (int b = 0; b < nbins; ++b) if (hist[b] > 0) { bmin = min(bmin, b); bmax = b; total += hist[b]; }
  if (bmax < 0) { bmin = 0; bmax = 0; }
  int gq[GM_NW + 1];
  {
    long cum = 0;
    int w = 1;
    gq[0] = bmin;
    for (int b = bmin; b <= bmax && w < GM_NW; ++b) {
      cum += hist[b];
      while (w < GM_NW && cum * GM_NW >= total * w) gq[w++] = b;   // bin that holds the w-th quantile
    }
    for (; w < GM_NW; ++w) gq[w] = bmax;
    gq[GM_NW] = bmax + 1;
  }
  int lo[GM_NW];
  for (int w = 0; w < GM_NW; ++w) {
    const int hi = max(gq[w + 1] - 1, gq[w]), slack = max(0, GM_WMAX - (hi - gq[w] + 1));
    lo[w] = max(0, gq[w] - slack / 2);
  }
  lo[0] = min(lo[0], bmin);
  for (int w = 1; w < GM_NW; ++w) lo[w] = min(max(lo[w], lo[w - 1]), lo[w - 1] + GM_WMAX);
  lo[GM_NW - 1] = max(lo[GM_NW - 1], bmax - (GM_WMAX - 1));
  for (int w = GM_NW - 2; w >= 0; --w) lo[w] = max(lo[w], lo[w + 1] - GM_WMAX);
  for (int w = 0; w < GM_NW; ++w) wlo[w] = lo[w];
  wlo[GM_NW] = GM_MAX_BINS;
}

// One wavefront per molecule: the molecule's pairs sorted (stably, by slot) by window start k0, then cut into GM_NW consecutive segments of (nearly) equal
// length -- segment w goes to wavefront w and must fit its rows: it takes every pair with k0 < wlo[w+1] (the next window cannot hold them) and otherwise
// fills up to ceil(pairs / GM_NW) with pairs of k0 < wlo[w] + GM_WMAX.  sched_ptr[m][w] = first entry of segment w relative to pb = lowptr[a0].
// sched[pb + i] = {slot of the lower edge (row n, source k < n), row offset (k0 - wlo[w]) << 26 | (n - a0) << 13 | (k - a0)}.
// Ranks come from ballots over the distinct keys of a 64-slot chunk (no atomics): the order is a function of the geometry only.
template <bool FILL>
__device__ __forceinline__ void pair_sched_pass(const NqGraphView& g, const int* __restrict__ dst, const float* __restrict__ RW, const int* __restrict__ wlo,
                                                const int* cut, int a0, int s0, int s1, int pb, int lane, int* cnt, int2* __restrict__ sched) {
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int c0 = s0; c0 < s1; c0 += 64) {
    const int s = c0 + lane;
    int key = -1, n = 0, k = 0;
    if (s < s1) {
      n = dst[s]; k = g.col[s];
      if (k < n) key = min(GM_MAX_BINS - 1, __float_as_int(RW[(long)s * RW_STRIDE + 13]));
    }
    unsigned long long todo = __ballot(key >= 0);
    while (todo) {   // one round per distinct key of the chunk
      const int lead = __builtin_ctzll(todo);
      const int kk = __builtin_amdgcn_readlane(key, lead);
      const unsigned long long mask = __ballot(key == kk);
      if (FILL && key == kk) {
        const int pos = cnt[kk] + __popcll(mask & lt);
        int w = 0;
#pragma unroll
        for (int v = 1; v < GM_NW; ++v) w += pos >= cut[v];
        sched[pb + pos] = make_int2(s, ((kk - wlo[w]) << 26) | ((n - a0) << 13) | (k - a0));
      }
      __builtin_amdgcn_wave_barrier();
      if (lane == lead) cnt[kk] += __popcll(mask);
      __builtin_amdgcn_wave_barrier();
      todo &= ~mask;
    }
  }
}
__global__ __launch_bounds__(64) void k_pair_sched(NqGraphView g, const int* __restrict__ dst, const float* __restrict__ RW, const int* __restrict__ wlo,
                                                    int2* __restrict__ sched, int* __restrict__ sched_ptr) {
  __shared__ int cnt[GM_MAX_BINS + 1];
  __shared__ int cut[GM_NW + 1];
  const int m = blockIdx.x, lane = threadIdx.x;
  const int a0 = g.mol_ptr[m], a1 = g.mol_ptr[m + 1];
  const int s0 = g.row_ptr[a0], s1 = g.row_ptr[a1], pb = g.lowptr[a0];
  cnt[lane] = 0; cnt[lane + 64] = 0;
  __syncthreads();
  pair_sched_pass<false>(g, dst, RW, wlo, cut, a0, s0, s1, pb, lane, cnt, sched);
  __syncthreads();
  if (lane == 0) {
    int run = 0;   // exclusive scan over the window starts (<= 128 values): cnt[b] = pairs with k0 < b
    for (int b = 0; b < GM_MAX_BINS; ++b) { const int v = cnt[b]; cnt[b] = run; run += v; }
    cnt[GM_MAX_BINS] = run;
    const int np = run, target = (np + GM_NW - 1) / GM_NW;
    int cur = 0;
    for (int w = 0; w < GM_NW; ++w) {
      cut[w] = cur;
      const int forced_end = w == GM_NW - 1 ? np : cnt[min(GM_MAX_BINS, wlo[w + 1])];
      const int optional_end = min(cur + target, cnt[min(GM_MAX_BINS, wlo[w] + GM_WMAX)]);
      cur = max(cur, max(forced_end, optional_end));
    }
    cut[GM_NW] = np;
  }
  __syncthreads();
  if (lane <= GM_NW) sched_ptr[(long)m * (GM_NW + 1) + lane] = cut[lane];
  pair_sched_pass<true>(g, dst, RW, wlo, cut, a0, s0, s1, pb, lane, cnt, sched);
}

// Expanded A operands, once per step (the window records depend on the geometry only): PA[pair][lane] = the value lane (row j = lane & 31, K index = lane >> 5)
// feeds to the matrix core for this pair: rho (K = 0) / drho (K = 1) tap j - off of the window record, 0 outside the 13 taps, and in row 31 the bias
// multiplier / its derivative.  One coalesced 256-byte load per pair in the gradient kernel instead of ~25 instructions of unpacking per pair, slice and layer.
__global__ __launch_bounds__(256) void k_pair_arec(const int2* __restrict__ sched, const float* __restrict__ RW, int npairs, float* __restrict__ PA) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int pr = (int)(idx >> 6), lane = (int)(idx & 63);
  if (pr >= npairs) return;
  const int2 en = sched[pr];
  const int off = (int)((unsigned)en.y >> 26), j = lane & 31, half = lane >> 5;
  const int t = j - off;
  const bool valid = ((unsigned)t < (unsigned)FWIN) | (j == 31);
  const int tap = j == 31 ? 14 : min(max(t, 0), FWIN - 1);
  const float v = RW[(long)en.x * RW_STRIDE + half * 16 + tap];
  PA[idx] = valid ? v : 0.f;
}
// Packed geometry records, once per backward sweep (the tangents t_d, t_r follow the force seeds): PG[pair] = {gx, gy, gz, t_d, t_r0, t_r1, t_r2, n << 13 | k}
// in schedule order, so the gradient kernel reads ONE sequential scalar stream per wavefront.
__global__ __launch_bounds__(256) void k_pair_grec(const int2* __restrict__ sched, const float4* __restrict__ geom, const float* __restrict__ TD,
                                                    const float* __restrict__ TR, int npairs, float4* __restrict__ PG) {
  const int pr = blockIdx.x * blockDim.x + threadIdx.x;
  if (pr >= npairs) return;
  const int2 en = sched[pr];
  const float4 gm = geom[en.x];
  PG[2 * (long)pr] = make_float4(gm.x, gm.y, gm.z, TD[en.x]);
  PG[2 * (long)pr + 1] = make_float4(TR[3 * (long)en.x], TR[3 * (long)en.x + 1], TR[3 * (long)en.x + 2], __int_as_float(en.y & 0x3ffffff));
}

// ---- the gradient kernel ----------------------------------------------------------------------------------------------------------
struct GwrMolArgs {
  NqGraphView g; int F; int nslices; int groups; int max_atoms;
  const float* XH; const float* V; const float* TXH; const float* TV;      // primal / tangent rows of the layer input side  [N][3F]
  const float* GX; const float* GV; const float* GTX; const float* GTV;    // adjoints of x_msg / vec_msg and of their tangents  [N][F], [N][3F]
  const int* sched_ptr;
  float* part;                                                              // [groups][nslices][GM_NW][GM_PART_FLOATS]
};

typedef float f16v __attribute__((ext_vector_type(16)));

// x and y hold one value per direction (lanes 0-31: n -> k, lanes 32-63: k -> n).  Returns the direction sum of x in lanes 0-31 and of y in lanes 32-63:
// v_permlane32_swap exchanges the upper half of its first operand with the lower half of its second.
__device__ __forceinline__ float gm_pair_sum(float x, float y) {
  if (GM_ABLATE & 16) return x + y;
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__global__ __launch_bounds__(GM_THREADS) void k_gwr_mol(GwrMolArgs q, const float* __restrict__ PA, const float* __restrict__ PG) {
  extern __shared__ __attribute__((aligned(16))) float rows[];   // [atom][5 blocks][32 channels][4]
  const int F = q.F, F3 = 3 * q.F;
  // Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8).  The nslices workgroups of one molecule group read the same per-pair records at about the
  // same time: they are placed on ONE XCD (one HBM fetch, the others hit that XCD's L2) when the grid allows it.
  int slice = blockIdx.x % q.nslices, group = blockIdx.x / q.nslices;
  if (q.groups % 8 == 0) { const int x = blockIdx.x & 7, r = blockIdx.x >> 3; slice = r % q.nslices; group = (r / q.nslices) * 8 + x; }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int half = lane >> 5, c = lane & 31;
  const int cb = slice * GM_CH;                  // first channel of the slice
  f16v acc0, acc1, acc2;                         // matrix-core accumulators of the three filter parts: [32 rows][32 channels] each
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = acc2[r] = 0.f;
  const float sgn = half ? 1.0f : -1.0f;         // unit vector of the lane's direction: n -> k is -geom, k -> n is +geom (same for its tangent)
  const unsigned hmask = half ? 0xffffffffu : 0u;
  const unsigned lbase = (unsigned)(c * 16);     // byte offset of the lane's channel inside a [32 channels][4] block
  float* const gring = rows + (size_t)q.max_atoms * GM_ATOM_FLOATS + wave * 256;   // this wavefront's 32 geometry records (1 kB)

  for (int m = group; m < q.g.B; m += q.groups) {
    const int a0 = __builtin_amdgcn_readfirstlane(q.g.mol_ptr[m]), na = __builtin_amdgcn_readfirstlane(q.g.mol_ptr[m + 1]) - a0;
    if (!(GM_ABLATE & 128)) __syncthreads();                             // every wavefront is done with the previous molecule's rows
    // ---- stage the 20 rows of the molecule: LDS block b (four rows, transposed to [channel][4 rows]) is filled by the wavefronts b, b + 5, b + 10;
    //      task = (atom, channel quad): four 16-byte loads (128 contiguous bytes per row and eight lanes), four ds_write_b128 ----
    if ((GM_ABLATE & 15) != 2 && wave < 15) {
      const int blk = wave % 5, sub = wave / 5;
      const float* s0; const float* s1; const float* s2; const float* s3;
      int st3 = F3;   // row stride of the fourth source (F for the scalar adjoints GX / GTX)
      switch (blk) {
        case 0: s0 = q.XH; s1 = q.XH + F; s2 = q.XH + 2 * F; s3 = q.TXH; break;
        case 1: s0 = q.TXH + F; s1 = q.TXH + 2 * F; s2 = q.V; s3 = q.V + F; break;
        case 2: s0 = q.V + 2 * F; s1 = q.TV; s2 = q.TV + F; s3 = q.TV + 2 * F; break;
        case 3: s0 = q.GV; s1 = q.GV + F; s2 = q.GV + 2 * F; s3 = q.GX; st3 = F; break;
        default: s0 = q.GTV; s1 = q.GTV + F; s2 = q.GTV + 2 * F; s3 = q.GTX; st3 = F; break;
      }
      for (int id = sub * 64 + lane; id < na * 8; id += 192) {
        const int at = id >> 3, qd = id & 7;
        const long n = a0 + at, o = n * F3 + cb + 4 * qd;
        const f4 r0 = *reinterpret_cast<const f4*>(s0 + o), r1 = *reinterpret_cast<const f4*>(s1 + o), r2 = *reinterpret_cast<const f4*>(s2 + o);
        const f4 r3 = *reinterpret_cast<const f4*>(s3 + n * st3 + cb + 4 * qd);
        float* d = rows + at * GM_ATOM_FLOATS + blk * (GM_CH * 4) + (4 * qd) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f4*>(d + i * 4) = f4{r0[i], r1[i], r2[i], r3[i]};
      }
    }
    if (!(GM_ABLATE & 128)) __syncthreads();
    // ---- this wavefront's pairs of the molecule: a contiguous run [p0, p1) of the schedule ----
    // Every per-pair record is a sequential stream per wavefront, streamed once per launch, i.e. every access misses to HBM (measured: an empty pair
    // loop with a 4-deep register ring took 400 ns per pair = latency / depth).  So: the geometry records of 32 pairs at a time travel through a 1-kB
    // LDS ring private to the wavefront (one coalesced 16-byte load per lane, requested one chunk = 32 pairs ahead; consumed as two broadcast
    // ds_read_b128: no scalar-memory waits mixed into lgkmcnt), the A operands through a register ring four pairs deep (R0-R3 below).
    const int pb = __builtin_amdgcn_readfirstlane(q.g.lowptr[a0]);
    const int* sp = q.sched_ptr + (long)m * (GM_NW + 1) + wave;
    const int p0 = pb + __builtin_amdgcn_readfirstlane(sp[0]), p1 = pb + ((GM_ABLATE & 15) == 1 ? __builtin_amdgcn_readfirstlane(sp[0]) : __builtin_amdgcn_readfirstlane(sp[1]));
    if (p0 < p1) {
      struct Ops { f4 P0, P1, P2, A, T; };
      struct Geo { f4 g0, g1; };   // {gx, gy, gz, t_d}, {t_r0, t_r1, t_r2, n << 13 | k}: the same value in every lane (broadcast LDS reads)
      const int last = p1 - 1;
      const f4* PG4 = reinterpret_cast<const f4*>(PG);
      auto a_load = [&](int pr) __attribute__((always_inline)) -> float {
        return (GM_ABLATE & 15) == 4 ? 1.f : PA[(long)min(pr, last) * 64 + lane];   // past the end the last pair is re-fetched: no branches in the pipeline
      };
      auto chunk_load = [&](int c0) __attribute__((always_inline)) -> f4 {   // lane L: float4 number L of the 32 records starting at pair c0
        return PG4[min(2 * (long)c0 + lane, 2 * (long)last + 1)];
      };
      auto geo_read = [&](Geo& o, int i) __attribute__((always_inline)) {       // record i of the wavefront's ring
        o.g0 = *reinterpret_cast<const f4*>(gring + (i & 31) * 8);
        o.g1 = *reinterpret_cast<const f4*>(gring + (i & 31) * 8 + 4);
      };
      // One pair.  Roles rotate statically over 12 steps per trip (A ring of 4, three scalar geometry sets, two LDS operand sets: no register copies):
      //   consumes  the A register `ra` (refilled 4 pairs ahead), the geometry scalars `gc` and the operand set `cur` (requested one step ago);
      //   requests  the operands of pair i + 1 into `nxt` (atom indices from `gn`) and the geometry record of pair i + 2 (broadcast LDS reads -> v_readfirstlane
      //             at the END of the step, when the data has arrived, into `gl`).
      struct GeoS { float gx, gy, gz, td, t0, t1, t2; unsigned nk; };
      auto to_scalar = [&](GeoS& o, const Geo& v) __attribute__((always_inline)) {
        o.gx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.g0[0]))); o.gy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.g0[1])));
        o.gz = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.g0[2]))); o.td = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.g0[3])));
        o.t0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.g1[0]))); o.t1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.g1[1])));
        o.t2 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.g1[2]))); o.nk = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(v.g1[3]));
      };
      auto ops_load = [&](Ops& o, unsigned nk) __attribute__((always_inline)) {
        const unsigned no = (nk >> 13) * (GM_ATOM_FLOATS * 4), ko = (nk & 0x1fff) * (GM_ATOM_FLOATS * 4);
        // operands: primal / tangent rows of the lane's SOURCE atom (n for lanes 0-31, k for 32-63), adjoint rows of its TARGET atom (k / n)
        const unsigned x = (no ^ ko) & hmask;
        const char* ps = reinterpret_cast<const char*>(rows) + ((no ^ x) + lbase);
        const char* pt = reinterpret_cast<const char*>(rows) + ((ko ^ x) + lbase);
        if (GM_ABLATE & 32) { const float z = __uint_as_float((no ^ x) + lbase) ; o.P0 = o.P1 = o.P2 = f4{z, z, 1.f, 2.f}; o.A = o.T = f4{1.f, z, __uint_as_float((ko ^ x) + lbase), 3.f}; return; }
        o.P0 = *reinterpret_cast<const f4*>(ps);                          // xa xb xc txa
        o.P1 = *reinterpret_cast<const f4*>(ps + GM_CH * 16);             // txb txc v0 v1
        o.P2 = *reinterpret_cast<const f4*>(ps + 2 * GM_CH * 16);         // v2 tv0 tv1 tv2
        o.A = *reinterpret_cast<const f4*>(pt + 3 * GM_CH * 16);          // A0 A1 A2 gma
        o.T = *reinterpret_cast<const f4*>(pt + 4 * GM_CH * 16);          // T0 T1 T2 gtma
      };
      auto step = [&](float& ra, const GeoS& gc, const GeoS& gn, GeoS& gl, const Ops& cur, Ops& nxt, int pr, int i, bool live) __attribute__((always_inline)) {
        const float a = live ? ra : 0.f;
        ops_load(nxt, gn.nk);
        Geo gv;
        geo_read(gv, i + 2);
        __builtin_amdgcn_sched_barrier(0);
        const float gx = gc.gx, gy = gc.gy, gz = gc.gz, td = gc.td, t0 = gc.t0, t1 = gc.t1, t2 = gc.t2;
        const f4 P0 = cur.P0, P1 = cur.P1, P2 = cur.P2, A = cur.A, T = cur.T;
        const float xa = P0[0], xb = P0[1], xc = P0[2], txa = P0[3], txb = P1[0], txc = P1[1], v0 = P1[2], v1 = P1[3], v2 = P2[0], tv0 = P2[1],
                    tv1 = P2[2], tv2 = P2[3];
        float ga, gb, gcc, ha, hb, hc;
        if (GM_ABLATE & 64) { ga = xa + gx; gb = xb + txa; gcc = xc + A[0]; ha = txb + T[0]; hb = v2 + gy + td + t0; hc = tv2 + v1 + t1 + t2 + gz + sgn; }
        else {
        const float gmb = A[0] * v0 + A[1] * v1 + A[2] * v2 + (T[0] * tv0 + T[1] * tv1 + T[2] * tv2);
        const float gtmb = T[0] * v0 + T[1] * v1 + T[2] * v2;
        const float gmc = sgn * ((A[0] * gx + A[1] * gy + A[2] * gz) + (T[0] * t0 + T[1] * t1 + T[2] * t2));
        const float gtmc = sgn * (T[0] * gx + T[1] * gy + T[2] * gz);
        const float gma = A[3], gtma = T[3];
        ga = gma * xa + gtma * txa; gb = gmb * xb + gtmb * txb; gcc = gmc * xc + gtmc * txc;
        ha = gtma * xa * td; hb = gtmb * xb * td; hc = gtmc * xc * td;
        }
        // both directions summed: lanes 0-31 <- gphi(n->k) + gphi(k->n), lanes 32-63 <- gpsi(n->k) + gpsi(k->n)
        const float ba = gm_pair_sum(ga, ha), bb = gm_pair_sum(gb, hb), bc = gm_pair_sum(gcc, hc);
        if ((GM_ABLATE & 15) == 3) { acc0[0] += a * ba; acc1[0] += a * bb; acc2[0] += a * bc; }
        else if (live) {   // wave-uniform: the padding steps of the last trip of a segment skip the matrix core (they were 13 % of its instructions)
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, ba, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bc, acc2, 0, 0, 0);
        }
        ra = a_load(pr + 4);
        to_scalar(gl, gv);
        // Without a side-effecting instruction between the prefetches and the end of the loop body, instcombine rewrites "phi of loads" into "load of a phi of
        // addresses", i.e. it sinks every prefetch to the top of the iteration that consumes it (first build: one exposed global latency per pair)
        __builtin_amdgcn_sched_barrier(0);
      };
      float R0 = a_load(p0), R1 = a_load(p0 + 1), R2 = a_load(p0 + 2), R3 = a_load(p0 + 3);
      f4 gq = chunk_load(p0);
      // chunks of 24 pairs (two trips of 12 steps) out of the 32 records the ring holds: the look-ahead of two records never leaves the ring
      for (int c0 = p0; c0 < p1; c0 += 24) {
        *reinterpret_cast<f4*>(gring + lane * 4) = gq;   // records of pairs [c0, c0 + 32): ordered behind this wavefront's reads of the previous chunk
        gq = chunk_load(c0 + 24);
        __builtin_amdgcn_sched_barrier(0);
        const int cn = min(24, p1 - c0);
        GeoS G0, G1, G2;
        Ops L0, L1;
        { Geo v0, v1; geo_read(v0, 0); geo_read(v1, 1); to_scalar(G0, v0); to_scalar(G1, v1); }
        ops_load(L0, G0.nk);
#pragma nounroll
        for (int i = 0; i < cn; i += 12) {
          const int pr = c0 + i;
          step(R0, G0, G1, G2, L0, L1, pr, i, true);
          step(R1, G1, G2, G0, L1, L0, pr + 1, i + 1, i + 1 < cn);
          step(R2, G2, G0, G1, L0, L1, pr + 2, i + 2, i + 2 < cn);
          step(R3, G0, G1, G2, L1, L0, pr + 3, i + 3, i + 3 < cn);
          if (i + 4 >= cn) break;
          step(R0, G1, G2, G0, L0, L1, pr + 4, i + 4, true);
          step(R1, G2, G0, G1, L1, L0, pr + 5, i + 5, i + 5 < cn);
          step(R2, G0, G1, G2, L0, L1, pr + 6, i + 6, i + 6 < cn);
          step(R3, G1, G2, G0, L1, L0, pr + 7, i + 7, i + 7 < cn);
          if (i + 8 >= cn) break;
          step(R0, G2, G0, G1, L0, L1, pr + 8, i + 8, true);
          step(R1, G0, G1, G2, L1, L0, pr + 9, i + 9, i + 9 < cn);
          step(R2, G1, G2, G0, L0, L1, pr + 10, i + 10, i + 10 < cn);
          step(R3, G2, G0, G1, L1, L0, pr + 11, i + 11, i + 11 < cn);
        }
      }
    }
  }
  // ---- one flush per launch.  C/D layout of the 32x32 tile: lane -> column (channel) lane & 31, register r -> row (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
  float* out = q.part + ((long)(group * q.nslices + slice) * GM_NW + wave) * GM_PART_FLOATS;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
    out[(row * 3 + 0) * GM_CH + c] = acc0[r];
    out[(row * 3 + 1) * GM_CH + c] = acc1[r];
    out[(row * 3 + 2) * GM_CH + c] = acc2[r];
  }
}

// gWr[col][k] = sum over wavefronts whose rows cover k, over the workgroups in order; gbr[col] = sum of the bias rows.  Fixed order: reproducible.
__global__ void k_gwr_mol_reduce(const float* __restrict__ part, const int* __restrict__ wlo, int groups, int nslices, int R, int F,
                                 float* __restrict__ gWr, float* __restrict__ gbr) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int F3 = 3 * F;
  if (idx >= (R + 1) * F3) return;
  const int k = idx / F3, col = idx % F3;
  const int p = col / F, ch = col % F, slice = ch / GM_CH, c = ch % GM_CH;
  float s = 0.f;
  for (int w = 0; w < GM_NW; ++w) {
    int row = GM_ROWS - 1;             // bias row
    if (k < R) {
      row = k - wlo[w];
      if (row < 0 || row >= GM_ROWS - 1) continue;
    }
    const int o = (row * 3 + p) * GM_CH + c;
#pragma unroll 8
    for (int g = 0; g < groups; ++g) s += part[((long)(g * nslices + slice) * GM_NW + w) * GM_PART_FLOATS + o];   // unrolled: eight loads in flight, same order
  }
  if (k < R) gWr[(long)col * R + k] = s;
  else gbr[col] = s;
}

// ---- host side -------------------------------------------------------------------------------------------------------------------
bool nq_molgw_supported(int F, int R, int max_mol_atoms) {
  if (F % GM_CH != 0 || R < FWIN || R - FWIN + 1 > GM_NW * GM_WMAX || R - FWIN + 1 > GM_MAX_BINS) return false;
  if (max_mol_atoms <= 0 || max_mol_atoms >= 8192) return false;
  return (size_t)max_mol_atoms * GM_ATOM_FLOATS * sizeof(float) + GM_NW * 1024 <= 160 * 1024;
}
static int molgw_groups(int B, int nslices) {
  int groups = 256 / nslices;            // one workgroup per CU
  if (groups < 1) groups = 1;
  if (groups > B) groups = B;
  return groups;
}
size_t nq_molgw_sched_ints(int E, int B) { return (size_t)E + (size_t)B * (GM_NW + 1) + GM_MAX_BINS + GM_NW + 1 + 16; }   // sched (int2 per pair) + sched_ptr + hist + wlo
size_t nq_molgw_rec_floats(int E) { return (size_t)(E / 2) * 64 + (size_t)(E / 2) * 8 + 16; }   // PA: 64 floats per pair (once per step), PG: 8 floats per pair (once per backward sweep)
size_t nq_molgw_part_floats(int F, int B) { const int ns = F / GM_CH; return (size_t)molgw_groups(B, ns) * ns * GM_NW * GM_PART_FLOATS; }

struct MolGwBufs { int2* sched; int* sched_ptr; int* hist; int* wlo; };
static MolGwBufs molgw_bufs(int* base, int E, int B) {
  MolGwBufs b;
  b.sched = reinterpret_cast<int2*>(base);                       // E / 2 pairs, 8 bytes each
  b.sched_ptr = base + (((size_t)E + 1) & ~(size_t)1);
  b.hist = b.sched_ptr + (size_t)B * (GM_NW + 1);
  b.wlo = b.hist + GM_MAX_BINS;
  return b;
}

// once per step, after the window records: histogram of the window starts over the pairs -> row windows of the wavefronts -> per-molecule pair lists -> expanded A operands
int nq_molgw_schedule(hipStream_t st, const NqGraphView& g, const int* dst, const float* RW, int R, int* sched_ints, float* recs) {
  NQ_PROF(st, "pair_schedule");
  const MolGwBufs b = molgw_bufs(sched_ints, g.E, g.B);
  NQ_HIP(hipMemsetAsync(b.hist, 0, GM_MAX_BINS * sizeof(int), st));
  int blocks = nq_cdiv(g.E, 256 * 8);
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  hipLaunchKernelGGL(k_pair_hist, dim3(blocks), dim3(256), 0, st, RW, dst, g.col, g.E, b.hist);
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_pair_windows, dim3(1), dim3(64), 0, st, b.hist, R - FWIN + 1, b.wlo);
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_pair_sched, dim3(g.B), dim3(64), 0, st, g, dst, RW, b.wlo, b.sched, b.sched_ptr);
  NQ_LAUNCH_CHECK();
  const int npairs = g.E / 2;
  hipLaunchKernelGGL(k_pair_arec, dim3(nq_cdiv((long)npairs * 64, 256)), dim3(256), 0, st, b.sched, RW, npairs, recs);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
// once per backward sweep, after nq_geom_tan: geometry + tangent + atom indices of every pair in schedule order
int nq_molgw_geometry(hipStream_t st, const NqGraphView& g, const float* TD, const float* TR, const int* sched_ints, float* recs) {
  NQ_PROF(st, "pair_geometry");
  const MolGwBufs b = molgw_bufs(const_cast<int*>(sched_ints), g.E, g.B);
  const int npairs = g.E / 2;
  float* PG = recs + (size_t)npairs * 64;
  hipLaunchKernelGGL(k_pair_grec, dim3(nq_cdiv(npairs, 256)), dim3(256), 0, st, b.sched, g.geom, TD, TR, npairs, reinterpret_cast<float4*>(PG));
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_gwr_mol(hipStream_t st, const NqGraphView& g, int F, int R, int max_mol_atoms, const float* XH, const float* V, const float* TXH, const float* TV,
               const float* GX, const float* GV, const float* GTX, const float* GTV, const int* sched_ints, const float* recs, float* part, float* gWr,
               float* gbr) {
  NQ_PROF(st, "gwr_mol");
  const MolGwBufs b = molgw_bufs(const_cast<int*>(sched_ints), g.E, g.B);
  GwrMolArgs q;
  q.g = g; q.F = F; q.nslices = F / GM_CH; q.groups = molgw_groups(g.B, q.nslices); q.max_atoms = max_mol_atoms;
  q.XH = XH; q.V = V; q.TXH = TXH; q.TV = TV; q.GX = GX; q.GV = GV; q.GTX = GTX; q.GTV = GTV;
  q.sched_ptr = b.sched_ptr; q.part = part;
  const float* PA = recs; const float* PG = recs + (size_t)(g.E / 2) * 64;
  const size_t lds = (size_t)max_mol_atoms * GM_ATOM_FLOATS * sizeof(float) + GM_NW * 1024;
  static size_t lds_set = 0;
  if (lds > lds_set) {
    NQ_HIP(hipFuncSetAttribute((const void*)k_gwr_mol, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_set = lds;
  }
  hipLaunchKernelGGL(k_gwr_mol, dim3(q.groups * q.nslices), dim3(GM_THREADS), lds, st, q, PA, PG);
  NQ_LAUNCH_CHECK();
  if ((GM_ABLATE & 15) != 5) hipLaunchKernelGGL(k_gwr_mol_reduce, dim3(nq_cdiv((long)(R + 1) * 3 * F, 256)), dim3(256), 0, st, part, b.wlo, q.groups, q.nslices, R, F, gWr, gbr);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
