// rbf_proj weight / bias gradient of the PaiNN dual-reverse sweep, molecule per workgroup (round 5; round 6: bf16-split matrix core, 8-pair batches,
// next molecule prefetched through registers).
// Reference semantics: painn.py:475-509 (rbfh = rbf_proj(edge_rbf); x = xh[j] * rbfh) differentiated twice -- the adjoints of
// phi = rbf_proj(rho(d)) and of its tangent psi * t_d, contracted with the 13-tap Gaussian window:
//     gWr[c][k0_p + t] += gphi_p[c] rho_t(d_p) + gpsi_p[c] drho_t(d_p),   p = undirected pair, both directions summed.
// Rounds 1-4 wrote gphi / gpsi per pair to HBM from the dual-reverse message kernel (2 x 1.5 kB per pair, 2.5 GB per layer at 2048
// conformers) and re-read them in k0 order (k_gwr_sorted).  gphi / gpsi do NOT depend on the filter, only on node rows:
//     direction s -> t:  gm_b = A_t.v_s + T_t.tv_s, gtm_b = T_t.v_s, gm_c = A_t.r + T_t.tr, gtm_c = T_t.r,
//                        gphi = (gma_t xa_s + gtma_t txa_s,  gm_b xb_s + gtm_b txb_s,  gm_c xc_s + gtm_c txc_s),  gpsi = t_d (gtma_t xa_s, gtm_b xb_s, gtm_c xc_s)
// (A, gma, T, gtma: adjoint rows of the target; x, tx, v, tv: primal / tangent rows of the source).  So here a workgroup stages the 20 rows
// of ONE molecule for a 32-channel slice in LDS with coalesced loads (2.5 kB per atom, <= 64 atoms) and every pair is recomputed where it is consumed:
//   * wavefront w keeps GM_WMAX + 12 accumulator rows (window starts [wlo[w], wlo[w] + GM_WMAX), placed on the w-th quantile range of the batch's k0
//     histogram) for the WHOLE launch as the C/D registers of the matrix core -- no atomics, no LDS accumulator, no sliding; neighbouring windows
//     overlap, and every molecule's k0-sorted pair list is cut into GM_NW equal segments that respect them: all wavefronts reach the barrier together;
//   * one pair step: lanes 0-31 the direction n -> k, lanes 32-63 the direction k -> n, one channel per lane; the 20 operands are five conflict-free
//     ds_read_b128; one v_permlane32_swap + add per filter part leaves gphi (both directions summed) in lanes 0-31 and gpsi in lanes 32-63;
//   * ROUND 6: the rank-2 update acc[j][c] += rho_j gphi[c] + drho_j gpsi[c] of EIGHT pairs is one K = 16 contraction of v_mfma_f32_32x32x16_bf16
//     (k = 0..7: gphi of the pairs, lanes 0-31; k = 8..15: gpsi, lanes 32-63 -- exactly what the swap leaves).  Both operands are split into two bf16
//     pieces (x = hi + lo, round-to-nearest twice; the window records once per step by k_pair_arec, gphi / gpsi in the loop: 3 VALU instructions per
//     value) and the product is hi hi' + hi lo' + lo hi' accumulated in f32: 9 matrix instructions of 32 cycles per 8 pairs instead of 24 of 64 cycles
//     (round 5: v_mfma_f32_32x32x2_f32 per pair, 192 of ~400 issue cycles per pair and slice; the f32 matrix path has no rate advantage over the VALU).
//     Error per product <= 3 x 2^-18 (dropped lo lo', two residuals), measured against float64 in tests/test_engine_gpu.py next to the exact-f32 pair rows;
//   * per-pair scalars (geometry, tangents, atom indices) arrive as ONE coalesced dword load per batch (lane 8 i + f = field f of pair i) and are
//     broadcast with v_readlane (static lane index): no LDS ring, no scalar-memory latency; the A operands as two 16-byte loads per lane and batch;
//     both requested two batches ahead (three static register sets);
//   * the rows of the NEXT molecule of the workgroup are requested into registers right after a wavefront's last pair (the pair loop's registers are free
//     again) and written to LDS behind the barrier that ends the molecule: their latency overlaps the wait for the slowest wavefront;
//   * the pair lists per (molecule, wavefront) are built once per step (the windows depend on the geometry only), sorted by window start, slot-ascending,
//     padded to whole batches, so the summation order is fixed: results are bitwise reproducible;
//   * per-workgroup partial rows (one flush per launch) are summed in workgroup order by k_gwr_mol_reduce.
// HBM traffic per launch: 20 rows x N x F x 4 B read once (0.88 GB at 2048 conformers) instead of 2.5 GB written + 2.5 GB read.
#include "common.h"
#include "lanes.h"
#include <type_traits>

#ifndef GM_THREADS
#define GM_THREADS 512    // 8 wavefronts = 2 per SIMD.  12 (the kernel fits 168 registers) measured 5.28 vs 5.13 ms per step: more segments per molecule = more
                          // padding and a longer wait at the per-molecule barrier, and the VALU is no longer the bound (profiles/r06_gwr_mol_variants.txt)
#endif
#define GM_STAGE_THREADS 512   // threads that stage rows: (atom, channel quad) = 64 x 8
#define GM_NW (GM_THREADS / 64)     // wavefronts per workgroup = owners of window-start ranges
#define GM_ROWS 32                  // rows of the matrix-core tile: 31 window rows + the bias row
#define GM_WMAX (GM_ROWS - 1 - (FWIN - 1))   // window starts per wavefront at most (19): row off + 12 <= 30
#define GM_CH 32                    // channels per slice
#define GM_BATCH 8                  // pairs per matrix-core contraction (K = 16 = 8 x {gphi, gpsi})
#ifndef GM_ATOM_PAD
#define GM_ATOM_PAD 4       // floats between atoms: with 2560-byte atoms the staging writes of the 8 atoms a wavefront holds fell on the same banks (16-way; SQ_LDS_BANK_CONFLICT
                         // 2.1e7 cycles per launch, profiles/r06_pmc_sq_*.txt); 16 bytes of padding spread them (two-way left: channel quads q and q + 4)
#endif
#define GM_ATOM_FLOATS (20 * GM_CH + GM_ATOM_PAD) // LDS floats per atom: 5 blocks [32 channels][4 rows] + padding
#define GM_ATOM_BYTES (GM_ATOM_FLOATS * 4)
#define GM_RING_BYTES 512           // per wavefront: the scalar records of two batches (2 x 8 pairs x 8 dwords), read back as broadcast ds_read_b128
#define GM_MAX_ATOMS ((160 * 1024 - GM_NW * GM_RING_BYTES) / GM_ATOM_BYTES)   // 62 with 8 wavefronts: the rows + the rings fill the 160 KB of LDS
#define GM_PART_FLOATS (GM_ROWS * 3 * GM_CH)   // per (workgroup, wavefront): [row][part][channel]
#define GM_MAX_BINS 128
#define GM_PA_DWORDS (2 * 64 * 4)   // A operands of one batch: [piece hi / lo][lane][8 bf16]
#define GM_PG_DWORDS 64             // scalars of one batch: [pair][8 fields]
#ifndef GM_STAGE_EARLY
#define GM_STAGE_EARLY 1
#endif
#ifndef GM_ABLATE
#define GM_ABLATE 0   // development only (results are wrong, timing only): 1 no pair steps, 2 no matrix-core updates, 4 no per-pair arithmetic, 8 no operand loads, 16 no permlane swaps, 32 no geometry broadcasts, 64 no bf16 split, 128 no barriers, 256 no per-step liveness branch
#endif

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __bf16 gm_bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 gm_bf2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
static_assert(GM_THREADS >= GM_STAGE_THREADS && GM_THREADS % 64 == 0, "the staging of k_gwr_mol deals (atom, channel quad) tasks to the first 512 threads");
static_assert(GM_MAX_ATOMS * 8 <= GM_STAGE_THREADS, "one staging task per (atom, channel quad)");

__device__ __forceinline__ unsigned gm_pack(float a, float b) {   // two f32 -> packed bf16 (round to nearest even): v_cvt_pk_bf16_f32
  gm_bf2 v;
  v[0] = (__bf16)a; v[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, v);
}
// (x0, x1) -> hi = bf16(x), lo = bf16(x - hi) (the subtraction is exact), both packed
__device__ __forceinline__ void gm_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  if (GM_ABLATE & 64) { hi = __float_as_uint(x0); lo = __float_as_uint(x1); return; }
  hi = gm_pack(x0, x1);
  lo = gm_pack(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u));
}

// first batch of molecule m's padded pair list: sum_w ceil(len_w / 8) <= pairs / 8 + GM_NW, so floor(lowptr / 8) + (GM_NW) m never overlaps the next molecule
__host__ __device__ __forceinline__ int gm_mol_batch0(int lowptr_a0, int m) { return (lowptr_a0 >> 3) + GM_NW * m; }
static inline size_t gm_max_batches(int E, int B) { return (size_t)(E / 2) / GM_BATCH + (size_t)GM_NW * B + GM_NW; }

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
// fills up to ceil(pairs / GM_NW) with pairs of k0 < wlo[w] + GM_WMAX.  sched_ptr[m][w] = first pair of segment w in the molecule's sorted list (cut points).
// Every segment starts on a batch boundary of the padded list: seg[m][w] = {first batch, pairs}; slot 8 batch + i holds
// {CSR slot of the lower edge (row n, source k < n) or -1 (padding), row offset (k0 - wlo[w]) << 26 | (n - a0) << 13 | (k - a0)}.
// Ranks come from ballots over the distinct keys of a 64-slot chunk (no atomics): the order is a function of the geometry only.
// A molecule of more than cap atoms gets empty segments (its pairs go through the pair-row kernels, engine.hip).
template <bool FILL>
__device__ __forceinline__ void pair_sched_pass(const NqGraphView& g, const int* __restrict__ dst, const float* __restrict__ RW, const int* __restrict__ wlo,
                                                const int* cut, const int* bst, int b0, int a0, int s0, int s1, int lane, int* cnt, int2* __restrict__ sched) {
  for (int c0 = s0; c0 < s1; c0 += 64) {
    const int s = c0 + lane;
    int key = -1, n = 0, k = 0;
    if (s < s1) {
      n = dst[s]; k = g.col[s];
      if (k < n) key = min(GM_MAX_BINS - 1, __float_as_int(RW[(long)s * RW_STRIDE + 13]));
    }
    // rank of the lane among the chunk's lanes with the same key (lower lane index first: slot order) and whether it is the last of them: 64 lane broadcasts,
    // no LDS and no loop over the distinct keys (that loop was a chain of ~10 dependent LDS operations per key: 0.22 ms per step at 2048 conformers)
    int below = 0;
    bool last = true;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      const bool same = __builtin_amdgcn_readlane(key, i) == key;
      below += (same && i < lane) ? 1 : 0;
      last = last && !(same && i > lane);
    }
    const int pos = key >= 0 ? cnt[key] + below : 0;
    if (FILL && key >= 0) {
      int w = 0;
#pragma unroll
      for (int v = 1; v < GM_NW; ++v) w += pos >= cut[v];
      sched[(long)GM_BATCH * (b0 + bst[w]) + (pos - cut[w])] = make_int2(s, ((key - wlo[w]) << 26) | ((n - a0) << 13) | (k - a0));
    }
    __builtin_amdgcn_wave_barrier();
    if (key >= 0 && last) cnt[key] = pos + 1;
    __builtin_amdgcn_wave_barrier();
  }
}
__global__ __launch_bounds__(64) void k_pair_sched(NqGraphView g, const int* __restrict__ dst, const float* __restrict__ RW, const int* __restrict__ wlo, int cap,
                                                    int2* __restrict__ sched, int* __restrict__ sched_ptr, int2* __restrict__ seg) {
  __shared__ int cnt[GM_MAX_BINS + 1];
  __shared__ int cut[GM_NW + 1];
  __shared__ int bst[GM_NW + 1];
  const int m = blockIdx.x, lane = threadIdx.x;
  const int a0 = g.mol_ptr[m], a1 = g.mol_ptr[m + 1];
  const int s0 = g.row_ptr[a0], s1 = g.row_ptr[a1], b0 = gm_mol_batch0(g.lowptr[a0], m);
  if (a1 - a0 > cap) {   // too large for the LDS of k_gwr_mol: nothing scheduled
    if (lane <= GM_NW) sched_ptr[(long)m * (GM_NW + 1) + lane] = 0;
    if (lane < GM_NW) seg[(long)m * GM_NW + lane] = make_int2(b0, 0);
    return;
  }
  cnt[lane] = 0; cnt[lane + 64] = 0;
  __syncthreads();
  pair_sched_pass<false>(g, dst, RW, wlo, cut, bst, b0, a0, s0, s1, lane, cnt, sched);
  __syncthreads();
  if (lane == 0) {
    int run = 0;   // exclusive scan over the window starts (<= 128 values): cnt[b] = pairs with k0 < b
    for (int b = 0; b < GM_MAX_BINS; ++b) { const int v = cnt[b]; cnt[b] = run; run += v; }
    cnt[GM_MAX_BINS] = run;
    const int np = run, target = (np + GM_NW - 1) / GM_NW;
    int cur = 0, nb = 0;
    for (int w = 0; w < GM_NW; ++w) {
      cut[w] = cur; bst[w] = nb;
      const int forced_end = w == GM_NW - 1 ? np : cnt[min(GM_MAX_BINS, wlo[w + 1])];
      const int optional_end = min(cur + target, cnt[min(GM_MAX_BINS, wlo[w] + GM_WMAX)]);
      const int end = max(cur, max(forced_end, optional_end));
      nb += (end - cur + GM_BATCH - 1) / GM_BATCH;
      cur = end;
    }
    cut[GM_NW] = np; bst[GM_NW] = nb;
  }
  __syncthreads();
  if (lane <= GM_NW) sched_ptr[(long)m * (GM_NW + 1) + lane] = cut[lane];
  if (lane < GM_NW) {
    const int len = cut[lane + 1] - cut[lane];
    seg[(long)m * GM_NW + lane] = make_int2(b0 + bst[lane], len);
    for (int i = len; i < (bst[lane + 1] - bst[lane]) * GM_BATCH; ++i) sched[(long)GM_BATCH * (b0 + bst[lane]) + i] = make_int2(-1, 0);   // padding of the last batch
  }
  pair_sched_pass<true>(g, dst, RW, wlo, cut, bst, b0, a0, s0, s1, lane, cnt, sched);
}

// Expanded A operands.  PA[batch][piece][lane] = 8 bf16: what lane (row j = lane & 31, K half = lane >> 5) feeds to v_mfma_f32_32x32x16_bf16 for the 8 pairs of
// the batch: rho (lanes 0-31) / t_d drho (lanes 32-63) tap j - off of each pair's window record, 0 outside the 13 taps and for padding, and in row 31 the bias
// multiplier / t_d times its derivative; piece 0 = bf16(x), piece 1 = bf16(x - piece 0).  The rho half depends on the geometry only: written once per step
// (TD == nullptr); the drho half carries the pair's tangent t_d (the factor of gpsi, folded in here so that the pair loop spends no instruction on it) and is
// written once per backward sweep (TD set).
__global__ __launch_bounds__(256) void k_pair_arec(const int2* __restrict__ sched, const int2* __restrict__ seg, const float* __restrict__ RW,
                                                    const float* __restrict__ TD, int nseg, u4* __restrict__ PA) {
  // four wavefronts per (molecule, wavefront-of-k_gwr_mol) segment; each HALF-wavefront takes every eighth batch (a segment has ~7: the loop is a chain of
  // dependent gathers; a launch writes only one half of the 64-lane records -- rho, or t_d drho -- so 32 lanes cover a batch)
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), sg = wid >> 2, lane = threadIdx.x & 63;
  if (sg >= nseg) return;
  const int2 se = seg[sg];
  const int nb = (se.y + GM_BATCH - 1) / GM_BATCH, j = lane & 31, sub = lane >> 5;
  const int half = TD != nullptr ? 1 : 0, olane = half * 32 + j;
  for (int b = (wid & 3) * 2 + sub; b < nb; b += 8) {
    const long batch = se.x + b;
    float v[GM_BATCH];
#pragma unroll
    for (int i = 0; i < GM_BATCH; ++i) {
      const int2 en = sched[batch * GM_BATCH + i];
      const int off = (int)((unsigned)en.y >> 26);
      const int t = j - off;
      const bool valid = en.x >= 0 && (((unsigned)t < (unsigned)FWIN) | (j == 31));
      const int tap = j == 31 ? 14 : min(max(t, 0), FWIN - 1);
      v[i] = valid ? RW[(long)max(en.x, 0) * RW_STRIDE + half * 16 + tap] * (TD ? TD[max(en.x, 0)] : 1.0f) : 0.f;
    }
    unsigned hi[4], lo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) gm_split2(v[2 * i], v[2 * i + 1], hi[i], lo[i]);
    PA[(batch * 2 + 0) * 64 + olane] = u4{hi[0], hi[1], hi[2], hi[3]};
    PA[(batch * 2 + 1) * 64 + olane] = u4{lo[0], lo[1], lo[2], lo[3]};
  }
}
// Packed scalars, once per backward sweep (the tangent t_r follows the force seeds): PG[batch][8 i + f] = field f of pair i:
// {gx, gy, gz, 0, t_r0, t_r1, t_r2, n << 16 | k}; padding pairs: zeros (atom 0 with itself; their A operands are zero).
__global__ __launch_bounds__(256) void k_pair_grec(const int2* __restrict__ sched, const int2* __restrict__ seg, const float4* __restrict__ geom,
                                                    const float* __restrict__ TR, int nseg, float* __restrict__ PG) {
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), sg = wid >> 2, lane = threadIdx.x & 63;
  if (sg >= nseg) return;
  const int2 se = seg[sg];
  const int nb = (se.y + GM_BATCH - 1) / GM_BATCH, i = lane >> 3, f = lane & 7;
  for (int b = wid & 3; b < nb; b += 4) {
    const long batch = se.x + b;
    const int2 en = sched[batch * GM_BATCH + i];
    float v = 0.f;
    if (en.x >= 0) {
      if (f < 3) v = reinterpret_cast<const float*>(geom)[4 * (long)en.x + f];
      else if (f == 3) v = 0.f;
      else if (f < 7) v = TR[3 * (long)en.x + (f - 4)];
      else v = __int_as_float((((en.y >> 13) & 0x1fff) << 16) | (en.y & 0x1fff));
    }
    PG[batch * GM_PG_DWORDS + lane] = v;
  }
}

// ---- the gradient kernel ----------------------------------------------------------------------------------------------------------
struct GwrMolArgs {
  NqGraphView g; int F; int nslices; int groups; int max_atoms;
  const float* XH; const float* V; const float* TXH; const float* TV;      // primal / tangent rows of the layer input side  [N][3F]
  const float* GX; const float* GV; const float* GTX; const float* GTV;    // adjoints of x_msg / vec_msg and of their tangents  [N][F], [N][3F]
  const int2* seg;                                                          // [B][GM_NW] {first batch, pairs}
  const int* order;                                                         // [B] molecules by descending pair count (k_mol_order) or null: identity
  float* part;                                                              // [groups][nslices][GM_NW][GM_PART_FLOATS]
};

// x and y hold one value per direction (lanes 0-31: n -> k, lanes 32-63: k -> n).  v_permlane32_swap exchanges the upper half of its first operand with the
// lower half of its second: afterwards r[0] = {x(n->k), y(n->k)}, r[1] = {x(k->n), y(k->n)}.  SUM: both directions added (lanes 0-31: x, lanes 32-63: y);
// DIFF: (k -> n) - (n -> k) -- the part that carries the pair's unit vector, whose sign flips with the direction, so the loop spends no multiply on the sign.
template <bool DIFF>
__device__ __forceinline__ float gm_pair_sum(float x, float y) {
  if (GM_ABLATE & 16) return DIFF ? y - x : x + y;
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return DIFF ? __uint_as_float(r[1]) - __uint_as_float(r[0]) : __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

struct GmOps { f4 P0, P1, P2, A, T; };
struct GmRec { u4 hi, lo; };   // the two A pieces of this lane for one batch

__global__ __launch_bounds__(GM_THREADS) void k_gwr_mol(GwrMolArgs q, const u4* __restrict__ PA, const unsigned* __restrict__ PG) {
  extern __shared__ __attribute__((aligned(16))) float rows[];   // [atom][5 blocks][32 channels][4], then one 512-byte scalar ring per wavefront
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
  const unsigned hmask = half ? 0xffffffffu : 0u;
  const unsigned lbase = (unsigned)(c * 16);     // byte offset of the lane's channel inside a [32 channels][4] block
  char* const ring = reinterpret_cast<char*>(rows) + (size_t)q.max_atoms * GM_ATOM_BYTES + wave * GM_RING_BYTES;   // [2 batches][8 pairs][8 dwords]

  // ---- staging: thread t < 512 owns (atom t >> 3, channel quad t & 7) of every row block: four 16-byte loads (128 contiguous bytes per row and eight lanes),
  //      transposed to [channel][4 rows] by four ds_write_b128 ----
  const int st_at = threadIdx.x >> 3, st_qd = threadIdx.x & 7;
  const bool stager = threadIdx.x < GM_STAGE_THREADS;   // wave-uniform
  auto blk_load = [&](int blk, int a0, int na, f4 (&r)[4]) __attribute__((always_inline)) {
    const float* s0; const float* s1; const float* s2; const float* s3;
    int st3 = F3;   // row stride of the fourth source (F for the scalar adjoints GX / GTX)
    switch (blk) {
      case 0: s0 = q.XH; s1 = q.XH + F; s2 = q.XH + 2 * F; s3 = q.TXH; break;
      case 1: s0 = q.TXH + F; s1 = q.TXH + 2 * F; s2 = q.V; s3 = q.V + F; break;
      case 2: s0 = q.V + 2 * F; s1 = q.TV; s2 = q.TV + F; s3 = q.TV + 2 * F; break;
      case 3: s0 = q.GV; s1 = q.GV + F; s2 = q.GV + 2 * F; s3 = q.GX; st3 = F; break;
      default: s0 = q.GTV; s1 = q.GTV + F; s2 = q.GTV + 2 * F; s3 = q.GTX; st3 = F; break;
    }
    const long n = a0 + min(st_at, na - 1), o = n * F3 + cb + 4 * st_qd;   // threads past the last atom re-read it (no branch around loads that stay in flight)
    r[0] = *reinterpret_cast<const f4*>(s0 + o); r[1] = *reinterpret_cast<const f4*>(s1 + o); r[2] = *reinterpret_cast<const f4*>(s2 + o);
    r[3] = *reinterpret_cast<const f4*>(s3 + n * st3 + cb + 4 * st_qd);
  };
  auto blk_store = [&](int blk, int na, const f4 (&r)[4]) __attribute__((always_inline)) {
    if (st_at < na) {
      float* d = rows + st_at * GM_ATOM_FLOATS + blk * (GM_CH * 4) + (4 * st_qd) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f4*>(d + i * 4) = f4{r[0][i], r[1][i], r[2][i], r[3][i]};
    }
  };
  auto ops_load = [&](GmOps& o, unsigned nk) __attribute__((always_inline)) {
    const unsigned no = (nk >> 16) * GM_ATOM_BYTES, ko = (nk & 0xffffu) * GM_ATOM_BYTES;
    // operands: primal / tangent rows of the lane's SOURCE atom (n for lanes 0-31, k for 32-63), adjoint rows of its TARGET atom (k / n)
    const unsigned x = (no ^ ko) & hmask;
    const char* ps = reinterpret_cast<const char*>(rows) + ((no ^ x) + lbase);
    const char* pt = reinterpret_cast<const char*>(rows) + ((ko ^ x) + lbase);
    if (GM_ABLATE & 8) { const float z = __uint_as_float((no ^ x) + lbase); o.P0 = o.P1 = o.P2 = f4{z, z, 1.f, 2.f}; o.A = o.T = f4{1.f, z, __uint_as_float((ko ^ x) + lbase), 3.f}; return; }
    o.P0 = *reinterpret_cast<const f4*>(ps);                          // xa xb xc txa
    o.P1 = *reinterpret_cast<const f4*>(ps + GM_CH * 16);             // txb txc v0 v1
    o.P2 = *reinterpret_cast<const f4*>(ps + 2 * GM_CH * 16);         // v2 tv0 tv1 tv2
    o.A = *reinterpret_cast<const f4*>(pt + 3 * GM_CH * 16);          // A0 A1 A2 gma
    o.T = *reinterpret_cast<const f4*>(pt + 4 * GM_CH * 16);          // T0 T1 T2 gtma
  };
  auto rec_load = [&](GmRec& r, long batch) __attribute__((always_inline)) {
    r.hi = PA[(batch * 2 + 0) * 64 + lane];
    r.lo = PA[(batch * 2 + 1) * 64 + lane];
  };
  auto geo_load = [&](long batch) __attribute__((always_inline)) -> unsigned { return PG[batch * GM_PG_DWORDS + lane]; };
  // scalar records of a batch: one dword per lane into the wavefront's ring (slot = parity of the batch inside its segment), read back as broadcast
  // ds_read_b128 -- every lane gets the pair's scalars as VECTOR operands: no v_readlane / v_readfirstlane per scalar (round 5 and the first round-6 build spent
  // 8 VALU instructions per pair on them), only the atom indices of the NEXT pair go through one v_readfirstlane for the address arithmetic
  auto ring_put = [&](int slot, unsigned g) __attribute__((always_inline)) { *reinterpret_cast<unsigned*>(ring + slot * 256 + lane * 4) = g; };
  auto ring_lo = [&](int slot, int i) __attribute__((always_inline)) -> f4 { return *reinterpret_cast<const f4*>(ring + slot * 256 + i * 32); };        // gx gy gz .
  auto ring_hi = [&](int slot, int i) __attribute__((always_inline)) -> f4 { return *reinterpret_cast<const f4*>(ring + slot * 256 + i * 32 + 16); };   // t0 t1 t2 n<<16|k
  // B operands of the matrix core: [part][pair / 2] packed bf16 pieces.  They persist across batches: a segment's last batch stops after its last live pair
  // and leaves the previous batch's (finite) values in the remaining slots -- the A operands of padding slots are zero.
  unsigned bh[3][4], bl[3][4];
#pragma unroll
  for (int p_ = 0; p_ < 3; ++p_)
#pragma unroll
    for (int j = 0; j < 4; ++j) bh[p_][j] = bl[p_][j] = 0u;
  // One batch = 8 pair steps + 9 matrix-core instructions.  `L0` holds the operands of pair 0 and `gc` its {t0, t1, t2, .} on entry (requested by the previous
  // batch / the prologue); step i requests the operands of pair i + 1 (the first pair of the next batch, ring slot `slot ^ 1`, in step 7) and consumes its own.
  // Steps go in twos (one packed bf16 pair per part): a two-step group whose first pair is padding ends the batch (wave-uniform branch, once per group);
  // an odd number of live pairs computes one padding step on atom 0's rows.
  auto batch = [&](const GmRec& rc, int slot, int live, bool more, GmOps& L0, GmOps& L1, f4& gc) __attribute__((always_inline)) {
    auto step = [&](int i, GmOps& cur, GmOps& nxt, float& ba, float& bb, float& bc) __attribute__((always_inline)) {
      const f4 gn = ring_hi(i < GM_BATCH - 1 ? slot : slot ^ 1, (i + 1) & 7);
      const f4 g0 = ring_lo(slot, i);
      unsigned nkn = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(gn[3]));
      if (i == GM_BATCH - 1 && !more) nkn = 0;   // past the segment: atom 0 with itself (rows that exist)
      ops_load(nxt, nkn);
      const float gx = g0[0], gy = g0[1], gz = g0[2], t0 = gc[0], t1 = gc[1], t2 = gc[2];
      const f4 P0 = cur.P0, P1 = cur.P1, P2 = cur.P2, A = cur.A, T = cur.T;
      const float xa = P0[0], xb = P0[1], xc = P0[2], txa = P0[3], txb = P1[0], txc = P1[1], v0 = P1[2], v1 = P1[3], v2 = P2[0], tv0 = P2[1],
                  tv1 = P2[2], tv2 = P2[3];
      float ga, gb, gcc, ha, hb, hc;
      if (GM_ABLATE & 4) { ga = xa + gx; gb = xb + txa; gcc = xc + A[0]; ha = txb + T[0]; hb = v2 + gy + t0; hc = tv2 + v1 + t1 + t2 + gz + txc + v0 + tv0 + tv1; }
      else {
        // the unit vector of the lane's direction is -geom (n -> k) / +geom (k -> n): gmc, gtmc are computed with +geom here and the sign is applied by
        // the DIFF form of the direction sum below; the factor t_d of gpsi sits in the drho half of the A operands (k_pair_arec)
        const float gmb = A[0] * v0 + A[1] * v1 + A[2] * v2 + (T[0] * tv0 + T[1] * tv1 + T[2] * tv2);
        const float gtmb = T[0] * v0 + T[1] * v1 + T[2] * v2;
        const float gmc = (A[0] * gx + A[1] * gy + A[2] * gz) + (T[0] * t0 + T[1] * t1 + T[2] * t2);
        const float gtmc = T[0] * gx + T[1] * gy + T[2] * gz;
        const float gma = A[3], gtma = T[3];
        ga = gma * xa + gtma * txa; gb = gmb * xb + gtmb * txb; gcc = gmc * xc + gtmc * txc;
        ha = gtma * xa; hb = gtmb * xb; hc = gtmc * xc;
      }
      // both directions combined: lanes 0-31 <- gphi(n->k) + gphi(k->n), lanes 32-63 <- gpsi(n->k) + gpsi(k->n) (part c: difference, see gm_pair_sum)
      ba = gm_pair_sum<false>(ga, ha); bb = gm_pair_sum<false>(gb, hb); bc = gm_pair_sum<true>(gcc, hc);
      gc = gn;
    };
    if (!(GM_ABLATE & 1)) {
#pragma unroll
      for (int j = 0; j < GM_BATCH / 2; ++j) {
        if (!(GM_ABLATE & 256) && 2 * j >= live) break;   // wave-uniform; only in a segment's last batch (an even number of steps has run: L0 / L1 keep their roles)
        float a0_, b0_, c0_, a1_, b1_, c1_;
        step(2 * j, L0, L1, a0_, b0_, c0_);
        step(2 * j + 1, L1, L0, a1_, b1_, c1_);
        gm_split2(a0_, a1_, bh[0][j], bl[0][j]);
        gm_split2(b0_, b1_, bh[1][j], bl[1][j]);
        gm_split2(c0_, c1_, bh[2][j], bl[2][j]);
      }
    }
    if (!(GM_ABLATE & 2)) {
      const gm_bf8 ah = __builtin_bit_cast(gm_bf8, rc.hi), al = __builtin_bit_cast(gm_bf8, rc.lo);
      const gm_bf8 h0 = __builtin_bit_cast(gm_bf8, (u4{bh[0][0], bh[0][1], bh[0][2], bh[0][3]})), l0 = __builtin_bit_cast(gm_bf8, (u4{bl[0][0], bl[0][1], bl[0][2], bl[0][3]}));
      const gm_bf8 h1 = __builtin_bit_cast(gm_bf8, (u4{bh[1][0], bh[1][1], bh[1][2], bh[1][3]})), l1 = __builtin_bit_cast(gm_bf8, (u4{bl[1][0], bl[1][1], bl[1][2], bl[1][3]}));
      const gm_bf8 h2 = __builtin_bit_cast(gm_bf8, (u4{bh[2][0], bh[2][1], bh[2][2], bh[2][3]})), l2 = __builtin_bit_cast(gm_bf8, (u4{bl[2][0], bl[2][1], bl[2][2], bl[2][3]}));
      // small terms first; the three parts interleave so that no instruction waits for the accumulator of the one before it
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, h0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, h1, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, h2, acc2, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, l0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, l1, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, l2, acc2, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, h0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, h1, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, h2, acc2, 0, 0, 0);
    } else { acc0[0] += __uint_as_float(bh[0][0] ^ bl[1][1] ^ bh[2][2] ^ bl[0][3] ^ bh[1][0] ^ bl[2][1] ^ bh[0][1] ^ bl[0][0] ^ bh[1][2] ^ bl[1][3] ^ bh[2][0] ^ bl[2][3] ^ rc.hi[0] ^ rc.lo[1]); }
  };

  // ---- molecules of this workgroup: positions r G + g (r even) / r G + G - 1 - g (r odd) of the list sorted by descending pair count (a serpentine deal: the
  //      groups' pair totals agree to ~0.3 %).  Dealing m, m + G, ... by index left the slowest group 8 % above the mean on random batches -- and 50 % on batches
  //      built from replicas of 64 molecules, where a group then owns every copy of ONE molecule.  The order is a function of the geometry only: reproducible.
  //      Molecules larger than the LDS are skipped: their pairs are not in the schedule. ----
  int rpos = -1;
  auto next_mol = [&]() __attribute__((always_inline)) -> int {
    for (++rpos;; ++rpos) {
      const long idx = (long)rpos * q.groups + ((rpos & 1) ? q.groups - 1 - group : group);
      if (idx >= q.g.B) return q.g.B;
      const int mm = q.order ? q.order[idx] : (int)idx;
      if (q.g.mol_ptr[mm + 1] - q.g.mol_ptr[mm] <= q.max_atoms) return mm;
    }
  };
  int m = next_mol();
  if (m < q.g.B && stager) {
    const int a0 = __builtin_amdgcn_readfirstlane(q.g.mol_ptr[m]), na = __builtin_amdgcn_readfirstlane(q.g.mol_ptr[m + 1]) - a0;
#pragma unroll
    for (int blk = 0; blk < 5; ++blk) { f4 r[4]; blk_load(blk, a0, na, r); blk_store(blk, na, r); }
  }
  // per segment: A operands of batch j in RX (even j) / RY (odd j), scalar records of batch j in ring slot j & 1; requested one batch ahead
  // (scalars: two), the first ones before the barrier that ends the previous molecule
  GmRec RX, RY;
  unsigned gq0 = 0, gq1 = 0;
  int2 sg = make_int2(0, 0);
  if (m < q.g.B) {
    sg = q.seg[(long)m * GM_NW + wave];
    sg.x = __builtin_amdgcn_readfirstlane(sg.x); sg.y = __builtin_amdgcn_readfirstlane(sg.y);
    rec_load(RX, sg.x); gq0 = geo_load(sg.x); gq1 = geo_load(sg.x + 1);
  }
  while (m < q.g.B) {
    const int mn = next_mol();
    int a0n = 0, nan = 1;
    int2 sgn_ = make_int2(0, 0);
    if (mn < q.g.B) {
      a0n = __builtin_amdgcn_readfirstlane(q.g.mol_ptr[mn]); nan = __builtin_amdgcn_readfirstlane(q.g.mol_ptr[mn + 1]) - a0n;
      sgn_ = q.seg[(long)mn * GM_NW + wave];
      sgn_.x = __builtin_amdgcn_readfirstlane(sgn_.x); sgn_.y = __builtin_amdgcn_readfirstlane(sgn_.y);
    }
    if (!(GM_ABLATE & 128)) __syncthreads();                           // the rows of molecule m are in LDS
    // ---- this wavefront's pairs of the molecule: batches [b0, b0 + nb) of the padded schedule ----
    const int nb = (sg.y + GM_BATCH - 1) / GM_BATCH;
    if (nb > 0) {
      const long b0 = sg.x, last = b0 + nb - 1;
      GmOps L0, L1;
      ring_put(0, gq0);
      f4 gc = ring_hi(0, 0);
      ops_load(L0, (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(gc[3])));
      int left = sg.y;
#pragma nounroll
      for (long b = b0; b <= last; b += 2) {
        ring_put(1, gq1);                    // scalars of batch b + 1 (requested two batches ago)
        rec_load(RY, min(b + 1, last));
        gq0 = geo_load(min(b + 2, last));
        __builtin_amdgcn_sched_barrier(0);   // keeps the prefetch above the pair steps (instcombine otherwise sinks a load through its loop-carried phi)
        batch(RX, 0, min(left, GM_BATCH), b < last, L0, L1, gc);
        if (b + 1 > last) break;
        ring_put(0, gq0);
        rec_load(RX, min(b + 2, last));
        gq1 = geo_load(min(b + 3, last));
        __builtin_amdgcn_sched_barrier(0);
        batch(RY, 1, min(left - GM_BATCH, GM_BATCH), b + 1 < last, L0, L1, gc);
        left -= 2 * GM_BATCH;
      }
    }
    // GM_STAGE_EARLY: the rows of the next molecule are requested into registers BEFORE the barrier that ends this molecule (the pair loop's registers are
    // free again) and written to LDS behind it, so that their latency overlaps the wait for the slowest wavefront; otherwise they are requested behind the barrier.
    // (Requesting them before / during the pair loop was measured and lost: the in-order load counter holds the younger record loads behind 100 kB of rows.)
    f4 stg[5][4];
    if (GM_STAGE_EARLY && mn < q.g.B && stager) {
#pragma unroll
      for (int blk = 0; blk < 5; ++blk) blk_load(blk, a0n, nan, stg[blk]);
    }
    // the first batches of the next molecule's segment: they arrive while the rows are being written
    if (mn < q.g.B) { rec_load(RX, sgn_.x); gq0 = geo_load(sgn_.x); gq1 = geo_load(sgn_.x + 1); }
    if (!(GM_ABLATE & 128)) __syncthreads();                           // every wavefront is done with the rows of molecule m
    if (mn < q.g.B && stager) {
#pragma unroll
      for (int blk = 0; blk < 5; ++blk) {
        if (!GM_STAGE_EARLY) blk_load(blk, a0n, nan, stg[blk]);
        blk_store(blk, nan, stg[blk]);
      }
    }
    m = mn; sg = sgn_;
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
// ACC: add to what is there (the pair-row kernels ran first for the molecules that do not fit the LDS).
// Four threads per output element, each over a quarter of the workgroups (a single thread's 128-load chain at 2.4 wavefronts per CU was latency-bound: 33 us), combined
// through LDS as (q0 + q1) + (q2 + q3).
#define GMR_Q 4
__global__ __launch_bounds__(256) void k_gwr_mol_reduce(const float* __restrict__ part, const int* __restrict__ wlo, int groups, int nslices, int R, int F,
                                                        float* __restrict__ gWr, float* __restrict__ gbr, int acc) {
  __shared__ float red[GMR_Q][256 / GMR_Q];
  const int e = threadIdx.x & (256 / GMR_Q - 1), qd = threadIdx.x / (256 / GMR_Q);
  const int idx = blockIdx.x * (256 / GMR_Q) + e;
  const int F3 = 3 * F;
  const bool live = idx < (R + 1) * F3;
  const int k = live ? idx / F3 : 0, col = live ? idx % F3 : 0;
  const int p = col / F, ch = col % F, slice = ch / GM_CH, c = ch % GM_CH;
  const int gper = (groups + GMR_Q - 1) / GMR_Q, g0 = min(groups, qd * gper), g1 = min(groups, g0 + gper);
  float s = 0.f;
  if (live)
    for (int w = 0; w < GM_NW; ++w) {
      int row = GM_ROWS - 1;             // bias row
      if (k < R) {
        row = k - wlo[w];
        if (row < 0 || row >= GM_ROWS - 1) continue;
      }
      const int o = (row * 3 + p) * GM_CH + c;
#pragma unroll 8
      for (int g = g0; g < g1; ++g) s += part[((long)(g * nslices + slice) * GM_NW + w) * GM_PART_FLOATS + o];   // unrolled: eight loads in flight, same order
    }
  red[qd][e] = s;
  __syncthreads();
  if (qd != 0 || !live) return;
  s = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
  if (k < R) gWr[(long)col * R + k] = acc ? gWr[(long)col * R + k] + s : s;
  else gbr[col] = acc ? gbr[col] + s : s;
}

// order[i] = the molecule with the i-th largest pair count (ties: lower index first); molecules above the LDS limit count 0 pairs.  ONE workgroup, bitonic
// sort of unique 32-bit keys (pairs << 20 | inverted index) in LDS: n2 = B rounded up to a power of two, <= GM_ORDER_MAX.
#define GM_ORDER_MAX 32768
__global__ __launch_bounds__(1024) void k_mol_order(NqGraphView g, int cap, int n2, int* __restrict__ order) {
  extern __shared__ unsigned okeys[];
  for (int i = threadIdx.x; i < n2; i += 1024) {
    unsigned key = 0;   // padding: below every real key
    if (i < g.B) {
      const int a0 = g.mol_ptr[i], a1 = g.mol_ptr[i + 1];
      const int pairs = a1 - a0 <= cap ? g.lowptr[a1] - g.lowptr[a0] : 0;
      key = ((unsigned)min(pairs, 2047) << 20) | (0xFFFFFu - (unsigned)i);
    }
    okeys[i] = key;
  }
  __syncthreads();
  for (int k = 2; k <= n2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n2; i += 1024) {
        const int x = i ^ j;
        if (x > i) {
          const unsigned a = okeys[i], b = okeys[x];
          const bool desc = (i & k) == 0;        // descending runs first: the final order is descending
          if ((a < b) == desc) { okeys[i] = b; okeys[x] = a; }
        }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < g.B; i += 1024) order[i] = (int)(0xFFFFFu - (okeys[i] & 0xFFFFFu));
}

// ---- host side -------------------------------------------------------------------------------------------------------------------
int nq_molgw_max_atoms(void) { return GM_MAX_ATOMS; }
bool nq_molgw_config_ok(int F, int R) {
  return F % GM_CH == 0 && R >= FWIN && R - FWIN + 1 <= GM_NW * GM_WMAX && R - FWIN + 1 <= GM_MAX_BINS;
}
bool nq_molgw_supported(int F, int R, int max_mol_atoms) {
  return nq_molgw_config_ok(F, R) && max_mol_atoms > 0 && max_mol_atoms <= GM_MAX_ATOMS;
}
static int molgw_groups(int B, int nslices) {
  int groups = 256 / nslices;            // one workgroup per CU
  if (groups < 1) groups = 1;
  if (groups > B) groups = B;
  return groups;
}
// int32 workspace: sched (int2 per padded pair slot) + seg (int2 per molecule and wavefront) + sched_ptr + hist + wlo
size_t nq_molgw_sched_ints(int E, int B) {
  return 2 * GM_BATCH * gm_max_batches(E, B) + 2 * (size_t)B * GM_NW + (size_t)B * (GM_NW + 1) + GM_MAX_BINS + GM_NW + 1 + (size_t)B + 16;
}
size_t nq_molgw_sched_slots(int E, int B) { return GM_BATCH * gm_max_batches(E, B); }
size_t nq_molgw_rec_floats(int E, int B) { return gm_max_batches(E, B) * (GM_PA_DWORDS + GM_PG_DWORDS) + 16; }   // PA once per step, PG once per backward sweep
size_t nq_molgw_part_floats(int F, int B) { const int ns = F / GM_CH; return (size_t)molgw_groups(B, ns) * ns * GM_NW * GM_PART_FLOATS; }

struct MolGwBufs { int2* sched; int2* seg; int* sched_ptr; int* hist; int* wlo; int* order; };
static MolGwBufs molgw_bufs(int* base, int E, int B) {
  MolGwBufs b;
  b.sched = reinterpret_cast<int2*>(base);
  b.seg = b.sched + GM_BATCH * gm_max_batches(E, B);
  b.sched_ptr = reinterpret_cast<int*>(b.seg + (size_t)B * GM_NW);
  b.hist = b.sched_ptr + (size_t)B * (GM_NW + 1);
  b.wlo = b.hist + GM_MAX_BINS;
  b.order = B <= GM_ORDER_MAX ? b.wlo + GM_NW + 1 : nullptr;   // larger batches: dealt by index
  return b;
}

// once per step, after the window records: histogram of the window starts over the pairs -> row windows of the wavefronts -> per-molecule pair lists -> expanded A operands
int nq_molgw_schedule(hipStream_t st, const NqGraphView& g, const int* dst, const float* RW, int R, int cap, int* sched_ints, float* recs) {
  NQ_PROF(st, "pair_schedule");
  const MolGwBufs b = molgw_bufs(sched_ints, g.E, g.B);
  NQ_HIP(hipMemsetAsync(b.hist, 0, GM_MAX_BINS * sizeof(int), st));
  int blocks = nq_cdiv(g.E, 256 * 8);
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  hipLaunchKernelGGL(k_pair_hist, dim3(blocks), dim3(256), 0, st, RW, dst, g.col, g.E, b.hist);
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_pair_windows, dim3(1), dim3(64), 0, st, b.hist, R - FWIN + 1, b.wlo);
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_pair_sched, dim3(g.B), dim3(64), 0, st, g, dst, RW, b.wlo, cap < GM_MAX_ATOMS ? cap : GM_MAX_ATOMS, b.sched, b.sched_ptr, b.seg);
  NQ_LAUNCH_CHECK();
  if (b.order) {
    int n2 = 1;
    while (n2 < g.B) n2 <<= 1;
    const size_t lds = (size_t)n2 * sizeof(unsigned);
    NQ_DYN_LDS(k_mol_order, lds);
    hipLaunchKernelGGL(k_mol_order, dim3(1), dim3(1024), lds, st, g, cap < GM_MAX_ATOMS ? cap : GM_MAX_ATOMS, n2, b.order);
    NQ_LAUNCH_CHECK();
  }
  const int nseg = g.B * GM_NW;
  hipLaunchKernelGGL(k_pair_arec, dim3(nseg), dim3(256), 0, st, b.sched, b.seg, RW, (const float*)nullptr, nseg, reinterpret_cast<u4*>(recs));   // rho half
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
// once per backward sweep, after nq_geom_tan: geometry + tangent + atom indices of every pair in schedule order
int nq_molgw_geometry(hipStream_t st, const NqGraphView& g, const float* RW, const float* TD, const float* TR, const int* sched_ints, float* recs) {
  NQ_PROF(st, "pair_geometry");
  const MolGwBufs b = molgw_bufs(const_cast<int*>(sched_ints), g.E, g.B);
  float* PG = recs + gm_max_batches(g.E, g.B) * GM_PA_DWORDS;
  const int nseg = g.B * GM_NW;
  hipLaunchKernelGGL(k_pair_grec, dim3(nseg), dim3(256), 0, st, b.sched, b.seg, g.geom, TR, nseg, PG);
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_pair_arec, dim3(nseg), dim3(256), 0, st, b.sched, b.seg, RW, TD, nseg, reinterpret_cast<u4*>(recs));   // t_d drho half
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

int nq_gwr_mol(hipStream_t st, const NqGraphView& g, int F, int R, int max_mol_atoms, const float* XH, const float* V, const float* TXH, const float* TV,
               const float* GX, const float* GV, const float* GTX, const float* GTV, const int* sched_ints, const float* recs, float* part, float* gWr,
               float* gbr, bool accumulate) {
  NQ_PROF(st, "gwr_mol");
  const MolGwBufs b = molgw_bufs(const_cast<int*>(sched_ints), g.E, g.B);
  GwrMolArgs q;
  q.g = g; q.F = F; q.nslices = F / GM_CH; q.groups = molgw_groups(g.B, q.nslices);
  q.max_atoms = max_mol_atoms < GM_MAX_ATOMS ? max_mol_atoms : GM_MAX_ATOMS;   // molecules above it are not in the schedule (pair-row kernels)
  q.XH = XH; q.V = V; q.TXH = TXH; q.TV = TV; q.GX = GX; q.GV = GV; q.GTX = GTX; q.GTV = GTV;
  q.seg = b.seg; q.order = b.order; q.part = part;
  const u4* PA = reinterpret_cast<const u4*>(recs);
  const unsigned* PG = reinterpret_cast<const unsigned*>(recs + gm_max_batches(g.E, g.B) * GM_PA_DWORDS);
  const size_t lds = (size_t)q.max_atoms * GM_ATOM_BYTES + GM_NW * GM_RING_BYTES;
  NQ_DYN_LDS(k_gwr_mol, lds);
  hipLaunchKernelGGL(k_gwr_mol, dim3(q.groups * q.nslices), dim3(GM_THREADS), lds, st, q, PA, PG);
  NQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_gwr_mol_reduce, dim3(nq_cdiv((long)(R + 1) * 3 * F, 256 / GMR_Q)), dim3(256), 0, st, part, b.wlo, q.groups, q.nslices, R, F, gWr, gbr, accumulate ? 1 : 0);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
