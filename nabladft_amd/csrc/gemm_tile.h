// The fp32 MFMA tile engine of the dense contractions (shared by gemm.hip and the kernel lab under scripts/lab/).
//
// What it computes is described at the top of gemm.hip.  How (round 3, rebuilt for gfx950):
//   * v_mfma_f32_32x32x2_f32 (exact f32, 64 cycles per instruction and SIMD): every wavefront owns TM x TN accumulators of 32x32.
//   * Both operand tiles are double-buffered in LDS: the global loads of k-tile t+1 are issued before the MFMAs of k-tile t and land in
//     the other buffer after them -> ONE barrier per k-tile, the load latency sits under the matrix pipe.
//   * k is consumed in blocks of 8: lane (r, h) of a 32-row sub-tile owns the four consecutive k's  kb + 4h + {0..3}  and feeds them to four
//     successive MFMAs (MFMA i contracts k = kb + i and kb + 4 + i).  For a K-contiguous operand ([row][k] in LDS, row stride BK + 4 floats)
//     that is one conflict-free ds_read_b128 per sub-tile and k-block instead of four ds_read_b32; an MN-contiguous operand ([k][col] in LDS)
//     reads four conflict-free ds_read_b32.  The summation order over k is fixed by this assignment (deterministic, not the natural order).
//   * Tiles are numbered so that the column tiles of one row slab run back to back on ONE XCD (block b executes on XCD b % 8): the slab's
//     A rows are fetched from HBM once and re-read from that XCD's L2.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { EPI_STORE = 0, EPI_SILU = 1, EPI_ACC = 2, EPI_PARTIAL = 3, EPI_SILU_RES = 4, EPI_DSILU = 5, EPI_RES = 6, EPI_DSILU2 = 7 };   // 4: C = v, C2 = ea * resid + eb * silu(v); 7: C = v, C2 = v * silu'(aux)

struct GemmArgs {
  const float* A; const float* B; float* C; const float* bias; float* C2;
  int M, N, K, lda, ldb, ldc;
  int k_per_split;       // EPI_PARTIAL: K range per blockIdx.z
  long part_stride;      // EPI_PARTIAL: floats between partial slabs
  float* bpart;          // EPI_PARTIAL, optional: per-split column sums of A over rows < brows (bias gradient), [splits][M]
  int brows;
  // spherical (row-mapped) launches: the operands are packed irreps tensors [rows][ncomp][F]; the logical row q of order L is the pair
  // (r, m) = (q / w, q % w), w = 2L+1, stored at packed row r * ncomp + L*L + m
  int rm_rows, rm_ncomp, rm_w, rm_base;   // RM == 2 (weight gradient of one order): w / base given here
  int rm_s;                               // RM == 3 (weight gradients of all orders): splits per component; blockIdx.z = component * rm_s + split
  const float* Bz[7];                     // RM == 1 (forward / input gradient of all orders, blockIdx.z = L): per-order weights
  const float* resid; float ea, eb;       // EPI_SILU_RES (resid nullable, same leading dimension as C)
  // batched launches of the tile engines (round 4: the spherical linears as one plain strided product per packed component): blockIdx.y = z shifts the
  // operands, A += z * bsA, C += z * bsC, B = Bz[order of component z] (bz) or B + z * bsB; bias and bias-gradient partials belong to z = 0 only
  int nbatch, bz; long bsA, bsB, bsC;
  // round 6: B operand already split (gemm_split.h PreStageB): three bf16 planes [piece][N][K], K-contiguous, made once per optimiser step (nq_gemm_presplit_kn):
  // the input-gradient products then load their weight tile with three 16-byte loads per thread instead of eight 4-byte loads + the split arithmetic
  const void* Bpre; int ldbpre; int bpre_plane_bytes;
};
__device__ __forceinline__ void gemm_apply_batch(GemmArgs& p) {
  const int z = blockIdx.y;
  const int L = z >= 36 ? 6 : z >= 25 ? 5 : z >= 16 ? 4 : z >= 9 ? 3 : z >= 4 ? 2 : z >= 1 ? 1 : 0;
  // (static indices only: a dynamically indexed member would move the whole argument block to scratch memory)
  const float* bz = L == 0 ? p.Bz[0] : L == 1 ? p.Bz[1] : L == 2 ? p.Bz[2] : L == 3 ? p.Bz[3] : L == 4 ? p.Bz[4] : L == 5 ? p.Bz[5] : p.Bz[6];
  p.A += (long)z * p.bsA;
  p.B = p.bz ? bz : p.B + (long)z * p.bsB;
  p.C += (long)z * p.bsC;
  if (z > 0) { p.bias = nullptr; p.bpart = nullptr; }
}

typedef unsigned int gemm_u32x4 __attribute__((ext_vector_type(4)));

// One operand tile: ROWS (the M or N extent) x BK, staged by NT threads.
//   KC  (K-contiguous source: element (r, k) at src[r*ld + k])   LDS image [row][k],  row stride BK + 4   (16-B aligned rows; 16 rows cover 64 banks)
//   !KC (MN-contiguous source: element (k, r) at src[k*ld + r])  LDS image [k][col],  row stride ROWS + 4
// Global -> registers goes through a buffer descriptor rebuilt per tile (base = the tile's first element, num_records = the bytes up to the
// tile's last valid element): buffer_load_dwordx4 with ONE 32-bit lane offset and scalar k / row-group offsets -- no per-load 64-bit vector
// address arithmetic, no exec-masked branches, and the hardware range check returns zeros for rows past the operand's end (KC: rows >= R,
// !KC: k >= kend).  What the check cannot see is zeroed or harmless: KC k-tail floats are zeroed by a select when the registers are written
// to LDS; !KC columns past R feed only output rows / columns that are never stored.
// Requirements (host-checked, else the generic kernel in gemm.hip runs): 16-B aligned base, ld % 4 == 0, K % 4 == 0 (KC) resp. R % 4 == 0 (!KC).
template <bool KC, int ROWS, int BK, int NT>
struct GemmStage {
  static constexpr int STRIDE = KC ? BK + 4 : ROWS + 4;
  static constexpr int FLOATS = (KC ? ROWS : BK) * STRIDE;
  static constexpr int NV = ROWS * BK / 4 / NT;   // float4 per thread and k-tile
  static constexpr int LPR = KC ? BK / 4 : ROWS / 4;   // lanes per tile row (KC: row = operand row, !KC: row = k)
  static_assert(ROWS * BK / 4 % NT == 0 && NT % LPR == 0, "tile not divisible by the workgroup");

  static __device__ __forceinline__ __amdgpu_buffer_rsrc_t descriptor(const float* __restrict__ src, long ld, int r0, int R, int kbeg, int kend) {
    const float* base = KC ? src + (long)r0 * ld + kbeg : src + (long)kbeg * ld + r0;
    const long rr = min(R - r0, ROWS), kl = kend - kbeg;
    const long bytes = (KC ? (rr - 1) * ld + kl : (kl - 1) * ld + rr) * 4;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)min(bytes, 0xffffffffL), 0x00020000);
  }
  // byte offset of this thread's first float4 inside the tile (k-tile 0)
  static __device__ __forceinline__ int lane_offset(int ld) {
    const int t = threadIdx.x;
    return ((t / LPR) * ld + (t % LPR) * 4) * 4;
  }
  // krel = first k of the k-tile relative to the descriptor's kbeg
  static __device__ __forceinline__ void fetch(float4 (&v)[NV], __amdgpu_buffer_rsrc_t rsrc, int voff, int ld, int krel) {
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int soff = KC ? (it * (NT / LPR) * ld + krel) * 4 : (it * (NT / LPR) + krel) * ld * 4;
      const gemm_u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
      v[it] = make_float4(__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w));
    }
  }
  // registers -> LDS (one ds_write_b128 per float4; both images keep the float4 contiguous).  kzero (KC): this thread's k-quad lies past kend.
  static __device__ __forceinline__ void stash(float* __restrict__ tile, const float4 (&v)[NV], bool kzero) {
    const int t = threadIdx.x;
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int idx = t + NT * it;
      float4 x = v[it];
      if (KC && kzero) x = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(tile + (idx / LPR) * STRIDE + (idx % LPR) * 4) = x;
    }
  }
  // MFMA operand of the 32-row sub-tile starting at `base`, k-block kb: lane (lr, lk) -> k = kb + 4 lk + {0..3}
  static __device__ __forceinline__ float4 frag(const float* __restrict__ tile, int base, int kb, int lr, int lk) {
    if (KC) return *reinterpret_cast<const float4*>(tile + (base + lr) * STRIDE + kb + 4 * lk);
    const float* q = tile + (kb + 4 * lk) * STRIDE + base + lr;
    return make_float4(q[0], q[STRIDE], q[2 * STRIDE], q[3 * STRIDE]);
  }
};

__device__ __forceinline__ float gemm_f4(const float4& f, int i) { return i == 0 ? f.x : i == 1 ? f.y : i == 2 ? f.z : f.w; }

// tile id of workgroup b among nwg (bijective): the workgroups of XCD x = b % 8 get one contiguous range of ids in dispatch order
__device__ __forceinline__ int gemm_xcd_tile(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = b & 7, s = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}

// Epilogue addressing: wave-uniform row base (scalar registers: tile origin + (reg part of the row) * ldc) + ONE 32-bit byte offset per lane -- the
// scalar-base form of global_load / global_store, no per-access 64-bit vector arithmetic (which the compiler would hoist out of the persistent
// loop and spill).  C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// FULL: the tile lies inside the matrix -> straight-line accesses.
// The auxiliary tile is READ through a buffer descriptor (base = its first element, num_records = the bytes to the matrix end, 0 for a null
// source): unconditional loads, rows past the end return 0, columns past N return neighbours that are never stored.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gemm_aux_descriptor(const float* src, long tile_off, long total_floats) {
  const long bytes = src ? (total_floats - tile_off) * 4 : 0;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src ? src + tile_off : src), 0, (int)min(bytes, 0xffffffffL), 0x00020000);
}
__device__ __forceinline__ void gemm_load_aux(f32x16& x, __amdgpu_buffer_rsrc_t rsrc, int ldc, int rl, int cl) {
  const int boff = (rl * ldc + cl) * 4;
#pragma unroll
  for (int r = 0; r < 16; ++r) x[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, boff, ((r & 3) + 8 * (r >> 2)) * ldc * 4, 0));
}
// One 32x32 accumulator -> C (and C2); x = the values of the auxiliary tile (EPI_ACC: C itself, else resid), fetched earlier by gemm_load_aux.
template <int EPI, bool FULL>
__device__ __forceinline__ void gemm_store_acc(const f32x16& a, const f32x16& x, const GemmArgs& p, float* __restrict__ tbase, int rl, int cl, int rows_left,
                                               int cols_left, float bv, long tile_off) {
  if (!FULL && cl >= cols_left) return;
  const unsigned boff = (unsigned)(rl * p.ldc + cl) * 4u;   // byte offset of (rl, cl) inside the tile's row band (< 2^32: 128 rows)
  const size_t row_bytes = (size_t)p.ldc * 4;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rc = (r & 3) + 8 * (r >> 2);
    if (!FULL && rl + rc >= rows_left) continue;
    float* cptr = reinterpret_cast<float*>(reinterpret_cast<char*>(tbase) + rc * row_bytes + boff);
    const float v = a[r] + bv;
    if (EPI == EPI_ACC) *cptr = x[r] + v;
    else if (EPI == EPI_DSILU) *cptr = p.eb * v * nq_dsilu_fast(x[r]);   // 5: C = eb * v * silu'(aux)   (adjoint of the activation of the layer below)
    else if (EPI == EPI_RES) *cptr = p.ea * x[r] + v;                    // 6: C = ea * aux + v          (skip connection of the adjoint)
    else *cptr = v;
    if (EPI == EPI_SILU || EPI == EPI_SILU_RES || EPI == EPI_DSILU2) {
      float* c2 = reinterpret_cast<float*>(reinterpret_cast<char*>(p.C2 + tile_off) + rc * row_bytes + boff);
      if (EPI == EPI_SILU) *c2 = nq_silu(v);
      else if (EPI == EPI_DSILU2) *c2 = v * nq_dsilu_fast(x[r]);   // 7: C = v, C2 = v * silu'(aux): tangent of a Linear + SiLU layer (aux = the primal pre-activation)
      else *c2 = p.resid ? p.ea * x[r] + p.eb * nq_silu_fast(v) : p.eb * nq_silu_fast(v);   // 4: C = v, C2 = ea * resid + eb * silu(v)   (resid nullable)
    }
  }
}

// BM x BN x BK tiles, NWM x NWN wavefronts (each (BM/NWM) x (BN/NWN) = TM x TN MFMA tiles).
// PERSISTENT over tiles: workgroup w contracts the tiles w, w + gridDim.x, ... (tile id = (split, row tile, column tile), column fastest).  The
// (tile, k-tile) steps form ONE software pipeline: the first k-tile of the next tile is fetched under the last MFMAs of the current one, and
// the C stores of a finished tile are issued and left in flight while the next tile's MFMAs run.  (A one-tile-per-workgroup grid holds its
// CU slot until the stores have drained and every workgroup of the chip reaches that point together: measured, the store phase then adds to
// the MFMA time instead of hiding under it -- 0.19 vs 0.14 ms without stores for [256728 x 128] x [128 x 256].)
// Epilogue operands that are READ (bias; the C / aux tile of EPI_ACC, EPI_DSILU, EPI_RES, EPI_SILU_RES) are fetched before the MFMAs of the
// tile's last k-tile, so nothing in the store sequence waits on memory.
// WPE: wavefronts per SIMD the register allocation must leave room for (workgroups per CU * NWM * NWN / 4).
// ABL (lab only): 1 = no C stores, 2 = no global loads after the first k-tile, 3 = both.
template <bool A_KC, bool B_KC, int EPI, int BM, int BN, int BK, int NWM, int NWN, int WPE = 2, int ABL = 0, bool BATCH = false>
__global__ __launch_bounds__(NWM * NWN * 64, WPE) void k_gemm2(GemmArgs p) {
  if (BATCH) gemm_apply_batch(p);
  constexpr int NT = NWM * NWN * 64, TM = BM / NWM / 32, TN = BN / NWN / 32;
  static_assert(TM >= 1 && TN >= 1 && BK % 8 == 0, "bad tile");
  using SA = GemmStage<A_KC, BM, BK, NT>;
  using SB = GemmStage<B_KC, BN, BK, NT>;
  constexpr int BUF = SA::FLOATS + SB::FLOATS;
  constexpr bool AUX = EPI == EPI_ACC || EPI == EPI_DSILU || EPI == EPI_RES || EPI == EPI_SILU_RES || EPI == EPI_DSILU2;
  __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int per_split = ntn * ntm;
  const int nsplit = EPI == EPI_PARTIAL ? (p.K + p.k_per_split - 1) / p.k_per_split : 1;
  const int ntiles = per_split * nsplit;
  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  const int wrow0 = wm * (TM * 32), wcol0 = wn * (TN * 32);
  const int lr = lane & 31, lk = lane >> 5;
  const int voff_a = SA::lane_offset(p.lda), voff_b = SB::lane_offset(p.ldb);
  const int kq_a = A_KC ? (threadIdx.x % SA::LPR) * 4 : 0, kq_b = B_KC ? (threadIdx.x % SB::LPR) * 4 : 0;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // coordinates of the current tile
  int split = tile / per_split, rem = tile - split * per_split;
  int m0 = (rem / ntn) * BM, n0 = (rem % ntn) * BN;
  int kbeg = EPI == EPI_PARTIAL ? split * p.k_per_split : 0;
  int kend = EPI == EPI_PARTIAL ? min(p.K, kbeg + p.k_per_split) : p.K;
  int k0 = kbeg;
  __amdgpu_buffer_rsrc_t da = SA::descriptor(p.A, p.lda, m0, p.M, kbeg, kend), db = SB::descriptor(p.B, p.ldb, n0, p.N, kbeg, kend);

  float bsum = 0.f;
  float4 ra[SA::NV], rb[SB::NV];
  SA::fetch(ra, da, voff_a, p.lda, 0);
  SB::fetch(rb, db, voff_b, p.ldb, 0);
  SA::stash(lds, ra, k0 + kq_a >= kend);
  SB::stash(lds + SA::FLOATS, rb, k0 + kq_b >= kend);
  __syncthreads();
  int par = 0;
  while (true) {
    const float* As = lds + par * BUF;
    const float* Bs = As + SA::FLOATS;
    // the step after this one: the next k-tile of this tile, or the first k-tile of this workgroup's next tile
    const bool last_k = k0 + BK >= kend;
    int tile_n = tile, split_n = split, m0_n = m0, n0_n = n0, kbeg_n = kbeg, kend_n = kend, k0_n = k0 + BK;
    if (last_k) {
      tile_n = tile + (int)gridDim.x;
      split_n = tile_n / per_split;
      const int rem_n = tile_n - split_n * per_split;
      m0_n = (rem_n / ntn) * BM; n0_n = (rem_n % ntn) * BN;
      kbeg_n = EPI == EPI_PARTIAL ? split_n * p.k_per_split : 0;
      kend_n = EPI == EPI_PARTIAL ? min(p.K, kbeg_n + p.k_per_split) : p.K;
      k0_n = kbeg_n;
    }
    const bool more = !last_k || tile_n < ntiles;
    if (last_k && more) {
      da = SA::descriptor(p.A, p.lda, m0_n, p.M, kbeg_n, kend_n);
      db = SB::descriptor(p.B, p.ldb, n0_n, p.N, kbeg_n, kend_n);
    }
    if (more && !(ABL & 2)) {   // global -> registers now, registers -> the other LDS buffer after the MFMAs
      SA::fetch(ra, da, voff_a, p.lda, k0_n - kbeg_n);
      SB::fetch(rb, db, voff_b, p.ldb, k0_n - kbeg_n);
    }
    // epilogue inputs of this tile (its last k-tile only): they arrive under the MFMAs below
    const long tile_off = (long)m0 * p.ldc + n0;
    const bool full = m0 + BM <= p.M && n0 + BN <= p.N;   // workgroup-uniform
    float bias_v[TN];
    f32x16 aux[AUX ? TM : 1][AUX ? TN : 1];
    if (last_k && !(ABL & 1)) {
      if (EPI != EPI_PARTIAL) {   // bias[n0 + column]: descriptor over the N valid entries (none for a null bias) -> zeros outside
        const __amdgpu_buffer_rsrc_t dbias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? p.N * 4 : 0, 0x00020000);
#pragma unroll
        for (int j = 0; j < TN; ++j) bias_v[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dbias, (n0 + wcol0 + j * 32 + lr) * 4, 0, 0));
      } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) bias_v[j] = 0.f;
      }
      if (AUX) {
        const __amdgpu_buffer_rsrc_t daux = gemm_aux_descriptor(EPI == EPI_ACC ? p.C : p.resid, tile_off, (long)(p.M - 1) * p.ldc + p.N);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) gemm_load_aux(aux[AUX ? i : 0][AUX ? j : 0], daux, p.ldc, wrow0 + i * 32 + 4 * lk, wcol0 + j * 32 + lr);
      }
    }
    if (EPI == EPI_PARTIAL && !A_KC) {
      // bias gradient for free: column sums of the A tile (= gy rows) that is already in LDS, primal rows only
      if (p.bpart && n0 == 0 && threadIdx.x < BM) {
        const int kmax = min(BK, p.brows - k0);
        if (kmax == BK) {   // whole k-tile: independent reads, four partial sums (the common case; a dependent chain would wait on LDS BK times)
          float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
          for (int kk = 0; kk < BK; kk += 4) {
            s0 += As[kk * SA::STRIDE + threadIdx.x]; s1 += As[(kk + 1) * SA::STRIDE + threadIdx.x];
            s2 += As[(kk + 2) * SA::STRIDE + threadIdx.x]; s3 += As[(kk + 3) * SA::STRIDE + threadIdx.x];
          }
          bsum += (s0 + s1) + (s2 + s3);
        } else {
          for (int kk = 0; kk < kmax; ++kk) bsum += As[kk * SA::STRIDE + threadIdx.x];
        }
      }
    }
    // fragments of k-block kb+8 are read before the MFMAs of k-block kb issue (two register sets, statically indexed after unrolling)
    float4 fa[2][TM], fb[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[0][i] = SA::frag(As, wrow0 + 32 * i, 0, lr, lk);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[0][j] = SB::frag(Bs, wcol0 + 32 * j, 0, lr, lk);
#pragma unroll
    for (int kb = 0; kb < BK; kb += 8) {
      const int c = (kb >> 3) & 1;
      if (kb + 8 < BK) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[c ^ 1][i] = SA::frag(As, wrow0 + 32 * i, kb + 8, lr, lk);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[c ^ 1][j] = SB::frag(Bs, wcol0 + 32 * j, kb + 8, lr, lk);
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the reads of the NEXT k-block above this k-block's MFMAs (hipcc otherwise sinks them to their first use)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(gemm_f4(fa[c][i], s), gemm_f4(fb[c][j], s), acc[i][j], 0, 0, 0);
    }
    if (more) {
      float* An = lds + (par ^ 1) * BUF;
      SA::stash(An, ra, k0_n + kq_a >= kend_n);
      SB::stash(An + SA::FLOATS, rb, k0_n + kq_b >= kend_n);
    }
    if (last_k) {
      // ---- epilogue of the finished tile; the stores stay in flight under the next tile's MFMAs.
      if (EPI == EPI_PARTIAL && !A_KC) {
        if (p.bpart && n0 == 0 && threadIdx.x < BM && m0 + (int)threadIdx.x < p.M) p.bpart[(long)split * p.M + m0 + threadIdx.x] = bsum;
        bsum = 0.f;
      }
      // everything the stores read has been requested a whole k-tile of MFMAs ago: one explicit wait here (it returns at once) tells the
      // compiler so -- otherwise it guards every store block with its own vmcnt(0), which would also wait for the stores issued before it.
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
      float* tbase = p.C + tile_off + (EPI == EPI_PARTIAL ? (long)split * p.part_stride : 0L);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (ABL & 1) {   // keep the accumulators alive without the stores
#if defined(__HIP_DEVICE_COMPILE__)   // the "v" constraint only exists in the device pass
            asm volatile("" ::"v"(acc[i][j]));
#endif
          } else {
            const int rl = wrow0 + i * 32 + 4 * lk, cl = wcol0 + j * 32 + lr;
            if (full) gemm_store_acc<EPI, true>(acc[i][j], aux[AUX ? i : 0][AUX ? j : 0], p, tbase, rl, cl, BM, BN, bias_v[j], tile_off);
            else gemm_store_acc<EPI, false>(acc[i][j], aux[AUX ? i : 0][AUX ? j : 0], p, tbase, rl, cl, p.M - m0, p.N - n0, bias_v[j], tile_off);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
      if (!more) break;
    }
    tile = tile_n; split = split_n; m0 = m0_n; n0 = n0_n; kbeg = kbeg_n; kend = kend_n; k0 = k0_n;
    __syncthreads();
    par ^= 1;
  }
}
