// f32 products on the bf16 matrix pipe: every f32 operand value is split EXACTLY into three bf16 pieces, x = h + m + l (8 + 8 + 8 significand bits,
// round-to-nearest at every step, the remainders x - h and x - h - m are exact in f32), while its tile is staged into LDS; the product of two values is
// then the sum of the piece products  h h' + h m' + m h' + h l' + l h' + m m'  (|x - h| <= 2^-9 |x|, |x - h - m| <= 2^-18 |x|: the dropped products m l',
// l m', l l' and the residual of the split are <= 2^-25 |x y| together, below the 2^-24 rounding of an f32 product), each accumulated in f32 by
// v_mfma_f32_32x32x16_bf16.  One such instruction contracts 16 k's in 32 cycles where the exact-f32 v_mfma_f32_32x32x2_f32 needs 8 x 64: six of them
// are 2.67x the f32 pipe's rate (peak 2.5 PF / 6 = 416 TFLOP/s of f32-accurate products vs 157).  TERMS = 9 keeps all nine piece products.
// Measured against float64 (scripts/lab/gemm_lab.hip, tests/test_engine_gpu.py): the error is equal to or below the exact-f32 engine's on every shape.
//
// Same skeleton as k_gemm2 (gemm_tile.h): persistent workgroups over (tile, k-step) steps, two LDS buffers, one barrier per step, buffer-descriptor
// loads, C stores left in flight under the next tile, epilogue inputs fetched under the last step.  128 x 128 tiles, 4 wavefronts of 64 x 64 (2 x 2
// accumulators: per k16 step 12 ds_read_b128 feed 24 MFMAs = 64 B/clk/CU of LDS reads, half the LDS rate), 49 KB of LDS and <= 168 registers: three
// workgroups per CU, whose barriers and staging phases interleave on the matrix pipe.
// Inf / NaN: a non-finite operand value gives NaN (inf - inf in the split) where the f32 pipe would give inf.
#pragma once
#include "gemm_tile.h"
#include <type_traits>

#ifndef NQ_GEMM3_TERMS
#define NQ_GEMM3_TERMS 6   // piece products kept (lab builds: 3 = the two-piece product, 9 = all)
#endif
typedef __bf16 sp_bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 sp_bf2 __attribute__((ext_vector_type(2)));
typedef unsigned int sp_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned sp_pack(float a, float b) {
  sp_bf2 v;
  v[0] = (__bf16)a; v[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, v);
}
// (x0, x1) -> packed bf16 pairs h, m, l (convert, widen, subtract; the subtractions are exact).  CHEAP (lab ablation): no arithmetic.
// (Measured and dropped: v_dot2_f32_bf16 with a (-1, 0) operand as a one-instruction "subtract the widened half" -- it does not return the exact
// remainder on gfx950 and is not faster than the shift / and / packed-subtract sequence the compiler emits for the lines below.)
template <bool CHEAP = false>
__device__ __forceinline__ void sp_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  if (CHEAP) { h = m = l = (__float_as_uint(x0) >> 16) | (__float_as_uint(x1) & 0xffff0000u); return; }
  h = sp_pack(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = sp_pack(r0, r1);
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
  l = sp_pack(s0, s1);
}

// One operand tile: 128 rows (the M or N extent) x 16 k, staged by 256 threads into three planes (h, m, l) of [row][16 bf16] = 32 bytes per row, unpadded:
// a quarter-wavefront's ds_read_b128 (8 rows x 16 bytes, 32 bytes apart) covers all 64 banks once, and so do the ds_write_b128 below.
//   KC source ([row][k] in memory): thread t loads the 8 consecutive k's  8 (t & 1) ..  of row t >> 1 (two 16-byte loads), splits them and writes one
//   ds_write_b128 per plane.
//   MC source ([k][col] in memory): wavefront w loads the 8 k rows  8 (w & 1) ..  of the 64 columns  64 (w >> 1) + lane  (dword loads, 256 contiguous bytes per
//   row and wavefront); the thread then holds 8 consecutive k of ONE column = the same single ds_write_b128 per plane (the transposition costs nothing).
template <bool KC>
struct SplitStage {
  static constexpr int ROWS = 128, BK = 16, NT = 256;
  static constexpr int PITCH = 32, PLANE = ROWS * PITCH, BYTES = 3 * PLANE;
  static constexpr int NREG = 8, NLOADS = KC ? 2 : 8;

  static __device__ __forceinline__ __amdgpu_buffer_rsrc_t descriptor(const float* __restrict__ src, long ld, int r0, int R, int kbeg, int kend) {
    const float* base = KC ? src + (long)r0 * ld + kbeg : src + (long)kbeg * ld + r0;
    const long rr = min(R - r0, ROWS), kl = kend - kbeg;
    const long bytes = (KC ? (rr - 1) * ld + kl : (kl - 1) * ld + rr) * 4;
    // everything here is workgroup-uniform; say so (readfirstlane), or a descriptor the compiler cannot prove uniform costs a waterfall loop around every load
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    const int nrec = __builtin_amdgcn_readfirstlane((int)max(0L, min(bytes, 0xffffffffL)));   // r0 >= R: zero range
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo), 0, nrec, 0x00020000);
  }
  static __device__ __forceinline__ int lane_offset(int ld) {
    const int t = threadIdx.x;
    if (KC) return ((t >> 1) * ld + (t & 1) * 8) * 4;
    return (((t >> 6) & 1) * 8 * ld + (t >> 7) * 64 + (t & 63)) * 4;
  }
  static __device__ __forceinline__ void fetch(float (&v)[NREG], __amdgpu_buffer_rsrc_t rsrc, int voff, int ld, int krel_) {
    const int krel = __builtin_amdgcn_readfirstlane(krel_);   // uniform (see descriptor()): the scalar-offset operand must not become a waterfall loop
    if (KC) {
      const gemm_u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, krel * 4, 0), y = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, krel * 4 + 16, 0);
      v[0] = __uint_as_float(x.x); v[1] = __uint_as_float(x.y); v[2] = __uint_as_float(x.z); v[3] = __uint_as_float(x.w);
      v[4] = __uint_as_float(y.x); v[5] = __uint_as_float(y.y); v[6] = __uint_as_float(y.z); v[7] = __uint_as_float(y.w);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (i + krel) * ld * 4, 0));
    }
  }
  // registers -> the three planes.  KC: krem = kend - (first k of the k-tile): quads of k past it are zeroed (the range check only sees the row end)
  // KTAIL = false: the caller guarantees whole k16 steps (no zeroing code at all)
  template <bool CHEAP = false, bool KTAIL = true>
  static __device__ __forceinline__ void stash(char* __restrict__ tile, const float (&v)[NREG], int krem) {
    const int t = threadIdx.x;
    float x[8];
    if (KC && KTAIL) {
      const int kq = (t & 1) * 8;
      const bool z0 = kq >= krem, z1 = kq + 4 >= krem;
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = (e < 4 ? z0 : z1) ? 0.f : v[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = v[e];
    }
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) sp_split2<CHEAP>(x[2 * e], x[2 * e + 1], h[e], m[e], l[e]);
    char* q = KC ? tile + (t >> 1) * PITCH + (t & 1) * 16 : tile + ((t >> 7) * 64 + (t & 63)) * PITCH + ((t >> 6) & 1) * 16;
    *reinterpret_cast<gemm_u32x4*>(q) = gemm_u32x4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<gemm_u32x4*>(q + PLANE) = gemm_u32x4{m[0], m[1], m[2], m[3]};
    *reinterpret_cast<gemm_u32x4*>(q + 2 * PLANE) = gemm_u32x4{l[0], l[1], l[2], l[3]};
  }
  // MFMA operand of plane pl, 32-row sub-tile at `base`: lane (lr, lk) -> k = 8 lk + {0..7}
  static __device__ __forceinline__ sp_bf8 frag(const char* __restrict__ tile, int pl, int base, int lr, int lk) {
    return *reinterpret_cast<const sp_bf8*>(tile + pl * PLANE + (base + lr) * PITCH + lk * 16);
  }
};


// B operand from PRE-SPLIT planes (GemmArgs::Bpre: [piece][N][K] bf16, K-contiguous): same LDS image and fragments as SplitStage<true>, no arithmetic.
// Thread t loads the 8 consecutive k's  8 (t & 1) ..  of row t >> 1 of each plane (three 16-byte loads) and writes one ds_write_b128 per plane.
// N % 128 == 0 and K % 16 == 0 (host-checked): nothing of a valid tile is out of range; a tile past the end gets a zero-range descriptor.
struct PreStageB {
  static constexpr int ROWS = 128, BK = 16, NT = 256;
  static constexpr int PITCH = 32, PLANE = ROWS * PITCH, BYTES = 3 * PLANE;
  static constexpr int NREG = 12, NLOADS = 3;
  static __device__ __forceinline__ __amdgpu_buffer_rsrc_t descriptor(const void* planes, long ldk, long plane_bytes, int r0, int R, int kbeg, int kend) {
    const char* base = reinterpret_cast<const char*>(planes) + ((long)r0 * ldk + kbeg) * 2;
    const long rr = min(R - r0, ROWS), kl = kend - kbeg;
    const long bytes = rr > 0 ? 2 * plane_bytes + ((rr - 1) * ldk + kl) * 2 : 0;
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    const int nrec = __builtin_amdgcn_readfirstlane((int)max(0L, min(bytes, 0xffffffffL)));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo), 0, nrec, 0x00020000);
  }
  static __device__ __forceinline__ int lane_offset(int ldk) { const int t = threadIdx.x; return ((t >> 1) * ldk + (t & 1) * 8) * 2; }
  // `plane_bytes` travels in the argument that carries the leading dimension of the other stages
  static __device__ __forceinline__ void fetch(float (&v)[NREG], __amdgpu_buffer_rsrc_t rsrc, int voff, int plane_bytes, int krel_) {
    const int krel = __builtin_amdgcn_readfirstlane(krel_);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const gemm_u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, krel * 2 + pl * plane_bytes, 0);
      v[4 * pl] = __uint_as_float(x.x); v[4 * pl + 1] = __uint_as_float(x.y); v[4 * pl + 2] = __uint_as_float(x.z); v[4 * pl + 3] = __uint_as_float(x.w);
    }
  }
  template <bool CHEAP = false, bool KTAIL = true>
  static __device__ __forceinline__ void stash(char* __restrict__ tile, const float (&v)[NREG], int) {
    const int t = threadIdx.x;
    char* q = tile + (t >> 1) * PITCH + (t & 1) * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      *reinterpret_cast<gemm_u32x4*>(q + pl * PLANE) = gemm_u32x4{__float_as_uint(v[4 * pl]), __float_as_uint(v[4 * pl + 1]), __float_as_uint(v[4 * pl + 2]), __float_as_uint(v[4 * pl + 3])};
  }
  static __device__ __forceinline__ sp_bf8 frag(const char* __restrict__ tile, int pl, int base, int lr, int lk) {
    return *reinterpret_cast<const sp_bf8*>(tile + pl * PLANE + (base + lr) * PITCH + lk * 16);
  }
};

struct SpStep { int tile, split, m0, n0, kbeg, kend, k0; };   // one (tile, k-tile) step of a workgroup; tile >= ntiles: past the end (kbeg == kend == 0)

// Software pipeline, prefetch distance TWO: a k16 step is 24 MFMAs = 768 cycles of the matrix pipe, less than one trip to L2 / HBM, so the loads of
// step s+2 are issued at the top of step s (two register sets), the registers of step s+1 are split and written to the other LDS buffer during
// step s, and step s's MFMAs read the buffer filled during step s-1.  Everything is branch-free up to the per-tile epilogue: past the last step the
// descriptors have zero range (the loads return zeros, the stash writes zeros nobody reads).
// ABL (lab only): 1 = no split arithmetic (the three planes get the truncated value), 2 = no global loads inside the loop, 4 = no ds_reads inside the loop
// KTAIL: the contraction length need not be a multiple of 16 (K-contiguous operands only; costs 18 VALU instructions per step).
// WBIAS (weight-gradient launches): also produce the bias gradient (GemmArgs::bpart).
// bid / nb: this workgroup's index and the number of workgroups that share the product (blockIdx.x / gridDim.x for a plain launch; a grouped launch deals
// ranges of its grid to several products: k_gemm3_tn_group below)
template <bool A_KC, bool B_KC, int EPI, int WPE = 3, bool KTAIL = true, bool WBIAS = false, int TERMS = NQ_GEMM3_TERMS, int ABL = 0, bool B_PRE = false>
__device__ __forceinline__ void gemm3_body(const GemmArgs& p, const int bid, const int nb) {
  constexpr int BM = 128, BN = 128, BK = 16, NWN = 2, TM = 2, TN = 2;
  using SA = SplitStage<A_KC>;
  using SB = std::conditional_t<B_PRE, PreStageB, SplitStage<B_KC>>;
  const int ldb_ = B_PRE ? p.bpre_plane_bytes : p.ldb;   // what the stage's fetch takes as its third argument (PreStageB: the plane stride)
  constexpr int BUF = SA::BYTES + SB::BYTES;   // 24 KB: two buffers (+ the bias-gradient lines) = 49 KB, three workgroups per CU
  constexpr int NLOADS = SA::NLOADS + SB::NLOADS;   // load instructions of one step's fetch
  constexpr bool AUX = EPI == EPI_ACC || EPI == EPI_DSILU || EPI == EPI_RES || EPI == EPI_SILU_RES || EPI == EPI_DSILU2;
  constexpr bool BIAS = EPI == EPI_PARTIAL && !A_KC && WBIAS;
  __shared__ __attribute__((aligned(16))) char lds[2 * BUF + (BIAS ? 2 * 128 * 4 : 0)];

  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int per_split = ntn * ntm;
  const int nsplit = EPI == EPI_PARTIAL ? (p.K + p.k_per_split - 1) / p.k_per_split : 1;
  const int ntiles = per_split * nsplit;
  if (bid >= ntiles) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  const int wrow0 = wm * (TM * 32), wcol0 = wn * (TN * 32);
  const int lr = lane & 31, lk = lane >> 5;
  const int voff_a = SA::lane_offset(p.lda), voff_b = SB::lane_offset(B_PRE ? p.ldbpre : p.ldb);

  auto first_step = [&](int tile) {
    SpStep s;
    s.tile = tile;
    if (tile >= ntiles) { s.split = 0; s.m0 = 0; s.n0 = 0; s.kbeg = 0; s.kend = 0; s.k0 = 0; return s; }
    s.split = tile / per_split;
    const int rem = tile - s.split * per_split;
    s.m0 = (rem / ntn) * BM; s.n0 = (rem % ntn) * BN;
    s.kbeg = EPI == EPI_PARTIAL ? s.split * p.k_per_split : 0;
    s.kend = EPI == EPI_PARTIAL ? min(p.K, s.kbeg + p.k_per_split) : p.K;
    s.k0 = s.kbeg;
    return s;
  };
  auto advance = [&](const SpStep& s) {
    if (s.k0 + BK < s.kend) { SpStep n = s; n.k0 += BK; return n; }
    return first_step(s.tile >= ntiles ? s.tile : s.tile + nb);
  };
  __amdgpu_buffer_rsrc_t da, db;
  auto descriptors = [&](const SpStep& s) {   // of the step's tile; zero range past the end
    const bool v = s.tile < ntiles;
    da = SA::descriptor(p.A, p.lda, v ? s.m0 : p.M, p.M, s.kbeg, s.kend);
    if constexpr (B_PRE) db = PreStageB::descriptor(p.Bpre, p.ldbpre, p.bpre_plane_bytes, v ? s.n0 : p.N, p.N, s.kbeg, s.kend);
    else db = SB::descriptor(p.B, p.ldb, v ? s.n0 : p.N, p.N, s.kbeg, s.kend);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // bias gradient of the weight-gradient contraction (EPI_PARTIAL, A = gy [rows][M]): column sums of A over the rows < brows, taken from the f32
  // registers on their way into LDS (a thread holds 8 of the step's 16 rows of ONE column); the two half-steps' sums meet in LDS at the end of the
  // tile, fixed order.  bnext: a tile's first step is staged while the previous tile is still being finished.
  const bool bias_on = BIAS && p.bpart != nullptr;   // (a grouped launch mixes products with and without a bias gradient)
  float bsum = 0.f, bnext = 0.f;
  float* lds_b = reinterpret_cast<float*>(lds + 2 * BUF);   // [2][128], only there for EPI_PARTIAL
  auto colsum = [&](const float (&v)[SA::NREG], const SpStep& s) {
    float o = 0.f;
    const int kb = s.k0 + (wave & 1) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (kb + i < p.brows) o += v[i];
    return o;
  };

  SpStep cur = first_step(bid), n1 = advance(cur), n2 = advance(n1);
  float ra[2][SA::NREG], rb[2][SB::NREG];
  descriptors(cur);
  SA::fetch(ra[0], da, voff_a, p.lda, 0);
  SB::fetch(rb[0], db, voff_b, ldb_, 0);
  if (n1.k0 == n1.kbeg) descriptors(n1);
  SA::fetch(ra[1], da, voff_a, p.lda, n1.k0 - n1.kbeg);
  SB::fetch(rb[1], db, voff_b, ldb_, n1.k0 - n1.kbeg);
  SA::template stash<false, KTAIL>(lds, ra[0], cur.kend - cur.k0);
  SB::template stash<false, KTAIL>(lds + SA::BYTES, rb[0], cur.kend - cur.k0);
  if (bias_on && cur.n0 == 0) bsum = colsum(ra[0], cur);
  __syncthreads();
  do {   // two steps per trip and ONE exit test: an exit between the halves would be a control-flow edge from the first half to the loop header, along which
         // the compiler must assume the first half's loads pending and guards every reuse of their registers with vmcnt(0)
#pragma unroll
    for (int u = 0; u < 2; ++u) {   // u = parity of the step: LDS buffer u holds it, register set u^1 the next one, register set u receives the one after
      const char* As = lds + u * BUF;
      const char* Bs = As + SA::BYTES;
      const bool last_k = cur.tile < ntiles && cur.k0 + BK >= cur.kend;   // (a trailing odd step past the end runs on zeros and stores nothing)
      const long tile_off = (long)cur.m0 * p.ldc + cur.n0;
      const bool full = cur.m0 + BM <= p.M && cur.n0 + BN <= p.N;
      // epilogue inputs of this tile FIRST (older than the prefetch below: the wait before the stores leaves the prefetch in flight)
      float bias_v[TN];
      f32x16 aux[AUX ? TM : 1][AUX ? TN : 1];
      if (last_k) {
        if (EPI != EPI_PARTIAL) {
          const __amdgpu_buffer_rsrc_t dbias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? p.N * 4 : 0, 0x00020000);
#pragma unroll
          for (int j = 0; j < TN; ++j) bias_v[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dbias, (cur.n0 + wcol0 + j * 32 + lr) * 4, 0, 0));
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
      descriptors(n2);   // every step (scalar work): a conditional update would be a block merge with loads pending, which costs a vmcnt(0)
      if (!(ABL & 2)) {
        SA::fetch(ra[u], da, voff_a, p.lda, n2.k0 - n2.kbeg);
        SB::fetch(rb[u], db, voff_b, ldb_, n2.k0 - n2.kbeg);
      }
      {
        sp_bf8 fa[3][TM], fb[3][TN];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          if (ABL & 4) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int e = 0; e < 8; ++e) { fa[pl][i][e] = (__bf16)(float)(lane + pl + i + e); fb[pl][i][e] = (__bf16)(float)(lane - pl - i - e); }
            continue;
          }
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[pl][i] = SA::frag(As, pl, wrow0 + 32 * i, lr, lk);
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[pl][j] = SB::frag(Bs, pl, wcol0 + 32 * j, lr, lk);
        }
        constexpr int PA[9] = {2, 2, 1, 1, 2, 0, 1, 0, 0}, PB[9] = {2, 1, 2, 1, 0, 2, 0, 1, 0};   // small terms first
#pragma unroll
        for (int tt = 9 - TERMS; tt < 9; ++tt)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[tt]][i], fb[PB[tt]][j], acc[i][j], 0, 0, 0);
      }
      {
        char* An = lds + (u ^ 1) * BUF;
        SA::template stash<(ABL & 1) != 0, KTAIL>(An, ra[u ^ 1], n1.kend - n1.k0);
        SB::template stash<(ABL & 1) != 0, KTAIL>(An + SA::BYTES, rb[u ^ 1], n1.kend - n1.k0);
        if (bias_on && n1.n0 == 0) {
          const float t2 = colsum(ra[u ^ 1], n1);
          if (n1.k0 == n1.kbeg) bnext = t2;
          else bsum += t2;
        }
      }
      if (last_k && bias_on) {
        if (cur.n0 == 0) lds_b[(wave & 1) * 128 + (wave >> 1) * 64 + lane] = bsum;
        __syncthreads();
        const int t = threadIdx.x;
        if (cur.n0 == 0 && t < BM && cur.m0 + t < p.M) p.bpart[(long)cur.split * p.M + cur.m0 + t] = lds_b[t] + lds_b[128 + t];
        bsum = bnext;
      }
      if (last_k) {
        // bias / aux were requested before this step's prefetch: wait for everything but the NLOADS youngest loads
        __builtin_amdgcn_s_waitcnt(0x0F70 | (NLOADS & 15) | ((NLOADS >> 4) << 14));
        float* tbase = p.C + tile_off + (EPI == EPI_PARTIAL ? (long)cur.split * p.part_stride : 0L);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int rl = wrow0 + i * 32 + 4 * lk, cl = wcol0 + j * 32 + lr;
            if (full) gemm_store_acc<EPI, true>(acc[i][j], aux[AUX ? i : 0][AUX ? j : 0], p, tbase, rl, cl, BM, BN, bias_v[j], tile_off);
            else gemm_store_acc<EPI, false>(acc[i][j], aux[AUX ? i : 0][AUX ? j : 0], p, tbase, rl, cl, p.M - cur.m0, p.N - cur.n0, bias_v[j], tile_off);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
          }
      }
      cur = n1; n1 = n2; n2 = advance(n2);
      __syncthreads();
    }
  } while (cur.tile < ntiles);
}

template <bool A_KC, bool B_KC, int EPI, int WPE = 3, bool KTAIL = true, bool WBIAS = false, int TERMS = NQ_GEMM3_TERMS, int ABL = 0, bool BATCH = false, bool B_PRE = false>
__global__ __launch_bounds__(256, WPE) void k_gemm3(GemmArgs p) {
  if (BATCH) gemm_apply_batch(p);
  gemm3_body<A_KC, B_KC, EPI, WPE, KTAIL, WBIAS, TERMS, ABL, B_PRE>(p, (int)blockIdx.x, (int)gridDim.x);
}

// Several weight-gradient contractions (EPI_PARTIAL: both operands row-major over the contracted rows) in ONE launch: workgroups [first[g], first[g + 1])
// belong to product g.  The five products of a PaiNN layer's reverse sweep share the chip's 512 workgroup slots in proportion to their rows x tiles:
// each workgroup streams ~10x more rows than when every product splits itself over all slots, and the per-split partial tiles (which a second kernel
// has to read back) shrink by the same factor.
#define GEMM_GROUP_MAX 5
struct GemmGroupArgs { GemmArgs a[GEMM_GROUP_MAX]; int first[GEMM_GROUP_MAX + 1]; int n; };
template <int UNUSED = 0>   // (a template only so that every translation unit that includes this header may hold its own copy)
__global__ __launch_bounds__(256, 2) void k_gemm3_tn_group(GemmGroupArgs q) {
  const int b = (int)blockIdx.x;
  // (static member indices only: a dynamically indexed argument block would live in scratch memory)
  const int g = b >= q.first[4] ? 4 : b >= q.first[3] ? 3 : b >= q.first[2] ? 2 : b >= q.first[1] ? 1 : 0;
  if (g == 0) gemm3_body<false, false, EPI_PARTIAL, 2, false, true>(q.a[0], b - q.first[0], q.first[1] - q.first[0]);
  else if (g == 1) gemm3_body<false, false, EPI_PARTIAL, 2, false, true>(q.a[1], b - q.first[1], q.first[2] - q.first[1]);
  else if (g == 2) gemm3_body<false, false, EPI_PARTIAL, 2, false, true>(q.a[2], b - q.first[2], q.first[3] - q.first[2]);
  else if (g == 3) gemm3_body<false, false, EPI_PARTIAL, 2, false, true>(q.a[3], b - q.first[3], q.first[4] - q.first[3]);
  else gemm3_body<false, false, EPI_PARTIAL, 2, false, true>(q.a[4], b - q.first[4], q.first[5] - q.first[4]);
}
