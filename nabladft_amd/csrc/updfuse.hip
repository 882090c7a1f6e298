// PaiNN update block (painn.py:535-548), forward and tangent sweeps, as ONE kernel per layer (round 6).
//   vec1, vec2 = split(vec_proj(vec_msg));  s = <vec1, vec2>;  n = sqrt(|vec2|^2 + 1e-8);  q = silu([x_msg | n] V1^T + c1);
//   (ya, yb, yc) = split(q V2^T + c2);  x_upd = x_msg + ya + yb s;  vec_upd = vec_msg + yc vec1
// Rounds 1-5 ran this as three products and two elementwise kernels per sweep (U, upd_a, V1, V2, upd_b): every intermediate went to HBM and came back,
// 42 N F floats of traffic per layer and sweep for 22 that are compulsory (4 read, 18 written: the reverse sweeps need every intermediate).  Here a workgroup
// owns 32 atoms x all F = 128 channels and keeps the chain on chip:
//   * four wavefronts, wavefront w = channels [32 w, 32 w + 32) of EVERY output block: its accumulators hold vec1 AND vec2 (columns f and F + f of the
//     vec_proj product) for the three components of its 32 atoms, later ya, yb, yc (columns f, F + f, 2F + f) -- the C/D layout of the 32x32 matrix
//     instruction (lane = column, register = row) puts everything an output element needs into ONE lane: s, n, x_upd, vec_upd are register arithmetic;
//   * products on the bf16 matrix pipe with the three-piece split of csrc/gemm_split.h (x = h + m + l exactly, six piece products: f32-accurate);
//     activations are split while they are staged into LDS ([k16 step][piece][row][16 bf16], conflict-free 16-byte fragment reads), the WEIGHTS are split
//     once per optimiser step into matrix-instruction fragment order (k_uf_presplit: VERDICT r5 item 2a) and stream from L2 as one 16-byte load per lane,
//     piece and fragment -- no LDS, no split arithmetic in the loop for the weight operand;
//   * n and q reach the next product through LDS (written as bf16 pieces straight from the accumulator layout);
//   * every intermediate the other sweeps read is still written (u, s, cat, zq, q, y) -- once, from registers.
// Other channel counts keep the unfused kernels (engine.hip).
#include "common.h"
#include "gemm_split.h"

#define UF_R 32                      // atoms per workgroup
#define UF_F 128
#define UF_NT 256
#define UF_PB96 (96 * 32 + 32)       // bytes of one (k16 step, piece) block of a 96-row operand: 32 bytes per row + 32 bytes that rotate the banks from step to step
#define UF_PB32 (32 * 32 + 32)
#define UF_A2 0                      // LDS offsets: [x_msg | n] of the second product, q of the third (the vec_msg operand of the first one dies before both)
#define UF_A3 (16 * 3 * UF_PB32)
#define UF_LDS (UF_A3 + 8 * 3 * UF_PB32)   // 76032 bytes (>= 8 * 3 * UF_PB96 = 74496): two workgroups per CU

#ifndef UF_ABLATE
#define UF_ABLATE 0   // development only (wrong results, timing): 1 no matrix instructions, 2 no weight-fragment loads, 4 no epilogue stores, 8 no activation staging loads
#endif
typedef unsigned int uf_u4 __attribute__((ext_vector_type(4)));

size_t nq_updfuse_frag_floats(int F) { return F == UF_F ? 2 * (size_t)(2 * F * F + F * 2 * F + 3 * F * F) * 6 / 4 : 0; }   // three pieces of two bytes per weight, two fragment orders (A W^T of the forward / tangent sweeps, G W of the reverse sweep)

// W [OUT][IN] (the product is A W^T) -> fragments [k16 step][column block of 32][piece][lane]: lane (n, kh) holds W[32 cb + n][16 ks + 8 kh + 0..7] as bf16
// mc = 1: the reverse products G W (W [K][N] row-major): the same fragments with the roles of the two indices exchanged -- lane (n, kh) holds W[16 ks + 8 kh + 0..7][32 cb + n]
struct UfSplitArgs { const float* W[6]; int OUT[6]; int IN[6]; uf_u4* out[6]; };   // matrices 0-2: forward order, 3-5: reverse order (mc)
__global__ __launch_bounds__(256) void k_uf_presplit(UfSplitArgs a) {
  int idx = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int mtx = 0; mtx < 6; ++mtx) {
    const bool mc = mtx >= 3;
    const int OUT = a.OUT[mtx], IN = a.IN[mtx], ncb = OUT / 32, total = (IN / 16) * ncb * 64;
    if (idx < total) {
      const int lane = idx & 63, cb = (idx >> 6) % ncb, ks = (idx >> 6) / ncb;
      float4 x, y;
      if (mc) {
        const float* src = a.W[mtx] + (long)(ks * 16 + (lane >> 5) * 8) * OUT + cb * 32 + (lane & 31);
        x = make_float4(src[0], src[OUT], src[2 * (long)OUT], src[3 * (long)OUT]);
        y = make_float4(src[4 * (long)OUT], src[5 * (long)OUT], src[6 * (long)OUT], src[7 * (long)OUT]);
      } else {
        const float* src = a.W[mtx] + (long)(cb * 32 + (lane & 31)) * IN + ks * 16 + (lane >> 5) * 8;
        x = *reinterpret_cast<const float4*>(src); y = *reinterpret_cast<const float4*>(src + 4);
      }
      unsigned h[4], m[4], l[4];
      sp_split2(x.x, x.y, h[0], m[0], l[0]); sp_split2(x.z, x.w, h[1], m[1], l[1]);
      sp_split2(y.x, y.y, h[2], m[2], l[2]); sp_split2(y.z, y.w, h[3], m[3], l[3]);
      uf_u4* o = a.out[mtx] + ((long)(ks * ncb + cb) * 3) * 64 + lane;
      o[0] = uf_u4{h[0], h[1], h[2], h[3]}; o[64] = uf_u4{m[0], m[1], m[2], m[3]}; o[128] = uf_u4{l[0], l[1], l[2], l[3]};
      return;
    }
    idx -= total;
  }
}

struct UpdFuseArgs {
  int N;
  const float* XM; const float* VM;                                  // [N][F], [N][3][F]
  float* UU; float* S; float* CAT; float* ZQ; float* Q; float* Y;    // [N][3][2F], [N][F], [N][2F], [N][F], [N][F], [N][3F]
  float* X1; float* V1;                                              // outputs [N][F], [N][3][F]
  const uf_u4* Uf; const uf_u4* V1f; const uf_u4* V2f;               // weight fragments
  const float* c1; const float* c2;
  // tangent sweep: tangents of the inputs, primal intermediates (read), tangent outputs
  const float* TXM; const float* TVM;
  float* TUU; float* TS; float* TCAT; float* TZQ; float* TQ; float* TY; float* TX1; float* TV1;
};

__device__ __forceinline__ sp_bf8 uf_frag(const char* lds, int off) { return *reinterpret_cast<const sp_bf8*>(lds + off); }
__device__ __forceinline__ sp_bf8 uf_wfrag(const uf_u4* p) {
  if (UF_ABLATE & 2) { const unsigned v = (unsigned)(size_t)p; return __builtin_bit_cast(sp_bf8, (uf_u4{v, v, v, v})); }
  return __builtin_bit_cast(sp_bf8, *p);
}

// six piece products, small terms first (gemm_split.h): pieces 0 = h, 1 = m, 2 = l
#define UF_TERMS(ACC, FA, FB)                                                                   \
  if (UF_ABLATE & 1) { ACC[0] += (float)FA[0][0] + (float)FA[1][1] + (float)FA[2][2] + (float)FB[0][0] + (float)FB[1][1] + (float)FB[2][2]; } else {  \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[1], FB[1], ACC, 0, 0, 0);                    \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[2], FB[0], ACC, 0, 0, 0);                    \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0], FB[2], ACC, 0, 0, 0);                    \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[1], FB[0], ACC, 0, 0, 0);                    \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0], FB[1], ACC, 0, 0, 0);                    \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0], FB[0], ACC, 0, 0, 0); }

// eight consecutive k of one row -> the three pieces, written to [ks][piece][row][16 bf16] (block pitch PB)
__device__ __forceinline__ void uf_stash8(char* lds, int PB, int row, int oct, const float4& x, const float4& y) {
  unsigned h[4], m[4], l[4];
  sp_split2(x.x, x.y, h[0], m[0], l[0]); sp_split2(x.z, x.w, h[1], m[1], l[1]);
  sp_split2(y.x, y.y, h[2], m[2], l[2]); sp_split2(y.z, y.w, h[3], m[3], l[3]);
  char* d = lds + ((oct >> 1) * 3) * PB + row * 32 + (oct & 1) * 16;
  *reinterpret_cast<uf_u4*>(d) = uf_u4{h[0], h[1], h[2], h[3]};
  *reinterpret_cast<uf_u4*>(d + PB) = uf_u4{m[0], m[1], m[2], m[3]};
  *reinterpret_cast<uf_u4*>(d + 2 * PB) = uf_u4{l[0], l[1], l[2], l[3]};
}
// one accumulator value pair (rows r0, r1 of column k) -> pieces, two-byte stores into [ks][piece][row][16 bf16]
__device__ __forceinline__ void uf_put2(char* lds, int PB, int k, int row0, int row1, float v0, float v1) {
  unsigned h, m, l;
  sp_split2(v0, v1, h, m, l);
  char* d = lds + ((k >> 4) * 3) * PB + (k & 15) * 2;
  *reinterpret_cast<unsigned short*>(d + row0 * 32) = (unsigned short)(h & 0xffffu);
  *reinterpret_cast<unsigned short*>(d + row1 * 32) = (unsigned short)(h >> 16);
  *reinterpret_cast<unsigned short*>(d + PB + row0 * 32) = (unsigned short)(m & 0xffffu);
  *reinterpret_cast<unsigned short*>(d + PB + row1 * 32) = (unsigned short)(m >> 16);
  *reinterpret_cast<unsigned short*>(d + 2 * PB + row0 * 32) = (unsigned short)(l & 0xffffu);
  *reinterpret_cast<unsigned short*>(d + 2 * PB + row1 * 32) = (unsigned short)(l >> 16);
}

// TAN = false: the forward sweep.  TAN = true: the tangent sweep -- the same three products on the tangents (tu = tvm U^T, tzq = [txm | tn] V1^T,
// ty = tq V2^T, no biases), the primal intermediates read back where the product rule needs them.
template <bool TAN>
__global__ __launch_bounds__(UF_NT, 2) void k_upd_fused(UpdFuseArgs q) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int F = UF_F, F2 = 2 * UF_F, F3 = 3 * UF_F;
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), lr = lane & 31, lk = lane >> 5;
  const int a0 = blockIdx.x * UF_R, nrows = min(UF_R, q.N - a0);
  const int f = 32 * w + lr;                       // this lane's channel
  const float* XMs = TAN ? q.TXM : q.XM;
  const float* VMs = TAN ? q.TVM : q.VM;

  // ---- product 1: u[(c, n)][0 .. 2F) = vec_msg[n][c][:] U^T.  Operand rows (c, n) -> LDS; this wavefront's columns f (vec1) and F + f (vec2) ----
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int id = t + UF_NT * j, row = id >> 4, oct = id & 15, c = row >> 5, n = min(row & 31, nrows - 1);
    const float* src = VMs + (long)(a0 + n) * F3 + c * F + oct * 8;
    uf_stash8(lds, UF_PB96, row, oct, *reinterpret_cast<const float4*>(src), *reinterpret_cast<const float4*>(src + 4));
  }
  // this lane's x_msg values (accumulator layout: register r = row (r & 3) + 8 (r >> 2) + 4 lk): requested here, they arrive under the first product.
  // (Loads that follow stores cannot be hoisted by the compiler -- the output arrays may alias -- so every epilogue below loads first, then stores.)
  float xm[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) xm[r] = XMs[(long)(a0 + min((r & 3) + 8 * (r >> 2) + 4 * lk, nrows - 1)) * F + f];
  __syncthreads();
  f32x16 acc[3][2];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][h][r] = 0.f;
  {
    const uf_u4* wp = q.Uf + (long)w * 3 * 64 + lane;   // fragment (ks, cb = 4 h + w, piece): ((ks * 8 + cb) * 3 + piece) * 64 + lane
    sp_bf8 fb[2][2][3];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int p = 0; p < 3; ++p) fb[0][h][p] = uf_wfrag(wp + ((0 * 8 + 4 * h) * 3 + p) * 64);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int cur = ks & 1;
      if (ks + 1 < 8) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int p = 0; p < 3; ++p) fb[cur ^ 1][h][p] = uf_wfrag(wp + (((ks + 1) * 8 + 4 * h) * 3 + p) * 64);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        sp_bf8 fa[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) fa[p] = uf_frag(lds, (ks * 3 + p) * UF_PB96 + (c * 32 + lr) * 32 + lk * 16);
        UF_TERMS(acc[c][0], fa, fb[cur][0])
        UF_TERMS(acc[c][1], fa, fb[cur][1])
      }
    }
  }
  // ---- s, n; u, s, cat stored; n (and x_msg) become the operand of product 2 ----
  float sv[TAN ? 1 : 16], nn[TAN ? 1 : 16];   // forward: s, n (the tangent sweep reads the primal ones where it needs them; its ts, tn live in ts_[], tn_[])
  float ts_[TAN ? 16 : 1], tn_[TAN ? 16 : 1];
  // the first fragments of product 2: requested before the epilogue's stores
  const uf_u4* wp2 = q.V1f + (long)w * 3 * 64 + lane;   // ((ks * 4 + w) * 3 + piece) * 64 + lane
  sp_bf8 fb2[4][3];
  if (!TAN) {   // (the tangent epilogue below has no registers to spare: its fragments are requested behind its stores)
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int p = 0; p < 3; ++p) fb2[d][p] = uf_wfrag(wp2 + ((d * 4) * 3 + p) * 64);
  }
  __syncthreads();                   // every wavefront is done with the vec_msg operand: the region is rewritten below
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
    const long atom = a0 + min(row, nrows - 1);
    const bool ok = row < nrows && !(UF_ABLATE & 4);
    float* uo = (TAN ? q.TUU : q.UU) + atom * 3 * F2 + f;
    if (!TAN) {
      const float a_0 = acc[0][0][r], a_1 = acc[1][0][r], a_2 = acc[2][0][r], b_0 = acc[0][1][r], b_1 = acc[1][1][r], b_2 = acc[2][1][r];
      sv[TAN ? 0 : r] = a_0 * b_0 + a_1 * b_1 + a_2 * b_2;
      nn[TAN ? 0 : r] = sqrtf(b_0 * b_0 + b_1 * b_1 + b_2 * b_2 + 1e-8f);
      if (ok) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { uo[c * F2] = acc[c][0][r]; uo[c * F2 + F] = acc[c][1][r]; }
        q.S[atom * F + f] = sv[TAN ? 0 : r]; q.CAT[atom * F2 + f] = xm[r]; q.CAT[atom * F2 + F + f] = nn[TAN ? 0 : r];
      }
    }
  }
  if (TAN) {
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += 8) {   // two groups of eight rows: 48 registers of primal vec1 / vec2 at a time, requested before the group's first store
      float pa[3][8], pb[3][8], pn[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = r0 + i;
        const long atom = a0 + min((r & 3) + 8 * (r >> 2) + 4 * lk, nrows - 1);
        const float* up = q.UU + atom * 3 * F2 + f;
#pragma unroll
        for (int c = 0; c < 3; ++c) { pa[c][i] = up[c * F2]; pb[c][i] = up[c * F2 + F]; }
        pn[i] = q.CAT[atom * F2 + F + f];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = r0 + i;
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
        const long atom = a0 + min(row, nrows - 1);
        const bool ok = row < nrows && !(UF_ABLATE & 4);
        float* uo = q.TUU + atom * 3 * F2 + f;
        const float ta0 = acc[0][0][r], ta1 = acc[1][0][r], ta2 = acc[2][0][r], tb0 = acc[0][1][r], tb1 = acc[1][1][r], tb2 = acc[2][1][r];
        ts_[TAN ? r : 0] = ta0 * pb[0][i] + pa[0][i] * tb0 + ta1 * pb[1][i] + pa[1][i] * tb1 + ta2 * pb[2][i] + pa[2][i] * tb2;
        tn_[TAN ? r : 0] = (pb[0][i] * tb0 + pb[1][i] * tb1 + pb[2][i] * tb2) / pn[i];
        if (ok) {
#pragma unroll
          for (int c = 0; c < 3; ++c) { uo[c * F2] = acc[c][0][r]; uo[c * F2 + F] = acc[c][1][r]; }
          q.TS[atom * F + f] = ts_[TAN ? r : 0]; q.TCAT[atom * F2 + f] = xm[r]; q.TCAT[atom * F2 + F + f] = tn_[TAN ? r : 0];
        }
      }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int p = 0; p < 3; ++p) fb2[d][p] = uf_wfrag(wp2 + ((d * 4) * 3 + p) * 64);
  }
#pragma unroll
  for (int r = 0; r < 16; r += 2) {   // registers r, r + 1 = rows row, row + 1
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
    uf_put2(lds + UF_A2, UF_PB32, F + f, row, row + 1, TAN ? tn_[TAN ? r : 0] : nn[TAN ? 0 : r], TAN ? tn_[TAN ? r + 1 : 0] : nn[TAN ? 0 : r + 1]);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {       // x_msg half of [x_msg | n]: k = 0 .. F
    const int id = t + UF_NT * j, row = id >> 4, oct = id & 15, n = min(row, nrows - 1);
    const float* src = XMs + (long)(a0 + n) * F + oct * 8;
    uf_stash8(lds + UF_A2, UF_PB32, row, oct, *reinterpret_cast<const float4*>(src), *reinterpret_cast<const float4*>(src + 4));
  }
  __syncthreads();
  // ---- product 2: zq = [x_msg | n] V1^T (+ c1), K = 2F ----
  f32x16 az;
#pragma unroll
  for (int r = 0; r < 16; ++r) az[r] = 0.f;
  // one k16 step is six matrix instructions (192 cycles): the fragments are requested three steps ahead (L2 latency), the first three before the epilogue above
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    if (ks + 3 < 16) {
#pragma unroll
      for (int p = 0; p < 3; ++p) fb2[(ks + 3) & 3][p] = uf_wfrag(wp2 + (((ks + 3) * 4) * 3 + p) * 64);
    }
    sp_bf8 fa[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) fa[p] = uf_frag(lds + UF_A2, (ks * 3 + p) * UF_PB32 + lr * 32 + lk * 16);
    UF_TERMS(az, fa, fb2[ks & 3])
  }
  // the first fragments of product 3 and the bias: requested before the stores below
  const uf_u4* wp3 = q.V2f + (long)w * 3 * 64 + lane;   // ((ks * 12 + 4 part + w) * 3 + piece) * 64 + lane
  sp_bf8 fb3[2][3][3];
#pragma unroll
  for (int pt = 0; pt < 3; ++pt)
#pragma unroll
    for (int p = 0; p < 3; ++p) fb3[0][pt][p] = uf_wfrag(wp3 + ((4 * pt) * 3 + p) * 64);
  // ---- zq, q stored; q becomes the operand of product 3 ----
  {
    const float bz = TAN ? 0.f : q.c1[f];
    float qv[16], pz[TAN ? 16 : 1];
    if (TAN) {
#pragma unroll
      for (int r = 0; r < 16; ++r) pz[TAN ? r : 0] = q.ZQ[(long)(a0 + min((r & 3) + 8 * (r >> 2) + 4 * lk, nrows - 1)) * F + f];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
      const long atom = a0 + min(row, nrows - 1);
      const float z = az[r] + bz;
      if (!TAN) {
        qv[r] = nq_silu(z);
        if (row < nrows && !(UF_ABLATE & 4)) { q.ZQ[atom * F + f] = z; q.Q[atom * F + f] = qv[r]; }
      } else {
        qv[r] = z * nq_dsilu_fast(pz[TAN ? r : 0]);   // the tangent of the activation (same form as the epilogue of nq_gemm_nt_dsilu)
        if (row < nrows && !(UF_ABLATE & 4)) { q.TZQ[atom * F + f] = z; q.TQ[atom * F + f] = qv[r]; }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
      uf_put2(lds + UF_A3, UF_PB32, f, row, row + 1, qv[r], qv[r + 1]);
    }
  }
  __syncthreads();
  // ---- product 3: y = q V2^T (+ c2): this wavefront's columns f, F + f, 2F + f ----
  f32x16 ay[3];
#pragma unroll
  for (int pt = 0; pt < 3; ++pt)
#pragma unroll
    for (int r = 0; r < 16; ++r) ay[pt][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < 8) {
#pragma unroll
      for (int pt = 0; pt < 3; ++pt)
#pragma unroll
        for (int p = 0; p < 3; ++p) fb3[cur ^ 1][pt][p] = uf_wfrag(wp3 + (((ks + 1) * 12 + 4 * pt) * 3 + p) * 64);
    }
    sp_bf8 fa[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) fa[p] = uf_frag(lds + UF_A3, (ks * 3 + p) * UF_PB32 + lr * 32 + lk * 16);
    UF_TERMS(ay[0], fa, fb3[cur][0])
    UF_TERMS(ay[1], fa, fb3[cur][1])
    UF_TERMS(ay[2], fa, fb3[cur][2])
  }
  // ---- y stored; x_upd = x_msg + ya + yb s, vec_upd[c] = vec_msg[c] + yc vec1[c] (and their tangents) ----
  {
    const float ca = TAN ? 0.f : q.c2[f], cbb = TAN ? 0.f : q.c2[F + f], cc = TAN ? 0.f : q.c2[F2 + f];
    constexpr int CHK = TAN ? 8 : 16;   // rows per group: everything a group reads is requested before its first store (tangent: two groups, 80 registers of reads each)
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += CHK) {
      float vmr[3][CHK], pv1[TAN ? 3 : 1][TAN ? CHK : 1], pyb[TAN ? CHK : 1], pyc[TAN ? CHK : 1], ps[TAN ? CHK : 1];
#pragma unroll
      for (int i = 0; i < CHK; ++i) {
        const int r = r0 + i;
        const long atom = a0 + min((r & 3) + 8 * (r >> 2) + 4 * lk, nrows - 1);
        const float* vm = VMs + atom * F3 + f;
#pragma unroll
        for (int c = 0; c < 3; ++c) vmr[c][i] = vm[c * F];
        if (TAN) {
          const float* yp = q.Y + atom * F3 + f;
          pyb[TAN ? i : 0] = yp[F]; pyc[TAN ? i : 0] = yp[F2];
          ps[TAN ? i : 0] = q.S[atom * F + f];
          const float* up = q.UU + atom * 3 * F2 + f;     // primal vec1
#pragma unroll
          for (int c = 0; c < 3; ++c) pv1[TAN ? c : 0][TAN ? i : 0] = up[c * F2];
        }
      }
#pragma unroll
      for (int i = 0; i < CHK; ++i) {
        const int r = r0 + i;
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row >= nrows || ((UF_ABLATE & 4) && ay[0][r] != 12345.f)) continue;
        const long atom = a0 + row;
        const float ya = ay[0][r] + ca, yb = ay[1][r] + cbb, yc = ay[2][r] + cc;
        float* yo = (TAN ? q.TY : q.Y) + atom * F3 + f;
        yo[0] = ya; yo[F] = yb; yo[F2] = yc;
        if (!TAN) {
          q.X1[atom * F + f] = xm[r] + ya + yb * sv[TAN ? 0 : r];
#pragma unroll
          for (int c = 0; c < 3; ++c) q.V1[atom * F3 + c * F + f] = vmr[c][i] + yc * acc[c][0][r];
        } else {
          q.TX1[atom * F + f] = xm[r] + ya + yb * ps[TAN ? i : 0] + pyb[TAN ? i : 0] * ts_[TAN ? r : 0];
#pragma unroll
          for (int c = 0; c < 3; ++c)   // t vec_msg + t yc vec1 + yc t vec1
            q.TV1[atom * F3 + c * F + f] = vmr[c][i] + yc * pv1[TAN ? c : 0][TAN ? i : 0] + pyc[TAN ? i : 0] * acc[c][0][r];
        }
      }
    }
  }
}

static size_t uf_set_u4(int F) { return (size_t)(2 * F * F + 2 * F * F + 3 * F * F) / 8 * 3; }   // uint4 fragments of one order
int nq_updfuse_presplit(hipStream_t st, const float* U, const float* V1, const float* V2, int F, float* frag) {
  NQ_PROF(st, "upd_presplit");
  if (F != UF_F) return nq_fail(NQ_ERR_ARG, "fused update block: hidden_channels must be 128");
  uf_u4* base = reinterpret_cast<uf_u4*>(frag);
  const int total = (2 * F * F + 2 * F * F + 3 * F * F) / 8;   // one thread per fragment lane (8 weights)
  UfSplitArgs a;
  // forward order (A W^T, W [OUT][IN]): U [2F][F], V1 [F][2F], V2 [3F][F]
  a.W[0] = U; a.OUT[0] = 2 * F; a.IN[0] = F; a.out[0] = base;
  a.W[1] = V1; a.OUT[1] = F; a.IN[1] = 2 * F; a.out[1] = base + (size_t)2 * F * F / 8 * 3;
  a.W[2] = V2; a.OUT[2] = 3 * F; a.IN[2] = F; a.out[2] = a.out[1] + (size_t)2 * F * F / 8 * 3;
  // reverse order (G W, W [K][N] row-major; "OUT" = N columns, "IN" = K): V2 [3F][F], V1 [F][2F], U [2F][F]
  uf_u4* rbase = base + uf_set_u4(F);
  a.W[3] = V2; a.OUT[3] = F; a.IN[3] = 3 * F; a.out[3] = rbase;
  a.W[4] = V1; a.OUT[4] = 2 * F; a.IN[4] = F; a.out[4] = rbase + (size_t)3 * F * F / 8 * 3;
  a.W[5] = U; a.OUT[5] = F; a.IN[5] = 2 * F; a.out[5] = a.out[4] + (size_t)2 * F * F / 8 * 3;
  hipLaunchKernelGGL(k_uf_presplit, dim3(nq_cdiv(2 * total, 256)), dim3(256), 0, st, a);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

// u: UpdArgs of the sweep (engine.hip); frag: this layer's weight fragments; Q / ZQ: the activation pair of the V1 product
int nq_upd_fused(hipStream_t st, const UpdArgs& u, const float* frag, const float* c1, const float* c2, float* ZQ, float* Q, float* TZQ, float* TQ, bool tan) {
  NQ_PROF(st, tan ? "upd_fused_tan" : "upd_fused");
  if (u.F != UF_F) return nq_fail(NQ_ERR_ARG, "fused update block: hidden_channels must be 128");
  const int F = u.F;
  UpdFuseArgs q{};
  q.N = u.N; q.XM = u.XM; q.VM = u.VM; q.UU = const_cast<float*>(u.U); q.S = u.S; q.CAT = u.CAT; q.ZQ = ZQ; q.Q = Q; q.Y = const_cast<float*>(u.Y);
  q.X1 = u.X1; q.V1 = u.V1;
  const uf_u4* base = reinterpret_cast<const uf_u4*>(frag);
  q.Uf = base; q.V1f = base + (size_t)2 * F * F / 8 * 3; q.V2f = q.V1f + (size_t)2 * F * F / 8 * 3;
  q.c1 = c1; q.c2 = c2;
  q.TXM = u.TXM; q.TVM = u.TVM; q.TUU = const_cast<float*>(u.TU); q.TS = u.TS; q.TCAT = u.TCAT; q.TZQ = TZQ; q.TQ = TQ; q.TY = const_cast<float*>(u.TY);
  q.TX1 = u.TX1; q.TV1 = u.TV1;
  const int grid = nq_cdiv(u.N, UF_R);
  if (tan) {
    NQ_DYN_LDS(k_upd_fused<true>, UF_LDS);
    hipLaunchKernelGGL(k_upd_fused<true>, dim3(grid), dim3(UF_NT), UF_LDS, st, q);
  } else {
    NQ_DYN_LDS(k_upd_fused<false>, UF_LDS);
    hipLaunchKernelGGL(k_upd_fused<false>, dim3(grid), dim3(UF_NT), UF_LDS, st, q);
  }
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}

// ---- force-adjoint sweep of the update block (painn.py:535-548 reversed; node.hip k_upd_rev1 / k_upd_rev2 and the three input-gradient products) ---------
//   gy = (gx, gx s, <gvec, vec1>);  gq = (gy V2) silu'(zq);  gcat = gq V1;  gx <- gx + gcat[:F];  gn = gcat[F:] / n;
//   gu[c] = (gvec[c] yc + gx yb vec2[c],  gn vec2[c] + gx yb vec1[c]);  gvec[c] <- gvec[c] + gu[c] U
// The force sweep computes no weight gradients, so NONE of gy, gq, gcat, gu has another consumer: they live in registers / LDS only.  HBM traffic per layer:
// 15 N F floats read (gx, gvec, s, u, yb, yc, n, zq) + 4 written, against 37 + 16 for the two elementwise kernels and three products of rounds 1-5.
struct UpdRevFuseArgs {
  int N;
  float* GX; float* GV;                                          // adjoints of x_upd / vec_upd in, of x_msg / vec_msg out (in place)  [N][F], [N][3][F]
  const float* S; const float* UU; const float* Y; const float* CAT; const float* ZQ;
  const uf_u4* V2n; const uf_u4* V1n; const uf_u4* Un;           // weight fragments, reverse order
};

__global__ __launch_bounds__(UF_NT, 2) void k_updrev_fused(UpdRevFuseArgs q) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int F = UF_F, F2 = 2 * UF_F, F3 = 3 * UF_F;
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), lr = lane & 31, lk = lane >> 5;
  const int a0 = blockIdx.x * UF_R, nrows = min(UF_R, q.N - a0);
  const int f = 32 * w + lr;
  auto atom_of = [&](int r) __attribute__((always_inline)) -> long { return a0 + min((r & 3) + 8 * (r >> 2) + 4 * lk, nrows - 1); };
  (void)t;
  // ---- gy -> LDS (K = 3F: parts a, b, c at k = f, F + f, 2F + f) ----
  float zq[16];
  {
    float gx[16], sv[16], gyc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long atom = atom_of(r);
      gx[r] = q.GX[atom * F + f]; sv[r] = q.S[atom * F + f];
      const float* gv = q.GV + atom * F3 + f;
      const float* up = q.UU + atom * 3 * F2 + f;
      gyc[r] = gv[0] * up[0] + gv[F] * up[F2] + gv[F2] * up[2 * F2];
      zq[r] = q.ZQ[atom * F + f];
    }
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
      uf_put2(lds, UF_PB32, f, row, row + 1, gx[r], gx[r + 1]);
      uf_put2(lds, UF_PB32, F + f, row, row + 1, gx[r] * sv[r], gx[r + 1] * sv[r + 1]);
      uf_put2(lds, UF_PB32, F2 + f, row, row + 1, gyc[r], gyc[r + 1]);
    }
  }
  // ---- product A: gq = (gy V2) silu'(zq), K = 3F ----
  const uf_u4* wpa = q.V2n + (long)w * 3 * 64 + lane;   // ((ks * 4 + w) * 3 + piece) * 64 + lane
  sp_bf8 fbr[4][3];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int p = 0; p < 3; ++p) fbr[d][p] = uf_wfrag(wpa + ((d * 4) * 3 + p) * 64);
  __syncthreads();
  f32x16 aq;
#pragma unroll
  for (int r = 0; r < 16; ++r) aq[r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 24; ++ks) {
    if (ks + 3 < 24) {
#pragma unroll
      for (int p = 0; p < 3; ++p) fbr[(ks + 3) & 3][p] = uf_wfrag(wpa + (((ks + 3) * 4) * 3 + p) * 64);
    }
    sp_bf8 fa[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) fa[p] = uf_frag(lds, (ks * 3 + p) * UF_PB32 + lr * 32 + lk * 16);
    UF_TERMS(aq, fa, fbr[ks & 3])
  }
  // first fragments of product B (columns f and F + f of gcat)
  const uf_u4* wpb = q.V1n + (long)w * 3 * 64 + lane;   // ((ks * 8 + 4 h + w) * 3 + piece) * 64 + lane
  sp_bf8 fbb[2][2][3];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int p = 0; p < 3; ++p) fbb[0][h][p] = uf_wfrag(wpb + ((4 * h) * 3 + p) * 64);
  __syncthreads();                   // every wavefront is done with the gy operand
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
    uf_put2(lds, UF_PB32, f, row, row + 1, aq[r] * nq_dsilu_fast(zq[r]), aq[r + 1] * nq_dsilu_fast(zq[r + 1]));
  }
  __syncthreads();
  // ---- product B: gcat = gq V1, K = F ----
  f32x16 ac[2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 16; ++r) ac[h][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < 8) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int p = 0; p < 3; ++p) fbb[cur ^ 1][h][p] = uf_wfrag(wpb + (((ks + 1) * 8 + 4 * h) * 3 + p) * 64);
    }
    sp_bf8 fa[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) fa[p] = uf_frag(lds, (ks * 3 + p) * UF_PB32 + lr * 32 + lk * 16);
    UF_TERMS(ac[0], fa, fbb[cur][0])
    UF_TERMS(ac[1], fa, fbb[cur][1])
  }
  // ---- gx out; gs = gx yb, gn = gcat_n / n ----
  float gs[16], gnn[16], yc[16];
  {
    float gx[16], yb[16], nn[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long atom = atom_of(r);
      gx[r] = q.GX[atom * F + f]; yb[r] = q.Y[atom * F3 + F + f]; yc[r] = q.Y[atom * F3 + F2 + f]; nn[r] = q.CAT[atom * F2 + F + f];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      gs[r] = gx[r] * yb[r];
      gnn[r] = ac[1][r] / nn[r];
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (row < nrows) q.GX[(long)(a0 + row) * F + f] = gx[r] + ac[0][r];
    }
  }
  // ---- per component: gu[c] -> LDS (K = 2F), gvec[c] += gu[c] U ----
  const uf_u4* wpc = q.Un + (long)w * 3 * 64 + lane;    // ((ks * 4 + w) * 3 + piece) * 64 + lane
#pragma nounroll   // (unrolled, the compiler requests the three components' vec1 / vec2 rows at once and spills)
  for (int c = 0; c < 3; ++c) {
    float gvc[16];
    {
      float va[16], vb[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long atom = atom_of(r);
        va[r] = q.UU[atom * 3 * F2 + c * F2 + f]; vb[r] = q.UU[atom * 3 * F2 + c * F2 + F + f]; gvc[r] = q.GV[atom * F3 + c * F + f];
      }
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int p = 0; p < 3; ++p) fbr[d][p] = uf_wfrag(wpc + ((d * 4) * 3 + p) * 64);
      __syncthreads();               // the previous operand (gq / gu of the component before) has been consumed
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
        uf_put2(lds, UF_PB32, f, row, row + 1, gvc[r] * yc[r] + gs[r] * vb[r], gvc[r + 1] * yc[r + 1] + gs[r + 1] * vb[r + 1]);
        uf_put2(lds, UF_PB32, F + f, row, row + 1, gnn[r] * vb[r] + gs[r] * va[r], gnn[r + 1] * vb[r + 1] + gs[r + 1] * va[r + 1]);
      }
    }
    __syncthreads();
    f32x16 av;
#pragma unroll
    for (int r = 0; r < 16; ++r) av[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (ks + 3 < 16) {
#pragma unroll
        for (int p = 0; p < 3; ++p) fbr[(ks + 3) & 3][p] = uf_wfrag(wpc + (((ks + 3) * 4) * 3 + p) * 64);
      }
      sp_bf8 fa[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) fa[p] = uf_frag(lds, (ks * 3 + p) * UF_PB32 + lr * 32 + lk * 16);
      UF_TERMS(av, fa, fbr[ks & 3])
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (row < nrows) q.GV[(long)(a0 + row) * F3 + c * F + f] = gvc[r] + av[r];
    }
  }
}

// force-adjoint sweep of the update block of one layer: u.GX / u.GV updated in place; frag = the layer's fragment block (nq_updfuse_presplit)
int nq_updrev_fused(hipStream_t st, const UpdRevArgs& u, const float* frag, const float* ZQ) {
  NQ_PROF(st, "updrev_fused");
  if (u.F != UF_F) return nq_fail(NQ_ERR_ARG, "fused update block: hidden_channels must be 128");
  const int F = u.F;
  UpdRevFuseArgs q{};
  q.N = u.N; q.GX = u.GX; q.GV = u.GV; q.S = u.S; q.UU = u.U; q.Y = u.Y; q.CAT = u.CAT; q.ZQ = ZQ;
  const uf_u4* rbase = reinterpret_cast<const uf_u4*>(frag) + uf_set_u4(F);
  q.V2n = rbase; q.V1n = rbase + (size_t)3 * F * F / 8 * 3; q.Un = q.V1n + (size_t)2 * F * F / 8 * 3;
  NQ_DYN_LDS(k_updrev_fused, UF_LDS);
  hipLaunchKernelGGL(k_updrev_fused, dim3(nq_cdiv(u.N, UF_R)), dim3(UF_NT), UF_LDS, st, q);
  NQ_LAUNCH_CHECK();
  return NQ_OK;
}
