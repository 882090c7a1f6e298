// Device helpers shared by the per-atom wavefront kernels (edge.hip: PaiNN messages, schnet.hip: continuous-filter convolution):
// lane broadcasts, CH-channels-per-lane vector loads/stores, the per-edge 13-tap window record and packed-fp32 vector ops.
#pragma once
#include "common.h"

#define FWIN 13        // taps of the Gaussian window that matter to fp32 (+-6 centres around the nearest one)
#define RW_STRIDE 32   // floats per edge window record: [0..12] rho, [13] k0 (int bits), [14] beta, [16..28] drho, [30] dbeta

__device__ __forceinline__ int bl_i(int v, int j) { return __builtin_amdgcn_readlane(v, j); }
__device__ __forceinline__ float bl_f(float v, int j) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j)); }

// ---- CH consecutive channels per lane: vector loads / stores ---------------------------------------------------
template <int CH> struct VecOf;
template <> struct VecOf<1> { typedef float T; };
template <> struct VecOf<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct VecOf<4> { typedef float T __attribute__((ext_vector_type(4))); };

template <int CH>
__device__ __forceinline__ void ldv(float (&o)[CH], const float* p) {
  if constexpr (CH == 1) { o[0] = *p; }
  else {
    const typename VecOf<CH>::T v = *reinterpret_cast<const typename VecOf<CH>::T*>(p);
#pragma unroll
    for (int c = 0; c < CH; ++c) o[c] = v[c];
  }
}
// Gathers of neighbour rows through a buffer descriptor: address = descriptor base + lane byte offset (VGPR, constant for the kernel) + row byte offset (SGPR:
// the neighbour index comes from a lane broadcast).  No vector address arithmetic per load (the flat form costs one v_lshl_add_u64 per load: 12 of the 138
// VALU instructions per edge of the tangent flavour).  The scalar offset is 32-bit: an array half must stay below 4 GB (2.8 M atoms at F = 128; the launchers check).
typedef unsigned int lanes_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int lanes_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const float* base) {
  const unsigned long long b = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo), 0, (int)0xffffffffu, 0x00020000);
}
template <int CH>
__device__ __forceinline__ void ldv_buf(float (&o)[CH], __amdgpu_buffer_rsrc_t r, int lane_bytes, int row_bytes) {
  if constexpr (CH == 1) o[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, lane_bytes, row_bytes, 0));
  else if constexpr (CH == 2) {
    const lanes_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, lane_bytes, row_bytes, 0);
    o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y);
  } else {
    const lanes_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lane_bytes, row_bytes, 0);
    o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
  }
}
template <int CH>
__device__ __forceinline__ void stv(float* p, const float (&o)[CH]) {
  if constexpr (CH == 1) { *p = o[0]; }
  else {
    typename VecOf<CH>::T v;
#pragma unroll
    for (int c = 0; c < CH; ++c) v[c] = o[c];
    *reinterpret_cast<typename VecOf<CH>::T*>(p) = v;
  }
}

// streaming store (gphi / gpsi: written once, read once by the weight-gradient kernel): keep it from evicting the node rows in L2
template <int CH>
__device__ __forceinline__ void stv_stream(float* p, const float (&o)[CH]) {
  if constexpr (CH == 1) { __builtin_nontemporal_store(o[0], p); }
  else {
    typename VecOf<CH>::T v;
#pragma unroll
    for (int c = 0; c < CH; ++c) v[c] = o[c];
    __builtin_nontemporal_store(v, reinterpret_cast<typename VecOf<CH>::T*>(p));
  }
}

// ---- per-edge window record (scalar loads: the record address is wave-uniform) -------------------------------
template <bool PSI>
struct WinRegs { float rr[16]; float dd[PSI ? 16 : 1]; };

template <bool PSI>
__device__ __forceinline__ void load_win(WinRegs<PSI>& w, const float* __restrict__ RW, int sp) {
  const float4* rw4 = reinterpret_cast<const float4*>(RW + (long)sp * RW_STRIDE);
#pragma unroll
  for (int v = 0; v < 4; ++v) *reinterpret_cast<float4*>(&w.rr[4 * v]) = rw4[v];
  if (PSI) {
#pragma unroll
    for (int v = 0; v < 4; ++v) *reinterpret_cast<float4*>(&w.dd[4 * v]) = rw4[4 + v];
  }
}

// LDS filter taps, never paired: the compiler merges two ds_read_b64 of one base address into ds_read2(st64)_b64, which the LDS serves at half the
// bytes per clock (measured on MI355X, profiles/r05_valu_lds_issue_rates_lab.txt: paired 261 B/ns per CU, single ds_read_b64 348, ds_read_b128 474;
// the 13-tap x 3-part filter of one edge: 66.5 ns paired vs 46.4 ns single at 16 wavefronts per CU).  A volatile access through an LDS-address-space
// pointer stays a single ds_read_b64 and keeps the compiler's own lgkmcnt accounting.
#ifndef NQ_LDS_SINGLE_READS
#define NQ_LDS_SINGLE_READS 1
#endif
template <int CH>
__device__ __forceinline__ typename VecOf<CH>::T lds_tap(const float* p) {
  typedef typename VecOf<CH>::T V;
#if NQ_LDS_SINGLE_READS
  if constexpr (CH == 2) return *(__attribute__((address_space(3))) const volatile V*)(p);
#endif
  return *reinterpret_cast<const V*>(p);
}

// CH-wide vector arithmetic: every FMA pair becomes one v_pk_fma_f32
template <int CH> struct VOps {
  typedef typename VecOf<CH>::T V;
  static __device__ __forceinline__ V splat(float x) { V v; for (int c = 0; c < CH; ++c) v[c] = x; return v; }
  static __device__ __forceinline__ V load(const float* p) { return *reinterpret_cast<const V*>(p); }
  static __device__ __forceinline__ V from(const float (&a)[CH]) { V v; for (int c = 0; c < CH; ++c) v[c] = a[c]; return v; }
  static __device__ __forceinline__ void to(float (&a)[CH], V v) { for (int c = 0; c < CH; ++c) a[c] = v[c]; }
  static __device__ __forceinline__ V fma(V a, V b, V c) { return __builtin_elementwise_fma(a, b, c); }
};
template <> struct VOps<1> {
  typedef float V;
  static __device__ __forceinline__ V splat(float x) { return x; }
  static __device__ __forceinline__ V load(const float* p) { return *p; }
  static __device__ __forceinline__ V from(const float (&a)[1]) { return a[0]; }
  static __device__ __forceinline__ void to(float (&a)[1], V v) { a[0] = v; }
  static __device__ __forceinline__ V fma(V a, V b, V c) { return fmaf(a, b, c); }
};
