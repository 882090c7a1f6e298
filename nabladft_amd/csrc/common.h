// nablaq -- MI355X (gfx950 / CDNA4) engine for the nablaDFT PaiNN hot path.
// Shared device helpers and host-side launch/error plumbing.  gfx950 only: wavefront = 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define NQ_WAVE 64

// ---- status codes (mirrored in include/nablaq.h) -------------------------------------------
#define NQ_OK 0
#define NQ_ERR_HIP 1          // a HIP call failed; see nq_last_error()
#define NQ_ERR_ARG 2          // bad argument (shape / alignment / unsupported size)
#define NQ_ERR_MOL_TOO_LARGE 3
#define NQ_ERR_WORKSPACE 4    // caller-provided workspace too small
#define NQ_ERR_NO_EDGES 5

extern thread_local char nq_err_buf[512];
int nq_fail(int code, const char* fmt, ...);

#define NQ_HIP(call)                                                                            \
  do {                                                                                          \
    hipError_t e__ = (call);                                                                    \
    if (e__ != hipSuccess)                                                                      \
      return nq_fail(NQ_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
  } while (0)

#define NQ_LAUNCH_CHECK() NQ_HIP(hipGetLastError())

#define NQ_TRY(call)            \
  do {                          \
    int rc__ = (call);          \
    if (rc__ != NQ_OK) return rc__; \
  } while (0)

#define NQ_MAX_LAYERS 64
// The opt-in to more than 64 KB of dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize) is a property of (device, kernel): a process that drives several
// GPUs has to set it on each of them.  nq_dyn_lds keeps the largest size granted per (device, kernel) under a mutex and calls hipFuncSetAttribute only when a
// launch asks for more (engine.hip); no process-global "already set" flags in the launchers.
int nq_dyn_lds(const void* kernel, size_t bytes);
#define NQ_DYN_LDS(kernel, bytes) NQ_TRY(nq_dyn_lds((const void*)(kernel), (bytes)))
static inline int nq_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- optional per-kernel timing (HIP events on the launch stream; used by bench.py for the roofline) ----
// Off by default (one branch per launch).  nq_profile_enable(1) makes every launcher record a
// start/stop event pair around its kernel(s); nq_profile_read() synchronises and sums by name.
struct NqProfScope {
  hipStream_t st; int slot;
  NqProfScope(hipStream_t s, const char* name);
  ~NqProfScope();
  void add_flops(double f);   // arithmetic of this launch (GEMM launchers: 2 M N K), summed per name and returned by nq_profile_read2
};
extern int nq_profile_on;
#define NQ_PROF(st, name) NqProfScope nq_prof_scope__((st), (name))
#define NQ_PROF_FLOPS(f) do { if (nq_profile_on) nq_prof_scope__.add_flops((double)(f)); } while (0)

// ---- device math ---------------------------------------------------------------------------
__device__ __forceinline__ float nq_sigmoid(float z) { return 1.0f / (1.0f + expf(-z)); }
// SiLU and its first two derivatives (oracle/painn_sweeps.py: silu, dsilu, d2silu)
__device__ __forceinline__ float nq_silu(float z) { return z * nq_sigmoid(z); }
__device__ __forceinline__ float nq_dsilu(float z) {
  float s = nq_sigmoid(z);
  return s * (1.0f + z * (1.0f - s));
}
__device__ __forceinline__ float nq_d2silu(float z) {
  float s = nq_sigmoid(z);
  float ds = s * (1.0f - s);
  return ds * (2.0f + z * (1.0f - 2.0f * s));
}

// Hardware-rate forms for GEMM epilogues (v_exp_f32 + v_rcp_f32, ~1 ulp each: 2e-7 relative, inside every parity budget of the paths that use them)
__device__ __forceinline__ float nq_sigmoid_fast(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }
__device__ __forceinline__ float nq_silu_fast(float z) { return z * nq_sigmoid_fast(z); }
__device__ __forceinline__ float nq_dsilu_fast(float z) {
  const float s = nq_sigmoid_fast(z);
  return s * (1.0f + z * (1.0f - s));
}

// Sum over the 64 lanes of the wavefront, result broadcast to every lane.  DPP only (no LDS round trips):
// an inclusive scan inside each 16-lane row (row_shr 1,2,4,8), then row_bcast:15 / row_bcast:31 carry the row totals
// upwards, so lane 63 holds the full sum; v_readlane broadcasts it as a scalar.  Fixed order -> deterministic.
__device__ __forceinline__ float nq_wave_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));  // row_shr:1
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, true));  // row_shr:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, true));  // row_shr:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, true));  // row_shr:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, true));  // row_bcast:15 -> rows 1,3
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, true));  // row_bcast:31 -> rows 2,3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// kernels (implemented in the .hip files, launched by engine.hip) ------------------------------
struct NqGraphView {
  int N, B, E;
  const int* mol_ptr;   // [B+1]
  const int* row_ptr;   // [N+1]  CSR by target atom
  const int* col;       // [E]    source atom of the in-edge stored at this slot
  const int* rev;       // [E]    slot of the reverse edge
  const float4* geom;   // [E]    {rx, ry, rz, d}: r = (pos[col]-pos[row])/d
  const int* z;         // [N]    atomic numbers
  const int* atom_mol;  // [N]
  const int* lowptr;    // [N+1] prefix count of lower (source < target) in-edges = numbering of the undirected pairs
};

// ---- kernel argument blocks (shared between the kernel files and engine.hip) ---------------------
struct GraphFillArgs {
  const float* pos; const int* mol_ptr; const int* row_ptr; const int* lowptr;
  float r2; int K;
  int* col; int* dst; int* rev; float4* geom; int* slot2canon;  // CSR outputs
  long long* c_src; long long* c_dst; float* c_dist; float* c_vec; long long* id_swap; long long* neighbors;  // canonical (nullable)
  int* atom_mol;
};

struct MsgArgs {
  NqGraphView g; int F;
  // primal
  const float* X; const float* V; const float* XH; const float* PHI; const float* PSI;
  float* XM; float* VM;
  // tangent
  const float* TX; const float* TV; const float* TXH; const float* TD; const float* TR;  // TD[E], TR[E][3]
  float* TXM; float* TVM;
};

// in-kernel radial filter: phi = br + Wr rho(d), psi = Wr drho(d) evaluated from an LDS-resident transposed Wr
struct FilterArgs {
  const float* WRT;   // [R][3F] = rbf_proj.weight^T (written once per forward by k_transpose)
  const float* br;    // [3F]
  const float* mu;    // [R] GaussianSmearing offsets
  const float* RW;    // [E][32] per-edge 13-tap window record written by k_rbf_window: [0..12] rho, [13] k0 (int bits), [16..28] drho
  int R;
  float inv_cutoff, p, a, b, c, coeff;
  int mode;           // 0: polynomial envelope inside the projection (painn_pyg); 1: cosine cutoff after the bias, unscaled Gaussians (spk PaiNN);
                      // 2: bare unscaled Gaussians, record slots 14/30 = fcut, fcut' (spk SchNet)
  float cutoff;
  int* row_ctr;       // fused message kernels: [8 XCDs][slices] zeroed row counters of THIS launch (dynamic row claiming), or null = static striding
};
#define NQ_ROWCTR_PAD 32                        // one counter per 128-byte line
#define NQ_ROWCTR_INTS (8 * 4 * 8 * NQ_ROWCTR_PAD)   // ints per launch: 8 XCDs x up to 4 channel slices x up to 8 claim groups, padded

struct MsgRevArgs {
  NqGraphView g; int F;
  const float* V; const float* XH; const float* PHI; const float* PSI;      // primal, layer input side
  const float* TV; const float* TXH; const float* TD; const float* TR;      // tangents (dual only)
  const float* GX; const float* GV;                                          // adjoints of x_msg / vec_msg   [N][F], [N][3F]
  const float* GTX; const float* GTV;                                        // adjoints of their tangents (dual)
  float* GXH; float* GTXH;                                                   // out: adjoint of xh (and t_xh)  [N][3F]
  float* GV_out; float* GTV_out;                                             // out: gvec_msg + scatter part   [N][3F]
  float* GPHI; float* GPSI;                                                  // dual out: [E][3F] each
  float4* GEDGE;                                                             // force mode: [nwaves][E] {gd, grx, gry, grz} (+=)
  float* GBR;                                                                // dual out: [N][3F] per-atom sums of gphi (bias gradient partials)
  int row_filter, mol_cap;                                                   // 0: every row; 1: only rows of molecules of <= mol_cap atoms; 2: only rows of larger molecules
  int lite;                                                                  // dual: GTXH / GTV_out are not written (they are the force sweep's GXH / GV_out, read from its per-layer store)
};

struct UpdArgs {
  int N, F;
  const float* XM; const float* VM; const float* U; const float* Y;        // [N][F], [N][3][F], [N][3][2F], [N][3F]
  float* S; float* CAT; float* X1; float* V1;                              // [N][F], [N][2F], outputs [N][F], [N][3][F]
  const float* TXM; const float* TVM; const float* TU; const float* TY;
  float* TS; float* TCAT; float* TX1; float* TV1;
};

struct UpdRevArgs {
  int N, F;
  const float* U; const float* Y; const float* S; const float* CAT;          // primal
  const float* TU; const float* TY; const float* TS; const float* TCAT;      // tangent (dual)
  float* GX; float* GV; float* GTX; float* GTV;                              // adjoints of x_upd/vec_upd (GX,GTX updated in rev2)
  float* GY; float* GTY;                                                     // [N][3F] out of rev1
  const float* GCAT; const float* GTCAT;                                     // [N][2F] in to rev2
  float* GU; float* GTU;                                                     // [N][3][2F] out of rev2
  // The tangent adjoints (GT*) of the second-order sweep obey the force-adjoint sweep's recursion with the same seeds: they ARE that sweep's adjoints.  lite = 1
  // (dual flavours): the GT* operands are read from where the force sweep stored them and nothing is written to them.  GX_out (force sweep, rev2): where
  // gx_upd + gcat[:F] goes (null: in place), so that the stage the second-order sweep reads stays intact.
  int lite; float* GX_out;
};

struct ReadoutArgs {
  int N, H;
  const float* ZO; const float* TZO; const float* w2; float o2_dummy; const float* o2;
  float* e_atom; float* te_atom;
  const float* ge; const float* gte;           // per-atom seeds
  float* GZO; float* GTZO;                     // [N][H]
  float* TMPW;                                 // dual: [N][H] per-atom contribution to grad of w2
};

// ---- launchers ----------------------------------------------------------------------------------
int nq_graph_count_impl(const float* pos, const int* mol_ptr, int N, int B, int max_mol_atoms, float cutoff2, int K, int* deg, int* lowdeg,
                        int* row_ptr, int* lowptr, int* E_host, hipStream_t st);
int nq_graph_fill_impl(GraphFillArgs args, int B, int max_mol_atoms, hipStream_t st);

// Bpre (optional): the weight pre-split into bf16 planes by nq_gemm_presplit_kn (same values as W): the split engine then loads it with 16-byte loads, no split arithmetic
int nq_gemm_nn_epi(hipStream_t st, const float* G, const float* W, float* C, int M, int Nout, int Kin, const float* aux, float ea, float eb, int mode,
                   const char* tag = nullptr, const void* Bpre = nullptr);
int nq_gemm_nn_dsilu2(hipStream_t st, const float* G, const float* W, float* C, float* C2, const float* aux, int M, int Nout, int Kin, const char* tag, const void* Bpre = nullptr);
int nq_gemm_presplit_kn(hipStream_t st, int n, const float* const* W, const int* Kc, const int* N, void* const* out);
int nq_gemm_nt_dsilu(hipStream_t st, const float* A, const float* W, float* C, float* C2, const float* aux, int M, int N, int K, const char* tag = nullptr);
int nq_gemm_nt_act(hipStream_t, const float* A, const float* W, float* C, float* C2, const float* resid, float ea, float eb, int M, int N, int K,
                   const char* tag = nullptr);
int nq_gemm_nt(hipStream_t, const float* A, const float* W, float* C, const float* bias, float* C2_silu, int M, int N, int K, int lda,
               int ldw, int ldc, const char* tag = nullptr);
int nq_gemm_nn(hipStream_t, const float* G, const float* W, float* C, int M, int Nout, int Kin, int ldg, int ldw, int ldc, int accumulate,
               const char* tag = nullptr, const void* Bpre = nullptr);
int nq_gemm_nt_res(hipStream_t st, const float* A, const float* W, float* C, const float* aux, float ea, int M, int N, int K);
size_t nq_gemm_tn_scratch_floats(long rows, int Mo, int No);
int nq_gemm_tn(hipStream_t, const float* GY, const float* X, float* out, long rows, int Mo, int No, int ldg, int ldx, float* scratch,
               const char* tag = nullptr, float* bias_out = nullptr, long bias_rows = 0);
// several weight-gradient products (+ bias gradients) in one launch and one reduction of the partial tiles; NQ_ERR_ARG = not eligible, nothing launched
struct NqTnSpec { const float* G; const float* X; float* out; long rows; int Mo, No, ldg, ldx; float* bias_out; long bias_rows; };
size_t nq_gemm_tn_group_scratch_floats(const NqTnSpec* sp, int n);
int nq_gemm_tn_group(hipStream_t, const NqTnSpec* sp, int n, float* scratch);
size_t nq_colsum_scratch_floats(long rows, int cols);
int nq_colsum(hipStream_t, const float* A, long rows, int cols, int lda, float* out, float* scratch);
int nq_reduce_partials(hipStream_t, const float* part, int nsplit, long stride, long count, float* out);

int nq_rbf(hipStream_t, const float4* geom, int E, int R, double cutoff, int env_p, float coeff, const float* offsets, float* rho,
           float* drho, int type = 0, const float* theta = nullptr);
int nq_rbf_param_grad(hipStream_t, const float4* geom, int E, int R, double cutoff, int env_p, const float* offsets, int type, const float* theta,
                      const float* grho, float* contrib);
int nq_msg_fwd(hipStream_t, const MsgArgs&, bool tangent);
bool nq_filter_fits_lds(int F, int R);
void nq_make_filter_args(FilterArgs* fa, const float* WRT, const float* br, const float* mu, const float* RW, int R, double cutoff, int env_p,
                         float coeff, int mode = 0);
int nq_rbf_window(hipStream_t, const float4* geom, int E, const FilterArgs& fa, float* RW);
int nq_transpose(hipStream_t, const float* in, int rows, int cols, float* out);
size_t nq_k0_sort_scratch_ints(int E, int R);
// mol_ptr / atom_mol / cap / count_out (optional, with row_of / col): only the lower slots of molecules of MORE than cap atoms; their number goes to *count_out
int nq_k0_sort(hipStream_t, const float* RW, int E, int R, int* order, int* scratch, const int* row_of = nullptr, const int* col = nullptr,
               const int* mol_ptr = nullptr, const int* atom_mol = nullptr, int cap = 0, int* count_out = nullptr);
size_t nq_gwr_scratch_floats(int E, int F, int R, int parts = 3);
// count_dev (optional): device int holding the number of valid entries of order[] (<= E)
int nq_gwr_sorted(hipStream_t, const float* GPHI, const float* GPSI, const float* RW, const int* order, int E, int F, int R, float* gWr,
                  float* scratch, int parts = 3, const int* count_dev = nullptr);
int nq_msgf_fwd(hipStream_t, const MsgArgs&, const FilterArgs&, bool tangent);
int nq_msgf_rev(hipStream_t, const MsgRevArgs&, const FilterArgs&, bool dual, bool pair_rows = true);   // pair_rows = false: the dual flavour neither writes gphi / gpsi nor GBR (molpair.hip computes the rbf_proj gradient); MsgRevArgs::row_filter selects the rows
// rbf_proj gradient with the molecule's node rows staged in LDS (molpair.hip): no gphi / gpsi arrays
int nq_molgw_max_atoms(void);                      // largest molecule whose 20 rows of a 32-channel slice fit the LDS (64)
bool nq_molgw_config_ok(int F, int R);             // channel count / window count supported
bool nq_molgw_supported(int F, int R, int max_mol_atoms);
size_t nq_molgw_sched_ints(int E, int B);
size_t nq_molgw_sched_slots(int E, int B);
size_t nq_molgw_rec_floats(int E, int B);
size_t nq_molgw_part_floats(int F, int B);
// cap: molecules of more atoms are left out of the schedule (their pairs go through the pair-row kernels)
int nq_molgw_schedule(hipStream_t, const NqGraphView&, const int* dst, const float* RW, int R, int cap, int* sched_ints, float* recs);
int nq_molgw_geometry(hipStream_t, const NqGraphView&, const float* RW, const float* TD, const float* TR, const int* sched_ints, float* recs);
int nq_gwr_mol(hipStream_t, const NqGraphView&, int F, int R, int max_mol_atoms, const float* XH, const float* V, const float* TXH, const float* TV,
               const float* GX, const float* GV, const float* GTX, const float* GTV, const int* sched_ints, const float* recs, float* part, float* gWr,
               float* gbr, bool accumulate = false);
int nq_msg_rev(hipStream_t, const MsgRevArgs&, bool dual);
int nq_geom_tan(hipStream_t, const NqGraphView&, const int* dst, const float* pos_dot, float* TD, float* TR);
int nq_geom_rev(hipStream_t, const NqGraphView&, const float4* GEDGE, int nwaves, float* forces);

// the whole update block of one layer and sweep as one kernel (updfuse.hip; hidden_channels = 128): weight fragments once per forward call, then one launch
bool nq_gemm_exact_f32_requested();   // the exact-f32 engine was asked for (NQ_GEMM_F32=1 / nq_set_gemm_variant(32)): kernels that only exist on the bf16 matrix pipe step aside
size_t nq_updfuse_frag_floats(int F);
int nq_updfuse_presplit(hipStream_t, const float* U, const float* V1, const float* V2, int F, float* frag);
int nq_upd_fused(hipStream_t, const UpdArgs&, const float* frag, const float* c1, const float* c2, float* ZQ, float* Q, float* TZQ, float* TQ, bool tan);
int nq_updrev_fused(hipStream_t, const UpdRevArgs&, const float* frag, const float* ZQ);   // force-adjoint sweep: rev1, three input-gradient products, rev2 in one kernel
int nq_upd_a(hipStream_t, const UpdArgs&, bool tan);
int nq_upd_b(hipStream_t, const UpdArgs&, bool tan);
int nq_silu_tan(hipStream_t, const float* Z, const float* TZ, float* TH, long count);
int nq_silu_rev(hipStream_t, const float* Z, const float* TZ, float* G, float* GT, long count, bool dual, bool lite = false);   // lite: GT = the tangent adjoint of the layer OUTPUT, read only
int nq_upd_rev(hipStream_t, const UpdRevArgs&, int stage, bool dual);
int nq_embed(hipStream_t, const int* z, const float* emb, int N, int F, float* X0);
size_t nq_embed_grad_scratch_floats(int N, int F, int T);
int nq_embed_grad(hipStream_t, const int* z, const float* GX, int N, int F, int T, float* out, float* scratch);
int nq_readout(hipStream_t, const ReadoutArgs&, int mode);
int nq_readout_rev(hipStream_t, const ReadoutArgs&, bool dual);
int nq_mol_sum(hipStream_t, const float* e_atom, const int* mol_ptr, int B, float* out);
int nq_atom_seeds(hipStream_t, const float* gE, const int* atom_mol, int N, float* ge, float* gte);
int nq_negate(hipStream_t, const float* in, float* out, long count);
int nq_axpy(hipStream_t, const float* x, float* y, long count);   // y += x
int nq_loss_impl(hipStream_t, const float* E, const float* y, int B, const float* Fc, const float* Ft, int N, float ce, float cf, float* loss,
                 float* gE, float* gF, bool mse);
int nq_adamw_impl(hipStream_t, float* p, const float* g, float* m, float* v, long count, float max_norm, float lr, float beta1, float beta2,
                  float eps, float wd, int step, float* scratch);
