// Host orchestration + C ABI (include/nablaq.h).  A PaiNN step is four sweeps over a caller-owned
// workspace (see oracle/painn_sweeps.py for the math and buffer names):
//   forward, force adjoint                 -> nq_painn_forward
//   tangent forward, dual reverse          -> nq_painn_backward
// Every dual-capable buffer is stored stacked [2][rows][width] (primal, tangent) so that the second-
// order sweep runs each GEMM once over 2x the rows and each weight gradient as ONE contraction.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/nablaq.h"
#include "common.h"

thread_local char nq_err_buf[512] = "";
int nq_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(nq_err_buf, sizeof(nq_err_buf), fmt, ap);
  va_end(ap);
  return code;
}

// ---- dynamic-LDS opt-in, per (device, kernel) ---------------------------------------------------------
#include <map>
#include <mutex>
#include <utility>
static std::mutex g_lds_mu;
static std::map<std::pair<int, const void*>, size_t> g_lds_granted;
int nq_dyn_lds(const void* kernel, size_t bytes) {
  int dev = 0;
  NQ_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_lds_mu);
  size_t& have = g_lds_granted[std::make_pair(dev, kernel)];
  if (bytes > have) {
    NQ_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    have = bytes;
  }
  return NQ_OK;
}

// ---- profiler ---------------------------------------------------------------------------------------
#include <map>
#include <mutex>
#include <utility>
#include <string>
#include <vector>
int nq_profile_on = 0;
struct ProfRec { hipEvent_t a, b; int name_id;  double flops = 0.0; };
static std::vector<ProfRec> g_prof_recs;
static std::vector<std::string> g_prof_names;
static std::map<std::string, int> g_prof_ids;
NqProfScope::NqProfScope(hipStream_t s, const char* name) : st(s), slot(-1) {
  if (!nq_profile_on) return;
  auto it = g_prof_ids.find(name);
  int id;
  if (it == g_prof_ids.end()) { id = (int)g_prof_names.size(); g_prof_names.push_back(name); g_prof_ids[name] = id; }
  else id = it->second;
  ProfRec r; r.name_id = id;
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
  (void)hipEventRecord(r.a, st);
  slot = (int)g_prof_recs.size();
  g_prof_recs.push_back(r);
}
NqProfScope::~NqProfScope() {
  if (slot >= 0) (void)hipEventRecord(g_prof_recs[slot].b, st);
}
void NqProfScope::add_flops(double f) {
  if (slot >= 0) g_prof_recs[slot].flops += f;
}

// ---- parameter layout --------------------------------------------------------------------------
struct MsgP { size_t W1, b1, W2, b2, Wr, br; };
struct UpdP { size_t U, V1, c1, V2, c2; };
struct ParamLayout {
  size_t emb, basis, n_basis, O1, o1, w2, o2, total;
  MsgP msg[64];
  UpdP upd[64];
};
static int make_param_layout(const nq_painn_cfg* c, ParamLayout* P) {
  const size_t F = c->hidden_channels, R = c->num_rbf, H = F / 2, T = c->num_elements;
  if (c->num_layers < 1 || c->num_layers > 64) return nq_fail(NQ_ERR_ARG, "num_layers=%d out of range [1,64]", c->num_layers);
  if (c->rbf_type < 0 || c->rbf_type > 2) return nq_fail(NQ_ERR_ARG, "rbf_type=%d unknown", c->rbf_type);
  if (c->rbf_type != 0 && c->filter_mode != 0) return nq_fail(NQ_ERR_ARG, "learnable bases are built for filter_mode 0 only");
  if (F % 64 != 0 || F < 64 || F > 1024) return nq_fail(NQ_ERR_ARG, "hidden_channels=%zu must be a multiple of 64 in [64,1024]", F);
  if (R < 1 || T < 1) return nq_fail(NQ_ERR_ARG, "num_rbf / num_elements must be positive");
  size_t o = 0;
  P->emb = o; o += T * F;
  P->basis = o; P->n_basis = c->rbf_type == 1 ? R : (c->rbf_type == 2 ? 1 : 0); o += P->n_basis;   // radial_basis.rbf.{frequencies | pregamma}
  for (int l = 0; l < c->num_layers; ++l) {
    MsgP& m = P->msg[l];
    m.W1 = o; o += F * F; m.b1 = o; o += F; m.W2 = o; o += 3 * F * F; m.b2 = o; o += 3 * F; m.Wr = o; o += 3 * F * R; m.br = o; o += 3 * F;
  }
  for (int l = 0; l < c->num_layers; ++l) {
    UpdP& u = P->upd[l];
    u.U = o; o += 2 * F * F; u.V1 = o; o += F * 2 * F; u.c1 = o; o += F; u.V2 = o; o += 3 * F * F; u.c2 = o; o += 3 * F;
  }
  P->O1 = o; o += H * F; P->o1 = o; o += H; P->w2 = o; o += H; P->o2 = o; o += 1;
  P->total = o;
  return NQ_OK;
}

// ---- workspace layout ------------------------------------------------------------------------
struct WsLayer {
  size_t Z1, Hh, XH, PHI, PSI, XM, VM, UU, S, CAT, ZQ, Q, Y;  // float offsets; dual buffers hold [2][rows][w]
  size_t WRT;                                                  // [R][3F] transposed rbf_proj.weight
  size_t UFRAG;                                                // bf16 fragments of the update block's weights (updfuse.hip), rebuilt by every forward call
  size_t WPRE;                                                 // bf16 planes of V2, V1, U, W2, W1 for the input-gradient products (gemm_split.h PreStageB), rebuilt by every forward call
  // Per-layer adjoint store (fused filter only).  The tangent adjoints of the second-order sweep ARE the adjoints of the force sweep (same recursion, same seeds),
  // so the force sweep writes its gy, gcat, gu, gxh into the SECOND halves of these stacked [2][rows][w] buffers and keeps the four stages of gx / gvec
  // (a: adjoint of x_upd / vec_upd, b: of x_msg / vec_msg); the second-order sweep fills the first halves and reads the second ones instead of recomputing them.
  size_t LGY, LGCAT, LGU, LGXH, LGXA, LGXB, LGVA, LGVB;
  size_t LGQ, LGH, LGQP, LGHP;   // stacked [2][N][F] adjoints of the two SiLU layers' pre-activations (second half: force sweep) and, [N][F], of their outputs
};
struct WsLayout {
  size_t X[65], V[65];
  WsLayer lay[64];
  size_t RHO2, RW, ORDER, ZO, e_atom, te_atom, TD, TR, pos_dot, ge, gte;
  size_t GX, GVa, GVb, GY, GQ, GCAT, GU, GXH, GH, GBR, GPHI2, GEDGE, GZO, TMPW, GRHO, BCON, ROWCTR, SCHED, GWREC, GWPART, scratch;
  size_t scratch_floats, total_floats;
  bool fused;   // radial filter evaluated inside the message kernels (WrT resident in LDS); PHI/PSI not materialised
};
static size_t a4(size_t x) { return (x + 3) & ~(size_t)3; }  // keep every buffer 16-byte aligned

static bool use_fused_filter(const nq_painn_cfg* c) {
  const char* off = getenv("NQ_NO_FUSED_FILTER");
  return c->rbf_type == 0 && nq_filter_fits_lds(c->hidden_channels, c->num_rbf) && !(off && off[0] == '1');   // the window needs compact Gaussians
}

static void tn_group_shapes(NqTnSpec (&sp)[5], long N, int F);
static void make_ws_layout(const nq_painn_cfg* c, size_t N, size_t E, size_t B, WsLayout* W) {
  const size_t F = c->hidden_channels, R = c->num_rbf, H = F / 2, L = c->num_layers, T = c->num_elements;
  W->fused = use_fused_filter(c);
  const size_t EP = W->fused ? 0 : E;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += a4(n); return r; };
  for (size_t l = 0; l <= L; ++l) { W->X[l] = take(2 * N * F); W->V[l] = take(2 * N * 3 * F); }
  for (size_t l = 0; l < L; ++l) {
    WsLayer& y = W->lay[l];
    y.Z1 = take(2 * N * F); y.Hh = take(2 * N * F); y.XH = take(2 * N * 3 * F);
    y.PHI = take(EP * 3 * F); y.PSI = take(EP * 3 * F); y.WRT = take(R * 3 * F);
    y.XM = take(2 * N * F); y.VM = take(2 * N * 3 * F); y.UU = take(2 * N * 6 * F);
    y.S = take(2 * N * F); y.CAT = take(2 * N * 2 * F); y.ZQ = take(2 * N * F); y.Q = take(2 * N * F); y.Y = take(2 * N * 3 * F);
    y.UFRAG = take(nq_updfuse_frag_floats((int)F));
    y.WPRE = take(11 * F * F * 6 / 4);
    const size_t st_ = W->fused ? 1 : 0;
    y.LGY = take(st_ * 2 * N * 3 * F); y.LGCAT = take(st_ * 2 * N * 2 * F); y.LGU = take(st_ * 2 * N * 6 * F); y.LGXH = take(st_ * 2 * N * 3 * F);
    y.LGXA = take(st_ * N * F); y.LGXB = take(st_ * N * F); y.LGVA = take(st_ * N * 3 * F); y.LGVB = take(st_ * N * 3 * F);
    y.LGQ = take(st_ * 2 * N * F); y.LGH = take(st_ * 2 * N * F); y.LGQP = take(st_ * N * F); y.LGHP = take(st_ * N * F);
  }
  W->RHO2 = take(2 * EP * R);   // full rho / drho rows only for the materialised-filter path (B operand of the gWr contraction)
  W->ORDER = take(W->fused ? E : 0);   // int32: CSR slots sorted by window start k0
  W->RW = take(W->fused ? E * 32 : 0);
  W->ZO = take(2 * N * H); W->e_atom = take(N); W->te_atom = take(N);
  W->TD = take(E); W->TR = take(3 * E); W->pos_dot = take(3 * N); W->ge = take(N); W->gte = take(N);
  W->GX = take(2 * N * F); W->GVa = take(2 * N * 3 * F); W->GVb = take(2 * N * 3 * F);
  W->GY = take(2 * N * 3 * F); W->GQ = take(2 * N * F); W->GCAT = take(2 * N * 2 * F); W->GU = take(2 * N * 6 * F);
  W->GXH = take(2 * N * 3 * F); W->GH = take(2 * N * F);
  W->GBR = take(N * 3 * F);   // per-atom sums of gphi (rbf_proj bias gradient): its own buffer so that the side stream may still be reading GY
  W->GPHI2 = take(2 * E * 3 * F);
  W->GEDGE = take((F / 64) * E * 4);
  W->GZO = take(2 * N * H); W->TMPW = take(N * H);
  W->GRHO = take(c->rbf_type ? 2 * E * R : 0); W->BCON = take(c->rbf_type ? E * R : 0);   // learnable bases: adjoints of rho / drho, per-edge contributions
  W->ROWCTR = take(5 * L * NQ_ROWCTR_INTS);   // int32 row counters of the fused message launches: [kind][layer][NQ_ROWCTR_INTS]; kind 4 = the second dual-reverse launch of a mixed batch
  W->SCHED = take(W->fused ? nq_molgw_sched_ints((int)E, (int)B) : 0);      // int32: per-molecule pair lists of the molecule-per-workgroup rbf_proj gradient (molpair.hip)
  W->GWREC = take(W->fused ? nq_molgw_rec_floats((int)E, (int)B) : 0);               // its per-pair records: expanded matrix-core A operands (per step) + packed geometry (per backward sweep)
  W->GWPART = take(W->fused ? nq_molgw_part_floats((int)F, (int)B) : 0);   // its per-workgroup partial rows
  // scratch for split-K partials / column sums / embedding partials: max over all uses
  size_t s = 0;
  auto mx = [&](size_t v) { if (v > s) s = v; };
  mx(nq_gemm_tn_scratch_floats(2 * N, 3 * F, F)); mx(nq_gemm_tn_scratch_floats(2 * N, F, 2 * F)); mx(nq_gemm_tn_scratch_floats(6 * N, 2 * F, F));
  mx(W->fused ? nq_gwr_scratch_floats((int)E, (int)F, (int)R) : nq_gemm_tn_scratch_floats(2 * E, 3 * F, R));
  if (W->fused) mx(nq_k0_sort_scratch_ints((int)E, (int)R));
  mx(nq_gemm_tn_scratch_floats(2 * N, F, F)); mx(nq_gemm_tn_scratch_floats(2 * N, H, F));
  { NqTnSpec sp[5]; tn_group_shapes(sp, (long)N, (int)F); mx(nq_gemm_tn_group_scratch_floats(sp, 5)); }
  mx(nq_colsum_scratch_floats(N > E ? N : E, 3 * F));
  if (c->rbf_type) { mx(nq_colsum_scratch_floats(E, R)); mx(nq_colsum_scratch_floats(E * R, 1)); }
  mx(nq_embed_grad_scratch_floats((int)N, (int)F, (int)T));
  W->scratch_floats = s;
  W->scratch = take(s);
  W->total_floats = o;
  (void)B;
}

// rbf_proj gradient from node rows staged per molecule in LDS (molpair.hip) instead of gphi / gpsi pair rows through HBM: needs the fused filter.
//   GW_PAIR_ROWS  the dual-reverse kernel writes gphi / gpsi per pair, k_gwr_sorted reads them back (small batches; NQ_NO_MOLGW=1 forces it)
//   GW_MOLECULE   every molecule fits the LDS of one workgroup (nq_graph::max_mol_atoms <= cap): k_gwr_mol, nothing stored per pair
//   GW_MIXED      some molecules are larger than cap: THEIR rows go through the pair-row kernels, every other molecule stays on k_gwr_mol
// The decision is taken ONCE per step, by the forward call (which builds the schedule), and remembered per workspace: the backward call reads it back instead
// of re-evaluating the environment (a changed NQ_MOLGW between the two calls would consume a schedule that was never built).
enum { GW_PAIR_ROWS = 0, GW_MOLECULE = 1, GW_MIXED = 2 };
struct GwMode { int mode; int cap; int lite; };   // lite: the force sweep of this step left its per-layer adjoints in the workspace (WsLayer::LG*)
static int molgw_cap() {
  const char* c = getenv("NQ_MOLGW_CAP");   // tests: a small cap sends ordinary molecules down the mixed path
  const int hw = nq_molgw_max_atoms();
  if (c && atoi(c) > 0 && atoi(c) < hw) return atoi(c);
  return hw;
}
static GwMode decide_molgw(const nq_painn_cfg* c, const nq_graph* g, const WsLayout& W) {
  GwMode r{GW_PAIR_ROWS, molgw_cap(), 0};
  const char* off = getenv("NQ_NO_MOLGW");
  if (!W.fused || (off && off[0] == '1') || !nq_molgw_config_ok(c->hidden_channels, c->num_rbf) || g->max_mol_atoms <= 0) return r;
  // small batches (the reference's 32 conformers): the schedule kernels and the two launches per layer are pure latency there (measured 3.99 vs 3.72 ms per
  // step at 32 conformers), the pair rows are a few MB; NQ_MOLGW=1 forces the per-molecule path at any size (tests)
  const char* on = getenv("NQ_MOLGW");
  if (!((on && on[0] == '1') || g->N >= 4096)) return r;
  r.mode = g->max_mol_atoms <= r.cap ? GW_MOLECULE : GW_MIXED;
  return r;
}
static std::mutex g_gw_mu;
static std::map<const void*, GwMode> g_gw_modes;   // by workspace: written by the forward call, read by the backward call of the same step
static void remember_molgw(const void* ws, GwMode m) { std::lock_guard<std::mutex> lock(g_gw_mu); g_gw_modes[ws] = m; }
static bool recall_molgw(const void* ws, GwMode* m) {
  std::lock_guard<std::mutex> lock(g_gw_mu);
  auto it = g_gw_modes.find(ws);
  if (it == g_gw_modes.end()) return false;
  *m = it->second;
  return true;
}

// update block of a layer as one kernel per sweep (updfuse.hip): hidden_channels = 128; NQ_NO_FUSED_UPDATE=1 keeps the five launches of rounds 1-5 (A/B runs, tests)
static bool use_fused_update(const nq_painn_cfg* c) {
  const char* off = getenv("NQ_NO_FUSED_UPDATE");
  // (the fused kernel's products exist on the split-bf16 matrix pipe only: a run that asks for the exact-f32 engine gets the five launches)
  return nq_updfuse_frag_floats(c->hidden_channels) > 0 && !(off && off[0] == '1') && !nq_gemm_exact_f32_requested();
}

// The five weight-gradient products of one layer's dual-reverse sweep as ONE grouped launch (gemm.hip: nq_gemm_tn_group): built, parity-tested,
// NOT the default.  Measured (profiles/r06_tn_group_ab.txt): 2048 conformers 49.56-49.85 ms per step grouped vs 49.29-49.40 one launch each; 32 conformers 4.19 vs
// 3.73 ms -- the separate launches run on the side stream UNDER the layer's critical path, the grouped launch can only start when the layer's last adjoint exists, and
// at the large size the 4.5x smaller partial-tile traffic does not pay for the longer serial row streams per workgroup.  NQ_TN_GROUP=1 selects it.
static bool use_tn_group() {
  const char* on = getenv("NQ_TN_GROUP");
  return on && on[0] == '1';
}
static void tn_group_shapes(NqTnSpec (&sp)[5], long N, int F) {   // rows / output shapes only (workspace sizing); the engine fills in the pointers
  const int F2 = 2 * F, F3 = 3 * F;
  sp[0] = NqTnSpec{nullptr, nullptr, nullptr, 2 * N, F3, F, F3, F, reinterpret_cast<float*>(1), N};    // V2: gy^T q  (+ c2)
  sp[1] = NqTnSpec{nullptr, nullptr, nullptr, 2 * N, F, F2, F, F2, reinterpret_cast<float*>(1), N};    // V1: gq^T cat (+ c1)
  sp[2] = NqTnSpec{nullptr, nullptr, nullptr, 6 * N, F2, F, F2, F, nullptr, 0};                        // U:  gu^T vec_msg
  sp[3] = NqTnSpec{nullptr, nullptr, nullptr, 2 * N, F3, F, F3, F, reinterpret_cast<float*>(1), N};    // W2: gxh^T h (+ b2)
  sp[4] = NqTnSpec{nullptr, nullptr, nullptr, 2 * N, F, F, F, F, reinterpret_cast<float*>(1), N};      // W1: gh^T x (+ b1)
}

// Weights pre-split into bf16 planes once per step for the input-gradient products of both reverse sweeps: they load their weight tile with
// three 16-byte loads per thread instead of eight 4-byte loads + the split arithmetic.  Measured SLOWER at 2048 conformers (49.53 / 49.56 ms per step against
// 48.81 / 48.95 ms, same box, profiles/r06_presplit_nn_ab.txt: the planes are 6 bytes per element against 4 and one epilogue flavour spills), so it is opt-in:
// NQ_PRESPLIT=1, hidden_channels % 128 == 0, never under the exact-f32 engine.
struct PrePlanes { const void* V2; const void* V1; const void* U; const void* W2; const void* W1; };
static bool use_presplit(const nq_painn_cfg* c) {
  const char* on = getenv("NQ_PRESPLIT");
  return c->hidden_channels % 128 == 0 && on && on[0] == '1' && !nq_gemm_exact_f32_requested();
}
static PrePlanes pre_planes(const nq_painn_cfg* c, float* base) {
  PrePlanes q{nullptr, nullptr, nullptr, nullptr, nullptr};
  if (!use_presplit(c)) return q;
  const size_t F = c->hidden_channels, FF = F * F;
  unsigned short* b = reinterpret_cast<unsigned short*>(base);
  q.V2 = b; q.V1 = b + 3 * 3 * FF; q.U = b + 3 * 5 * FF; q.W2 = b + 3 * 7 * FF; q.W1 = b + 3 * 10 * FF;
  return q;
}

static NqGraphView view_of(const nq_graph* g) {
  NqGraphView v;
  v.N = g->N; v.B = g->B; v.E = g->E;
  v.mol_ptr = g->mol_ptr; v.row_ptr = g->row_ptr; v.col = g->col; v.rev = g->rev;
  v.geom = reinterpret_cast<const float4*>(g->geom); v.z = g->z; v.atom_mol = g->atom_mol; v.lowptr = g->lowptr;
  return v;
}

static int check_common(const nq_painn_cfg* cfg, const nq_graph* g, const void* ws, size_t ws_bytes, WsLayout* W, ParamLayout* P) {
  if (!cfg || !g || !ws) return nq_fail(NQ_ERR_ARG, "null argument");
  NQ_TRY(make_param_layout(cfg, P));
  if (g->N <= 0 || g->B <= 0) return nq_fail(NQ_ERR_ARG, "empty batch");
  if (g->E <= 0) return nq_fail(NQ_ERR_NO_EDGES, "batch has no edges within the cutoff");
  if ((reinterpret_cast<uintptr_t>(ws) & 15) != 0) return nq_fail(NQ_ERR_ARG, "workspace must be 16-byte aligned");
  make_ws_layout(cfg, g->N, g->E, g->B, W);
  if (cfg->filter_mode != 0 && cfg->filter_mode != 1) return nq_fail(NQ_ERR_ARG, "filter_mode must be 0 (painn_pyg) or 1 (schnetpack)");
  if (cfg->filter_mode == 1 && !W->fused)
    return nq_fail(NQ_ERR_ARG, "filter_mode 1 (schnetpack) needs the fused-filter path: hidden_channels in {64,128,256} and WrT fitting the LDS");
  if (ws_bytes < W->total_floats * sizeof(float))
    return nq_fail(NQ_ERR_WORKSPACE, "workspace too small: %zu < %zu bytes", ws_bytes, W->total_floats * sizeof(float));
  return NQ_OK;
}

// ================================================================================================
extern "C" {

int nq_abi_version(void) { return NQ_ABI_VERSION; }
int32_t nq_painn_molecule_lds_atoms(void) { return nq_molgw_max_atoms(); }

void nq_profile_enable(int32_t on) { nq_profile_on = on; }
// Synchronises the device, folds all recorded event pairs into per-name totals and clears them.
// Fills up to `cap` entries; returns the number of distinct names.
int nq_profile_read2(char* names, int32_t name_stride, double* total_ms, int64_t* counts, double* flops, int32_t cap) {
  (void)hipDeviceSynchronize();
  std::vector<double> tot(g_prof_names.size(), 0.0), fl(g_prof_names.size(), 0.0);
  std::vector<long long> cnt(g_prof_names.size(), 0);
  for (auto& r : g_prof_recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { tot[r.name_id] += ms; cnt[r.name_id] += 1; fl[r.name_id] += r.flops; }
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
  }
  g_prof_recs.clear();
  int n = (int)g_prof_names.size();
  for (int i = 0; i < n && i < cap; ++i) {
    snprintf(names + (size_t)i * name_stride, name_stride, "%s", g_prof_names[i].c_str());
    total_ms[i] = tot[i]; counts[i] = cnt[i];
    if (flops) flops[i] = fl[i];
  }
  return n;
}
int nq_profile_read(char* names, int32_t name_stride, double* total_ms, int64_t* counts, int32_t cap) {
  return nq_profile_read2(names, name_stride, total_ms, counts, nullptr, cap);
}
const char* nq_last_error(void) { return nq_err_buf; }

int nq_graph_count(const float* pos, const int32_t* mol_ptr, int32_t N, int32_t B, int32_t max_mol_atoms, double cutoff,
                   int32_t max_neighbors, int32_t* deg, int32_t* lowdeg, int32_t* row_ptr, int32_t* lowptr, int32_t* E_host, void* stream) {
  if (!pos || !mol_ptr || !deg || !lowdeg || !row_ptr || !lowptr || !E_host) return nq_fail(NQ_ERR_ARG, "null argument");
  if (max_neighbors < 1) return nq_fail(NQ_ERR_ARG, "max_neighbors must be >= 1");
  return nq_graph_count_impl(pos, mol_ptr, N, B, max_mol_atoms, (float)(cutoff * cutoff), max_neighbors, deg, lowdeg, row_ptr, lowptr,
                             E_host, (hipStream_t)stream);
}

int nq_graph_fill(const float* pos, const int32_t* mol_ptr, int32_t N, int32_t B, int32_t E, int32_t max_mol_atoms, double cutoff,
                  int32_t max_neighbors, const int32_t* row_ptr, const int32_t* lowptr, int32_t* col, int32_t* dst, int32_t* rev, float* geom,
                  int32_t* slot2canon, int32_t* atom_mol, int64_t* edge_index, float* edge_dist, float* edge_vector, int64_t* id_swap,
                  int64_t* neighbors, void* stream) {
  if (!pos || !mol_ptr || !row_ptr || !lowptr || !col || !dst || !rev || !geom || !slot2canon || !atom_mol)
    return nq_fail(NQ_ERR_ARG, "null argument");
  if (edge_index && (!edge_dist || !edge_vector || !id_swap)) return nq_fail(NQ_ERR_ARG, "canonical outputs must be all set or all NULL");
  if ((reinterpret_cast<uintptr_t>(geom) & 15) != 0) return nq_fail(NQ_ERR_ARG, "geom must be 16-byte aligned");
  GraphFillArgs a;
  a.pos = pos; a.mol_ptr = mol_ptr; a.row_ptr = row_ptr; a.lowptr = lowptr;
  a.r2 = (float)(cutoff * cutoff); a.K = max_neighbors;
  a.col = col; a.dst = dst; a.rev = rev; a.geom = reinterpret_cast<float4*>(geom); a.slot2canon = slot2canon;
  a.c_src = (long long*)edge_index; a.c_dst = edge_index ? (long long*)edge_index + E : nullptr;
  a.c_dist = edge_dist; a.c_vec = edge_vector; a.id_swap = (long long*)id_swap; a.neighbors = (long long*)neighbors;
  a.atom_mol = atom_mol;
  (void)N;
  return nq_graph_fill_impl(a, B, max_mol_atoms, (hipStream_t)stream);
}

size_t nq_painn_num_params(const nq_painn_cfg* cfg) {
  ParamLayout P;
  if (!cfg || make_param_layout(cfg, &P) != NQ_OK) return 0;
  return P.total;
}

size_t nq_painn_workspace_bytes(const nq_painn_cfg* cfg, int32_t N, int32_t E, int32_t B) {
  ParamLayout P;
  if (!cfg || make_param_layout(cfg, &P) != NQ_OK || N <= 0 || E < 0) return 0;
  WsLayout W;
  make_ws_layout(cfg, N, E, B, &W);
  return W.total_floats * sizeof(float);
}

int nq_painn_ws_lookup(const nq_painn_cfg* cfg, int32_t N, int32_t E, int32_t B, const char* name, int32_t l, int32_t tangent,
                       size_t* off, size_t* count) {
  ParamLayout P;
  if (!cfg || !name || !off || !count) return nq_fail(NQ_ERR_ARG, "null argument");
  NQ_TRY(make_param_layout(cfg, &P));
  WsLayout W;
  make_ws_layout(cfg, N, E, B, &W);
  const size_t F = cfg->hidden_channels, R = cfg->num_rbf, H = F / 2, n = N, e = E;
  const int L = cfg->num_layers;
  size_t base = 0, rows = 0, w = 0;
  bool dual = true;
#define LAYER_OK(maxl) if (l < 0 || l > (maxl)) return nq_fail(NQ_ERR_ARG, "layer %d out of range", l)
  if (!strcmp(name, "x_in")) { LAYER_OK(L); base = W.X[l]; rows = n; w = F; }
  else if (!strcmp(name, "vec_in")) { LAYER_OK(L); base = W.V[l]; rows = n; w = 3 * F; }
  else if (!strcmp(name, "z1")) { LAYER_OK(L - 1); base = W.lay[l].Z1; rows = n; w = F; }
  else if (!strcmp(name, "h")) { LAYER_OK(L - 1); base = W.lay[l].Hh; rows = n; w = F; }
  else if (!strcmp(name, "xh")) { LAYER_OK(L - 1); base = W.lay[l].XH; rows = n; w = 3 * F; }
  else if (!strcmp(name, "phi") || !strcmp(name, "psi")) {
    LAYER_OK(L - 1);
    if (W.fused) return nq_fail(NQ_ERR_ARG, "buffer '%s' is not materialised (filter is fused into the message kernels)", name);
    base = name[1] == 'h' ? W.lay[l].PHI : W.lay[l].PSI; rows = e; w = 3 * F; dual = false;
  }
  else if (!strcmp(name, "x_msg")) { LAYER_OK(L - 1); base = W.lay[l].XM; rows = n; w = F; }
  else if (!strcmp(name, "vec_msg")) { LAYER_OK(L - 1); base = W.lay[l].VM; rows = n; w = 3 * F; }
  else if (!strcmp(name, "u")) { LAYER_OK(L - 1); base = W.lay[l].UU; rows = n; w = 6 * F; }
  else if (!strcmp(name, "s")) { LAYER_OK(L - 1); base = W.lay[l].S; rows = n; w = F; }
  else if (!strcmp(name, "cat")) { LAYER_OK(L - 1); base = W.lay[l].CAT; rows = n; w = 2 * F; }
  else if (!strcmp(name, "zq")) { LAYER_OK(L - 1); base = W.lay[l].ZQ; rows = n; w = F; }
  else if (!strcmp(name, "q")) { LAYER_OK(L - 1); base = W.lay[l].Q; rows = n; w = F; }
  else if (!strcmp(name, "y")) { LAYER_OK(L - 1); base = W.lay[l].Y; rows = n; w = 3 * F; }
  else if (!strcmp(name, "rho")) {                                            // tangent=1 -> drho
    if (W.fused) return nq_fail(NQ_ERR_ARG, "buffer 'rho' is not materialised (filter is fused into the message kernels)");
    base = W.RHO2; rows = e; w = R;
  }
  else if (!strcmp(name, "order")) {                                          // int32 payload
    if (!W.fused) return nq_fail(NQ_ERR_ARG, "buffer 'order' exists only with the fused filter");
    base = W.ORDER; rows = e; w = 1; dual = false;
  }
  else if (!strcmp(name, "pair_sched")) {                                     // int32 payload: {slot, off << 26 | n << 13 | k} per pair (molpair.hip)
    if (!W.fused) return nq_fail(NQ_ERR_ARG, "buffer 'pair_sched' exists only with the fused filter");
    base = W.SCHED; rows = 2 * nq_molgw_sched_slots((int)e, B); w = 1; dual = false;
  }
  else if (!strcmp(name, "pair_sched_meta")) {                                // int32: seg [B][NW] {first batch, pairs}, sched_ptr [B][NW + 1], hist [128], wlo [NW + 1], order [B], 16 spare
    if (!W.fused) return nq_fail(NQ_ERR_ARG, "buffer 'pair_sched_meta' exists only with the fused filter");
    base = W.SCHED + 2 * nq_molgw_sched_slots((int)e, B); rows = nq_molgw_sched_ints((int)e, B) - 2 * nq_molgw_sched_slots((int)e, B); w = 1; dual = false;
  }
  else if (!strcmp(name, "rw")) {
    if (!W.fused) return nq_fail(NQ_ERR_ARG, "buffer 'rw' exists only with the fused filter");
    base = W.RW; rows = e; w = 32; dual = false;
  }
  else if (!strcmp(name, "zo")) { base = W.ZO; rows = n; w = H; }
  else if (!strcmp(name, "e_atom")) { base = tangent ? W.te_atom : W.e_atom; rows = n; w = 1; dual = false; tangent = 0; }
  else if (!strcmp(name, "t_d")) { base = W.TD; rows = e; w = 1; dual = false; }
  else if (!strcmp(name, "t_r")) { base = W.TR; rows = e; w = 3; dual = false; }
  else if (!strcmp(name, "gedge")) { base = W.GEDGE; rows = (F / 64) * e; w = 4; dual = false; }
  else return nq_fail(NQ_ERR_ARG, "unknown workspace buffer '%s'", name);
#undef LAYER_OK
  if (tangent && !dual) return nq_fail(NQ_ERR_ARG, "buffer '%s' has no tangent half", name);
  *off = base + (tangent ? rows * w : 0);
  *count = rows * w;
  return NQ_OK;
}

// ------------------------------------------------------------------------------------------------
int nq_painn_forward(const nq_painn_cfg* cfg, const float* params, const float* rbf_offsets, const nq_graph* graph, void* workspace,
                     size_t workspace_bytes, float* energy, float* forces, void* stream) {
  WsLayout W; ParamLayout P;
  NQ_TRY(check_common(cfg, graph, workspace, workspace_bytes, &W, &P));
  if (!params || !rbf_offsets || !energy) return nq_fail(NQ_ERR_ARG, "null argument");
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const NqGraphView g = view_of(graph);
  const int N = g.N, E = g.E, F = cfg->hidden_channels, R = cfg->num_rbf, H = F / 2, L = cfg->num_layers;
  const size_t NF = (size_t)N * F;
  int* const rowctr0 = reinterpret_cast<int*>(ws + W.ROWCTR);   // kinds 0 (forward) and 1 (force adjoint) are zeroed here, 2 / 3 in the backward call
  NQ_HIP(hipMemsetAsync(rowctr0, 0, (size_t)2 * L * NQ_ROWCTR_INTS * sizeof(int), st));
  // claimed rows pay off once every wavefront gets at least a row or two (measured: 10.7 k atoms 12.16 vs 12.57 ms per step); with fewer rows than
  // wavefronts the claim atomics are a visible part of each kernel (1.3 k atoms: dual sweep 57 -> 112 us), so small batches keep the static striding
  auto row_ctr = [&](int kind, int l) { return N >= 4096 ? rowctr0 + ((size_t)kind * L + l) * NQ_ROWCTR_INTS : (int*)nullptr; };

  // embedding; zero vec_in0 and the tangent halves of layer-0 inputs (d x0 / d pos = 0)
  NQ_TRY(nq_embed(st, g.z, params + P.emb, N, F, ws + W.X[0]));
  NQ_HIP(hipMemsetAsync(ws + W.X[0] + NF, 0, NF * sizeof(float), st));
  NQ_HIP(hipMemsetAsync(ws + W.V[0], 0, 6 * NF * sizeof(float), st));
  float* rho = ws + W.RHO2; float* drho = rho + (size_t)E * R;
  GwMode gw = decide_molgw(cfg, graph, W);   // GW_PAIR_ROWS without the fused filter
  // the force sweep below stores its per-layer adjoints for the second-order sweep (WsLayer::LG*): fused filter, the five-launch update reverse, NQ_NO_LITE unset
  const bool lite_store = W.fused && forces != nullptr && !(getenv("NQ_FUSED_UPDATE_REV") && getenv("NQ_FUSED_UPDATE_REV")[0] == '1') &&
                          !(getenv("NQ_NO_LITE") && getenv("NQ_NO_LITE")[0] == '1');
  gw.lite = lite_store ? 1 : 0;
  remember_molgw(workspace, gw);
  if (W.fused) {
    FilterArgs fa0;
    nq_make_filter_args(&fa0, nullptr, nullptr, rbf_offsets, ws + W.RW, R, cfg->cutoff, cfg->envelope_exponent, cfg->rbf_coeff, cfg->filter_mode);
    NQ_TRY(nq_rbf_window(st, g.geom, E, fa0, ws + W.RW));
    int* const sched = reinterpret_cast<int*>(ws + W.SCHED);
    if (gw.mode != GW_PAIR_ROWS) NQ_TRY(nq_molgw_schedule(st, g, graph->dst, ws + W.RW, R, gw.cap, sched, ws + W.GWREC));
    if (gw.mode == GW_PAIR_ROWS) NQ_TRY(nq_k0_sort(st, ws + W.RW, E, R, reinterpret_cast<int*>(ws + W.ORDER), reinterpret_cast<int*>(ws + W.scratch), graph->dst, g.col));   // lower slots only
    else if (gw.mode == GW_MIXED)   // lower slots of the molecules above the cap only; their number stays on the device (last spare int of the schedule block)
      NQ_TRY(nq_k0_sort(st, ws + W.RW, E, R, reinterpret_cast<int*>(ws + W.ORDER), reinterpret_cast<int*>(ws + W.scratch), graph->dst, g.col, g.mol_ptr, g.atom_mol,
                        gw.cap, sched + nq_molgw_sched_ints(E, g.B) - 1));
  } else {
    NQ_TRY(nq_rbf(st, g.geom, E, R, cfg->cutoff, cfg->envelope_exponent, cfg->rbf_coeff, rbf_offsets, rho, drho, cfg->rbf_type, params + P.basis));
  }

  const bool fused_upd = use_fused_update(cfg);
  for (int l = 0; l < L; ++l) {
    const WsLayer& y = W.lay[l]; const MsgP& mp = P.msg[l]; const UpdP& up = P.upd[l];
    if (use_presplit(cfg)) {
      const PrePlanes pp = pre_planes(cfg, ws + y.WPRE);
      const float* Wm[5] = {params + up.V2, params + up.V1, params + up.U, params + mp.W2, params + mp.W1};
      const int Kc[5] = {3 * F, F, 2 * F, 3 * F, F}, Nc[5] = {F, 2 * F, F, F, F};
      void* outp[5] = {const_cast<void*>(pp.V2), const_cast<void*>(pp.V1), const_cast<void*>(pp.U), const_cast<void*>(pp.W2), const_cast<void*>(pp.W1)};
      NQ_TRY(nq_gemm_presplit_kn(st, 5, Wm, Kc, Nc, outp));
    }
    NQ_TRY(nq_gemm_nt(st, ws + W.X[l], params + mp.W1, ws + y.Z1, params + mp.b1, ws + y.Hh, N, F, F, F, F, F, "W1"));
    NQ_TRY(nq_gemm_nt(st, ws + y.Hh, params + mp.W2, ws + y.XH, params + mp.b2, nullptr, N, 3 * F, F, F, F, 3 * F, "W2"));
    MsgArgs m{};
    m.g = g; m.F = F; m.X = ws + W.X[l]; m.V = ws + W.V[l]; m.XH = ws + y.XH; m.PHI = ws + y.PHI; m.PSI = ws + y.PSI;
    m.XM = ws + y.XM; m.VM = ws + y.VM;
    if (W.fused) {
      FilterArgs fa;
      NQ_TRY(nq_transpose(st, params + mp.Wr, 3 * F, R, ws + y.WRT));
      nq_make_filter_args(&fa, ws + y.WRT, params + mp.br, rbf_offsets, ws + W.RW, R, cfg->cutoff, cfg->envelope_exponent, cfg->rbf_coeff, cfg->filter_mode);
      fa.row_ctr = row_ctr(0, l);
      NQ_TRY(nq_msgf_fwd(st, m, fa, false));
    } else {
      NQ_TRY(nq_gemm_nt(st, rho, params + mp.Wr, ws + y.PHI, params + mp.br, nullptr, E, 3 * F, R, R, R, 3 * F, "Wr"));
      NQ_TRY(nq_gemm_nt(st, drho, params + mp.Wr, ws + y.PSI, nullptr, nullptr, E, 3 * F, R, R, R, 3 * F, "Wr"));
      NQ_TRY(nq_msg_fwd(st, m, false));
    }
    UpdArgs u{};
    u.N = N; u.F = F; u.XM = ws + y.XM; u.VM = ws + y.VM; u.U = ws + y.UU; u.Y = ws + y.Y; u.S = ws + y.S; u.CAT = ws + y.CAT;
    u.X1 = ws + W.X[l + 1]; u.V1 = ws + W.V[l + 1];
    if (fused_upd) {
      NQ_TRY(nq_updfuse_presplit(st, params + up.U, params + up.V1, params + up.V2, F, ws + y.UFRAG));
      NQ_TRY(nq_upd_fused(st, u, ws + y.UFRAG, params + up.c1, params + up.c2, ws + y.ZQ, ws + y.Q, nullptr, nullptr, false));
      continue;
    }
    NQ_TRY(nq_gemm_nt(st, ws + y.VM, params + up.U, ws + y.UU, nullptr, nullptr, 3 * N, 2 * F, F, F, F, 2 * F, "U"));
    NQ_TRY(nq_upd_a(st, u, false));
    NQ_TRY(nq_gemm_nt(st, ws + y.CAT, params + up.V1, ws + y.ZQ, params + up.c1, ws + y.Q, N, F, 2 * F, 2 * F, 2 * F, F, "V1"));
    NQ_TRY(nq_gemm_nt(st, ws + y.Q, params + up.V2, ws + y.Y, params + up.c2, nullptr, N, 3 * F, F, F, F, 3 * F, "V2"));
    NQ_TRY(nq_upd_b(st, u, false));
  }
  NQ_TRY(nq_gemm_nt(st, ws + W.X[L], params + P.O1, ws + W.ZO, params + P.o1, nullptr, N, H, F, F, F, H, "O1"));
  ReadoutArgs r{};
  r.N = N; r.H = H; r.ZO = ws + W.ZO; r.w2 = params + P.w2; r.o2 = params + P.o2; r.e_atom = ws + W.e_atom;
  NQ_TRY(nq_readout(st, r, 0));
  NQ_TRY(nq_mol_sum(st, ws + W.e_atom, g.mol_ptr, g.B, energy));
  if (!forces) return NQ_OK;

  // ---- force adjoint sweep: seeds dE_tot/de_i = 1 --------------------------------------------
  NQ_TRY(nq_atom_seeds(st, nullptr, g.atom_mol, N, ws + W.ge, nullptr));
  r.ge = ws + W.ge; r.GZO = ws + W.GZO;
  NQ_TRY(nq_readout_rev(st, r, false));
  // lite_store: every stage of the sweep goes to its own per-layer buffer (no in-place updates), in the layout the second-order sweep reads as its tangent adjoints
  float* gx_cur = lite_store ? ws + W.lay[L - 1].LGXA : ws + W.GX;     // adjoint of x_upd of the layer being processed
  NQ_TRY(nq_gemm_nn(st, ws + W.GZO, params + P.O1, gx_cur, N, H, F, H, F, F, 0, "O1"));
  float* gv_cur = lite_store ? ws + W.lay[L - 1].LGVA : ws + W.GVa; float* gv_oth = ws + W.GVb;
  NQ_HIP(hipMemsetAsync(gv_cur, 0, 3 * NF * sizeof(float), st));
  const int nwaves = F / 64;
  NQ_HIP(hipMemsetAsync(ws + W.GEDGE, 0, (size_t)nwaves * E * 4 * sizeof(float), st));
  for (int l = L - 1; l >= 0; --l) {
    const WsLayer& y = W.lay[l]; const MsgP& mp = P.msg[l]; const UpdP& up = P.upd[l];
    const PrePlanes pp = pre_planes(cfg, ws + y.WPRE);
    float* const GY = lite_store ? ws + y.LGY + 3 * NF : ws + W.GY;
    float* const GCAT = lite_store ? ws + y.LGCAT + 2 * NF : ws + W.GCAT;
    float* const GU = lite_store ? ws + y.LGU + 6 * NF : ws + W.GU;
    float* const GXH = lite_store ? ws + y.LGXH + 3 * NF : ws + W.GXH;
    float* const gx_msg = lite_store ? ws + y.LGXB : gx_cur;            // adjoint of x_msg: gx_upd + gcat[:F]
    float* const gv_msg = lite_store ? ws + y.LGVB : gv_cur;            // adjoint of vec_msg: gvec_upd + gu U
    float* const gx_next = lite_store ? (l > 0 ? ws + W.lay[l - 1].LGXA : ws + W.GX) : gx_cur;
    float* const gv_next = lite_store ? (l > 0 ? ws + W.lay[l - 1].LGVA : ws + W.GVb) : gv_oth;
    UpdRevArgs u{};
    u.N = N; u.F = F; u.U = ws + y.UU; u.Y = ws + y.Y; u.S = ws + y.S; u.CAT = ws + y.CAT;
    u.GX = gx_cur; u.GV = gv_cur; u.GY = GY; u.GCAT = GCAT; u.GU = GU;
    u.GX_out = lite_store ? gx_msg : nullptr;
    // Force-adjoint flavour of the fused update block: built, parity-tested, NOT the default -- 3.47 ms per step against 3.34 ms for the five launches below
    // (five dependent products with ten barriers and 400 four-byte loads per lane at eight wavefronts per CU: profiles/r06_fused_update_ab.txt); NQ_FUSED_UPDATE_REV=1 selects it.
    if (fused_upd && getenv("NQ_FUSED_UPDATE_REV") && getenv("NQ_FUSED_UPDATE_REV")[0] == '1') {
      NQ_TRY(nq_updrev_fused(st, u, ws + y.UFRAG, ws + y.ZQ));   // no weight gradients in this sweep: gy, gq, gcat, gu never leave the chip
    } else {
    NQ_TRY(nq_upd_rev(st, u, 1, false));
    // G_Q = (G_Y V2) * silu'(Z_Q): the activation's adjoint in the epilogue of the input-gradient product (no separate k_silu_rev pass)
    float* const GQ = lite_store ? ws + y.LGQ + NF : ws + W.GQ;
    if (lite_store) NQ_TRY(nq_gemm_nn_dsilu2(st, GY, params + up.V2, ws + y.LGQP, GQ, ws + y.ZQ, N, 3 * F, F, "V2", pp.V2));   // keeps G_Y V2 as well (silu'' term of the second-order sweep)
    else NQ_TRY(nq_gemm_nn_epi(st, GY, params + up.V2, GQ, N, 3 * F, F, ws + y.ZQ, 0.f, 1.f, 1, "V2", pp.V2));
    NQ_TRY(nq_gemm_nn(st, GQ, params + up.V1, GCAT, N, F, 2 * F, F, 2 * F, 2 * F, 0, "V1", pp.V1));
    NQ_TRY(nq_upd_rev(st, u, 2, false));
    if (lite_store) NQ_TRY(nq_gemm_nn_epi(st, GU, params + up.U, gv_msg, 3 * N, 2 * F, F, gv_cur, 1.f, 0.f, 0, "U", pp.U));   // gv_msg = gv_upd + gu U
    else NQ_TRY(nq_gemm_nn(st, GU, params + up.U, gv_cur, 3 * N, 2 * F, F, 2 * F, F, F, 1, "U", pp.U));
    }
    MsgRevArgs m{};
    m.g = g; m.F = F; m.V = ws + W.V[l]; m.XH = ws + y.XH; m.PHI = ws + y.PHI; m.PSI = ws + y.PSI;
    m.GX = gx_msg; m.GV = gv_msg; m.GXH = GXH; m.GV_out = gv_next; m.GEDGE = reinterpret_cast<float4*>(ws + W.GEDGE);
    if (W.fused) {
      FilterArgs fa;
      nq_make_filter_args(&fa, ws + y.WRT, params + mp.br, rbf_offsets, ws + W.RW, R, cfg->cutoff, cfg->envelope_exponent, cfg->rbf_coeff, cfg->filter_mode);
      fa.row_ctr = row_ctr(1, l);
      NQ_TRY(nq_msgf_rev(st, m, fa, false));
    } else {
      NQ_TRY(nq_msg_rev(st, m, false));
    }
    float* const GH = lite_store ? ws + y.LGH + NF : ws + W.GH;
    if (lite_store) NQ_TRY(nq_gemm_nn_dsilu2(st, GXH, params + mp.W2, ws + y.LGHP, GH, ws + y.Z1, N, 3 * F, F, "W2", pp.W2));
    else NQ_TRY(nq_gemm_nn_epi(st, GXH, params + mp.W2, GH, N, 3 * F, F, ws + y.Z1, 0.f, 1.f, 1, "W2", pp.W2));
    if (lite_store) {
      NQ_TRY(nq_gemm_nn_epi(st, GH, params + mp.W1, gx_next, N, F, F, gx_msg, 1.f, 0.f, 0, "W1", pp.W1));   // gx_upd of the layer below = gx_msg + gz1 W1
      gx_cur = gx_next; gv_cur = gv_next;
    } else {
      { float* t = gv_cur; gv_cur = gv_oth; gv_oth = t; }
      NQ_TRY(nq_gemm_nn(st, GH, params + mp.W1, gx_cur, N, F, F, F, F, F, 1, "W1", pp.W1));
    }
  }
  NQ_TRY(nq_geom_rev(st, g, reinterpret_cast<const float4*>(ws + W.GEDGE), nwaves, forces));
  return NQ_OK;
}

// ------------------------------------------------------------------------------------------------
// seeded = true: first-order reverse for the direct-force model (painn.py:130-133): no tangent sweep; the tangent halves of every
// stacked buffer are zeroed so that the dual-reverse kernels reduce to the plain reverse, and the adjoints of the final node state
// (from the PaiNNOutput head, evaluated by the caller) are added to the seeds.
// ---- side stream for the weight gradients ---------------------------------------------------------------------------------------------------
// Nothing downstream of the reverse sweep waits for dL/dW (split-K TN GEMMs, the k0-sorted rbf_proj gradient, bias column sums): only the
// optimiser does.  They are issued on a second HIP stream and run under the critical path (input-gradient GEMMs, node kernels, message sweeps),
// which leaves matrix-core and bandwidth gaps: the dual message kernel waits on gathers for half of its cycles.  Ordering is by events: a side
// launch waits for its producer on the main stream; the main stream waits before it overwrites a buffer the side stream may still read.
// The side stream and its event pool belong to one (device, main stream) pair (a process that drives two GPUs, or two streams from two threads, gets
// one pair each); an early error return between fork() and join() still joins (destructor), so a stream capture is never left forked.
struct SideStream {
  hipStream_t main = nullptr, side = nullptr;
  bool on = false;
  std::vector<hipEvent_t>* pool = nullptr;
  size_t used = 0;
  bool forked = false, joined = false;
  hipEvent_t last_read[8] = {};
  SideStream() = default;
  SideStream(const SideStream&) = delete;
  SideStream& operator=(const SideStream&) = delete;
  ~SideStream() { if (on && forked && !joined) join(); }
  hipEvent_t next() {
    if (used == pool->size()) { hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr; pool->push_back(e); }
    return (*pool)[used++];
  }
  hipStream_t fork() {                       // the side stream sees everything issued on main so far
    if (!on) return main;
    hipEvent_t e = next();
    (void)hipEventRecord(e, main); (void)hipStreamWaitEvent(side, e, 0);
    forked = true; joined = false;
    return side;
  }
  void read_by_side(int buf) { if (on) { hipEvent_t e = next(); (void)hipEventRecord(e, side); last_read[buf] = e; } }
  void before_main_writes(int buf) { if (on && last_read[buf]) { (void)hipStreamWaitEvent(main, last_read[buf], 0); last_read[buf] = nullptr; } }
  void join() { if (on) { hipEvent_t e = next(); (void)hipEventRecord(e, side); (void)hipStreamWaitEvent(main, e, 0); joined = true; } }
};
enum { SB_GY = 0, SB_GQ, SB_GU, SB_GXH, SB_GH, SB_GPHI, SB_GBR };
struct SidePool { hipStream_t side = nullptr; std::vector<hipEvent_t> events; };
static std::mutex g_side_mu;
static std::map<std::pair<int, hipStream_t>, SidePool*> g_side_pools;
static void side_stream_init(SideStream& s, hipStream_t main, int n_atoms) {
  s.main = main;
  // Measured (profiles/r02_side_stream_ab.txt): at 2048 conformers / step 60.5 ms with the side stream vs 60.1 ms without; at 32 conformers (1.3 k atoms, the step is
  // a chain of ~330 small dependent kernels) 4.40 vs 4.71 ms, at 256 conformers 11.57 vs 11.78 ms: the weight gradients leave the critical path.  Re-measured on
  // the round-6 kernels (profiles/r06_helper_kernels_ab.txt, item 6): now a small gain at every size (2048 conformers 45.45 -> 45.1-45.2 ms, 512: 13.74 -> 13.14),
  // but two streams sharing the chip make every per-kernel duration (HIP events and rocprofv3 alike) depend on what ran beside it, and the record's roofline
  // is a per-kernel figure: the default stays "on up to 16 k atoms"; NQ_SIDE_STREAM=0 / 1 forces it.
  const char* env = getenv("NQ_SIDE_STREAM");
  const bool want = env && (env[0] == '0' || env[0] == '1') ? env[0] == '1' : n_atoms <= 16384;
  if (!want) return;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  std::lock_guard<std::mutex> lock(g_side_mu);
  SidePool*& pool = g_side_pools[std::make_pair(dev, main)];
  if (!pool) {
    SidePool* p = new SidePool();
    if (hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking) != hipSuccess) { delete p; return; }
    pool = p;
  }
  s.side = pool->side; s.pool = &pool->events; s.on = true;
}

static int painn_backward_impl(const nq_painn_cfg* cfg, const float* params, const float* rbf_offsets, const nq_graph* graph, void* workspace,
                               size_t workspace_bytes, const float* grad_energy, const float* grad_forces, float* grad_params, void* stream,
                               bool seeded, const float* seed_x, const float* seed_vec, void* const* layer_events = nullptr) {
  WsLayout W; ParamLayout P;
  NQ_TRY(check_common(cfg, graph, workspace, workspace_bytes, &W, &P));
  if (!params || !grad_params || !rbf_offsets) return nq_fail(NQ_ERR_ARG, "null argument");
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  float* gp = grad_params;
  float* scr = ws + W.scratch;
  const NqGraphView g = view_of(graph);
  const int N = g.N, E = g.E, F = cfg->hidden_channels, R = cfg->num_rbf, H = F / 2, L = cfg->num_layers, T = cfg->num_elements;
  const size_t NF = (size_t)N * F;
  int* const rowctr0 = reinterpret_cast<int*>(ws + W.ROWCTR);
  NQ_HIP(hipMemsetAsync(rowctr0 + (size_t)2 * L * NQ_ROWCTR_INTS, 0, (size_t)3 * L * NQ_ROWCTR_INTS * sizeof(int), st));
  // claimed rows pay off once every wavefront gets at least a row or two (measured: 10.7 k atoms 12.16 vs 12.57 ms per step); with fewer rows than
  // wavefronts the claim atomics are a visible part of each kernel (1.3 k atoms: dual sweep 57 -> 112 us), so small batches keep the static striding
  auto row_ctr = [&](int kind, int l) { return N >= 4096 ? rowctr0 + ((size_t)kind * L + l) * NQ_ROWCTR_INTS : (int*)nullptr; };

  const size_t NH = (size_t)N * H;
  ReadoutArgs r{};
  r.N = N; r.H = H; r.ZO = ws + W.ZO; r.TZO = ws + W.ZO + NH; r.w2 = params + P.w2; r.o2 = params + P.o2;
  r.e_atom = ws + W.e_atom; r.te_atom = ws + W.te_atom;
  if (seeded) {
    // zero every tangent half (the forward sweep filled the primal halves only)
    auto zero = [&](size_t off, size_t n) { return hipMemsetAsync(ws + off, 0, n * sizeof(float), st); };
    for (int l = 0; l <= L; ++l) { NQ_HIP(zero(W.X[l] + NF, NF)); NQ_HIP(zero(W.V[l] + 3 * NF, 3 * NF)); }
    for (int l = 0; l < L; ++l) {
      const WsLayer& y = W.lay[l];
      NQ_HIP(zero(y.Z1 + NF, NF)); NQ_HIP(zero(y.Hh + NF, NF)); NQ_HIP(zero(y.XH + 3 * NF, 3 * NF)); NQ_HIP(zero(y.XM + NF, NF));
      NQ_HIP(zero(y.VM + 3 * NF, 3 * NF)); NQ_HIP(zero(y.UU + 6 * NF, 6 * NF)); NQ_HIP(zero(y.S + NF, NF)); NQ_HIP(zero(y.CAT + 2 * NF, 2 * NF));
      NQ_HIP(zero(y.ZQ + NF, NF)); NQ_HIP(zero(y.Q + NF, NF)); NQ_HIP(zero(y.Y + 3 * NF, 3 * NF));
    }
    NQ_HIP(zero(W.TD, E)); NQ_HIP(zero(W.TR, 3 * (size_t)E)); NQ_HIP(zero(W.ZO + (size_t)N * H, (size_t)N * H)); NQ_HIP(zero(W.te_atom, N));
  } else {
  // ---- tangent forward along pos_dot = -dL/dF --------------------------------------------------
  if (grad_forces) NQ_TRY(nq_negate(st, grad_forces, ws + W.pos_dot, 3L * N));
  else NQ_HIP(hipMemsetAsync(ws + W.pos_dot, 0, 3 * (size_t)N * sizeof(float), st));
  NQ_TRY(nq_geom_tan(st, g, graph->dst, ws + W.pos_dot, ws + W.TD, ws + W.TR));
  for (int l = 0; l < L; ++l) {
    const WsLayer& y = W.lay[l]; const MsgP& mp = P.msg[l]; const UpdP& up = P.upd[l];
    float* TZ1 = ws + y.Z1 + NF; float* TH = ws + y.Hh + NF; float* TXH = ws + y.XH + 3 * NF;
    // tangent pre-activation and tangent activation TH = TZ1 * silu'(Z1) in one pass
    NQ_TRY(nq_gemm_nt_dsilu(st, ws + W.X[l] + NF, params + mp.W1, TZ1, TH, ws + y.Z1, N, F, F, "W1"));
    NQ_TRY(nq_gemm_nt(st, TH, params + mp.W2, TXH, nullptr, nullptr, N, 3 * F, F, F, F, 3 * F, "W2"));
    MsgArgs m{};
    m.g = g; m.F = F; m.X = ws + W.X[l]; m.V = ws + W.V[l]; m.XH = ws + y.XH; m.PHI = ws + y.PHI; m.PSI = ws + y.PSI;
    m.TX = ws + W.X[l] + NF; m.TV = ws + W.V[l] + 3 * NF; m.TXH = TXH; m.TD = ws + W.TD; m.TR = ws + W.TR;
    m.TXM = ws + y.XM + NF; m.TVM = ws + y.VM + 3 * NF;
    if (W.fused) {
      FilterArgs fa;
      nq_make_filter_args(&fa, ws + y.WRT, params + mp.br, rbf_offsets, ws + W.RW, R, cfg->cutoff, cfg->envelope_exponent, cfg->rbf_coeff, cfg->filter_mode);
      fa.row_ctr = row_ctr(2, l);
      NQ_TRY(nq_msgf_fwd(st, m, fa, true));
    } else {
      NQ_TRY(nq_msg_fwd(st, m, true));
    }
    UpdArgs u{};
    u.N = N; u.F = F; u.XM = ws + y.XM; u.VM = ws + y.VM; u.U = ws + y.UU; u.Y = ws + y.Y; u.S = ws + y.S; u.CAT = ws + y.CAT;
    u.TXM = ws + y.XM + NF; u.TVM = ws + y.VM + 3 * NF; u.TU = ws + y.UU + 6 * NF; u.TY = ws + y.Y + 3 * NF;
    u.TS = ws + y.S + NF; u.TCAT = ws + y.CAT + 2 * NF; u.TX1 = ws + W.X[l + 1] + NF; u.TV1 = ws + W.V[l + 1] + 3 * NF;
    float* TZQ = ws + y.ZQ + NF; float* TQ = ws + y.Q + NF;
    // Tangent flavour of the fused update block: built, parity-tested (tests/test_engine_gpu.py), NOT the default -- it re-reads the primal intermediates in the
    // accumulator layout (4-byte loads) and measured 3.58 ms per step against 3.0 ms for the five launches below (profiles/r06_fused_update_ab.txt); NQ_FUSED_UPDATE_TAN=1 selects it.
    if (use_fused_update(cfg) && getenv("NQ_FUSED_UPDATE_TAN") && getenv("NQ_FUSED_UPDATE_TAN")[0] == '1') {   // the fragments are the forward call's (same weights: one step)
      NQ_TRY(nq_upd_fused(st, u, ws + y.UFRAG, nullptr, nullptr, ws + y.ZQ, ws + y.Q, TZQ, TQ, true));
      continue;
    }
    NQ_TRY(nq_gemm_nt(st, ws + y.VM + 3 * NF, params + up.U, ws + y.UU + 6 * NF, nullptr, nullptr, 3 * N, 2 * F, F, F, F, 2 * F, "U"));
    NQ_TRY(nq_upd_a(st, u, true));
    NQ_TRY(nq_gemm_nt_dsilu(st, ws + y.CAT + 2 * NF, params + up.V1, TZQ, TQ, ws + y.ZQ, N, F, 2 * F, "V1"));
    NQ_TRY(nq_gemm_nt(st, TQ, params + up.V2, ws + y.Y + 3 * NF, nullptr, nullptr, N, 3 * F, F, F, F, 3 * F, "V2"));
    NQ_TRY(nq_upd_b(st, u, true));
  }
  NQ_TRY(nq_gemm_nt(st, ws + W.X[L] + NF, params + P.O1, ws + W.ZO + NH, nullptr, nullptr, N, H, F, F, F, H, "O1"));
  r.N = N; r.H = H; r.ZO = ws + W.ZO; r.TZO = ws + W.ZO + NH; r.w2 = params + P.w2; r.o2 = params + P.o2;
  r.e_atom = ws + W.e_atom; r.te_atom = ws + W.te_atom;
  NQ_TRY(nq_readout(st, r, 1));

  }
  // ---- dual reverse: seeds (dL/dE_b, 1) on (E_b, Edot) -----------------------------------------
  if (grad_energy) NQ_TRY(nq_atom_seeds(st, grad_energy, g.atom_mol, N, ws + W.ge, ws + W.gte));
  else {
    NQ_HIP(hipMemsetAsync(ws + W.ge, 0, (size_t)N * sizeof(float), st));
    NQ_TRY(nq_atom_seeds(st, nullptr, g.atom_mol, N, ws + W.gte, nullptr));  // gte = 1
  }
  if (seeded) NQ_HIP(hipMemsetAsync(ws + W.gte, 0, (size_t)N * sizeof(float), st));   // no Edot term
  r.ge = ws + W.ge; r.gte = ws + W.gte; r.GZO = ws + W.GZO; r.GTZO = ws + W.GZO + NH; r.TMPW = ws + W.TMPW;
  NQ_TRY(nq_readout_rev(st, r, true));
  GwMode gw;
  if (!recall_molgw(workspace, &gw)) return nq_fail(NQ_ERR_ARG, "nq_painn_backward: no forward call has prepared this workspace");
  // lite: the tangent adjoints (every GT* operand below) are the force sweep's adjoints of this step, stored per layer by nq_painn_forward: they are read, not
  // recomputed -- every input-gradient product runs over the primal-adjoint rows only, the elementwise and message kernels skip their GT* stores.
  // (The force sweep's V2 / W2 products keep both forms of their result: the adjoint of the SiLU layer's output, for the silu'' term here, and of its pre-activation.)
  const bool lite = gw.lite && !seeded && W.fused;
  SideStream ss;
  if (gw.mode != GW_MIXED) side_stream_init(ss, st, N);   // mixed batches: the pair-row contraction of the large molecules and the per-molecule kernel add into one
                                                           // gradient and share the scratch with the split-K products -- everything stays on the main stream (a rare path)
  hipStream_t sd = ss.fork();          // sd == st when the side stream is off
  NQ_TRY(nq_colsum(sd, ws + W.TMPW, N, H, H, gp + P.w2, scr));
  NQ_TRY(nq_colsum(sd, ws + W.ge, N, 1, 1, gp + P.o2, scr));
  NQ_TRY(nq_gemm_tn(sd, ws + W.GZO, ws + W.X[L], gp + P.O1, 2L * N, H, F, H, F, scr, "O1", gp + P.o1, N));
  NQ_TRY(nq_gemm_nn(st, ws + W.GZO, params + P.O1, ws + W.GX, lite ? N : 2 * N, H, F, H, F, F, 0, "O1"));
  float* gv_cur = ws + W.GVa; float* gv_oth = ws + W.GVb;
  NQ_HIP(hipMemsetAsync(gv_cur, 0, (lite ? 3 : 6) * NF * sizeof(float), st));
  if (seeded) {   // adjoints of the final (x, vec) coming from the force head
    if (seed_x) NQ_TRY(nq_axpy(st, seed_x, ws + W.GX, (long)NF));
    if (seed_vec) NQ_HIP(hipMemcpyAsync(gv_cur, seed_vec, 3 * NF * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  float* gphi = ws + W.GPHI2; float* gpsi = gphi + (size_t)E * 3 * F;
  const bool molgw = gw.mode != GW_PAIR_ROWS, mixed = gw.mode == GW_MIXED;
  const bool tn_group = use_tn_group();
  const int* const sched = reinterpret_cast<const int*>(ws + W.SCHED);
  const int* const n_big_pairs = sched + nq_molgw_sched_ints(E, g.B) - 1;
  if (molgw) NQ_TRY(nq_molgw_geometry(st, g, ws + W.RW, ws + W.TD, ws + W.TR, sched, ws + W.GWREC));
  if (mixed) NQ_HIP(hipMemsetAsync(ws + W.GBR, 0, 3 * NF * sizeof(float), st));   // only the rows of the large molecules are written below; the column sum runs over all atoms
  for (int l = L - 1; l >= 0; --l) {
    const WsLayer& y = W.lay[l]; const MsgP& mp = P.msg[l]; const UpdP& up = P.upd[l];
    const PrePlanes pp = pre_planes(cfg, ws + y.WPRE);
    float* const GYs = lite ? ws + y.LGY : ws + W.GY;         // stacked [2][N][3F]: the second half is the force sweep's (lite) or written below
    float* const GCATs = lite ? ws + y.LGCAT : ws + W.GCAT;
    float* const GUs = lite ? ws + y.LGU : ws + W.GU;
    float* const GXHs = lite ? ws + y.LGXH : ws + W.GXH;
    float* const GQs = lite ? ws + y.LGQ : ws + W.GQ;         // stacked [2][N][F] adjoints of zq (lite: second half = the force sweep's)
    float* const GHs = lite ? ws + y.LGH : ws + W.GH;
    UpdRevArgs u{};
    u.N = N; u.F = F; u.U = ws + y.UU; u.Y = ws + y.Y; u.S = ws + y.S; u.CAT = ws + y.CAT;
    u.TU = ws + y.UU + 6 * NF; u.TY = ws + y.Y + 3 * NF; u.TS = ws + y.S + NF; u.TCAT = ws + y.CAT + 2 * NF;
    u.GX = ws + W.GX; u.GTX = lite ? ws + y.LGXA : ws + W.GX + NF; u.GV = gv_cur; u.GTV = lite ? ws + y.LGVA : gv_cur + 3 * NF;
    u.GY = GYs; u.GTY = GYs + 3 * NF; u.GCAT = GCATs; u.GTCAT = GCATs + 2 * NF;
    u.GU = GUs; u.GTU = GUs + 6 * NF;
    u.lite = lite ? 1 : 0;
    ss.before_main_writes(SB_GY);
    NQ_TRY(nq_upd_rev(st, u, 1, true));
    if (!tn_group) {
    sd = ss.fork();
    NQ_TRY(nq_gemm_tn(sd, GYs, ws + y.Q, gp + up.V2, 2L * N, 3 * F, F, 3 * F, F, scr, "V2", gp + up.c2, N));
    ss.read_by_side(SB_GY);
    }
    ss.before_main_writes(SB_GQ);
    NQ_TRY(nq_gemm_nn(st, GYs, params + up.V2, GQs, lite ? N : 2 * N, 3 * F, F, 3 * F, F, F, 0, "V2", pp.V2));
    NQ_TRY(nq_silu_rev(st, ws + y.ZQ, ws + y.ZQ + NF, GQs, lite ? ws + y.LGQP : GQs + NF, (long)NF, true, lite));
    if (!tn_group) {
    sd = ss.fork();
    NQ_TRY(nq_gemm_tn(sd, GQs, ws + y.CAT, gp + up.V1, 2L * N, F, 2 * F, F, 2 * F, scr, "V1", gp + up.c1, N));
    ss.read_by_side(SB_GQ);
    }
    NQ_TRY(nq_gemm_nn(st, GQs, params + up.V1, GCATs, lite ? N : 2 * N, F, 2 * F, F, 2 * F, 2 * F, 0, "V1", pp.V1));
    ss.before_main_writes(SB_GU);
    NQ_TRY(nq_upd_rev(st, u, 2, true));
    if (!tn_group) {
    sd = ss.fork();
    NQ_TRY(nq_gemm_tn(sd, GUs, ws + y.VM, gp + up.U, 6L * N, 2 * F, F, 2 * F, F, scr, "U"));
    ss.read_by_side(SB_GU);
    }
    NQ_TRY(nq_gemm_nn(st, GUs, params + up.U, gv_cur, lite ? 3 * N : 6 * N, 2 * F, F, 2 * F, F, F, 1, "U", pp.U));
    MsgRevArgs m{};
    m.g = g; m.F = F; m.V = ws + W.V[l]; m.XH = ws + y.XH; m.PHI = ws + y.PHI; m.PSI = ws + y.PSI;
    m.TV = ws + W.V[l] + 3 * NF; m.TXH = ws + y.XH + 3 * NF; m.TD = ws + W.TD; m.TR = ws + W.TR;
    m.GX = ws + W.GX; m.GV = gv_cur; m.GTX = lite ? ws + y.LGXB : ws + W.GX + NF; m.GTV = lite ? ws + y.LGVB : gv_cur + 3 * NF;
    m.GXH = GXHs; m.GTXH = GXHs + 3 * NF; m.GV_out = gv_oth; m.GTV_out = gv_oth + 3 * NF;
    m.lite = lite ? 1 : 0;
    m.GPHI = gphi; m.GPSI = gpsi; m.GBR = ws + W.GBR;
    ss.before_main_writes(SB_GXH); ss.before_main_writes(SB_GPHI); ss.before_main_writes(SB_GBR);
    if (W.fused) {
      FilterArgs fa;
      nq_make_filter_args(&fa, ws + y.WRT, params + mp.br, rbf_offsets, ws + W.RW, R, cfg->cutoff, cfg->envelope_exponent, cfg->rbf_coeff, cfg->filter_mode);
      fa.row_ctr = row_ctr(3, l);
      m.mol_cap = gw.cap;
      m.row_filter = mixed ? 1 : 0;
      NQ_TRY(nq_msgf_rev(st, m, fa, true, !molgw));
      if (mixed) {   // the molecules that do not fit the LDS of k_gwr_mol: pair rows, k0-sorted contraction and per-atom bias sums as in rounds 1-4, for THEIR rows only
        m.row_filter = 2;
        fa.row_ctr = row_ctr(4, l);
        NQ_TRY(nq_msgf_rev(st, m, fa, true, true));
        NQ_TRY(nq_gwr_sorted(st, gphi, gpsi, ws + W.RW, reinterpret_cast<const int*>(ws + W.ORDER), E / 2, F, R, gp + mp.Wr, scr, 3, n_big_pairs));
        NQ_TRY(nq_colsum(st, ws + W.GBR, N, 3 * F, 3 * F, gp + mp.br, scr));
      }
      // rbf_proj weight and bias gradient from the same node rows, staged per molecule in LDS (main stream: it reads the adjoints this layer's input-gradient
      // products overwrite next; the fork below orders the side stream and the layer event behind it); mixed: added to what the pair-row kernels left
      if (molgw)
        NQ_TRY(nq_gwr_mol(st, g, F, R, gw.cap, m.XH, m.V, m.TXH, m.TV, m.GX, m.GV, m.GTX, m.GTV, sched, ws + W.GWREC, ws + W.GWPART, gp + mp.Wr, gp + mp.br, mixed));
    } else {
      NQ_TRY(nq_msg_rev(st, m, true));
    }
    { float* t = gv_cur; gv_cur = gv_oth; gv_oth = t; }
    sd = ss.fork();
    if (W.fused && !molgw) NQ_TRY(nq_gwr_sorted(sd, gphi, gpsi, ws + W.RW, reinterpret_cast<const int*>(ws + W.ORDER), E / 2, F, R, gp + mp.Wr, scr));   // one row per pair
    else if (!W.fused) NQ_TRY(nq_gemm_tn(sd, gphi, ws + W.RHO2, gp + mp.Wr, 2L * E, 3 * F, R, 3 * F, R, scr, "Wr"));
    if (cfg->rbf_type)   // adjoints of rho / drho (shared by all layers): [gphi; gpsi] Wr, accumulated over the layers
      NQ_TRY(nq_gemm_nn(st, gphi, params + mp.Wr, ws + W.GRHO, 2 * E, 3 * F, R, 3 * F, R, R, l == L - 1 ? 0 : 1, "Wr"));
    if (!molgw) NQ_TRY(nq_colsum(sd, ws + W.GBR, N, 3 * F, 3 * F, gp + mp.br, scr));
    if (!tn_group) NQ_TRY(nq_gemm_tn(sd, GXHs, ws + y.Hh, gp + mp.W2, 2L * N, 3 * F, F, 3 * F, F, scr, "W2", gp + mp.b2, N));
    ss.read_by_side(SB_GPHI); ss.read_by_side(SB_GBR); if (!tn_group) ss.read_by_side(SB_GXH);
    ss.before_main_writes(SB_GH);
    NQ_TRY(nq_gemm_nn(st, GXHs, params + mp.W2, GHs, lite ? N : 2 * N, 3 * F, F, 3 * F, F, F, 0, "W2", pp.W2));
    NQ_TRY(nq_silu_rev(st, ws + y.Z1, ws + y.Z1 + NF, GHs, lite ? ws + y.LGHP : GHs + NF, (long)NF, true, lite));
    sd = ss.fork();
    if (!tn_group) {
      NQ_TRY(nq_gemm_tn(sd, GHs, ws + W.X[l], gp + mp.W1, 2L * N, F, F, F, F, scr, "W1", gp + mp.b1, N));
      ss.read_by_side(SB_GH);
    } else {
      // all five weight-gradient products of the layer (and their bias gradients) in one launch: gy, gq, gu, gxh, gh are final and stay untouched until the
      // next layer's kernels overwrite them (each of those waits for this launch through its before_main_writes)
      NqTnSpec sp[5];
      tn_group_shapes(sp, N, F);
      sp[0].G = GYs; sp[0].X = ws + y.Q; sp[0].out = gp + up.V2; sp[0].bias_out = gp + up.c2;
      sp[1].G = GQs; sp[1].X = ws + y.CAT; sp[1].out = gp + up.V1; sp[1].bias_out = gp + up.c1;
      sp[2].G = GUs; sp[2].X = ws + y.VM; sp[2].out = gp + up.U;
      sp[3].G = GXHs; sp[3].X = ws + y.Hh; sp[3].out = gp + mp.W2; sp[3].bias_out = gp + mp.b2;
      sp[4].G = GHs; sp[4].X = ws + W.X[l]; sp[4].out = gp + mp.W1; sp[4].bias_out = gp + mp.b1;
      if (nq_gemm_tn_group(sd, sp, 5, scr) != NQ_OK) {   // not eligible for the split engine (exact-f32 engine selected, unaligned operands): one launch each
        NQ_TRY(nq_gemm_tn(sd, sp[0].G, sp[0].X, sp[0].out, 2L * N, 3 * F, F, 3 * F, F, scr, "V2", sp[0].bias_out, N));
        NQ_TRY(nq_gemm_tn(sd, sp[1].G, sp[1].X, sp[1].out, 2L * N, F, 2 * F, F, 2 * F, scr, "V1", sp[1].bias_out, N));
        NQ_TRY(nq_gemm_tn(sd, sp[2].G, sp[2].X, sp[2].out, 6L * N, 2 * F, F, 2 * F, F, scr, "U"));
        NQ_TRY(nq_gemm_tn(sd, sp[3].G, sp[3].X, sp[3].out, 2L * N, 3 * F, F, 3 * F, F, scr, "W2", sp[3].bias_out, N));
        NQ_TRY(nq_gemm_tn(sd, sp[4].G, sp[4].X, sp[4].out, 2L * N, F, F, F, F, scr, "W1", sp[4].bias_out, N));
      }
      ss.read_by_side(SB_GY); ss.read_by_side(SB_GQ); ss.read_by_side(SB_GU); ss.read_by_side(SB_GXH); ss.read_by_side(SB_GH);
    }
    NQ_TRY(nq_gemm_nn(st, GHs, params + mp.W1, ws + W.GX, lite ? N : 2 * N, F, F, F, F, F, 1, "W1", pp.W1));
    // every gradient slice of layer l (and, for l = L-1, of the read-out head) is final once the weight-gradient stream gets here: the caller's
    // collective stream may start reducing it
    if (layer_events && layer_events[L - 1 - l]) NQ_HIP(hipEventRecord((hipEvent_t)layer_events[L - 1 - l], ss.on ? ss.side : st));
  }
  ss.join();
  NQ_TRY(nq_embed_grad(st, g.z, ws + W.GX, N, F, T, gp + P.emb, scr));
  if (cfg->rbf_type) {   // dL/d(frequencies) [R] or dL/d(pregamma) [1]
    NQ_TRY(nq_rbf_param_grad(st, g.geom, E, R, cfg->cutoff, cfg->envelope_exponent, rbf_offsets, cfg->rbf_type, params + P.basis, ws + W.GRHO, ws + W.BCON));
    if (cfg->rbf_type == 1) NQ_TRY(nq_colsum(st, ws + W.BCON, E, R, R, gp + P.basis, scr));
    else NQ_TRY(nq_colsum(st, ws + W.BCON, (long)E * R, 1, 1, gp + P.basis, scr));
  }
  return NQ_OK;
}

int nq_painn_backward(const nq_painn_cfg* cfg, const float* params, const float* rbf_offsets, const nq_graph* graph, void* workspace,
                      size_t workspace_bytes, const float* grad_energy, const float* grad_forces, float* grad_params, void* stream) {
  return painn_backward_impl(cfg, params, rbf_offsets, graph, workspace, workspace_bytes, grad_energy, grad_forces, grad_params, stream, false, nullptr,
                             nullptr);
}
int nq_painn_backward_events(const nq_painn_cfg* cfg, const float* params, const float* rbf_offsets, const nq_graph* graph, void* workspace,
                             size_t workspace_bytes, const float* grad_energy, const float* grad_forces, float* grad_params, void* const* layer_events_host,
                             void* stream) {
  return painn_backward_impl(cfg, params, rbf_offsets, graph, workspace, workspace_bytes, grad_energy, grad_forces, grad_params, stream, false, nullptr,
                             nullptr, layer_events_host);
}
int nq_painn_layer_param_ranges(const nq_painn_cfg* cfg, int64_t* ranges_host) {
  ParamLayout P;
  if (!cfg || !ranges_host) return nq_fail(NQ_ERR_ARG, "null argument");
  NQ_TRY(make_param_layout(cfg, &P));
  const int L = cfg->num_layers;
  for (int l = 0; l < L; ++l) {
    const size_t m0 = P.msg[l].W1, m1 = l + 1 < L ? P.msg[l + 1].W1 : P.upd[0].U;
    const size_t u0 = P.upd[l].U, u1 = l + 1 < L ? P.upd[l + 1].U : P.O1;
    ranges_host[4 * l + 0] = (int64_t)m0; ranges_host[4 * l + 1] = (int64_t)(m1 - m0);
    ranges_host[4 * l + 2] = (int64_t)u0; ranges_host[4 * l + 3] = (int64_t)(u1 - u0);
  }
  ranges_host[4 * L + 0] = (int64_t)P.O1; ranges_host[4 * L + 1] = (int64_t)(P.total - P.O1);       // read-out head: final with layer L-1
  ranges_host[4 * L + 2] = 0; ranges_host[4 * L + 3] = (int64_t)P.msg[0].W1;                          // embedding (+ basis parameters): final at the end
  return NQ_OK;
}
int nq_painn_backward_seeded(const nq_painn_cfg* cfg, const float* params, const float* rbf_offsets, const nq_graph* graph, void* workspace,
                             size_t workspace_bytes, const float* grad_energy, const float* grad_x, const float* grad_vec, float* grad_params,
                             void* stream) {
  return painn_backward_impl(cfg, params, rbf_offsets, graph, workspace, workspace_bytes, grad_energy, nullptr, grad_params, stream, true, grad_x,
                             grad_vec);
}

// ------------------------------------------------------------------------------------------------
int nq_loss_l1_l2(const float* energy, const float* y, int32_t B, const float* forces, const float* f_target, int32_t N, float coef_e,
                  float coef_f, float* loss, float* grad_energy, float* grad_forces, void* stream) {
  if (!energy || !y || !forces || !f_target || !loss || !grad_energy || !grad_forces) return nq_fail(NQ_ERR_ARG, "null argument");
  return nq_loss_impl((hipStream_t)stream, energy, y, B, forces, f_target, N, coef_e, coef_f, loss, grad_energy, grad_forces, false);
}
int nq_loss_mse(const float* energy, const float* y, int32_t B, const float* forces, const float* f_target, int32_t N, float coef_e,
                  float coef_f, float* loss, float* grad_energy, float* grad_forces, void* stream) {
  if (!energy || !y || !forces || !f_target || !loss || !grad_energy || !grad_forces) return nq_fail(NQ_ERR_ARG, "null argument");
  return nq_loss_impl((hipStream_t)stream, energy, y, B, forces, f_target, N, coef_e, coef_f, loss, grad_energy, grad_forces, true);
}

int nq_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t count, float max_norm, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int32_t step, float* scratch, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !scratch || step < 1) return nq_fail(NQ_ERR_ARG, "bad argument");
  return nq_adamw_impl((hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, (long)count, max_norm, lr, beta1, beta2, eps, weight_decay,
                       step, scratch);
}

int nq_linear_forward(const float* A, const float* Wt, const float* bias, float* C, float* C_silu, int32_t M, int32_t N, int32_t K, void* stream) {
  return nq_gemm_nt((hipStream_t)stream, A, Wt, C, bias, C_silu, M, N, K, K, K, N);
}
int nq_linear_forward_act(const float* A, const float* Wt, float* C, float* C_act, const float* resid, float alpha, float beta, int32_t M, int32_t N, int32_t K,
                          void* stream) {
  if (!A || !Wt || !C || !C_act) return nq_fail(NQ_ERR_ARG, "null argument");
  return nq_gemm_nt_act((hipStream_t)stream, A, Wt, C, C_act, resid, alpha, beta, M, N, K);
}
int nq_linear_forward_res(const float* A, const float* Wt, const float* aux, float alpha, float* C, int32_t M, int32_t N, int32_t K, void* stream) {
  if (!A || !Wt || !C || !aux) return nq_fail(NQ_ERR_ARG, "null argument");
  if (aux == C) return nq_fail(NQ_ERR_ARG, "nq_linear_forward_res: aux must not alias C");
  return nq_gemm_nt_res((hipStream_t)stream, A, Wt, C, aux, alpha, M, N, K);
}
int nq_linear_input_grad(const float* G, const float* Wt, float* C, int32_t M, int32_t N, int32_t K, int32_t accumulate, void* stream) {
  return nq_gemm_nn((hipStream_t)stream, G, Wt, C, M, N, K, N, K, K, accumulate);
}
int nq_linear_input_grad_epi(const float* G, const float* Wt, float* C, int32_t M, int32_t N, int32_t K, const float* aux, float alpha, float beta, int32_t mode,
                             void* stream) {
  if (!G || !Wt || !C || !aux || (mode != 1 && mode != 2)) return nq_fail(NQ_ERR_ARG, "bad argument");
  return nq_gemm_nn_epi((hipStream_t)stream, G, Wt, C, M, N, K, aux, alpha, beta, mode);
}
size_t nq_column_sum_scratch_floats(int64_t rows, int32_t cols) { return nq_colsum_scratch_floats((long)rows, cols); }
int nq_column_sum(const float* A, int64_t rows, int32_t cols, int64_t lda, float* out, float* scratch, void* stream) {
  if (!A || !out || !scratch || cols < 1 || lda < cols) return nq_fail(NQ_ERR_ARG, "column_sum: bad argument");
  return nq_colsum((hipStream_t)stream, A, (long)rows, cols, (int)lda, out, scratch);
}
size_t nq_weight_grad_scratch_floats(int64_t rows, int32_t N, int32_t K) { return nq_gemm_tn_scratch_floats(rows, N, K); }
int nq_linear_weight_grad(const float* G, const float* X, float* gW, int64_t rows, int32_t N, int32_t K, float* scratch, void* stream) {
  return nq_gemm_tn((hipStream_t)stream, G, X, gW, rows, N, K, N, K, scratch);
}
int nq_linear_weight_grad_bias(const float* G, const float* X, float* gW, float* gb, int64_t rows, int32_t N, int32_t K, float* scratch, void* stream) {
  if (!gb) return nq_fail(NQ_ERR_ARG, "null bias gradient");
  return nq_gemm_tn((hipStream_t)stream, G, X, gW, rows, N, K, N, K, scratch, nullptr, gb, rows);
}

}  // extern "C"
