/* nablaq.h -- C ABI of libnablaq.so, the MI355X (gfx950) engine for the nablaDFT PaiNN hot path.
 *
 * Drop-in boundary.  The reference is pure Python; its "FFI" for this path is the set
 * of third-party native ops it calls from nablaDFT/painn_pyg (paths relative to /root/reference/):
 *
 *   nq_graph_count / nq_graph_fill   replace torch_cluster.radius_graph (painn.py:411-416), the edge
 *                                    geometry (painn.py:418-423, :319-321), compute_neighbors
 *                                    (utils.py:469-481) and symmetrize_edges (painn.py:168-304)
 *   nq_painn_forward                 replaces PaiNN.forward (painn.py:89-148): RadialBasis
 *                                    (layers.py:181-185), AtomEmbedding (layers.py:215-222), 6x
 *                                    PaiNNMessage (painn.py:475-509; PyG propagate + torch_scatter.scatter)
 *                                    and PaiNNUpdate (painn.py:535-548), out_energy + scatter
 *                                    (painn.py:127-128) and forces = -autograd.grad (painn.py:135-146)
 *   nq_painn_backward                replaces loss.backward() through the create_graph=True force graph
 *                                    (painn.py:142; Lightning backward after painn.py:655-668)
 *   nq_loss_l1_l2                    replaces _calculate_loss (painn.py:741-745) with L1Loss + L2Loss
 *                                    (gemnet_oc/loss.py:5-22; config/model/painn-oc.yaml:36-43)
 *   nq_adamw_step                    replaces clip_grad_norm (config/painn-oc.yaml:18-19) + torch.optim.AdamW
 *
 * Conventions: every pointer is a DEVICE pointer owned by the caller (torch tensors) unless it is
 * named *_host; all arrays are contiguous row-major fp32 / int32 / int64 as stated; `stream` is a
 * hipStream_t passed as void*; functions are re-entrant (no global mutable state except the
 * thread-local error string); return value 0 = NQ_OK, otherwise an NQ_ERR_* code and
 * nq_last_error() describes it.  Nothing here allocates device memory.
 */
#ifndef NABLAQ_H
#define NABLAQ_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NQ_OK 0
#define NQ_ERR_HIP 1
#define NQ_ERR_ARG 2
#define NQ_ERR_MOL_TOO_LARGE 3
#define NQ_ERR_WORKSPACE 4
#define NQ_ERR_NO_EDGES 5

#define NQ_ABI_VERSION 16

/* Model hyper-parameters = constructor arguments of nablaDFT.painn_pyg.PaiNN (painn.py:28-45). */
typedef struct nq_painn_cfg {
  int32_t hidden_channels;   /* F, multiple of 64, <= 1024 */
  int32_t num_layers;        /* L */
  int32_t num_rbf;           /* R */
  int32_t num_elements;      /* rows of atom_emb.embeddings.weight */
  int32_t max_neighbors;     /* K of radius_graph */
  int32_t envelope_exponent; /* PolynomialEnvelope(exponent) (layers.py:14-33); 0 selects ExponentialEnvelope (layers.py:36-48) */
  double cutoff;             /* Angstrom */
  float rbf_coeff;           /* GaussianSmearing.coeff = -0.5/(offset[1]-offset[0])^2 */
  int32_t filter_mode;       /* 0: painn_pyg  filter = rbf_proj(envelope(d/rc) * gauss(d/rc)) + bias   (layers.py:181-185)
                              * 1: schnetpack filter = cosine_cutoff(d) * (filter_net(gauss(d)) + bias), Gaussians on the unscaled
                              *    distance (config/model/painn.yaml:9-16); needs the fused-filter path */
  int32_t rbf_type;          /* RadialBasis(rbf=...) (layers.py:168-179): 0 gaussian; 1 spherical_bessel (learnable frequencies [R]);
                              * 2 bernstein (learnable pregamma).  1 and 2 use the materialised-filter path; their parameters sit in the
                              * flat buffer right after atom_emb.embeddings.weight (state_dict order); for 2 `rbf_offsets` carries the
                              * BernsteinBasis.prefactor buffer (binomial coefficients). */
  int32_t reserved;
} nq_painn_cfg;

/* Neighbour list in engine layout (CSR by target atom, sources ascending; symmetric). */
typedef struct nq_graph {
  int32_t N, B, E;
  int32_t max_mol_atoms;    /* largest molecule of the batch (the value passed to nq_graph_count); 0 = unknown: the per-molecule LDS kernels
                             * (rbf_proj gradient with node rows staged per molecule, csrc/molpair.hip) are then replaced by the row-gather path */
  const int32_t* mol_ptr;   /* [B+1] first atom of each molecule */
  const int32_t* row_ptr;   /* [N+1] */
  const int32_t* col;       /* [E] source atom of the in-edge at this slot */
  const int32_t* dst;       /* [E] target atom of the slot (row index) */
  const int32_t* rev;       /* [E] slot of the reverse edge */
  const float* geom;        /* [E][4] {rx, ry, rz, d}, r = (pos[col]-pos[dst])/d */
  const int32_t* z;         /* [N] atomic numbers */
  const int32_t* atom_mol;  /* [N] molecule of each atom */
  const int32_t* lowptr;    /* [N+1] prefix count of in-edges with source < target (the first entries of each row); output of
                             * nq_graph_count.  Numbers the undirected pairs: pair(i <- j, j < i) = lowptr[i] + position in row i. */
} nq_graph;

int nq_abi_version(void);
const char* nq_last_error(void);

/* ---- neighbour list ------------------------------------------------------------------------ */
/* Pass 1: degrees and prefix sums; returns the number of directed edges E in *E_host (synchronises
 * `stream`).  deg/lowdeg: int32[N] scratch; row_ptr/lowptr: int32[N+1] outputs. */
int nq_graph_count(const float* pos, const int32_t* mol_ptr, int32_t N, int32_t B, int32_t max_mol_atoms, double cutoff,
                   int32_t max_neighbors, int32_t* deg, int32_t* lowdeg, int32_t* row_ptr, int32_t* lowptr, int32_t* E_host,
                   void* stream);
/* Pass 2: fills the CSR arrays (col, dst, rev, geom[E][4], slot2canon, atom_mol[N]) and, if edge_index != NULL,
 * the reference's canonical outputs: edge_index int64[2][E] (row0 = source j, row1 = target i),
 * edge_dist f32[E], edge_vector f32[E][3], id_swap int64[E], neighbors int64[B]. */
int nq_graph_fill(const float* pos, const int32_t* mol_ptr, int32_t N, int32_t B, int32_t E, int32_t max_mol_atoms, double cutoff,
                  int32_t max_neighbors, const int32_t* row_ptr, const int32_t* lowptr, int32_t* col, int32_t* dst, int32_t* rev,
                  float* geom, int32_t* slot2canon, int32_t* atom_mol, int64_t* edge_index, float* edge_dist, float* edge_vector,
                  int64_t* id_swap, int64_t* neighbors, void* stream);

/* ---- model --------------------------------------------------------------------------------- */
/* Flat parameter buffer: tensors concatenated in state_dict order of the reference PaiNN
 * (atom_emb.embeddings.weight, message_layers.{l}.{x_proj.0,x_proj.2,rbf_proj}.{weight,bias} for all l,
 * update_layers.{l}.{vec_proj.weight, xvec_proj.0.{weight,bias}, xvec_proj.2.{weight,bias}} for all l,
 * out_energy.{0,2}.{weight,bias}). */
size_t nq_painn_num_params(const nq_painn_cfg* cfg);
size_t nq_painn_workspace_bytes(const nq_painn_cfg* cfg, int32_t N, int32_t E, int32_t B);

/* energy[B] and (if forces != NULL) forces[N][3] = -dE_tot/dpos.  Keeps every activation in `workspace`
 * for nq_painn_backward -- and, when forces are computed, the per-layer adjoints of that force sweep: they are the
 * tangent adjoints of the second-order sweep, which nq_painn_backward reads instead of recomputing (same parameters
 * required in both calls, as before).  rbf_offsets: f32[R] = buffer radial_basis.rbf.offset. */
int nq_painn_forward(const nq_painn_cfg* cfg, const float* params, const float* rbf_offsets, const nq_graph* graph, void* workspace,
                     size_t workspace_bytes, float* energy, float* forces, void* stream);
/* Given dL/dE[B] and dL/dF[N][3] (either may be NULL = zeros) writes dL/dparams[num_params] (overwrites).
 * Must follow nq_painn_forward (with forces) on the same workspace and graph. */
int nq_painn_backward(const nq_painn_cfg* cfg, const float* params, const float* rbf_offsets, const nq_graph* graph, void* workspace,
                      size_t workspace_bytes, const float* grad_energy, const float* grad_forces, float* grad_params, void* stream);
/* nq_painn_backward with completion events: layer_events_host[i] (HOST array of L hipEvent_t, entries may be NULL) is recorded on `stream` as soon as
 * every gradient slice of layer L-1-i is final (the reverse sweep runs from the last layer to the first; the read-out head's slices are final with
 * event 0, the embedding's when the call's work is complete), so that a data-parallel caller can all-reduce finished slices on a side stream while the
 * earlier layers are still being differentiated (reference: torch DDP's bucketed all-reduce under Lightning's DDPStrategy, utils/pipelines.py:65-68).
 * nq_painn_layer_param_ranges: ranges_host int64[4 (L+1)] = per layer {message offset, count, update offset, count} in floats of the flat buffer,
 * then {head offset, count, embedding offset, count}. */
int nq_painn_backward_events(const nq_painn_cfg* cfg, const float* params, const float* rbf_offsets, const nq_graph* graph, void* workspace,
                             size_t workspace_bytes, const float* grad_energy, const float* grad_forces, float* grad_params, void* const* layer_events_host,
                             void* stream);
int nq_painn_layer_param_ranges(const nq_painn_cfg* cfg, int64_t* ranges_host);
/* First-order reverse for the direct-force model (direct_forces=True, painn.py:130-133; the PaiNNOutput head itself, painn.py:551-620, is
 * evaluated by the caller on the final node state "x_in"/"vec_in" at layer L of the workspace): given dL/dE[B] and the adjoints of the
 * final x [N][F] and vec [N][3][F] (any may be NULL) writes dL/dparams.  Must follow nq_painn_forward (forces == NULL) on the same workspace. */
int nq_painn_backward_seeded(const nq_painn_cfg* cfg, const float* params, const float* rbf_offsets, const nq_graph* graph, void* workspace,
                             size_t workspace_bytes, const float* grad_energy, const float* grad_x, const float* grad_vec, float* grad_params,
                             void* stream);
/* Elementwise pieces of GatedEquivariantBlock (painn.py:583-620), the GEMMs in between are nq_linear_*:
 * nq_scaled_silu: out = silu(z)/0.6 (grad_y == NULL) or grad_y * dsilu(z)/0.6;  nq_geb_cat: cat[N][2h] = [x | ||v1||_xyz], v1 [N][3][h];
 * nq_geb_gate: (xo | gate) = split(o2 [N][2o]); xout = ScaledSiLU(xo), vout [N][3][o] = gate * v2. */
int nq_scaled_silu(const float* z, const float* grad_y, float* out, int64_t count, void* stream);
int nq_geb_cat(const float* x, const float* v1, int64_t N, int32_t h, float* cat, void* stream);
int nq_geb_cat_backward(const float* grad_cat, const float* v1, int64_t N, int32_t h, float* grad_x, float* grad_v1, void* stream);
int nq_geb_gate(const float* o2, const float* v2, int64_t N, int32_t o, float* xout, float* vout, void* stream);
int nq_geb_gate_backward(const float* o2, const float* v2, const float* grad_xout, const float* grad_vout, int64_t N, int32_t o, float* grad_o2,
                         float* grad_v2, void* stream);
/* Largest molecule (atoms) whose rbf_proj gradient runs on the per-molecule LDS kernel (csrc/molpair.hip): its 20 rows of a 32-channel slice must fit one
 * workgroup's LDS.  Larger molecules of a batch take the pair-row kernels, the rest of the batch is not affected (painn.py:475-509 semantics either way). */
int32_t nq_painn_molecule_lds_atoms(void);
/* Test/inspection hook: offset (in floats) and element count of a named workspace buffer, e.g.
 * ("x_msg", 2, tangent=0).  Returns NQ_ERR_ARG for unknown names. */
int nq_painn_ws_lookup(const nq_painn_cfg* cfg, int32_t N, int32_t E, int32_t B, const char* name, int32_t layer, int32_t tangent,
                       size_t* offset_floats, size_t* count);

/* ---- SchNet (schnetpack 2.0.4 representation.SchNet + Atomwise + Forces as instantiated by config/model/schnet.yaml:4-28;
 *      third-party arithmetic, PARITY UNPINNED -- oracle/spk_schnet_ref.py) -------------------------------------------------- */
typedef struct nq_schnet_cfg {
  int32_t n_atom_basis;    /* F in {64,128,256} (n_filters = n_atom_basis) */
  int32_t n_interactions;  /* L */
  int32_t n_rbf;           /* GaussianRBF(n_rbf, cutoff) */
  int32_t max_z;           /* rows of representation.embedding.weight (row 0 = padding) */
  double cutoff;           /* CosineCutoff(cutoff) = GaussianRBF cutoff, Angstrom */
  float rbf_coeff;         /* -0.5 / width^2, width = offsets[1] - offsets[0] */
  int32_t reserved;
} nq_schnet_cfg;
/* Flat parameter buffer = the tensors in module order: representation.embedding.weight [max_z][F]; per interaction l:
 * in2f.weight [F][F], filter_network.{0.weight [F][R], 0.bias, 1.weight [F][F], 1.bias}, f2out.{0.weight, 0.bias, 1.weight, 1.bias};
 * output_modules.0.outnet.{0.weight [F/2][F], 0.bias, 1.weight [1][F/2], 1.bias}. */
size_t nq_schnet_num_params(const nq_schnet_cfg* cfg);
size_t nq_schnet_workspace_bytes(const nq_schnet_cfg* cfg, int32_t N, int32_t E, int32_t B);
/* Same contracts as nq_painn_forward / nq_painn_backward; rbf_offsets = GaussianRBF.offsets f32[n_rbf] (unscaled distances);
 * the graph is the full list inside the cutoff (nq_graph_count/fill with max_neighbors >= molecule size). */
int nq_schnet_forward(const nq_schnet_cfg* cfg, const float* params, const float* rbf_offsets, const nq_graph* graph, void* workspace,
                      size_t workspace_bytes, float* energy, float* forces, void* stream);
int nq_schnet_backward(const nq_schnet_cfg* cfg, const float* params, const float* rbf_offsets, const nq_graph* graph, void* workspace,
                       size_t workspace_bytes, const float* grad_energy, const float* grad_forces, float* grad_params, void* stream);

/* ---- Hamiltonian block assembly (QHNet.build_final_matrix, qhnet/qhnet.py:293-321; + H + H^T, qhnet.py:237; HamiltonianLoss,
 *      qhnet/loss.py:9-16).  Packed result layout: molecule after molecule, M_b x M_b row-major, M_b = orbitals of molecule b;
 *      pack_ptr[b] = sum_{b' < b} M_b'^2, mol_orb_ptr[b] = sum_{b' < b} M_b'.  All index arrays are device pointers. ------------- */
/* Tables for one batch: orb_atom / orb_slot [sum M_b] (global orbital -> atom, slot of the padded S x S block = mask[Z][t], the table
 * of QHNet._get_mask, qhnet.py:323-342), look [sum_b n_b^2] (ordered atom pair -> index into the pair-block array; e_dst = row 0 and
 * e_src = row 1 of data.full_edge_index, qhnet.py:296).  orb_ptr [N+1]: prefix of mask_count[z]; pair_base [B]: prefix of n_b^2.
 * err_flag (device int32) becomes non-zero if a pair joins two molecules (1) or, later, a needed pair is missing (2). */
int nq_hblock_tables(const int32_t* z, const int32_t* atom_mol, const int32_t* mol_ptr, int32_t N, int32_t B, const int64_t* orb_ptr,
                     const int64_t* pair_base, const int64_t* e_dst, const int64_t* e_src, int64_t P, const int32_t* mask_table,
                     const int32_t* mask_count, int32_t S, int32_t* orb_atom, int32_t* orb_slot, int32_t* look, int64_t look_count,
                     int32_t* err_flag, void* stream);
/* out_packed[total]: block (dst rows, src columns) = diag[atom] or nondiag[pair(dst, src)] on the atoms' orbital slots; symmetrize != 0
 * adds the transposed element (H + H^T). */
int nq_hblock_assemble(const float* diag, const float* nondiag, const int32_t* mol_ptr, const int64_t* pair_base, const int64_t* pack_ptr,
                       const int64_t* mol_orb_ptr, const int32_t* orb_atom, const int32_t* orb_slot, const int32_t* look, int32_t B, int32_t S,
                       int32_t symmetrize, int64_t total, float* out_packed, int32_t* err_flag, void* stream);
/* Reverse of nq_hblock_assemble: grad_diag [N][S][S], grad_nondiag [P][S][S] (unused slots get 0).  inv_table [Zt][S]: slot -> local
 * orbital of that atom type or -1. */
int nq_hblock_assemble_backward(const float* grad_packed, const int32_t* z, const int32_t* atom_mol, const int64_t* orb_ptr, const int64_t* pack_ptr,
                                const int64_t* mol_orb_ptr, const int32_t* inv_table, const int64_t* e_dst, const int64_t* e_src, int32_t N,
                                int64_t P, int32_t S, int32_t symmetrize, float* grad_diag, float* grad_nondiag, void* stream);
/* to_dense != 0: dense [m_total][m_total] = block_diag of the packed blocks (zero elsewhere; what the reference returns);
 * to_dense == 0: packed <- the diagonal blocks of dense. */
int nq_hblock_packed_dense(float* packed, float* dense, const int64_t* pack_ptr, const int64_t* mol_orb_ptr, int32_t B, int64_t total, int64_t m_total,
                           int32_t to_dense, void* stream);
/* PhiSNet irreps -> matrix (NeuralNetwork.matrix_block / generate_matrix_from_irreps + the collection loops of forward,
 * phisnet/nn/neural_network.py:636-706, 859-918): packed result as above; block(i,j)[(n_i,m_i),(n_j,m_j)] =
 * sum_L sum_M cg_table[l_i][l_j][L][m_i][m_j][M] * f[row][L*L+M][idx(type_i, type_j, n_i, n_j, L)], f = f_ii[atom] or f_ij[pair(i, j)]
 * ([rows][ncomp][Fo]); cg_table [3][3][5][5][5][9] = sqrt(2L+1) * the model's Clebsch-Gordan tensors; shell tables [T][32]; idx_ii
 * [T][S][S][5], idx_ij [T][T][S][S][5] (-1 = absent).  orb_local: local orbital index inside its atom (nq_hblock_tables with identity masks).
 * unit_diagonal: overlap matrices (diagonal = 1, neural_network.py:964-965).  Backward: inv_ii [T][5][Fo], inv_ij [T][T][5][Fo] = n_i*S+n_j or -1. */
int nq_irreps_assemble(const float* f_ii, const float* f_ij, const int32_t* z, const int32_t* mol_ptr, const int64_t* pair_base, const int64_t* pack_ptr,
                       const int64_t* mol_orb_ptr, const int64_t* orb_ptr, const int32_t* orb_atom, const int32_t* orb_local, const int32_t* look,
                       int32_t B, int32_t ncomp, int32_t Fo, const int32_t* tz, const int32_t* sh_n, const int32_t* sh_l, const int32_t* sh_m,
                       const int32_t* sh_off, const int32_t* idx_ii, const int32_t* idx_ij, const float* cg_table, int32_t T, int32_t S,
                       int32_t symmetrize, int32_t unit_diagonal, int64_t total, float* out_packed, int32_t* err_flag, void* stream);
int nq_irreps_assemble_backward(const float* grad_packed, const int32_t* z, const int32_t* atom_mol, const int64_t* e_i, const int64_t* e_j,
                                const int64_t* pack_ptr, const int64_t* mol_orb_ptr, const int64_t* orb_ptr, const int32_t* inv_ii, const int32_t* inv_ij,
                                int64_t N, int64_t P, int32_t ncomp, int32_t Fo, const int32_t* tz, const int32_t* sh_n, const int32_t* sh_l,
                                const int32_t* sh_m, const int32_t* sh_off, const float* cg_table, int32_t T, int32_t S, int32_t symmetrize,
                                int32_t unit_diagonal, float* grad_f_ii, float* grad_f_ij, void* stream);
/* stats3 = {loss = sqrt(sum d^2 / total) + sum|d| / total, rmse, sum|d|} (mask.sum() == total for block-diagonal targets);
 * grad_packed (nullable) = grad_scale * dloss/dpred.  scratch: 512 doubles. */
int nq_hamiltonian_loss(const float* pred_packed, const float* target_packed, int64_t total, float grad_scale, float* stats3, float* grad_packed,
                        double* scratch, void* stream);

/* ---- SO(3) Clebsch-Gordan mixing (PhiSNet PairMixing / SelfMixing contraction: phisnet/nn/modules/pair_mixing.py:47-69,
 *      self_mixing.py:55-83).  x1 [rows][(order1+1)^2][F], x2 [rows][(order2+1)^2][F], y [rows][(order_out+1)^2][F], orders <= 4,
 *      components of all orders concatenated (offset l*l + m + l).  path_index_host: HOST int8[65], for every path (l1,l2,L) in the
 *      loop order of PairMixing.forward over orders 4/4/4 the index of its coefficient among the enabled paths, or -1.
 *      coeff [rows][n_enabled][F] (coeff_row_stride = n_enabled*F) or [n_enabled][F] shared by all rows (coeff_row_stride = 0).
 *      keep (nullable) [keep_orders][F]: y_L += keep_L * x1_L.  The kernels use the canonical tensors of nabladft_amd/cg.py; a model's
 *      own sign convention is folded into coeff by the caller. ------------------------------------------------------------------ */
int nq_so3_mix_forward(const float* x1, const float* x2, const float* coeff, const float* keep, int64_t rows, int32_t F, int32_t order1,
                       int32_t order2, int32_t order_out, const int8_t* path_index_host, int64_t coeff_row_stride, int32_t keep_orders, float* y,
                       void* stream);
/* grad_coeff_rows [rows][n_enabled][F] and grad_keep_rows [rows][keep_orders][F] are per-row contributions (sum over rows for shared
 * coefficients); grad_x1 / grad_x2 have the shapes of x1 / x2 (for x1 == x2 add them). */
int nq_so3_mix_backward(const float* x1, const float* x2, const float* coeff, const float* keep, const float* grad_y, int64_t rows, int32_t F,
                        int32_t order1, int32_t order2, int32_t order_out, const int8_t* path_index_host, int64_t coeff_row_stride,
                        int32_t keep_orders, float* grad_x1, float* grad_x2, float* grad_coeff_rows, float* grad_keep_rows, void* stream);

/* The same reverse pass for coefficients SHARED by all rows (SelfMixing; QHNet's SelfNetLayer): dL/dcoeff is reduced over rows inside the kernel;
 * grad_coeff_partials [nq_so3_mix_partial_blocks(rows, F)][n_enabled][F] are per-workgroup partial sums (sum them in order for the result).
 * F must divide 256. */
int64_t nq_so3_mix_partial_blocks(int64_t rows, int32_t F);
int nq_so3_mix_backward_shared(const float* x1, const float* x2, const float* coeff, const float* keep, const float* grad_y, int64_t rows, int32_t F,
                               int32_t order1, int32_t order2, int32_t order_out, const int8_t* path_index_host, int32_t keep_orders, float* grad_x1,
                               float* grad_x2, float* grad_coeff_partials, float* grad_keep_rows, void* stream);

/* ---- QHNet SO(3) tensor-product layers (nablaDFT/qhnet/layers.py; e3nn 0.5.1 TensorProduct / Linear / Norm semantics restated in
 *      oracle/e3nn_mini.py, PARITY UNPINNED for the e3nn part).  Irreps features are [rows][(lmax+1)^2][C], component l*l + m + l, channel
 *      fastest.  Graph arrays (int32, device): row_ptr [N+1] CSR by owner atom = "src" = row 1 of the reference's edge_index (qhnet.py:262),
 *      col [R] = "dst" = row 0 ascending, own [R] = owner of each slot, rev [R] = slot of the reverse edge (the neighbour relation is
 *      symmetric).  Path weights carry e3nn's sign and path normalisation (folded in by the caller). ------------------------------------------------------------------------------------- */
/* s0 [R][(2+lmax) C] = [x_0[dst] | x_0[dst] (conv, layers.py:240-246) or x_0[src] (pair, layers.py:469-475) | <x_l[dst], x_l[src]>/(2l+1)]
 * = InnerProduct (layers.py:277-294) + the concatenations of ConvLayer.forward / PairNetLayer.forward. */
int nq_qh_invariants_forward(const float* x, int64_t N, int32_t ncomp, int32_t C, const int32_t* own, const int32_t* col, int64_t R,
                             int32_t second_from_owner, float* s0, void* stream);
int nq_qh_invariants_backward(const float* x, const float* grad_s0, int64_t N, int32_t ncomp, int32_t C, const int32_t* row_ptr, const int32_t* col,
                              const int32_t* rev, int32_t second_from_owner, float* grad_x, void* stream);
/* Tensor products per row (edge or ordered pair), the arithmetic of e3nn's TensorProduct in QHNet:
 *   sh != NULL ('uvu', ConvLayer.tp_node, layers.py:262): y[r] = sum_paths w1 w2 CG . x[idx1[r]] . sh[r];  sh [R][25] real spherical harmonics of
 *       pos[dst] - pos[src];  path_set 1 = the 42 paths with even l1+l2+L (x [N][25][C]), 2 = the 5 paths of the first layer (x [N][1][C]);
 *   sh == NULL ('uuu', PairNetLayer.tp_node_pair, layers.py:481): second operand x[idx2[r]]; path_set 0 = all 65 paths.
 * w1, w2 (nullable second factor, multiplied in-kernel): [R][nq_qh_tp_num_paths(path_set)][C] in e3nn instruction order.  y_rows [R][25][C].
 * Backward: grad_y rows are read at idx_gy[r] (NULL = r); per-row operand adjoints grad_x1_rows [R][ncomp1][C], grad_x2_rows [R][25][C]
 * (uuu only) are summed over each atom's rows by nq_qh_pair_reduce; grad_w1 / grad_w2 like w1 / w2. */
/* 'uuu' forward with the two weight factors generated inside the kernel (csrc/qhgen.hip; layers.py:476-481: weight = fc_node_pair(edge_attr) * fc(s0)):
 *   w1[r] = h1[r] W1 (* col_scale), w2[r] = h2[r] W2^T + bias2, h1 / h2 [R][K] the hidden activations of the two generators (K = 32 / 64 / 128).
 * nq_qh_gen_presplit turns a generator's last weight matrix (layout 0: [K][65 C] as x @ W; 1: [65 C][K] as nn.Linear) into nq_qh_gen_fragment_floats(C, K)
 * floats of bf16 matrix-instruction fragments, once per optimiser step.  The weight factors never exist in HBM; nq_qh_tp_backward_gen is the reverse sweep with the same
 * in-kernel generation (it returns the adjoints of both factors, [R][65][C] each: the operands of the generators' own weight gradients). */
size_t nq_qh_gen_fragment_floats(int32_t C, int32_t K);
int nq_qh_gen_presplit(const float* W, const float* col_scale, int32_t K, int32_t C, int32_t layout, float* fragments, void* stream);
int nq_qh_tp_forward_gen(const float* x, const int32_t* idx1, const int32_t* idx2, const float* h1, const float* h2, const float* frag1, const float* frag2,
                         const float* bias2, int64_t R, int32_t C, int32_t K, float* y_rows, void* stream);
int nq_qh_tp_backward_gen(const float* x, const int32_t* idx1, const int32_t* idx2, const float* h1, const float* h2, const float* frag1, const float* frag2,
                          const float* bias2, const float* grad_y, int64_t R, int32_t C, int32_t K, float* grad_x1_rows, float* grad_x2_rows, float* grad_w1,
                          float* grad_w2, void* stream);
int nq_qh_tp_num_paths(int32_t path_set);
void nq_qh_set_tp_variant(int32_t variant);   /* tuning hook (process-global): 0 per-path weight loads, 1 / 2 chunked prefetch with / without scheduling barriers */
int nq_qh_tp_forward(const float* x, int32_t ncomp1, const int32_t* idx1, const float* sh, const int32_t* idx2, const float* w1, const float* w2, int64_t R,
                     int32_t C, int32_t path_set, float* y_rows, void* stream);
int nq_qh_tp_backward(const float* x, int32_t ncomp1, const int32_t* idx1, const float* sh, const int32_t* idx2, const float* w1, const float* w2,
                      const float* grad_y, const int32_t* idx_gy, int64_t R, int32_t C, int32_t path_set, float* grad_x1_rows, float* grad_x2_rows,
                      float* grad_w1, float* grad_w2, void* stream);
/* out[n][k] = base[n][k] + sum_{r in row n} (rows_own[r][k] + rows_nbr[rev[r]][k]), k < width (each operand may be NULL = zeros): the scatter of
 * ConvLayer (torch_scatter.scatter, layers.py:268: messages arrive along the reverse slots of the receiver's own row) and the reverse of every
 * per-row gather; fixed order, no atomics. */
int nq_qh_pair_reduce(const float* rows_own, const float* rows_nbr, const float* base, const int32_t* row_ptr, const int32_t* rev, int64_t N, int32_t width,
                      float* out, void* stream);
/* NormGate pieces (layers.py:141-147; e3nn o3.Norm + ElementwiseTensorProduct): nq_qh_normcat: out [rows][(lmax+1) C] = [x_0 | ||x_1|| | ...]
 * (grad_f0 == NULL) or the adjoint of x [rows][ncomp][C] given grad_f0;  nq_qh_gate: y = [gates_0 | x_l * gates_l] (grad_y == NULL) or
 * (grad_x, grad_gates) given grad_y. */
int nq_qh_normcat(const float* x, const float* grad_f0, int64_t rows, int32_t C, int32_t lmax, float* out, void* stream);
int nq_qh_gate(const float* x, const float* gates, const float* grad_y, int64_t rows, int32_t C, int32_t lmax, float* y, float* grad_x, float* grad_gates,
               void* stream);
/* out = cst * act(x) (grad_y == NULL) or grad_y * cst * act'(x); kind 0 SiLU, 1 shifted softplus (layers.py:21-22). */
int nq_qh_act(const float* x, const float* grad_y, int32_t kind, float cst, int64_t count, float* out, void* stream);
/* Expansion.forward (layers.py:598-662): x [R][25][Cb], weights [R][n_weights], bias [R][n_bias] (nullable) -> out [R][S][S];
 * shells_host: HOST int32[3] = number of s, p, d shells of the padded block (S = s + 3p + 5d); w3j: device [19][5][5][9] real Wigner 3j
 * tensors w3j(l1, l2, l_in)[i][j][k] in the instruction order of get_expansion_path (layers.py:664-671), zero padded. */
int nq_qh_expansion_forward(const float* x, const float* weights, const float* bias, int64_t R, int32_t Cb, const int32_t* shells_host, int32_t n_weights,
                            int32_t n_bias, const float* w3j, float* out, void* stream);
int nq_qh_expansion_backward(const float* x, const float* weights, const float* grad_out, int64_t R, int32_t Cb, const int32_t* shells_host,
                             int32_t n_weights, int32_t n_bias, const float* w3j, float* grad_x, float* grad_weights, float* grad_bias, void* stream);

/* ---- geometry bases of the Hamiltonian models --------------------------------------------------------------------------------- */
/* out [P][(order+1)^2]: real spherical harmonics Y_0..Y_order (order <= 4) of unit vectors [P][3], PhiSNet convention
 * (phisnet/nn/spherical_harmonics/spherical_harmonics.py:10-25: no 1/sqrt(4 pi), Condon-Shortley, m = -l..l). */
int nq_sph_harm(const float* unit_vectors, int64_t P, int32_t order, float* out, void* stream);
/* Adjoint of nq_sph_harm w.r.t. the vectors: gu[P][3] = sum_c grad_out[P][c] dY_c/d(x,y,z), the closed forms differentiated as polynomials of a free vector
 * (what autograd does with phisnet/nn/spherical_harmonics/spherical_harmonics_any_order.py before the chain rule through u = r/|r|): the angular half of
 * PhiSNet's forces = -dE/dR (phisnet/nn/neural_network.py:737, :981-984). */
int nq_sph_harm_backward(const float* unit_vectors, const float* grad_out, int64_t P, int32_t order, float* gu, void* stream);
/* out [P][K] = cutoff_function(r) * exp(logc_k + n_k x + v_k log(1 - e^x)), x = -alpha r  (ExponentialBernsteinRadialBasisFunctions.forward,
 * phisnet/nn/modules/exponential_bernstein_radial_basis_functions.py:36-41 == qhnet/layers.py:115-120); alpha = softplus(_alpha). */
int nq_bernstein_rbf(const float* r, int64_t P, int32_t K, float alpha, float cutoff, const float* logc, const float* n, const float* v, float* out,
                     void* stream);
int nq_bernstein_rbf_grad_alpha(const float* r, const float* grad_out, int64_t P, int32_t K, float alpha, float cutoff, const float* logc, const float* n,
                                const float* v, float* galpha_rows, void* stream);

/* gr[p] = sum_k grad_out[p][k] d rbf[p][k] / d r (cutoff function included): the radial half of PhiSNet's forces; alpha from a device scalar. */
int nq_bernstein_rbf_grad_r_dev(const float* r, const float* grad_out, int64_t P, int32_t K, const float* alpha_dev, float cutoff, const float* logc,
                                const float* n, const float* v, float* gr, void* stream);

/* The same two calls with alpha = softplus(_alpha) read from a DEVICE scalar: no host read of the learnable parameter, so a whole training step can be
 * captured into a HIP graph (trainer.GraphedStep). */
int nq_bernstein_rbf_dev(const float* r, int64_t P, int32_t K, const float* alpha_dev, float cutoff, const float* logc, const float* n, const float* v, float* out,
                         void* stream);
int nq_bernstein_rbf_grad_alpha_dev(const float* r, const float* grad_out, int64_t P, int32_t K, const float* alpha_dev, float cutoff, const float* logc,
                                    const float* n, const float* v, float* galpha_rows, void* stream);

/* The other radial bases of PhiSNet (neural_network.py:210-221), all times cutoff_function(r): kind 1 gaussian (t0 = centres, width), 2 exp-gaussian
 * (t0 = centres, width, alpha), 3 overlap-bernstein (t0 = logc, t1 = n, t2 = v, alpha), 4 bernstein (t0 = logc, t1 = n, t2 = v).  out [P][K];
 * nq_radial_basis_grad_alpha (kinds 2, 3): per-row dL/dalpha given grad_out [P][K]. */
int nq_radial_basis(int32_t kind, const float* r, int64_t P, int32_t K, float alpha, float cutoff, float width, const float* t0, const float* t1, const float* t2,
                    float* out, void* stream);
int nq_radial_basis_grad_alpha(int32_t kind, const float* r, const float* grad_out, int64_t P, int32_t K, float alpha, float cutoff, float width, const float* t0,
                               const float* t1, const float* t2, float* galpha_rows, void* stream);

/* Learnable feature-wise activations of PhiSNet on [rows][F]: kind 0 = Swish (swish.py:23-24), kind 1 = ShiftedSoftplus
 * (shifted_softplus.py:26-32).  Backward writes grad_x and per-element partials of dL/dalpha, dL/dbeta ([rows][F], sum over rows). */
int nq_feature_act(const float* x, const float* alpha, const float* beta, int64_t rows, int32_t F, int32_t kind, float* y, void* stream);
int nq_feature_act_backward(const float* x, const float* alpha, const float* beta, const float* grad_y, int64_t rows, int32_t F, int32_t kind,
                            float* grad_x, float* grad_alpha_rows, float* grad_beta_rows, void* stream);

/* The same activation on the scalar component of a packed irreps tensor [rows][ncomp][F] (PhiSNet activates xs[0] only, residual_block.py:58-64);
 * the other components are copied.  Backward partials of alpha / beta are [rows][F]. */
int nq_packed_act0(const float* x, const float* alpha, const float* beta, int64_t rows, int32_t ncomp, int32_t F, int32_t kind, float* y, void* stream);
int nq_packed_act0_backward(const float* x, const float* alpha, const float* beta, const float* grad_y, int64_t rows, int32_t ncomp, int32_t F, int32_t kind,
                            float* grad_x, float* grad_alpha_rows, float* grad_beta_rows, void* stream);

/* PhiSNet SphericalLinear (phisnet/nn/modules/spherical_linear.py:50-59; also EquiformerV2's SO3_LinearV2,
 * equiformer_v2/so3.py:587-625, order <= 6) on packed irreps tensors x, y: [rows][(order+1)^2][F]: one Linear per order,
 * y_L = x_L W_L^T (+ bias0 on the scalars), W: HOST array of order+1 device pointers to [Fout][Fin] matrices.  Forward and input gradient are one launch
 * for all orders; the weight gradient is one split-K contraction per order (scratch: nq_sph_weight_grad_scratch_floats floats; fixed summation order). */
int nq_sph_linear_forward(const float* x, const float* const* W_host, const float* bias0, float* y, int64_t rows, int32_t order, int32_t Fin, int32_t Fout,
                          void* stream);
int nq_sph_linear_input_grad(const float* grad_y, const float* const* W_host, float* grad_x, int64_t rows, int32_t order, int32_t Fin, int32_t Fout,
                             void* stream);
size_t nq_sph_weight_grad_scratch_floats(int64_t rows, int32_t order, int32_t Fin, int32_t Fout);
int nq_sph_linear_weight_grad(const float* grad_y, const float* x, float* const* grad_W_host, float* grad_bias0, int64_t rows, int32_t order, int32_t Fin,
                              int32_t Fout, float* scratch, void* stream);

/* Pair <-> atom data movement of the interaction blocks (interaction_block.py:135-142).  Rows of C floats.
 * nq_gather_rows: out[p] = x[idx[p]].  nq_segment_sum: out[n] = base[n] (nullable) + sum_{q in [seg_ptr[n], seg_ptr[n+1])} rows[order ? order[q] : q]
 * -- fixed summation order, no atomics (the reference's index_add on a GPU is not reproducible). */
int nq_gather_rows(const float* x, const int64_t* idx, int64_t P, int32_t C, float* out, void* stream);
int nq_segment_sum(const float* rows, const int64_t* order, const int64_t* seg_ptr, const float* base, int64_t N, int32_t C, float* out, void* stream);

/* ---- GemNet-OC (reference nablaDFT/gemnet_oc/) ------------------------------------------------------------------ */
/* One edge set stored as a CSR by target atom (sources ascending): ptr [atoms+1], src / dst [n], geom [n][4] = {unit vector source -> target, distance}. */
typedef struct nq_gn_set {
  int32_t n, reserved;
  const int32_t* ptr;
  const int32_t* src;
  const int32_t* dst;
  const float* geom;
} nq_gn_set;
/* All graphs of gemnet_oc.py:892-958 (get_graphs_and_indices) derived from the a2a graph (nq_graph_count / nq_graph_fill with cutoff_aint, CSR arrays
 * row_ptr / col / rev / dst; geom [E][4] is REWRITTEN with the reference's CPU arithmetic: d = torch.norm(pos[j] - pos[i]), unit vector = difference / d):
 * sub-graph selection by cutoff and the K nearest per target (utils.py:408-500, enforce_max_strictly), the symmetrised main
 * graph (gemnet_oc.py:694-775: source < target kept, flips added) with the counter-edge slot m_rev (= id_swap), the a2ee2a and qint graphs; triplet and
 * quadruplet lists (interaction_indices.py) are not materialised.  The caller owns every array: [N] counters, [N+1] prefix arrays, [E] flags / mpos / apos /
 * a_of_rev, and -- sized from the four totals nq_gn_graph_count returns -- the per-graph arrays. */
typedef struct nq_gn_graphs {
  int32_t N, E, k_main, k_aea, k_qint, reserved;
  double cutoff_main, cutoff_aea, cutoff_qint;
  const int32_t* row_ptr;
  const int32_t* col;
  const int32_t* rev;
  const int32_t* dst;
  const float* pos;
  float* geom;
  uint8_t* flags;
  int32_t *degm, *lowm, *cnt_a, *cnt_q, *tin_atom;
  int32_t *ptr_m, *lowptr_m, *ptr_a, *ptr_q, *tin_aptr;
  int32_t *m_src, *m_dst, *m_rev, *m_slot;
  float* m_geom;
  int32_t *a_src, *a_dst;
  float* a_geom;
  int32_t* a_of_rev;
  int32_t *q_src, *q_dst;
  float* q_geom;
  int32_t* tin_ptr;
  int32_t *mpos, *apos;
  int32_t *tin_main, *q_of_rev, *qpos;
} nq_gn_graphs;
/* totals_host[4] = {main edges, a2ee2a edges, qint edges, rows of the (qint edge, main in-edge of its source) table}; synchronises `stream`.
 * max_degree: largest a2a in-degree (<= 512). */
int nq_gn_graph_count(const void* graphs /* nq_gn_graphs* */, int32_t max_degree, int32_t* totals_host, void* stream);
int nq_gn_graph_fill(const void* graphs /* nq_gn_graphs* */, void* stream);
/* RadialBasis (layers/radial_basis.py:196-220): out[n][R] = scale * envelope_p(d/cutoff) * exp(coeff (d/cutoff - offset_k)^2); d = geom[.][3]. */
int nq_gn_radial_basis(const float* geom, int64_t n, int32_t num_radial, const float* offset, double cutoff, double exponent, float scale, float* out,
                       void* stream);
/* EfficientInteractionBilinear's first contraction (layers/efficient.py:213-229) for the triplet families (interaction_indices.py:13-118):
 * S[o][s][c] = sum over in-edges p of target(o) in `in_set` with source(p) != source(o) of Y_s0(clamp(v_o . v_p)) * scale * x[p][c];  backward: d x. */
int nq_gn_triplet_forward(const void* out_set, const void* in_set, const float* x, int32_t C, int32_t NS, float scale, float* S, void* stream);
int nq_gn_triplet_backward(const void* out_set, const void* in_set, const float* dS, int32_t C, int32_t NS, float scale, float* dx, void* stream);
/* Quadruplets (interaction_indices.py:121-282, angles gemnet_oc.py:597-655, "legendre_outer" basis spherical_basis.py:104-110): x rows are indexed
 * tin_ptr[q] + j (j-th main in-edge of source(q)); S[o][l * NS + l'][c].  backward scratch: f32[main edges * KQ * NS * C], KQ >= largest qint in-degree. */
int nq_gn_quad_forward(const void* main_set, const void* qint_set, const int32_t* tin_ptr, int32_t n_atoms, const float* x, int32_t C, int32_t NS, float scale,
                       float* S, void* stream);
int nq_gn_quad_backward(const void* main_set, const void* qint_set, const int32_t* tin_ptr, int32_t n_atoms, int64_t T, const float* dS, int32_t C, int32_t NS,
                        int32_t KQ, float scale, float* scratch, float* dx, void* stream);
/* Tuning hook (process-global): 1 = one workgroup per centre atom with LDS-shared bases (default), 0 = one thread per (edge, channel). */
void nq_gn_set_quad_variant(int32_t variant);
/* BasisEmbedding without inner index (layers/efficient.py:136-141): cir[(q, j)][i] = sum_s rad_w1[q][i * NS + s] Y_s0(v_q . v_p) * scale. */
int nq_gn_cir_forward(const void* main_set, const void* qint_set, const int32_t* tin_ptr, int64_t T, const float* rad_w1, int32_t I, int32_t NS, float scale,
                      float* cir, void* stream);
int nq_gn_cir_backward(const void* main_set, const void* qint_set, const int32_t* tin_ptr, const float* dcir, int32_t I, int32_t NS, float scale,
                       float* d_rad_w1, void* stream);
/* rad_W1 @ sph_m (layers/efficient.py:231-244): out[o][i][c] = sum_s R[o][i * NSS + s] S[o][s][c]; backward writes dR and / or dS (nullable). */
int nq_gn_rowmm_forward(const float* R, const float* S, int64_t n, int32_t I, int32_t NSS, int32_t C, float* out, void* stream);
int nq_gn_rowmm_backward(const float* R, const float* S, const float* dout, int64_t n, int32_t I, int32_t NSS, int32_t C, float* dR, float* dS, void* stream);
/* PairInteraction (layers/interaction_block.py:721-733): out[a][r][c] = sum_{p in row(a)} rad_w[p][r] x[source(p)][c]; the a2a graph is symmetric (rev). */
int nq_gn_pair_forward(const void* a2a_set, const float* rad_w, const float* x, int32_t N, int32_t Rr, int32_t C, float* out, void* stream);
int nq_gn_pair_backward(const void* a2a_set, const int32_t* rev, const float* rad_w, const float* x, const float* dout, int32_t N, int32_t Rr, int32_t C,
                        float* d_rad_w, float* dx, void* stream);
/* Adjoint of the row gather x_tin[row] = x[tin_main[row]] of the quadruplet path (interaction_block.py:579: x_db[idx["triplet_in"]["in"]]):
 * out[p] = sum over the qint edges q whose source is target(p) of g[tin_ptr[q] + position of p in its row]; a2a_row_ptr / q_of_rev from nq_gn_graphs. */
int nq_gn_tin_scatter(const void* main_set, const int32_t* a2a_row_ptr, const int32_t* q_of_rev, const int32_t* tin_ptr, const float* g, int32_t C,
                      float* out, void* stream);
/* EdgeEmbedding input (layers/embedding_block.py:85-90): cat[e] = [h[source] | h[target] | m[e]]; backward_h: dh from the first two blocks of dcat. */
int nq_gn_cat_forward(const void* main_set, const float* h, const float* m, int32_t A, int32_t Em, float* cat, void* stream);
int nq_gn_cat_backward_h(const void* main_set, const int32_t* rev, const float* dcat, int32_t N, int32_t A, int32_t Em, float* dh, void* stream);
/* AtomUpdateBlock / OutputBlock head (layers/atom_update_block.py:88-93): out[a][c] = sum_{p in row(a)} m[p][c] r[p][c]. */
int nq_gn_mulsum_forward(const void* main_set, const float* m, const float* r, int32_t N, int32_t C, float* out, void* stream);
int nq_gn_mulsum_backward(const void* main_set, const float* m, const float* r, const float* dout, int32_t C, float* dm, float* dr, void* stream);
/* Direct forces (gemnet_oc.py:1216-1243): per-edge scalar, averaged with the counter-edge if coupled, times the edge vector, summed per target atom. */
int nq_gn_forces_forward(const void* main_set, const int32_t* rev, const float* f_edge, int32_t N, int32_t coupled, float* forces, void* stream);
int nq_gn_forces_backward(const void* main_set, const int32_t* rev, const float* d_forces, int32_t coupled, float* d_f_edge, void* stream);
/* out[p] = x[idx[p]] (* y[p] if y); out[n] = sum_{q in [ptr[n], ptr[n+1])} rows[r] (* y[r] if y), r = order ? order[q] : q, entries r < 0 skipped;
 * out = a * b;  out = alpha a + beta b (b nullable);  dW[t] = sum_{n: z[n] == t + 1} g[n] (Embedding(z - 1), layers/embedding_block.py:39-53). */
int nq_gn_gather(const float* x, const int32_t* idx, const float* y, int64_t P, int32_t C, float* out, void* stream);
int nq_gn_segment_sum(const float* rows, const float* y, const int32_t* order, const int32_t* ptr, int64_t N, int32_t C, float* out, void* stream);
int nq_gn_mul(const float* a, const float* b, int64_t n, float* out, void* stream);
int nq_gn_lincomb(const float* a, const float* b, float alpha, float beta, int64_t n, float* out, void* stream);
/* out = scale * g * d/dz [silu(z) / 0.6] (adjoint of ScaledSiLU with a folded constant). */
int nq_gn_ssilu_backward(const float* z, const float* g, float scale, int64_t n, float* out, void* stream);
int nq_gn_embed_grad(const int32_t* z, const float* g, int32_t N, int32_t num_elements, int32_t C, float* dW, void* stream);

/* ---- eSCN building blocks (reference nablaDFT/escn/escn.py, so3.py) --------------------------------------------------- */
/* Directed radius graph of escn.py:253-255 (radius_graph(pos, r, batch, max_num_neighbors)): per target atom the first K atoms of its molecule (index
 * order) with d^2 < r^2.  Pass 1 (count): deg [N], ptr [N+1], *E_host (synchronises).  Pass 2 (fill): src, dst [E], geom [E][4] = {pos[j] - pos[i], |.|}. */
int nq_es_graph_count(const float* pos, const int32_t* mol_ptr, const int32_t* atom_mol, int32_t N, double cutoff, int32_t K, int32_t* deg, int32_t* ptr,
                      int32_t* E_host, void* stream);
int nq_es_graph_fill(const float* pos, const int32_t* mol_ptr, const int32_t* atom_mol, int32_t N, double cutoff, int32_t K, const int32_t* ptr, int32_t* src,
                     int32_t* dst, float* geom, void* stream);
/* Edge rotation matrices rot [E][3][3] (escn.py:435-487; deterministic helper axis instead of the reference's random vector: same model output). */
int nq_es_frames(const float* geom, int32_t E, float* rot, void* stream);
/* Wigner-D rows (so3.py:377-425): W [E][n_red][n_full], row b = row red_row[b] of the degree-red_l[b] block D^l = Z(alpha) J_l Z(beta) J_l Z(gamma);
 * J: the J_l matrices back to back (J_offset[l]); scratch: 8-byte aligned, 3 E doubles (the Euler angles and all trigonometry are evaluated in float64:
 * the rows are then exact to float32 rounding, the reference's float32 evaluation carries several 1e-6 near the polar axis). */
int nq_es_wigner(const float* rot, int32_t E, const float* J, const int32_t* J_offset, const int32_t* red_l, const int32_t* red_row, int32_t n_red, int32_t n_full,
                 int32_t lmax, float* scratch, float* W, void* stream);
/* GaussianSmearing (smearing.py:14-31): out[e][k] = exp(coeff (geom[e][3] - offset[k])^2). */
int nq_es_smearing(const float* geom, int64_t E, int32_t K, const float* offset, float coeff, float* out, void* stream);
/* One small matrix per row times the row's [coefficients][channels] block (SO3_Embedding._rotate / _rotate_inv / to_grid / from_grid, so3.py:265-375):
 * transpose == 0: out[o][i][c] (+)= sum_s R_o[i][s] X_r[s][c];  else out[o][s][c] (+)= sum_i R_o[i][s] X_r[i][c];  R_o = R + o * r_stride (0: shared),
 * r = index ? index[o] : o; strides in floats; R is I x NSS. */
int nq_rowop(const float* R, int64_t r_stride, const float* X, int64_t x_stride, const int32_t* index, float* out, int64_t out_stride, int64_t n, int32_t I,
             int32_t NSS, int32_t C, int32_t transpose, int32_t accumulate, void* stream);

/* nq_rowop with one side living in per-block tensors (the m-blocks an SO(2) layer consumes / produces): seg_side 0 = the I side (rows of R), 1 = the S side
 * (columns); seg_rows[nseg] (host) rows per block, seg_ptrs[nseg] (host array of device pointers) the contiguous [n][rows_k][C] tensors; x_or_out / stride: the
 * other side (transpose == 0: S side in, I side out; else I side in, S side out).  nseg <= 8. */
int nq_rowop_blocks(const float* R, int64_t r_stride, float* x_or_out, int64_t stride, const int32_t* index, int32_t seg_side, int32_t nseg,
                    const int32_t* seg_rows, float* const* seg_ptrs, int64_t n, int32_t I, int32_t NSS, int32_t C, int32_t transpose, int32_t accumulate,
                    void* stream);
/* Fused S2 activation of the SO(2) message blocks (escn/so3.py:301-318 to_grid / from_grid around the point-wise SiLU; equiformer_v2/activation.py:155-176):
 * y_blocks = from_grid(SiLU(to_grid(x_blocks))) in ONE kernel on the matrix cores, the [n][G][C] grid tensor never reaches memory; backward != 0: dx_blocks from
 * (x_blocks, dy_blocks) with the grid recomputed.  T, F: [G][S] to_grid / from_grid matrices; blocks: nseg contiguous tensors [n][seg_rows[k]][C] (HOST arrays of
 * device pointers), S = sum seg_rows.  Requires C % 64 == 0, S <= 32, G <= 96 (NQ_ERR_ARG otherwise). */
int nq_s2_activation_blocks(const float* T, const float* F, int32_t G, int32_t S, int32_t C, int64_t n, int32_t nseg, const int32_t* seg_rows, float* const* x_ptrs,
                            float* const* gy_ptrs, float* const* out_ptrs, int32_t backward, void* stream);

/* Rotations that use the degree-block structure of the Wigner rows (235 of 29 x 49 entries at lmax 6 / mmax 2), no LDS.  nq_es_rotate (SO3_Embedding._rotate):
 * block(i)[o][row][c_off + c] = sum_k W_o[i][l_i^2 + k] x[index ? index[o] : o][l_i^2 + k][c]; red_l [n_red] (HOST) = degree of every kept row, rows in W's
 * order; blocks = nseg contiguous tensors [E][seg_rows_k][c_stride] (HOST arrays; two rotations may fill two channel halves of one block).
 * nq_es_rotate_back (_rotate_inv fused with _reduce_edge): out[n][s][c] = coef_scale[s] sum_{q in [ptr[n], ptr[n+1])} sum_i W_o[i][s] block(i)[o][row][c_off + c],
 * o = order ? order[q] : q; ptr == NULL: one output row per edge; coef_scale nullable. */
int nq_es_rotate(const float* W, int64_t w_stride, const float* x, int64_t x_stride, const int32_t* index, int32_t nseg, const int32_t* seg_rows,
                 float* const* seg_ptrs, int32_t c_stride, int32_t c_off, int64_t E, const int32_t* red_l, int32_t n_red, int32_t lmax, int32_t C, void* stream);
int nq_es_rotate_back(const float* W, int64_t w_stride, int32_t nseg, const int32_t* seg_rows, float* const* seg_ptrs, int32_t c_stride, int32_t c_off,
                      const int32_t* ptr, const int32_t* order, const float* coef_scale, float* out, int64_t n_out, const int32_t* red_l, int32_t n_red,
                      int32_t lmax, int32_t C, void* stream);

/* ---- EquiformerV2 building blocks (reference nablaDFT/equiformer_v2/; the graph, rotations, S2 grids and SO(2) GEMMs are the eSCN entries
 * above, SO3_LinearV2 is nq_sph_linear_*) ------------------------------------------------------------------------------------------------------------------ */
/* torch.nn.LayerNorm over rows of width W (radial_function.py:20, transformer_block.py:151): rows may be strided (x_stride / y_stride floats); stats [rows][2]
 * (mean, rstd) out.  Backward: grad_x, grad_weight [W], grad_bias [W] (fixed summation order); scratch nq_eq_layernorm_scratch_floats floats. */
int nq_eq_layernorm_forward(const float* x, int64_t x_stride, const float* weight, const float* bias, int64_t rows, int32_t W, float eps, float* y, int64_t y_stride,
                            float* stats, void* stream);
size_t nq_eq_layernorm_scratch_floats(int64_t rows, int32_t W);
int nq_eq_layernorm_backward(const float* x, int64_t x_stride, const float* weight, const float* grad_y, int64_t g_stride, const float* stats, int64_t rows, int32_t W,
                             float* grad_x, int64_t gx_stride, float* grad_weight, float* grad_bias, float* scratch, void* stream);
/* EquivariantLayerNormArraySphericalHarmonics (layer_norm.py:117-215, normalization "component", std_balance_degrees): x, y [N][(lmax+1)^2][C]; LayerNorm
 * (w0, b0) on the scalars; all l > 0 scaled by (mean_c sum_i balance_weight[i] x_ic^2 + eps)^-1/2 affine_weight[l-1][c]; stats [N][3] out; 1 <= lmax <= 6. */
int nq_eq_norm_sh_forward(const float* x, const float* w0, const float* b0, const float* affine_weight, const float* balance_weight, int64_t N, int32_t lmax,
                          int32_t C, float eps, float* y, float* stats, void* stream);
size_t nq_eq_norm_sh_scratch_floats(int64_t N, int32_t lmax, int32_t C);
int nq_eq_norm_sh_backward(const float* x, const float* w0, const float* affine_weight, const float* balance_weight, const float* grad_y, const float* stats,
                           int64_t N, int32_t lmax, int32_t C, float* grad_x, float* grad_w0, float* grad_b0, float* grad_affine_weight, float* scratch,
                           void* stream);
/* Attention logits (transformer_block.py:343-350; activation.py:52-61): z[e][h] = sum_a alpha_dot[h][a] SmoothLeakyReLU_0.2(x[e][h][a]). */
int nq_eq_logits_forward(const float* x, const float* alpha_dot, int64_t E, int32_t H, int32_t A, float* z, void* stream);
size_t nq_eq_logits_scratch_floats(int64_t E, int32_t H, int32_t A);
int nq_eq_logits_backward(const float* x, const float* alpha_dot, const float* grad_z, int64_t E, int32_t H, int32_t A, float* grad_x, float* grad_alpha_dot,
                          float* scratch, void* stream);
/* torch_geometric.utils.softmax(z, edge_index[1]) (transformer_block.py:352) for edges sorted by target: the in-edges of atom n are [ptr[n], ptr[n+1]). */
int nq_eq_softmax_forward(const float* z, const int32_t* ptr, int64_t N, int32_t H, float* out, void* stream);
int nq_eq_softmax_backward(const float* y, const float* grad_y, const int32_t* ptr, int64_t N, int32_t H, float* grad_z, void* stream);
/* Messages times attention weights (transformer_block.py:357-370) on the per-block tensors x_b [E][rows_b][H V] (host arrays of device pointers, nseg <= 8),
 * alpha [E][H].  grad == NULL: out_b = x_b alpha.  Else out_b = grad_b alpha and (grad_alpha != NULL) grad_alpha [E][H] = sum_b sum_rows,v grad_b x_b. */
int nq_eq_head_scale(int32_t nseg, const int32_t* rows, const float* const* x, const float* const* grad, const float* alpha, int64_t E, int32_t H, int32_t V,
                     float* const* out, float* grad_alpha, void* stream);
/* out[n][i][c] = x[n][i][c] * row_scale[row_index ? row_index[n] : n] * coef_scale[i] (nullable factors): the rescale of SO3_Rotation.rotate_inv
 * (so3.py:121-136,338-343) and GraphDropPath (drop.py:57-71). */
int nq_eq_scale(const float* x, const float* row_scale, const int32_t* row_index, const float* coef_scale, int64_t N, int32_t I, int32_t C, float* out, void* stream);

/* ---- loss / optimizer ------------------------------------------------------------------------ */
/* loss[1] = coef_e * mean|E-y| + coef_f * mean_i ||F_i - Ft_i||_2 ; grad_energy[B], grad_forces[N][3] */
int nq_loss_l1_l2(const float* energy, const float* y, int32_t B, const float* forces, const float* f_target, int32_t N, float coef_e,
                  float coef_f, float* loss, float* grad_energy, float* grad_forces, void* stream);
/* loss[1] = coef_e * mean (E-y)^2 + coef_f * mean_{i,c} (F_ic - Ft_ic)^2  -- torch.nn.MSELoss on both outputs, the loss of the
 * schnetpack task (config/model/painn.yaml:30-46, weights loss_weight) */
int nq_loss_mse(const float* energy, const float* y, int32_t B, const float* forces, const float* f_target, int32_t N, float coef_e,
                float coef_f, float* loss, float* grad_energy, float* grad_forces, void* stream);
/* clip_grad_norm_(max_norm) (skipped if max_norm <= 0) + AdamW step over the flat buffers; step >= 1;
 * scratch: f32[512]. */
int nq_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t count, float max_norm, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int32_t step, float* scratch, void* stream);

/* ---- measurement hook ------------------------------------------------------------------------- */
/* Opt-in per-kernel timing with HIP events recorded on the launch stream (bench.py roofline leg).
 * Process-global and not thread-safe: enable it only around a single-threaded measurement.
 * nq_profile_read synchronises the device, writes up to `cap` rows {name (name_stride bytes, NUL
 * terminated), total milliseconds, launch count}, clears the records and returns the number of names. */
void nq_profile_enable(int32_t on);
int nq_profile_read(char* names_host, int32_t name_stride, double* total_ms_host, int64_t* counts_host, int32_t cap);
/* As nq_profile_read, plus flops_host[i] (nullable) = arithmetic of the launches of name i: the dense-product launchers record 2 M N K per call, so a
 * class's TFLOP/s is flops / time without re-deriving shapes from names (every other launcher records 0).  ABI 11. */
int nq_profile_read2(char* names_host, int32_t name_stride, double* total_ms_host, int64_t* counts_host, double* flops_host, int32_t cap);

/* Engine / tuning switch of the dense products (process-global; default 1):
 *   bits 0-1  generic round-1 kernel flavour (bit0 = 8 wavefronts per 128x128 tile, bit1 = register prefetch),
 *   bit 3 (8)   never the small-problem kernel,  bit 4 (16)  only the generic kernels (no k_gemm2 / k_gemm3),
 *   bit 5 (32)  exact-f32 MFMA only: every product on v_mfma_f32_32x32x2_f32.  Without it (the default) launches of >= 192 tiles of 128x128 run on the
 *               split-bf16 engine (csrc/gemm_split.h): each f32 operand value is split EXACTLY into three bf16 pieces and a product is the sum of six
 *               piece products on v_mfma_f32_32x32x16_bf16 with f32 accumulation -- f32-accurate (error vs float64 measured <= the exact engine's),
 *               1.3-1.7x faster, but a row's bits then depend on which engine the launch size selects, and non-finite operands give NaN where the
 *               exact engine would give inf.  The environment variable NQ_GEMM_F32=1 (read once) has the same effect as bit 5.
 *   bit 6 (64)  the split-bf16 engine for EVERY launch it can run, whatever the size (tests: the golden vectors with all products on it).
 *   bit 8 (256) spherical linears on the row-mapped generic kernel (default since round 4: one plain strided product per packed component as ONE batched
 *               launch of the tile engines, grid.y = component).
 *   bit 7 (128) never split the contraction of a forward / input-gradient product (default: contractions of >= 2048 with fewer than 256 output tiles of
 *               128x128 are cut into up to 16 ranges, partial slabs in a library-owned per-stream scratch, fixed-order reduction; round 4).
 * The weight-gradient scratch size does not depend on the switch. */
void nq_set_gemm_variant(int32_t variant);
/* The split-K scratch (bit 7 above) is one grow-only device buffer per (device, stream), owned by the library.  A buffer that was handed out while its stream
 * was capturing is never freed or moved afterwards (the captured graph replays into the raw pointer): when a later eager call needs more, the old buffer is
 * retired and a new one allocated; inside a capture nothing is allocated (a too small buffer means the plain, unsplit launch).  nq_gemm_splitk_release frees
 * every buffer of the current device, retired ones included -- call it only when no graph that used them will be replayed again.  nq_gemm_splitk_state is the
 * test hook: current pointer, size in floats, captured flag and number of retired buffers of one stream. */
void nq_gemm_splitk_release(void);
int nq_gemm_splitk_state(void* stream, uint64_t* ptr, uint64_t* floats, int32_t* captured, int32_t* retired);

/* ---- building blocks exported for unit tests ----------------------------------------------- */
/* C[M,N] = A[M,K] W[N,K]^T (+bias[N]); if C_silu != NULL also writes silu(C). */
int nq_linear_forward(const float* A, const float* W, const float* bias, float* C, float* C_silu, int32_t M, int32_t N, int32_t K,
                      void* stream);
/* C[M,N] = A W^T and C_act = alpha * resid (nullable) + beta * silu(C): Dense + ScaledSiLU (+ residual) of gemnet_oc/layers/base_layers.py:11-97 in one pass. */
int nq_linear_forward_act(const float* A, const float* W, float* C, float* C_act, const float* resid, float alpha, float beta, int32_t M, int32_t N,
                          int32_t K, void* stream);
/* C[M,N] = alpha * aux + A[M,K] W[N,K]^T  (aux [M,N], not aliasing C): the second product of a two-term sum in the epilogue of the GEMM -- the +-m pairs of
 * the SO(2) convolutions (escn.py:858-877 `x_r[:, 0] - x_i[:, 1]`, `x_r[:, 1] + x_i[:, 0]`; equiformer_v2 so2_ops.py:53-61) without a separate pass.  ABI 13. */
int nq_linear_forward_res(const float* A, const float* W, const float* aux, float alpha, float* C, int32_t M, int32_t N, int32_t K, void* stream);
/* bf16 MFMA variants (fp32 operands in HBM rounded to bf16 on the way into LDS, fp32 accumulation; BASELINE.json configs[2] names bf16 for GemNet-OC):
 * nq_bf16_pack: W [N][K] fp32 -> Wb [N][K], WbT [K][N] (bf16, round to nearest even).  nq_linear_forward_bf16: as nq_linear_forward_act with W = Wb (C_act
 * nullable -> plain store); K % 32 == 0.  nq_linear_input_grad_bf16: C[M,K] (+)= G[M,N] W[N,K] with W given as WbT; N % 32 == 0. */
int nq_bf16_pack(const float* W, int32_t N, int32_t K, void* Wb, void* WbT, void* stream);
int nq_linear_forward_bf16(const float* A, const void* Wb, float* C, float* C_act, const float* resid, float alpha, float beta, int32_t M, int32_t N,
                           int32_t K, void* stream);
int nq_linear_input_grad_bf16(const float* G, const void* WbT, float* C, int32_t M, int32_t N, int32_t K, int32_t accumulate, void* stream);
/* gW[N][K] = G[rows][N]^T X[rows][K] on the bf16 MFMA (operands transposed into bf16 copies inside `scratch`, split over the rows, fixed-order reduction). */
size_t nq_weight_grad_bf16_scratch_bytes(int64_t rows, int32_t N, int32_t K);
int nq_linear_weight_grad_bf16(const float* G, const float* X, float* gW, int64_t rows, int32_t N, int32_t K, void* scratch, void* stream);
/* C[M,K] (+)= G[M,N] W[N,K] */
int nq_linear_input_grad(const float* G, const float* W, float* C, int32_t M, int32_t N, int32_t K, int32_t accumulate, void* stream);
/* Input gradient with a fused epilogue on C[M,K] = G W: mode 1: C = beta * (G W) * silu'(aux) (adjoint of the activation below), mode 2: C = alpha * aux + G W
 * (skip connection of the adjoint); aux [M,K].  _bf16_epi: the same on the bf16 MFMA kernel (W given as WbT, N % 32 == 0). */
int nq_linear_input_grad_epi(const float* G, const float* W, float* C, int32_t M, int32_t N, int32_t K, const float* aux, float alpha, float beta, int32_t mode,
                             void* stream);
int nq_linear_input_grad_bf16_epi(const float* G, const void* WbT, float* C, int32_t M, int32_t N, int32_t K, const float* aux, float alpha, float beta,
                                  int32_t mode, void* stream);
/* gW[N,K] = G[rows,N]^T X[rows,K]; scratch: f32[nq_weight_grad_scratch_floats(rows,N,K)] */
/* out[c] = sum over rows of A[r][c] (row stride lda floats) in a fixed order (per-chunk partial sums, then the chunks in order): the bias gradient of a
 * Linear layer (torch.nn.Linear backward); scratch: nq_column_sum_scratch_floats(rows, cols) floats. */
size_t nq_column_sum_scratch_floats(int64_t rows, int32_t cols);
int nq_column_sum(const float* A, int64_t rows, int32_t cols, int64_t lda, float* out, float* scratch, void* stream);
size_t nq_weight_grad_scratch_floats(int64_t rows, int32_t N, int32_t K);
int nq_linear_weight_grad(const float* G, const float* X, float* gW, int64_t rows, int32_t N, int32_t K, float* scratch, void* stream);
/* The same launch also produces the bias gradient gb[N] = column sums of G (taken from the operand registers on their way into LDS: no separate pass over G;
 * torch.nn.Linear backward); same scratch size. */
int nq_linear_weight_grad_bias(const float* G, const float* X, float* gW, float* gb, int64_t rows, int32_t N, int32_t K, float* scratch, void* stream);

/* ---- Data-parallel gradient exchange over RCCL (round 4; SURVEY 8(b) `nq_allreduce`) -------------------------------------------------------
 * Replaces: Lightning DDPStrategy's gradient all-reduce (nablaDFT/utils/pipelines.py:65-68, `strategy: ddp` of the config yaml files).  One process
 * per GPU; the only thing ever exchanged is the flat fp32 gradient buffer (one conformer = one graph: no data-path collective).  librccl.so is bound with
 * dlopen at the first call (no link-time dependency; nq_rccl_available() = 0 and NQ_ERR_ARG from the other entry points if it cannot be loaded).
 * Job set-up: rank 0 calls nq_rccl_unique_id (128 bytes) and hands the bytes to every rank through any side channel (nabladft_amd/dist.py uses the
 * torch.distributed store); every rank calls nq_rccl_comm_create with its HIP device current.  The collectives are enqueued on `stream` (ordered after the
 * kernels that wrote `buf`, before the optimiser kernel; no host synchronisation).  nq_allreduce: buf <- sum over ranks (in place, fp32);
 * nq_allreduce_mean: sum, then * 1/world on the same stream; nq_rccl_broadcast: buf of `root` -> all (initial parameters). */
int nq_rccl_available(void);
int nq_rccl_unique_id(void* id128);
int nq_rccl_comm_create(const void* id128, int32_t world, int32_t rank, void** comm);
int nq_rccl_comm_destroy(void* comm);
int nq_rccl_comm_count(void* comm, int32_t* count);   /* ncclCommCount: the ranks RCCL itself sees (bench.py reports it as config.collective.ranks_seen) */
int nq_allreduce(float* buf, size_t n, void* comm, void* stream);
int nq_allreduce_mean(float* buf, size_t n, void* comm, void* stream);
int nq_rccl_broadcast(float* buf, size_t n, int32_t root, void* comm, void* stream);

#ifdef __cplusplus
}
#endif
#endif
