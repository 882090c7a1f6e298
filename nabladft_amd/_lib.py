"""ctypes binding of libnablaq.so (C ABI: include/nablaq.h).  No fallback: if the library or a
symbol is missing this module raises, it never routes around the HIP engine."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NABLAQ_LIB") or os.path.join(_HERE, "libnablaq.so")   # NABLAQ_LIB: development builds (scripts/ablate.sh)
ABI_VERSION = 16

NQ_OK, NQ_ERR_HIP, NQ_ERR_ARG, NQ_ERR_MOL_TOO_LARGE, NQ_ERR_WORKSPACE, NQ_ERR_NO_EDGES = range(6)


class NablaqError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libnablaq error {code}: {msg}")
        self.code = code


class PainnCfg(C.Structure):
    _fields_ = [("hidden_channels", C.c_int32), ("num_layers", C.c_int32), ("num_rbf", C.c_int32),
                ("num_elements", C.c_int32), ("max_neighbors", C.c_int32), ("envelope_exponent", C.c_int32),
                ("cutoff", C.c_double), ("rbf_coeff", C.c_float), ("filter_mode", C.c_int32), ("rbf_type", C.c_int32), ("reserved", C.c_int32)]


class SchnetCfg(C.Structure):
    _fields_ = [("n_atom_basis", C.c_int32), ("n_interactions", C.c_int32), ("n_rbf", C.c_int32), ("max_z", C.c_int32),
                ("cutoff", C.c_double), ("rbf_coeff", C.c_float), ("reserved", C.c_int32)]


class Graph(C.Structure):
    _fields_ = [("N", C.c_int32), ("B", C.c_int32), ("E", C.c_int32), ("max_mol_atoms", C.c_int32),
                ("mol_ptr", C.c_void_p), ("row_ptr", C.c_void_p), ("col", C.c_void_p), ("dst", C.c_void_p),
                ("rev", C.c_void_p), ("geom", C.c_void_p), ("z", C.c_void_p), ("atom_mol", C.c_void_p), ("lowptr", C.c_void_p)]


class GnSet(C.Structure):
    _fields_ = [("n", C.c_int32), ("reserved", C.c_int32), ("ptr", C.c_void_p), ("src", C.c_void_p), ("dst", C.c_void_p), ("geom", C.c_void_p)]


class GnGraphs(C.Structure):
    _fields_ = ([("N", C.c_int32), ("E", C.c_int32), ("k_main", C.c_int32), ("k_aea", C.c_int32), ("k_qint", C.c_int32), ("reserved", C.c_int32),
                 ("cutoff_main", C.c_double), ("cutoff_aea", C.c_double), ("cutoff_qint", C.c_double)]
                + [(k, C.c_void_p) for k in ("row_ptr", "col", "rev", "dst", "pos", "geom", "flags", "degm", "lowm", "cnt_a", "cnt_q", "tin_atom", "ptr_m", "lowptr_m", "ptr_a",
                                             "ptr_q", "tin_aptr", "m_src", "m_dst", "m_rev", "m_slot", "m_geom", "a_src", "a_dst", "a_geom", "a_of_rev",
                                             "q_src", "q_dst", "q_geom", "tin_ptr", "mpos", "apos", "tin_main", "q_of_rev", "qpos")])


_P, _I32, _I64, _F, _D, _SZ = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes); every symbol include/nablaq.h declares
SYMBOLS = {
    "nq_abi_version": (C.c_int, []),
    "nq_painn_molecule_lds_atoms": (C.c_int32, []),
    "nq_last_error": (C.c_char_p, []),
    "nq_graph_count": (C.c_int, [_P, _P, _I32, _I32, _I32, _D, _I32, _P, _P, _P, _P, C.POINTER(C.c_int32), _P]),
    "nq_graph_fill": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _D, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nq_painn_num_params": (_SZ, [C.POINTER(PainnCfg)]),
    "nq_painn_workspace_bytes": (_SZ, [C.POINTER(PainnCfg), _I32, _I32, _I32]),
    "nq_painn_forward": (C.c_int, [C.POINTER(PainnCfg), _P, _P, C.POINTER(Graph), _P, _SZ, _P, _P, _P]),
    "nq_painn_backward": (C.c_int, [C.POINTER(PainnCfg), _P, _P, C.POINTER(Graph), _P, _SZ, _P, _P, _P, _P]),
    "nq_painn_backward_events": (C.c_int, [C.POINTER(PainnCfg), _P, _P, C.POINTER(Graph), _P, _SZ, _P, _P, _P, _P, _P]),
    "nq_painn_layer_param_ranges": (C.c_int, [C.POINTER(PainnCfg), _P]),
    "nq_painn_backward_seeded": (C.c_int, [C.POINTER(PainnCfg), _P, _P, C.POINTER(Graph), _P, _SZ, _P, _P, _P, _P, _P]),
    "nq_scaled_silu": (C.c_int, [_P, _P, _P, _I64, _P]),
    "nq_geb_cat": (C.c_int, [_P, _P, _I64, _I32, _P, _P]),
    "nq_geb_cat_backward": (C.c_int, [_P, _P, _I64, _I32, _P, _P, _P]),
    "nq_geb_gate": (C.c_int, [_P, _P, _I64, _I32, _P, _P, _P]),
    "nq_geb_gate_backward": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _P, _P, _P]),
    "nq_painn_ws_lookup": (C.c_int, [C.POINTER(PainnCfg), _I32, _I32, _I32, C.c_char_p, _I32, _I32, C.POINTER(_SZ), C.POINTER(_SZ)]),
    "nq_schnet_num_params": (_SZ, [C.POINTER(SchnetCfg)]),
    "nq_schnet_workspace_bytes": (_SZ, [C.POINTER(SchnetCfg), _I32, _I32, _I32]),
    "nq_schnet_forward": (C.c_int, [C.POINTER(SchnetCfg), _P, _P, C.POINTER(Graph), _P, _SZ, _P, _P, _P]),
    "nq_schnet_backward": (C.c_int, [C.POINTER(SchnetCfg), _P, _P, C.POINTER(Graph), _P, _SZ, _P, _P, _P, _P]),
    "nq_hblock_tables": (C.c_int, [_P, _P, _P, _I32, _I32, _P, _P, _P, _P, _I64, _P, _P, _I32, _P, _P, _P, _I64, _P, _P]),
    "nq_hblock_assemble": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I64, _P, _P, _P]),
    "nq_hblock_assemble_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I64, _I32, _I32, _P, _P, _P]),
    "nq_hblock_packed_dense": (C.c_int, [_P, _P, _P, _P, _I32, _I64, _I64, _I32, _P]),
    "nq_irreps_assemble": (C.c_int, [_P] * 11 + [_I32, _I32, _I32] + [_P] * 8 + [_I32, _I32, _I32, _I32, _I64, _P, _P, _P]),
    "nq_irreps_assemble_backward": (C.c_int, [_P] * 10 + [_I64, _I64, _I32, _I32] + [_P] * 6 + [_I32, _I32, _I32, _I32, _P, _P, _P]),
    "nq_hamiltonian_loss": (C.c_int, [_P, _P, _I64, _F, _P, _P, _P, _P]),
    "nq_so3_mix_forward": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _P, _I64, _I32, _P, _P]),
    "nq_so3_mix_backward": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _P, _I64, _I32, _P, _P, _P, _P, _P]),
    "nq_qh_invariants_forward": (C.c_int, [_P, _I64, _I32, _I32, _P, _P, _I64, _I32, _P, _P]),
    "nq_qh_invariants_backward": (C.c_int, [_P, _P, _I64, _I32, _I32, _P, _P, _P, _I32, _P, _P]),
    "nq_qh_tp_num_paths": (C.c_int, [_I32]),
    "nq_qh_gen_fragment_floats": (C.c_size_t, [_I32, _I32]),
    "nq_qh_gen_presplit": (C.c_int, [_P, _P, _I32, _I32, _I32, _P, _P]),
    "nq_qh_tp_forward_gen": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _P, _P]),
    "nq_qh_tp_backward_gen": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _P, _P, _P, _P, _P]),
    "nq_qh_set_tp_variant": (None, [_I32]),
    "nq_qh_tp_forward": (C.c_int, [_P, _I32, _P, _P, _P, _P, _P, _I64, _I32, _I32, _P, _P]),
    "nq_qh_tp_backward": (C.c_int, [_P, _I32, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _P, _P, _P, _P, _P]),
    "nq_qh_pair_reduce": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _P, _P]),
    "nq_qh_normcat": (C.c_int, [_P, _P, _I64, _I32, _I32, _P, _P]),
    "nq_qh_gate": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _P, _P, _P, _P]),
    "nq_qh_act": (C.c_int, [_P, _P, _I32, _F, _I64, _P, _P]),
    "nq_qh_expansion_forward": (C.c_int, [_P, _P, _P, _I64, _I32, _P, _I32, _I32, _P, _P, _P]),
    "nq_qh_expansion_backward": (C.c_int, [_P, _P, _P, _I64, _I32, _P, _I32, _I32, _P, _P, _P, _P, _P]),
    "nq_so3_mix_partial_blocks": (C.c_int64, [_I64, _I32]),
    "nq_so3_mix_backward_shared": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _P, _I32, _P, _P, _P, _P, _P]),
    "nq_sph_harm": (C.c_int, [_P, _I64, _I32, _P, _P]),
    "nq_sph_harm_backward": (C.c_int, [_P, _P, _I64, _I32, _P, _P]),
    "nq_bernstein_rbf_grad_r_dev": (C.c_int, [_P, _P, _I64, _I32, _P, _F, _P, _P, _P, _P, _P]),
    "nq_bernstein_rbf": (C.c_int, [_P, _I64, _I32, _F, _F, _P, _P, _P, _P, _P]),
    "nq_bernstein_rbf_grad_alpha": (C.c_int, [_P, _P, _I64, _I32, _F, _F, _P, _P, _P, _P, _P]),
    "nq_bernstein_rbf_dev": (C.c_int, [_P, _I64, _I32, _P, _F, _P, _P, _P, _P, _P]),
    "nq_bernstein_rbf_grad_alpha_dev": (C.c_int, [_P, _P, _I64, _I32, _P, _F, _P, _P, _P, _P, _P]),
    "nq_radial_basis": (C.c_int, [_I32, _P, _I64, _I32, _F, _F, _F, _P, _P, _P, _P, _P]),
    "nq_radial_basis_grad_alpha": (C.c_int, [_I32, _P, _P, _I64, _I32, _F, _F, _F, _P, _P, _P, _P, _P]),
    "nq_feature_act": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _P, _P]),
    "nq_feature_act_backward": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _P, _P, _P, _P]),
    "nq_packed_act0": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _I32, _P, _P]),
    "nq_packed_act0_backward": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _P, _P, _P, _P]),
    "nq_sph_linear_forward": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _P]),
    "nq_sph_linear_input_grad": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _I32, _P]),
    "nq_sph_weight_grad_scratch_floats": (_SZ, [_I64, _I32, _I32, _I32]),
    "nq_sph_linear_weight_grad": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _P, _P]),
    "nq_gather_rows": (C.c_int, [_P, _P, _I64, _I32, _P, _P]),
    "nq_segment_sum": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _P, _P]),
    "nq_gn_graph_count": (C.c_int, [_P, _I32, C.POINTER(C.c_int32), _P]),
    "nq_gn_graph_fill": (C.c_int, [_P, _P]),
    "nq_gn_radial_basis": (C.c_int, [_P, _I64, _I32, _P, _D, _D, _F, _P, _P]),
    "nq_gn_triplet_forward": (C.c_int, [_P, _P, _P, _I32, _I32, _F, _P, _P]),
    "nq_gn_triplet_backward": (C.c_int, [_P, _P, _P, _I32, _I32, _F, _P, _P]),
    "nq_gn_quad_forward": (C.c_int, [_P, _P, _P, _I32, _P, _I32, _I32, _F, _P, _P]),
    "nq_gn_quad_backward": (C.c_int, [_P, _P, _P, _I32, _I64, _P, _I32, _I32, _I32, _F, _P, _P, _P]),
    "nq_gn_set_quad_variant": (None, [_I32]),
    "nq_gn_tin_scatter": (C.c_int, [_P, _P, _P, _P, _P, _I32, _P, _P]),
    "nq_gn_cir_forward": (C.c_int, [_P, _P, _P, _I64, _P, _I32, _I32, _F, _P, _P]),
    "nq_gn_cir_backward": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _F, _P, _P]),
    "nq_gn_rowmm_forward": (C.c_int, [_P, _P, _I64, _I32, _I32, _I32, _P, _P]),
    "nq_gn_rowmm_backward": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _I32, _P, _P, _P]),
    "nq_gn_pair_forward": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _P, _P]),
    "nq_gn_pair_backward": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P]),
    "nq_gn_cat_forward": (C.c_int, [_P, _P, _P, _I32, _I32, _P, _P]),
    "nq_gn_cat_backward_h": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _P, _P]),
    "nq_gn_mulsum_forward": (C.c_int, [_P, _P, _P, _I32, _I32, _P, _P]),
    "nq_gn_mulsum_backward": (C.c_int, [_P, _P, _P, _P, _I32, _P, _P, _P]),
    "nq_gn_forces_forward": (C.c_int, [_P, _P, _P, _I32, _I32, _P, _P]),
    "nq_gn_forces_backward": (C.c_int, [_P, _P, _P, _I32, _P, _P]),
    "nq_gn_gather": (C.c_int, [_P, _P, _P, _I64, _I32, _P, _P]),
    "nq_gn_segment_sum": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _P, _P]),
    "nq_gn_mul": (C.c_int, [_P, _P, _I64, _P, _P]),
    "nq_gn_lincomb": (C.c_int, [_P, _P, _F, _F, _I64, _P, _P]),
    "nq_gn_ssilu_backward": (C.c_int, [_P, _P, _F, _I64, _P, _P]),
    "nq_linear_forward_act": (C.c_int, [_P, _P, _P, _P, _P, _F, _F, _I32, _I32, _I32, _P]),
    "nq_linear_forward_res": (C.c_int, [_P, _P, _P, _F, _P, _I32, _I32, _I32, _P]),
    "nq_bf16_pack": (C.c_int, [_P, _I32, _I32, _P, _P, _P]),
    "nq_linear_forward_bf16": (C.c_int, [_P, _P, _P, _P, _P, _F, _F, _I32, _I32, _I32, _P]),
    "nq_linear_input_grad_bf16": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "nq_linear_input_grad_epi": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _P, _F, _F, _I32, _P]),
    "nq_linear_input_grad_bf16_epi": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _P, _F, _F, _I32, _P]),
    "nq_weight_grad_bf16_scratch_bytes": (_SZ, [_I64, _I32, _I32]),
    "nq_linear_weight_grad_bf16": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _P, _P]),
    "nq_gn_embed_grad": (C.c_int, [_P, _P, _I32, _I32, _I32, _P, _P]),
    "nq_es_graph_count": (C.c_int, [_P, _P, _P, _I32, _D, _I32, _P, _P, C.POINTER(C.c_int32), _P]),
    "nq_es_graph_fill": (C.c_int, [_P, _P, _P, _I32, _D, _I32, _P, _P, _P, _P, _P]),
    "nq_es_frames": (C.c_int, [_P, _I32, _P, _P]),
    "nq_es_wigner": (C.c_int, [_P, _I32, _P, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P]),
    "nq_es_smearing": (C.c_int, [_P, _I64, _I32, _P, _F, _P, _P]),
    "nq_rowop": (C.c_int, [_P, _I64, _P, _I64, _P, _P, _I64, _I64, _I32, _I32, _I32, _I32, _I32, _P]),
    "nq_rowop_blocks": (C.c_int, [_P, _I64, _P, _I64, _P, _I32, _I32, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P]),
    "nq_s2_activation_blocks": (C.c_int, [_P, _P, _I32, _I32, _I32, _I64, _I32, _P, _P, _P, _P, _I32, _P]),
    "nq_column_sum_scratch_floats": (_SZ, [_I64, _I32]),
    "nq_column_sum": (C.c_int, [_P, _I64, _I32, _I64, _P, _P, _P]),
    "nq_es_rotate": (C.c_int, [_P, _I64, _P, _I64, _P, _I32, _P, _P, _I32, _I32, _I64, _P, _I32, _I32, _I32, _P]),
    "nq_es_rotate_back": (C.c_int, [_P, _I64, _I32, _P, _P, _I32, _I32, _P, _P, _P, _P, _I64, _P, _I32, _I32, _I32, _P]),
    "nq_eq_layernorm_forward": (C.c_int, [_P, _I64, _P, _P, _I64, _I32, _F, _P, _I64, _P, _P]),
    "nq_eq_layernorm_scratch_floats": (_SZ, [_I64, _I32]),
    "nq_eq_layernorm_backward": (C.c_int, [_P, _I64, _P, _P, _I64, _P, _I64, _I32, _P, _I64, _P, _P, _P, _P]),
    "nq_eq_norm_sh_forward": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _I32, _F, _P, _P, _P]),
    "nq_eq_norm_sh_scratch_floats": (_SZ, [_I64, _I32, _I32]),
    "nq_eq_norm_sh_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _P, _P, _P, _P, _P, _P]),
    "nq_eq_logits_forward": (C.c_int, [_P, _P, _I64, _I32, _I32, _P, _P]),
    "nq_eq_logits_scratch_floats": (_SZ, [_I64, _I32, _I32]),
    "nq_eq_logits_backward": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _P, _P, _P, _P]),
    "nq_eq_softmax_forward": (C.c_int, [_P, _P, _I64, _I32, _P, _P]),
    "nq_eq_softmax_backward": (C.c_int, [_P, _P, _P, _I64, _I32, _P, _P]),
    "nq_eq_head_scale": (C.c_int, [_I32, _P, _P, _P, _P, _I64, _I32, _I32, _P, _P, _P]),
    "nq_eq_scale": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _P, _P]),
    "nq_loss_l1_l2": (C.c_int, [_P, _P, _I32, _P, _P, _I32, _F, _F, _P, _P, _P, _P]),
    "nq_loss_mse": (C.c_int, [_P, _P, _I32, _P, _P, _I32, _F, _F, _P, _P, _P, _P]),
    "nq_adamw_step": (C.c_int, [_P, _P, _P, _P, _SZ, _F, _F, _F, _F, _F, _F, _I32, _P, _P]),
    "nq_profile_enable": (None, [_I32]),
    "nq_profile_read": (C.c_int, [C.c_char_p, _I32, C.POINTER(C.c_double), C.POINTER(C.c_int64), _I32]),
    "nq_profile_read2": (C.c_int, [C.c_char_p, _I32, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), _I32]),
    "nq_set_gemm_variant": (None, [_I32]),
    "nq_gemm_splitk_release": (None, []),
    "nq_gemm_splitk_state": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "nq_linear_forward": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _P]),
    "nq_linear_input_grad": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "nq_weight_grad_scratch_floats": (_SZ, [_I64, _I32, _I32]),
    "nq_linear_weight_grad": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _P, _P]),
    "nq_linear_weight_grad_bias": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _P, _P]),
    "nq_rccl_available": (C.c_int, []),
    "nq_rccl_unique_id": (C.c_int, [_P]),
    "nq_rccl_comm_create": (C.c_int, [_P, _I32, _I32, C.POINTER(C.c_void_p)]),
    "nq_rccl_comm_destroy": (C.c_int, [_P]),
    "nq_rccl_comm_count": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "nq_allreduce": (C.c_int, [_P, _SZ, _P, _P]),
    "nq_allreduce_mean": (C.c_int, [_P, _SZ, _P, _P]),
    "nq_rccl_broadcast": (C.c_int, [_P, _SZ, _I32, _P, _P]),
}

_lib = None


def load():
    """Loads the shared library (once) and binds every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found -- build it with `python -m nabladft_amd.build` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.nq_abi_version() != ABI_VERSION:
        raise ImportError(f"libnablaq ABI {lib.nq_abi_version()} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc):
    if rc != NQ_OK:
        raise NablaqError(rc, load().nq_last_error().decode(errors="replace"))


def ptr(t):
    """Device (or host) pointer of a torch tensor, None -> NULL."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    """The current HIP stream of the current device as a C pointer.  Asked once per launch by every wrapper: the raw-handle query of torch (what its own
    compiled-kernel launchers use) instead of building a ``torch.cuda.Stream`` object each time."""
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is None:
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return C.c_void_p(raw(torch.cuda.current_device()))


def profile_enable(on: bool):
    load().nq_profile_enable(1 if on else 0)


def profile_read(cap=4096, stride=64):
    """-> {name: (total_ms, launches, flops)} of everything recorded since the last read; flops = 2 M N K summed over the launches of a dense-product class, 0 for other kernels (``cap`` distinct names: a process that ran several models has one name per
    GEMM shape of each of them -- 256 was too few for the default bench record, which dropped EquiformerV2's weight-gradient GEMMs from its table)."""
    names = C.create_string_buffer(cap * stride)
    tot = (C.c_double * cap)()
    cnt = (C.c_int64 * cap)()
    fl = (C.c_double * cap)()
    n = load().nq_profile_read2(names, stride, tot, cnt, fl, cap)
    out = {}
    for i in range(min(n, cap)):
        nm = names.raw[i * stride:(i + 1) * stride].split(b"\0", 1)[0].decode()
        out[nm] = (tot[i], cnt[i], fl[i])
    return out


def geometry_key(data):
    """What a prepared batch (net.prepare(data)) is tied to: the identity AND the in-place version of the position tensor plus the atom count.  A forward with
    ``data.prepared`` compares it, so positions updated in place (MD, geometry optimisation, refreshed inputs of a captured step) or another batch of the
    same size cannot silently reuse a stale neighbour list / frames / Wigner rows."""
    pos = data.pos
    return (int(pos.data_ptr()), int(pos._version), tuple(pos.shape), int(data.z.data_ptr()) if getattr(data, "z", None) is not None and hasattr(data.z, "data_ptr") else 0)


def check_prepared(prep, data):
    key = getattr(prep, "geometry_key", None)
    if key is not None and key != geometry_key(data):
        raise ValueError("data.prepared was built for another geometry (positions changed in place, or another batch): call net.prepare(data) again")
