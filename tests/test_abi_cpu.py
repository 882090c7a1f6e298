"""CPU: the C-ABI library loads and exports every symbol include/nablaq.h declares; host-side logic
(parameter layout, config struct, workspace sizing, state_dict surface, error paths) without any compute."""
import ctypes as C
import os
import re

import pytest
import torch

from oracle import painn_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from nabladft_amd.build import build
    build(verbose=False)
    from nabladft_amd import _lib
    return _lib.load()


def test_header_symbols_all_exported(lib):
    from nabladft_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "nablaq.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(nq_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/nablaq.h but not exported by libnablaq.so"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    assert lib.nq_abi_version() == _lib.ABI_VERSION


def test_param_layout_matches_reference_state_dict(lib):
    import nabladft_amd as nq
    cfg = R.PaiNNConfig()
    m = nq.PaiNN(cfg.hidden_channels, cfg.num_layers, cfg.num_rbf, cfg.cutoff, cfg.max_neighbors, {"name": "gaussian"},
                 {"name": "polynomial", "exponent": 5}, True, False, False, True, cfg.num_elements)
    names = [(k, tuple(p.shape)) for k, p in m.named_parameters()]
    assert names == R.param_shapes(cfg)                       # == reference named_parameters (asserted in make_golden.py)
    assert m.num_params == 1341313 == lib.nq_painn_num_params(C.byref(m._cfg))
    sd = m.state_dict()
    assert len(sd) == 72 and "radial_basis.rbf.offset" in sd
    assert torch.equal(sd["radial_basis.rbf.offset"], torch.linspace(0, 1, 100))
    # flat buffer: parameters become views in state_dict order, load_state_dict writes through
    flat = m.flat_parameters()
    assert flat.numel() == 1341313
    params = R.make_params(cfg, seed=23)
    m.load_state_dict(params, strict=False)
    assert m.flat_parameters().data_ptr() == flat.data_ptr()
    assert torch.equal(flat, torch.cat([params[k].reshape(-1) for k, _ in names]))


def test_workspace_and_lookup(lib):
    import nabladft_amd as nq
    m = nq.PaiNN(64, 2, 20, 3.0, 6, {"name": "gaussian"}, {"name": "polynomial", "exponent": 5}, True, False, False, True, 100)
    cfg = C.byref(m._cfg)
    total = lib.nq_painn_workspace_bytes(cfg, 100, 900, 4)
    assert total > 0 and total % 16 == 0
    off, cnt = C.c_size_t(), C.c_size_t()
    seen = []
    for name, lay, tan in [("x_in", 0, 0), ("x_in", 2, 1), ("vec_msg", 1, 0), ("xh", 0, 0), ("rw", 0, 0), ("zo", 0, 0), ("gedge", 0, 0)]:
        assert lib.nq_painn_ws_lookup(cfg, 100, 900, 4, name.encode(), lay, tan, C.byref(off), C.byref(cnt)) == 0
        assert (off.value + cnt.value) * 4 <= total and off.value % 4 == 0
        seen.append((off.value, cnt.value))
    assert len(set(seen)) == len(seen)
    assert lib.nq_painn_ws_lookup(cfg, 100, 900, 4, b"nonsense", 0, 0, C.byref(off), C.byref(cnt)) != 0
    assert b"unknown workspace buffer" in lib.nq_last_error()
    assert lib.nq_painn_ws_lookup(cfg, 100, 900, 4, b"phi", 0, 0, C.byref(off), C.byref(cnt)) != 0   # fused filter: not materialised
    os.environ["NQ_NO_FUSED_FILTER"] = "1"
    try:
        assert lib.nq_painn_ws_lookup(cfg, 100, 900, 4, b"phi", 0, 0, C.byref(off), C.byref(cnt)) == 0
        assert lib.nq_painn_ws_lookup(cfg, 100, 900, 4, b"phi", 0, 1, C.byref(off), C.byref(cnt)) != 0   # no tangent half
        assert lib.nq_painn_workspace_bytes(cfg, 100, 900, 4) != total
    finally:
        del os.environ["NQ_NO_FUSED_FILTER"]


def test_unsupported_configs_fail_loudly():
    import nabladft_amd as nq
    kw = dict(rbf={"name": "gaussian"}, envelope={"name": "polynomial", "exponent": 5}, regress_forces=True, direct_forces=False,
              use_pbc=False, otf_graph=True, num_elements=100)
    with pytest.raises(ValueError):
        nq.PaiNN(100, 6, 100, 5.0, 100, **kw)                # hidden_channels not a multiple of 64
    with pytest.raises(ValueError):                           # unknown names fail like the reference (layers.py:166,179)
        nq.PaiNN(128, 6, 100, 5.0, 100, **{**kw, "rbf": {"name": "chebyshev"}})
    with pytest.raises(ValueError):
        nq.PaiNN(128, 6, 100, 5.0, 100, **{**kw, "envelope": {"name": "cosine"}})
    # every RadialBasis option of the reference is built: parameter surface of the learnable bases
    mb = nq.PaiNN(64, 1, 12, 4.0, 100, **{**kw, "rbf": {"name": "spherical_bessel"}})
    assert tuple(mb.state_dict()["radial_basis.rbf.frequencies"].shape) == (12,) and "radial_basis.rbf.offset" not in mb.state_dict()
    mn = nq.PaiNN(64, 1, 12, 4.0, 100, **{**kw, "rbf": {"name": "bernstein"}, "envelope": {"name": "exponential"}})
    assert tuple(mn.state_dict()["radial_basis.rbf.pregamma"].shape) == () and "radial_basis.rbf.prefactor" not in mn.state_dict()
    assert [n for n, _ in mn.named_parameters()][:2] == ["atom_emb.embeddings.weight", "radial_basis.rbf.pregamma"]
    with pytest.raises(NotImplementedError):
        nq.PaiNN(128, 6, 100, 5.0, 100, **{**kw, "use_pbc": True})
    md = nq.PaiNN(64, 2, 20, 4.0, 100, **{**kw, "direct_forces": True})        # direct-force head: reference parameter surface
    cfg_d = R.PaiNNConfig(hidden_channels=64, num_layers=2, num_rbf=20, cutoff=4.0, direct_forces=True)
    assert [(n, tuple(p.shape)) for n, p in md.named_parameters()] == R.param_shapes(cfg_d)
    m = nq.PaiNN(128, 1, 100, 5.0, 100, **kw)
    pos, z, batch, _, _ = R.gen_conformers(0, 1, size=5)
    with pytest.raises(RuntimeError, match="MI355X only"):      # no CPU fallback
        m(nq.Batch(pos, z, batch))


def test_bad_cfg_rejected_by_abi(lib):
    from nabladft_amd import _lib
    cfg = _lib.PainnCfg()
    cfg.hidden_channels, cfg.num_layers, cfg.num_rbf, cfg.num_elements = 96, 6, 100, 100
    assert lib.nq_painn_num_params(C.byref(cfg)) == 0
    assert lib.nq_painn_workspace_bytes(C.byref(cfg), 10, 10, 1) == 0


_CTYPE = {"int32_t": "c_int", "int64_t": "c_long", "double": "c_double", "float": "c_float", "size_t": "c_ulong"}


def _header_structs():
    """{struct name: [(field, C type or 'ptr')]} parsed from include/nablaq.h."""
    hdr = open(os.path.join(ROOT, "include", "nablaq.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    out = {}
    for name, body in re.findall(r"typedef struct (\w+) \{(.*?)\} \1;", hdr, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"(const )?(\w+)(\s*\*)?\s*(.*)", decl)
            typ, is_ptr, names = m.group(2), m.group(3) is not None, m.group(4)
            for nm in names.split(","):
                nm = nm.strip()
                fields.append((nm.lstrip("* "), "ptr" if (is_ptr or nm.startswith("*")) else typ))
        out[name] = fields
    return out


def _ctypes_fields(cls):
    out = []
    for nm, ct in cls._fields_:
        if ct is C.c_void_p:
            out.append((nm, "ptr"))
        else:
            out.append((nm, {C.c_int32: "int32_t", C.c_int64: "int64_t", C.c_double: "double", C.c_float: "float", C.c_size_t: "size_t"}[ct]))
    return out


def test_struct_layouts_agree_between_header_binding_and_integration_doc():
    """include/nablaq.h is the source of truth: nabladft_amd/_lib.py and every ctypes.Structure shown in INTEGRATION.md must list the same fields in
    the same order with the same types (a stale 40-byte nq_painn_cfg in the doc once made copy-pasting bindings read past the struct)."""
    from nabladft_amd import _lib
    hs = _header_structs()
    assert set(hs) == {"nq_painn_cfg", "nq_schnet_cfg", "nq_graph", "nq_gn_set", "nq_gn_graphs"}
    for cname, cls in (("nq_painn_cfg", _lib.PainnCfg), ("nq_schnet_cfg", _lib.SchnetCfg), ("nq_graph", _lib.Graph), ("nq_gn_set", _lib.GnSet),
                       ("nq_gn_graphs", _lib.GnGraphs)):
        assert _ctypes_fields(cls) == hs[cname], cname
    assert C.sizeof(_lib.PainnCfg) == 48
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    snippets = re.findall(r"class (nq_\w+)\(ctypes\.Structure\):.*?_fields_ = \[(.*?)\]\n", doc, flags=re.S)
    assert {n for n, _ in snippets} >= {"nq_painn_cfg", "nq_graph"}
    for name, body in snippets:
        fields = [(f, "ptr" if t == "c_void_p" else {"c_int32": "int32_t", "c_int64": "int64_t", "c_double": "double", "c_float": "float"}[t])
                  for f, t in re.findall(r'\("(\w+)", ctypes\.(\w+)\)', body)]
        assert fields == hs[name], f"INTEGRATION.md shows a stale {name}"
    m = re.search(r"lib\.nq_abi_version\(\) == (\d+)", doc)
    assert m and int(m.group(1)) == _lib.ABI_VERSION
