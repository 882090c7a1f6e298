"""Development aid: per-stage errors of nabladft_amd.GemNetOC against tests/golden/gemnet_small.npz (run on the GPU box)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gemnet_gpu as T
d = np.load(os.path.join(ROOT, "tests/golden/gemnet_small.npz"))
dev = torch.device("cuda:0")
net = T.build(T.SMALL, d, dev, True)
data = T.Data(d, dev)
E, F, inter = net(data, return_intermediates=True)
G = inter["graphs"]; R = G.to_reference(); rid = R["main_ref_id"]; B = inter["bases"]
def show(k, got, ref):
    print(f"{k:40s} {T.rel(got.detach().cpu().numpy(), ref):.3e}")
for k, v, m in (("basis:rad_main_raw", B["rad_main_raw"], 1), ("basis:atom_update", B["atom_update"], 1), ("basis:output", B["output"], 1), ("basis:e2e:rad", B["e2e"]["rad"], 1),
                ("basis:a2e:rad", B["a2e"]["rad"], 0), ("basis:qint:rad", B["qint"]["rad"], 1), ("basis:e2a:rad", B["e2a"]["rad"], 1)):
    ref = d["f32:" + k]; show(k, v, ref[rid] if m else ref)
tni, tgt = d["f32:graph:a2a:target_neighbor_idx"], d["f32:graph:a2a:edge_index"][1]
show("basis:a2a_rad", B["a2a_rad"], d["f32:basis:a2a_rad"][tgt, :, tni])
show("e2e cir", B["e2e"]["cir"], d["f32:basis:e2e:cir:rad_W1"].reshape(len(rid), -1)[rid])
show("qint sph", B["qint"]["sph"], d["f32:basis:qint:sph:rad_W1"].reshape(len(rid), -1)[rid])
show("a2e cir", B["a2e"]["cir"], d["f32:basis:a2e:cir:rad_W1"].reshape(len(rid), -1)[rid])
show("e2a cir (padded ref)", B["e2a"]["cir"], B["e2a"]["cir"].detach().cpu().numpy())
# qint cir: reference rows are the masked triplet_in list; ours are padded (q, j) rows
tin_q, tin_in = d["f32:quad:triplet_in:out"], d["f32:quad:triplet_in:in"]
t = {k: v.cpu().numpy() for k, v in G.t.items() if torch.is_tensor(v)}
inv = np.argsort(rid)       # ref main id -> slot
rows = t["tin_ptr"][tin_q] + (inv[tin_in] - t["ptr_m"][t["m_dst"][inv[tin_in]]])
show("qint cir", B["qint"]["cir"][torch.tensor(rows, device=dev)], d["f32:basis:qint:cir"])
for k, v, m in [("atom_emb", inter["atom_emb"], 0), ("edge_emb", inter["edge_emb"], 1), ("out0:0", inter["out0"][0], 0), ("out0:1", inter["out0"][1], 1)]:
    ref = d["f64:" + k]; show(k, v, ref[rid] if m else ref)
# sub-interactions of block 0
blk = net.int_blocks[0]
h, m = inter["atom_emb"], inter["edge_emb"]
NS = net.num_spherical
show("int0:e2e", blk.trip_interaction(m, B["e2e"], G, "e2e", NS), d["f64:int0:e2e"][rid])
show("int0:qint", blk.quad_interaction(m, B["qint"], G, NS), d["f64:int0:qint"][rid])
show("int0:a2e", blk.atom_edge_interaction(h, B["a2e"], G, "a2e", NS), d["f64:int0:a2e"][rid])
show("int0:e2a", blk.edge_atom_interaction(m, B["e2a"], G, "e2a", NS), d["f64:int0:e2a"])
show("int0:a2a", blk.atom_interaction(h, B["a2a_rad"], G), d["f64:int0:a2a"])
for i in range(2):
    show(f"int{i}:0", inter[f"int{i}"][0], d[f"f64:int{i}:0"]); show(f"int{i}:1", inter[f"int{i}"][1], d[f"f64:int{i}:1"][rid])
show("E", E, d["f64:E"]); show("F", F, d["f64:F"])
loss = T._loss(E, F, data); loss.backward()
print("loss", float(loss), float(d["f64:loss"]))
for name, p in net.named_parameters():
    if not p.requires_grad: continue
    ref = d["f64:grad:" + name]; g = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref)
    e = np.abs(g - ref).max() / max(np.abs(ref).max(), 1e-30)
    if e > 5e-5: print(f"GRAD {name:70s} {e:.3e}  |ref| {np.abs(ref).max():.3e}")
print("---- edge_emb pieces")
with torch.no_grad():
    from nabladft_amd import gemnet_oc as GO
    h = inter["atom_emb"].detach(); rb = B["rad_main_raw"].detach()
    cat = GO._CatFn.apply(h, rb, G)
    src, dst = G.t["m_src"].long(), G.t["m_dst"].long()
    cat_t = torch.cat([h[src], h[dst], rb], dim=1)
    print("cat", (cat - cat_t).abs().max().item())
    W = net.edge_emb.dense.linear.weight
    y = GO._DenseFn.apply(cat_t, W, False); print("gemm", (y - cat_t @ W.T).abs().max().item(), (cat_t @ W.T).abs().max().item())
    y2 = GO._DenseFn.apply(cat_t, W, True); z = cat_t @ W.T; print("silu", (y2 - torch.nn.functional.silu(z) / 0.6).abs().max().item())
    ref = torch.tensor(d["f64:edge_emb"][rid], device=dev)
    print("vs ref", (y2 - ref).abs().max().item(), ref.abs().max().item())
    Wref = torch.tensor(d["state:edge_emb.dense.linear.weight"], device=dev); print("W", (W - Wref).abs().max().item())
    href = torch.tensor(d["f32:atom_emb"], device=dev); print("h", (h - href).abs().max().item())
    ei = torch.tensor(d["f32:graph:main:edge_index"], device=dev)
    rbref = torch.tensor(d["f32:basis:rad_main_raw"], device=dev)
    mref = torch.nn.functional.silu(torch.cat([href[ei[0]], href[ei[1]], rbref], 1) @ Wref.T) / 0.6
    print("torch-recomputed ref vs fixture", (mref - torch.tensor(d["f32:edge_emb"], device=dev)).abs().max().item())
