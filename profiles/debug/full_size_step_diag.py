import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import gnn_tableextraction_amd as gte
from gnn_tableextraction_amd import graph as G
from gnn_tableextraction_amd.data import synthetic as S
from gnn_tableextraction_amd.models.engine import FusedGcnSageStep
from oracle import gcnsage_cpu as oc
pages = S.make_pages(100, in_feats=831)
src, dst, w, feat, label, off = S.concat_pages(pages)
n = int(off[-1])
torch.manual_seed(42)
model = gte.GcnSAGE(831, 256, 9, 3, torch.nn.functional.relu, 0)
state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
og = oc.OracleGraph(src, dst, n, w)
xt, yt = torch.from_numpy(feat), torch.from_numpy(label)
tr = oc.OracleTrainer(state0, lr=0.01, weight_decay=5e-4)
want_loss, _ = tr.step(og, xt, yt)
want_grads = {k: v.numpy() for k, v in tr.grads().items()}
want_state = {k: v.detach().numpy() for k, v in tr.state.items()}
model = model.to("cuda:0")
g = G.PageGraph(src, dst, n, device="cuda:0")
g.ndata["feat"], g.edata["feat"] = torch.from_numpy(feat).cuda(), torch.from_numpy(w).cuda()
fused = FusedGcnSageStep(model, lr=0.01, weight_decay=5e-4)
out3 = fused.step(g, torch.from_numpy(label).cuda().float())
print("loss", float(out3[0]), want_loss)
for k, p in model.named_parameters():
    got_g = fused._gslice[id(p)].cpu().numpy(); ref = want_grads[k]
    got, want = p.detach().cpu().numpy(), want_state[k]
    bad = ~np.isclose(got, want, rtol=1e-5, atol=1e-5)
    ge = np.abs(ref + 5e-4 * state0[k].numpy())
    print(k, "grad max", np.abs(ref).max(), "grad err max", np.abs(got_g - ref).max(), "bad", bad.sum(), "of", bad.size,
          "max g_eff at bad", ge[bad].max() if bad.any() else 0, "max |dparam|", np.abs(got - want).max())
    if bad.any():
        i = np.argmax(np.where(bad, ge, 0))
        print("   worst:", "g_ref", ref.reshape(-1)[i], "g_got", got_g.reshape(-1)[i], "p0", state0[k].numpy().reshape(-1)[i], "want", want.reshape(-1)[i], "got", got.reshape(-1)[i])
