"""Developer tool (GPU): per-tensor gradient diagnostics of the headline configuration (UNet3D 4->3, 128^3, batch 2, fp32) against the
fp32 / fp64 CPU oracle and the one-ulp conditioning probe -- the numbers tests/test_headline_parity_gpu.py's recorded bounds come from.
    python tools/headline_diag.py [out.json]"""
import importlib, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import op_cases as C
from oracle import conditioning, torch_ops as O, unet3d_ref as R
unet = importlib.import_module("3dunetcnn_amd.unet"); losses = importlib.import_module("3dunetcnn_amd.losses")
torch.set_num_threads(min(64, os.cpu_count() or 1))
torch.manual_seed(1234)
m = unet.HipUNet3D(n_features=4, n_outputs=3).cuda().eval()
x, y = R.synthetic_case(2, 4, (128, 128, 128), 3)
sd0 = m.state_dict()
out = m(x.cuda()); loss = losses.HipDiceLoss(sigmoid=True)(out, y.cuda()); loss.backward()
grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}

def run(dt, per_sample=False):
    sd = {k: v.detach().cpu().clone().to(dt).requires_grad_(True) for k, v in sd0.items()}
    n = x.shape[0]
    for a, b in ([(i, i + 1) for i in range(n)] if per_sample else [(0, n)]):
        l = O.dice_loss(R.unet3d_forward(sd, x[a:b].to(dt)), y[a:b]) * ((b - a) / n)
        l.backward()
    return {k: v.grad for k, v in sd.items()}
t0 = time.perf_counter(); g32 = run(torch.float32); t32 = time.perf_counter() - t0
t0 = time.perf_counter(); g64 = run(torch.float64, True); t64 = time.perf_counter() - t0
t0 = time.perf_counter(); floor, pert = conditioning.noise_floor(R, lambda: run(torch.float32), return_evals=True); tfl = time.perf_counter() - t0
rows = {k: dict(e32=C.rel_err(grads[k], g32[k]), e64=C.rel_err(grads[k], g64[k]), ref32_vs_64=C.rel_err(g32[k], g64[k]), floor=floor[k],
                pert=min(C.rel_err(grads[k], p[k]) for p in pert)) for k in grads}
w = C.grad_parity(grads, g32, g64, floor, 1e-3, perturbed=pert)
res = dict(t32=t32, t64=t64, tfloor=tfl, worst=w, rows=rows)
print(json.dumps(dict(t32=t32, t64=t64, tfloor=tfl, worst=w)))
for k, r in sorted(rows.items(), key=lambda kv: -kv[1]["e32"])[:25]:
    print(f"{k:60s} e32 {r['e32']:.2e} e64 {r['e64']:.2e} ref32-64 {r['ref32_vs_64']:.2e} floor {r['floor']:.2e} pert {r['pert']:.2e}")
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"))
