"""Developer tool: per-parameter gradient errors of HipUNet3D vs the CPU oracle (needs an MI355X)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import op_cases as C
from oracle import torch_ops as O, unet3d_ref as R
unet = importlib.import_module("3dunetcnn_amd.unet")
losses = importlib.import_module("3dunetcnn_amd.losses")
ops = importlib.import_module("3dunetcnn_amd.ops")

def run(dhw, n):
    torch.manual_seed(1234)
    m = unet.HipUNet3D(n_features=4, n_outputs=3).cuda().eval()
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    x, y = R.synthetic_case(n, 4, dhw, 3)
    ref = R.unet3d_forward(sd, x); lref = O.dice_loss(ref, y); lref.backward()
    crit = losses.HipDiceLoss(sigmoid=True)
    gs = []
    for rep in range(2):
        for p in m.parameters(): p.grad = None
        out = m(x.cuda()); loss = crit(out, y.cuda()); loss.backward()
        gs.append({k: p.grad.detach().clone() for k, p in m.named_parameters()})
    print("==", dhw, n, "logits", C.rel_err(out, ref))
    for k, p in m.named_parameters():
        e = C.rel_err(gs[0][k], sd[k].grad)
        same = torch.equal(gs[0][k], gs[1][k])
        flag = " <<<<" if e > 1e-4 else ""
        print(f"{k:60s} {tuple(p.shape)!s:24s} err {e:.2e} maxref {float(sd[k].grad.abs().max()):.2e} det {same}{flag}")

if __name__ == "__main__":
    be = ops.default_backend()
    for seed in range(3):
        print("op wgrad 128->128 16^3 n1 norm seed", seed, C.case_conv_wgrad(be, 1, 128, 128, (16, 16, 16), norm=True, seed=seed))
    print("op wgrad 64->128 16^3", C.case_conv_wgrad(be, 1, 64, 128, (16, 16, 16), norm=True))
    print("op dgrad 128->128 16^3", C.case_conv_dgrad(be, 1, 128, 128, (16, 16, 16)))
    print("gn 128 16^3", C.case_gn(be, 1, 128, (16, 16, 16), 8))
    run((64, 64, 64), 1)
    run((32, 32, 32), 2)
