"""Developer tool: HipUNet3D gradients vs the CPU oracle run in fp64 and in fp32 (needs an MI355X)."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import op_cases as C
from oracle import torch_ops as O, unet3d_ref as R
unet = importlib.import_module("3dunetcnn_amd.unet")
losses = importlib.import_module("3dunetcnn_amd.losses")
torch.set_num_threads(min(32, os.cpu_count()))

def run(dhw, n):
    torch.manual_seed(1234)
    m = unet.HipUNet3D(n_features=4, n_outputs=3).cuda().eval()
    x, y = R.synthetic_case(n, 4, dhw, 3)
    res = {}
    for dt in (torch.float32, torch.float64):
        sd = {k: v.detach().cpu().clone().to(dt).requires_grad_(True) for k, v in m.state_dict().items()}
        t = time.time()
        out = R.unet3d_forward(sd, x.to(dt)); l = O.dice_loss(out, y); l.backward()
        res[dt] = (out.detach(), {k: v.grad for k, v in sd.items()})
        print(dt, "cpu time", time.time() - t, flush=True)
    crit = losses.HipDiceLoss(sigmoid=True)
    out = m(x.cuda()); loss = crit(out, y.cuda()); loss.backward()
    print("==", dhw, n, "logits gpu-vs-64", C.rel_err(out, res[torch.float64][0]), "cpu32-vs-64", C.rel_err(res[torch.float32][0], res[torch.float64][0]))
    for k, p in m.named_parameters():
        g64 = res[torch.float64][1][k]
        e_gpu = C.rel_err(p.grad, g64); e_cpu = C.rel_err(res[torch.float32][1][k], g64)
        flag = " <<<<" if e_gpu > max(3 * e_cpu, 1e-5) else ""
        print(f"{k:58s} gpu-vs-64 {e_gpu:.2e}  cpu32-vs-64 {e_cpu:.2e}{flag}")

if __name__ == "__main__":
    run((64, 64, 64), 1)
