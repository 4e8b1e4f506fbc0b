"""Developer tool: bisects the head of the backward chain (dice grad -> proj bwd -> first dgrad) against fp64 math on the
tensors of a real step (needs an MI355X)."""
import importlib, os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import op_cases as C
from oracle import unet3d_ref as R, torch_ops as O
unet = importlib.import_module("3dunetcnn_amd.unet")
losses = importlib.import_module("3dunetcnn_amd.losses")
ops = importlib.import_module("3dunetcnn_amd.ops")
torch.set_num_threads(32)
be = ops.default_backend()
cap = {}
orig_proj_bwd, orig_conv, orig_dice = be.proj_bwd, be.conv_fwd, be.dice
def dice_wrap(logits, target, **kw):
    loss, d = orig_dice(logits, target, **kw)
    z = logits.detach().cpu().double().requires_grad_(True)
    l = O.dice_loss(z, target.cpu()); l.backward()
    print("dice: loss", abs(float(loss) - float(l)) / float(l), "dlogits vs fp64", C.rel_err(d, z.grad), "max|dz|", float(z.grad.abs().max()))
    # error relative per element where it matters
    rel = ((d.cpu().double() - z.grad).abs() / z.grad.abs().clamp_min(1e-30))
    big = z.grad.abs() > 1e-3 * z.grad.abs().max()
    print("   elementwise rel err (elements > 1e-3 max): max", float(rel[big].max()), "mean", float(rel[big].mean()))
    return loss, d
def proj_wrap(x, w, dlogits, dx, dw, dbias):
    orig_proj_bwd(x, w, dlogits, dx, dw, dbias)
    ref = torch.einsum("ncdhw,ck->ndhwk", dlogits.double(), w.double())
    print("proj_bwd dx vs fp64", C.rel_err(dx.tensor(), ref))
    cap["d_last"] = dx
def conv_wrap(x, wp, y, kd, stride=1, **kw):
    orig_conv(x, wp, y, kd, stride, **kw)
    if "d_last" in cap and x is cap["d_last"] and "done" not in cap:
        cap["done"] = 1
        wt = cap["w2"]          # OIDHW of the last block's conv2
        dy = x.tensor().permute(0, 4, 1, 2, 3).cpu().double()
        ref = F.conv_transpose3d(dy, wt.cpu().double(), None, stride=1, padding=1)
        got = y.tensor().permute(0, 4, 1, 2, 3).cpu().double()
        print("first dgrad vs fp64 (max-norm)", C.rel_err(got, ref))
        s_got, s_ref = got.sum((0, 2, 3, 4)), ref.sum((0, 2, 3, 4))
        print("   per-channel plain sums rel err", float(((s_got - s_ref).abs() / s_ref.abs()).max()))
be.dice, be.proj_bwd, be.conv_fwd = dice_wrap, proj_wrap, conv_wrap
torch.manual_seed(1234)
m = unet.HipUNet3D(n_features=4, n_outputs=3).cuda().eval()
cap["w2"] = m.decoder.layers[-1].blocks[0].conv2.conv.weight.detach()
x, y = R.synthetic_case(1, 4, (64, 64, 64), 3)
crit = losses.HipDiceLoss(sigmoid=True)
out = m(x.cuda()); loss = crit(out, y.cuda()); loss.backward()
