"""Developer tool (GPU): error of the first norm's gradients for a 3-channel input under the (side stream, fused statistics) switches."""
import sys, importlib, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import torch_ops as O, unet3d_ref as R
unet = importlib.import_module("3dunetcnn_amd.unet"); losses = importlib.import_module("3dunetcnn_amd.losses")
ops = importlib.import_module("3dunetcnn_amd.ops")
be = ops.default_backend()
cin = 3; dhw = (16, 20, 24)
x, y = R.synthetic_case(1, cin, dhw, 2)
torch.manual_seed(3)
m0 = unet.HipUNet3D(n_features=cin, n_outputs=2, base_width=8, encoder_blocks=[1, 1]).eval()
res = {}
for dt in (torch.float32, torch.float64):
    sd = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in m0.state_dict().items()}
    l = O.dice_loss(R.unet3d_forward(sd, x.to(dt), (1, 1)), y); l.backward(); res[dt] = {k: v.grad for k, v in sd.items()}
orig = be.conv_fwd
for side in (False,):
    for fused in ("none", "fwd", "fwd-sync", "fwd-sync-stats-only"):
        for rep in range(1):
            be.fused_stats = fused != "none"
            def patched(*a, _f=fused, **k):
                if _f.startswith("fwd"): k.pop("gnb", None)
                if _f == "bwd": k.pop("moments", None)
                r = orig(*a, **k)
                if _f == "fwd-sync": torch.cuda.synchronize()
                return r
            be.conv_fwd = patched
            if "ogs" not in globals(): ogs = be.gn_stats
            def pgs(*a, _f=fused, **k):
                if "sync" in _f: torch.cuda.synchronize()
                r = ogs(*a, **k)
                if "sync" in _f: torch.cuda.synchronize()
                return r
            be.gn_stats = pgs
            if fused == "both-nocat":
                om = be.moments
                be.moments = lambda x: []

            torch.manual_seed(3)
            m = unet.HipUNet3D(n_features=cin, n_outputs=2, base_width=8, encoder_blocks=[1, 1]).cuda().eval()
            m.backward_side_stream = side
            crit = losses.HipDiceLoss(sigmoid=True)
            loss = crit(m(x.cuda()), y.cuda()); loss.backward(); torch.cuda.synchronize()
            out = []
            for k, p in m.named_parameters():
                if "layers.0.blocks.0.conv1.norm1" in k:
                    e32 = float((p.grad.cpu() - res[torch.float32][k]).abs().max()); e64 = float((p.grad.cpu().double() - res[torch.float64][k]).abs().max())
                    out.append("%s e32 %.2e e64 %.2e" % (k.split(".")[-1], e32, e64))

            print("side", side, "fused", fused, "|", " | ".join(out), flush=True)
            if fused == "both-nocat":
                be.moments = om
