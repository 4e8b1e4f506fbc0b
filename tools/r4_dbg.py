"""Developer tool (GPU): the norm-backward (gnb) form of the plane-ring kernel, one case per process step with a sync after every launch."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if os.environ.get("ONE_CONV_LIB"):
    importlib.import_module("3dunetcnn_amd._lib").LIB_PATH = os.path.abspath(os.environ["ONE_CONV_LIB"])
ops = importlib.import_module("3dunetcnn_amd.ops")
be = ops.default_backend()
be.set_precision("bf16")
n, c, s = 1, 32, (5, 8, 16)
g = torch.Generator().manual_seed(0)
x = be.empty_act(n, *s, c); x.buf.normal_()
dy = be.empty_act(n, *s, c); dy.buf.normal_()
dA = be.empty_act(n, *s, c)
w = torch.randn(c, c, 3, 3, 3, device=be.device) * 0.05
wp = be.pack_weight(w, 1)
st = be.gn_stats(x, 8, 1e-5, torch.ones(c, device=be.device), torch.zeros(c, device=be.device))
torch.cuda.synchronize(); print("stats ok", flush=True)
be.conv_fwd(dy, wp, dA, 3, 1); torch.cuda.synchronize(); print("plain ok", flush=True)
be.conv_fwd(dy, wp, dA, 3, 1, moments=True); torch.cuda.synchronize(); print("moments ok", flush=True)
parts = be.conv_fwd(dy, wp, dA, 3, 1, gnb=(x, st, 8, 0.0)); torch.cuda.synchronize(); print("gnb ok", parts[1], float(parts[0].abs().sum()), flush=True)
