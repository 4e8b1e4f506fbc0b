"""Developer tool (GPU): the fused first-layer backward alone, for a rocprofv3 kernel trace (tools/r6_call.sh)."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("3dunetcnn_amd.ops")
S, N = 128, 2
be = ops.default_backend()
g = torch.Generator(device="cuda").manual_seed(0)
x = be.empty_act(N, S, S, S, 4); x.buf.normal_(generator=g)
dy = be.empty_act(N, S, S, S, 32); dy.buf.normal_(generator=g)
w = (torch.randn(32, 4, 3, 3, 3, device="cuda", generator=g) * 0.1).contiguous()
dw = torch.empty_like(w)
gamma = torch.ones(4, device="cuda"); beta = torch.zeros(4, device="cuda")
mr, sc, sh = be.gn_stats(x, 4, 1e-5, gamma, beta)
wpd = be.pack_weight(w, 1)
dg, db = torch.empty(4, device="cuda"), torch.empty(4, device="cuda")
for _ in range(10):
    be.c4_bwd(x, dy, wpd, dw, 4, gamma, mr, sc, sh, dg, db)
torch.cuda.synchronize()
