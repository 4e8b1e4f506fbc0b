"""Developer tool (GPU): after a few training steps, where does the flat gradient of the fused-statistics run differ from the plain run?"""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
unet = importlib.import_module("3dunetcnn_amd.unet"); losses = importlib.import_module("3dunetcnn_amd.losses")
optim = importlib.import_module("3dunetcnn_amd.optim"); R = importlib.import_module("3dunetcnn_amd.synthetic")
ops = importlib.import_module("3dunetcnn_amd.ops")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
x, y = R.synthetic_case(2, 4, (S, S, S)); x, y = x.cuda(), y.cuda()
be = ops.default_backend()
res = {}
for fused in (False, True):
    be.fused_stats = fused
    torch.manual_seed(0)
    m = unet.HipUNet3D(n_features=4, n_outputs=3).cuda().train()
    crit = losses.HipDiceLoss(sigmoid=True); opt = optim.HipAdam(m.parameters(), lr=1e-3)
    ls = []
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        l = crit(m(x), y); l.backward(); opt.step(); ls.append(float(l))
    m.eval(); opt.zero_grad(set_to_none=True); l = crit(m(x), y); l.backward(); torch.cuda.synchronize()
    res[fused] = (m.flat_grad().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}, ls, float(l), m._offsets, [p.numel() for p in m.parameters()])
    print("fused", fused, "losses", ["%.5f" % v for v in ls], "eval loss %.6f" % float(l), "flat grad max %.3e" % float(res[fused][0].abs().max()), flush=True)
g0, g1 = res[False][0], res[True][0]
d = (g0 - g1).abs()
i = int(d.argmax())
print("max |diff| %.3e at flat index %d (values %.3e vs %.3e); max|g0| %.3e" % (float(d.max()), i, float(g0[i]), float(g1[i]), float(g0.abs().max())))
offs, nums = res[True][4], res[True][5]
names = [k for k, _ in unet.HipUNet3D(n_features=4, n_outputs=3).named_parameters()]
for k, (o, n) in enumerate(zip(offs, nums)):
    if o <= i < o + n: print("  inside parameter", names[k], "offset", o, "numel", n, "element", i - o)
    if o + n <= i < (offs[k + 1] if k + 1 < len(offs) else 10**12): print("  in the ALIGNMENT GAP after", names[k])
for k in res[True][1]:
    a, b = res[False][1][k], res[True][1][k]
    e = float((a - b).abs().max() / a.abs().max().clamp_min(1e-30))
    if e > 0.5: print("  param", k, "rel diff %.2e  max|plain| %.2e max|fused| %.2e" % (e, float(a.abs().max()), float(b.abs().max())))
