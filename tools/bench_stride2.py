"""Developer tool (GPU): the three down-sampling convolutions of UNet3D 128^3 batch 2 (3x3x3 stride 2, channels kept: 32 @128^3 -> 64^3,
64 @64^3 -> 32^3, 128 @32^3 -> 16^3): forward (with the moments epilogue, as the network runs it), zero-insert dgrad, weight gradient; ms per
launch and the result's error against a CPU reference on a crop.   [MI355_S2_KC16=1] python tools/bench_stride2.py"""
import importlib, os, sys, time
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("3dunetcnn_amd.ops")
be = ops.default_backend()
g = torch.Generator(device="cuda").manual_seed(0)
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
tot = [0.0, 0.0, 0.0]
for c, s in ((32, 128), (64, 64), (128, 32)):
    x = be.empty_act(2, s, s, s, c); x.buf.normal_(generator=g)
    y = be.empty_act(2, s // 2, s // 2, s // 2, c)
    dy = be.empty_act(2, s // 2, s // 2, s // 2, c); dy.buf.normal_(generator=g)
    dx = be.empty_act(2, s, s, s, c)
    w = (torch.randn(c, c, 3, 3, 3, device="cuda", generator=g) * 0.05).contiguous()
    dw = torch.empty_like(w)
    wp, wpd = be.pack_weight(w, 0), be.pack_weight(w, 1)
    t_f = timeit(lambda: be.conv_fwd(x, wp, y, 3, 2, moments=True))
    t_d = timeit(lambda: be.conv_fwd(dy, wpd, dx, 3, 1, in_mode=ops.IN_ZERO_INSERT, pad=1))
    t_w = timeit(lambda: be.conv_wgrad(x, dy, dw, 3, 2))
    # correctness of the forward on a corner crop
    xc = x.tensor()[:1, :9, :9, :9].permute(0, 4, 1, 2, 3).double().cpu()
    ref = F.conv3d(F.pad(xc, (1, 0, 1, 0, 1, 0)), w.double().cpu(), stride=2)[..., :4, :4, :4]
    got = y.tensor()[:1, :4, :4, :4].permute(0, 4, 1, 2, 3).double().cpu()
    err = float((got - ref).abs().max() / ref.abs().max())
    print(f"{c}->{c} @{s}^3 s2: fwd+moments {t_f:.3f} ms  zero-insert dgrad {t_d:.3f} ms  wgrad {t_w:.3f} ms   fwd err {err:.1e}", flush=True)
    tot[0] += t_f; tot[1] += t_d; tot[2] += t_w
print(f"sum: fwd {tot[0]:.3f}  dgrad {tot[1]:.3f}  wgrad {tot[2]:.3f}  = {sum(tot):.3f} ms per step")
