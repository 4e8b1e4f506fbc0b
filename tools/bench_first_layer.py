"""Developer tool (GPU): the north-star layer -- 3x3x3 conv, 4 input channels -> 32, 128^3, batch 2 -- forward / wgrad / dgrad in every
precision mode: ms per launch and ALGORITHMIC HBM rate (4 * (N*Cin*V + N*Cout*V + 27*Cin*Cout) bytes per pass = 604 MB)
against the 8 TB/s roofline.   python tools/bench_first_layer.py [size] [batch]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("3dunetcnn_amd.ops")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2
be = ops.default_backend()
g = torch.Generator(device="cuda").manual_seed(0)
x = be.empty_act(N, S, S, S, 4); x.buf.normal_(generator=g)
y = be.empty_act(N, S, S, S, 32)
dy = be.empty_act(N, S, S, S, 32); dy.buf.normal_(generator=g)
dx = be.empty_act(N, S, S, S, 4)
w = (torch.randn(32, 4, 3, 3, 3, device="cuda", generator=g) * 0.1).contiguous()
dw = torch.empty_like(w)
gamma = torch.ones(4, device="cuda"); beta = torch.zeros(4, device="cuda")
mr, sc, sh = be.gn_stats(x, 4, 1e-5, gamma, beta)
kw = dict(in_mode=ops.IN_AFFINE_ACT, scale=sc, shift=sh)
V = N * S ** 3
alg = 4.0 * (4 * V + 32 * V + 27 * 4 * 32)
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for mode in ("fp32", "bf16x6", "bf16x3", "bf16"):
    be.set_precision(mode)
    wp = be.pack_weight(w, 0); wpd = be.pack_weight(w, 1)
    t_f = timeit(lambda: be.conv_fwd(x, wp, y, 3, 1, **kw))
    t_fm = timeit(lambda: be.conv_fwd(x, wp, y, 3, 1, moments=True, **kw))
    t_w = timeit(lambda: be.conv_wgrad(x, dy, dw, 3, 1, **kw))
    t_d = timeit(lambda: be.conv_fwd(dy, wpd, dx, 3, 1))
    tot = t_f + t_w + t_d
    print(f"{mode:7s} fwd {t_f*1e3:.3f} ms ({alg/t_f/1e12:.2f} TB/s = {alg/t_f/8e12*100:.0f} %)  fwd+moments {t_fm*1e3:.3f} ms  wgrad {t_w*1e3:.3f} ms ({alg/t_w/1e12:.2f} TB/s)  "
          f"dgrad {t_d*1e3:.3f} ms ({alg/t_d/1e12:.2f} TB/s)  | fwd+bwd {tot*1e3:.3f} ms = {3*alg/tot/1e12:.2f} TB/s = {3*alg/tot/8e12*100:.0f} % of 8 TB/s", flush=True)
be.set_precision("fp32")
# round 6: the fused backward (csrc/conv3d_c4_bwd.hip: weight gradient + norm-backward sums in one pass over dy, no data-gradient tensor)
wpd = be.pack_weight(w, 1)
dg, db = torch.empty(4, device="cuda"), torch.empty(4, device="cuda")
if be.c4_bwd_supported(x, dy, ops.IN_AFFINE_ACT, 0.0, sc, sh):
    t_b = timeit(lambda: be.c4_bwd(x, dy, wpd, dw, 4, gamma, mr, sc, sh, dg, db))
    wp = be.pack_weight(w, 0)
    t_fm = timeit(lambda: be.conv_fwd(x, wp, y, 3, 1, moments=True, **kw))
    print(f"fp32 fused backward (conv3d_c4_bwd + reduce + gn_bwd_params) {t_b*1e3:.3f} ms = {2*alg/t_b/1e12:.2f} TB/s of the two passes it replaces; "
          f"fwd+moments + fused bwd {1e3*(t_fm+t_b):.3f} ms = {3*alg/(t_fm+t_b)/8e12*100:.0f} % of 8 TB/s", flush=True)
# reference rates of the memory system for the same 537 MB output tensor: a pure write (fill) and a read+write (scale in place)
t_fill = timeit(lambda: y.buf.fill_(1.0))
t_rw = timeit(lambda: y.buf.mul_(1.0001))
t_copy = timeit(lambda: dy.buf.copy_(y.buf))
nb = y.buf.numel() * 4
print(f"memory system on the {nb/1e6:.0f} MB output: fill {t_fill*1e3:.3f} ms ({nb/t_fill/1e12:.2f} TB/s write), in-place scale {t_rw*1e3:.3f} ms ({2*nb/t_rw/1e12:.2f} TB/s r+w), "
      f"copy {t_copy*1e3:.3f} ms ({2*nb/t_copy/1e12:.2f} TB/s r+w)")
