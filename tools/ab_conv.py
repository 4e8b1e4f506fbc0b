"""Developer tool (GPU): interleaved A/B timing of the fp32 conv forward between two builds of the library.
    python tools/ab_conv.py tools/libold_r0.so 3dunetcnn_amd/libmi355unet3d.so"""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib_mod = importlib.import_module("3dunetcnn_amd._lib"); ops = importlib.import_module("3dunetcnn_amd.ops")


def loose(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in lib_mod.SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            continue
        fn.restype = res; fn.argtypes = args
    if not hasattr(lib, "mi355_conv3d_uses_bf16"):
        lib.mi355_conv3d_uses_bf16 = lambda d: 0
    return lib


bes = [ops.Backend(lib=loose(p)) for p in sys.argv[1:3]]
n = 2
for cin, cout, s, mode in ((32, 32, 128, 1), (32, 32, 128, 0), (64, 32, 128, 1), (64, 64, 64, 1), (128, 128, 64, 1), (256, 256, 32, 1)):
    res = [[], []]
    setups = []
    for be in bes:
        x = be.empty_act(n, s, s, s, cin); x.buf.normal_()
        y = be.empty_act(n, s, s, s, cout)
        w = torch.randn(cout, cin, 3, 3, 3, device=be.device) * 0.05
        wp = be.pack_weight(w, 0)
        sc = torch.ones(n, cin, device=be.device); sh = torch.zeros(n, cin, device=be.device)
        kw = dict(in_mode=ops.IN_AFFINE_ACT, scale=sc, shift=sh) if mode else {}
        setups.append((be, x, wp, y, kw))
    for rnd in range(4):
        for i, (be, x, wp, y, kw) in enumerate(setups):
            for _ in range(2): be.conv_fwd(x, wp, y, 3, 1, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): be.conv_fwd(x, wp, y, 3, 1, **kw)
            e1.record(); torch.cuda.synchronize()
            res[i].append(e0.elapsed_time(e1) / 5)
    fl = 2.0 * n * s ** 3 * cin * cout * 27
    print(f"{cin}->{cout} @{s}^3 mode{mode}: A min {min(res[0]):.3f} ms ({fl / min(res[0]) / 1e9:.1f} TF/s)  B min {min(res[1]):.3f} ms ({fl / min(res[1]) / 1e9:.1f} TF/s)", flush=True)
