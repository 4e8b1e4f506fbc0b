"""Error behaviour of the C ABI (include/mi355_unet3d.h: negative MI355_STATUS_* codes, no exceptions, no partial launches) and of
the ctypes layer above it (RuntimeError naming the op). Runs on the emulated library: argument validation is host code shared
with the HIP build."""
import ctypes
import importlib

import pytest
import torch

ops = importlib.import_module("3dunetcnn_amd.ops")
_lib = importlib.import_module("3dunetcnn_amd._lib")
EINVAL, EUNSUPPORTED, EWORKSPACE = -1, -2, -4


def _act(be, n, d, h, w, c, ld=None):
    return be.zeros_act(n, d, h, w, c, ld)


def _desc(**kw):
    d = _lib.MiConvDesc()
    d.kd, d.stride, d.pad = 3, 1, 1
    for k, v in kw.items():
        setattr(d, k, v)
    return d


def test_conv_fwd_rejects_bad_arguments(emu_backend):
    be, lib = emu_backend, emu_backend.lib
    x, y = _act(be, 1, 4, 4, 8, 8), _act(be, 1, 4, 4, 8, 32)
    w = torch.zeros(32, 8, 3, 3, 3)
    wp = be.pack_weight(w, 0).f32()
    xd, yd = x.desc(), y.desc()
    ok = _desc(out_d=4, out_h=4, out_w=8)
    assert lib.mi355_conv3d_fwd(ctypes.byref(xd), wp.data_ptr(), ctypes.byref(yd), ctypes.byref(ok), 0) == 0
    assert lib.mi355_conv3d_fwd(None, wp.data_ptr(), ctypes.byref(yd), ctypes.byref(ok), 0) == EINVAL           # null tensor
    assert lib.mi355_conv3d_fwd(ctypes.byref(xd), None, ctypes.byref(yd), ctypes.byref(ok), 0) == EINVAL        # null weights
    bad = _desc(out_d=4, out_h=4, out_w=8, kd=5)
    assert lib.mi355_conv3d_fwd(ctypes.byref(xd), wp.data_ptr(), ctypes.byref(yd), ctypes.byref(bad), 0) == EUNSUPPORTED
    bad = _desc(out_d=4, out_h=4, out_w=8, in_mode=1)                                                           # affine without scale
    assert lib.mi355_conv3d_fwd(ctypes.byref(xd), wp.data_ptr(), ctypes.byref(yd), ctypes.byref(bad), 0) == EINVAL
    bad = _desc(out_d=4, out_h=4, out_w=8, precision=9)
    assert lib.mi355_conv3d_fwd(ctypes.byref(xd), wp.data_ptr(), ctypes.byref(yd), ctypes.byref(bad), 0) == EINVAL
    bad = _desc(out_d=4, out_h=4, out_w=8, wformat=1)                                                           # raw-weight form needs 4 channels
    assert lib.mi355_conv3d_fwd(ctypes.byref(xd), wp.data_ptr(), ctypes.byref(yd), ctypes.byref(bad), 0) == EUNSUPPORTED
    x6 = _lib.MiAct(x.ptr(), 1, 4, 4, 8, 6, 8)                                                                   # channels not a float4 multiple
    assert lib.mi355_conv3d_fwd(ctypes.byref(x6), wp.data_ptr(), ctypes.byref(yd), ctypes.byref(ok), 0) == EINVAL


def test_wgrad_workspace_is_checked(emu_backend):
    be, lib = emu_backend, emu_backend.lib
    x, dy = _act(be, 1, 4, 4, 8, 32), _act(be, 1, 4, 4, 8, 32)
    dw = torch.zeros(32, 32, 3, 3, 3)
    xd, dyd, d = x.desc(), dy.desc(), _desc(out_d=4, out_h=4, out_w=8)
    need = lib.mi355_conv3d_wgrad_workspace(ctypes.byref(xd), ctypes.byref(dyd), ctypes.byref(d))
    assert need > 0
    ws = torch.zeros(need // 4)
    assert lib.mi355_conv3d_wgrad(ctypes.byref(xd), ctypes.byref(dyd), dw.data_ptr(), ctypes.byref(d), ws.data_ptr(), need, 0) == 0
    assert lib.mi355_conv3d_wgrad(ctypes.byref(xd), ctypes.byref(dyd), dw.data_ptr(), ctypes.byref(d), ws.data_ptr(), need - 4, 0) == EWORKSPACE
    d5 = _desc(out_d=4, out_h=4, out_w=8, kd=5)
    assert lib.mi355_conv3d_wgrad_workspace(ctypes.byref(xd), ctypes.byref(dyd), ctypes.byref(d5)) == 0        # unsupported -> 0 bytes
    assert lib.mi355_conv3d_wgrad(ctypes.byref(xd), ctypes.byref(dyd), dw.data_ptr(), ctypes.byref(d5), ws.data_ptr(), need, 0) == EUNSUPPORTED


def test_norm_loss_and_pointwise_reject_bad_arguments(emu_backend):
    be, lib = emu_backend, emu_backend.lib
    x = _act(be, 1, 4, 4, 4, 32)
    xd = x.desc()
    f = torch.zeros(4096)
    assert lib.mi355_gn_stats(ctypes.byref(xd), 5, 1e-5, None, None, f.data_ptr(), f.data_ptr(), f.data_ptr(), f.data_ptr(), 16384, 0) == EINVAL  # 32 % 5
    assert lib.mi355_gn_stats(ctypes.byref(xd), 8, 1e-5, None, None, f.data_ptr(), f.data_ptr(), f.data_ptr(), f.data_ptr(), 8, 0) == EWORKSPACE
    z = torch.zeros(1, 3, 4, 4, 4)
    assert lib.mi355_dice_fwd_bwd(z.data_ptr(), z.data_ptr(), 0, 1, 3, 64, 1, 0, 0, 7, 1, 1e-5, 1e-5, f.data_ptr(), None, 1.0, f.data_ptr(), 16384, 0) == EINVAL
    assert lib.mi355_dice_fwd_bwd(z.data_ptr(), z.data_ptr(), 0, 1, 1, 64, 1, 0, 0, 0, 0, 1e-5, 1e-5, f.data_ptr(), None, 1.0, f.data_ptr(), 16384, 0) == EINVAL
    MiDiceOpts = importlib.import_module("3dunetcnn_amd._lib").MiDiceOpts
    def opts(**kw):
        base = dict(activation=1, target_kind=0, batch=0, squared_pred=0, include_background=1, jaccard=0, reduction=0, smooth_nr=1e-5,
                    smooth_dr=1e-5, class_weight=None)
        base.update(kw)
        return ctypes.byref(MiDiceOpts(**base))
    ex_f, ex_b = lib.mi355_dice_ex_forward, lib.mi355_dice_ex_backward
    assert ex_f(opts(), z.data_ptr(), z.data_ptr(), 1, 3, 64, f.data_ptr(), f.data_ptr(), 16384, 0) == 0
    assert ex_f(opts(activation=3), z.data_ptr(), z.data_ptr(), 1, 3, 64, f.data_ptr(), f.data_ptr(), 16384, 0) == EINVAL
    assert ex_f(opts(target_kind=5), z.data_ptr(), z.data_ptr(), 1, 3, 64, f.data_ptr(), f.data_ptr(), 16384, 0) == EINVAL
    assert ex_f(opts(reduction=3), z.data_ptr(), z.data_ptr(), 1, 3, 64, f.data_ptr(), f.data_ptr(), 16384, 0) == EINVAL
    assert ex_f(opts(include_background=0), z.data_ptr(), z.data_ptr(), 1, 1, 64, f.data_ptr(), f.data_ptr(), 16384, 0) == EINVAL
    assert ex_f(opts(), z.data_ptr(), z.data_ptr(), 1, 17, 64, f.data_ptr(), f.data_ptr(), 16384, 0) == EUNSUPPORTED      # more than 16 classes
    assert ex_f(opts(), z.data_ptr(), z.data_ptr(), 1, 3, 64, f.data_ptr(), f.data_ptr(), 16, 0) == EWORKSPACE
    assert ex_b(opts(reduction=2), z.data_ptr(), z.data_ptr(), 1, 3, 64, f.data_ptr(), 2, z.data_ptr(), f.data_ptr(), 0) == EINVAL   # 3 terms, 2 upstream values
    assert lib.mi355_ce_fwd_bwd(z.data_ptr(), z.data_ptr(), 0, 1, 17, 64, 0, 1.0, f.data_ptr(), 0, None, 0, 1.0, f.data_ptr(), 16384, 0) == EINVAL
    assert lib.mi355_ce_fwd_bwd(z.data_ptr(), z.data_ptr(), 0, 1, 3, 64, 0, 1.0, f.data_ptr(), 0, None, 0, 1.0, f.data_ptr(), 16, 0) == EWORKSPACE
    m = (ctypes.c_float * 12)(*([0.0] * 12))
    assert lib.mi355_resample_affine(z.data_ptr(), f.data_ptr(), 3, 4, 4, 4, 2, 2, 2, m, 5, 0, 0) == EINVAL
    assert lib.mi355_postprocess(z.data_ptr(), 3, 64, 3, 0.5, None, 0, 0, f.data_ptr(), None, 0) == EINVAL


def test_python_layer_raises_runtime_error(emu_backend):
    be = emu_backend
    x, y = _act(be, 1, 4, 4, 8, 8), _act(be, 1, 4, 4, 8, 32)
    wp = be.pack_weight(torch.zeros(32, 8, 3, 3, 3), 0)
    with pytest.raises(RuntimeError, match="conv3d_fwd failed"):
        be.conv_fwd(x, wp, y, 5)
    with pytest.raises(RuntimeError, match="unsupported"):
        be.conv_wgrad(x, y, torch.zeros(32, 8, 5, 5, 5), 5)
