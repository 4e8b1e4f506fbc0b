"""TEST INFRASTRUCTURE (oracle) -- conditioning probe for whole-network gradient parity.

Freshly initialised U-Nets with Instance/GroupNorm and a Dice loss have parameter gradients that are badly conditioned
with respect to rounding: a normalisation layer removes every per-channel constant, so the gradient that survives is a
small difference of large cancelling sums. Injecting unbiased relative noise of 1e-7 (one fp32 ulp) into every convolution
output and output-gradient of the CPU fp32 oracle moves some DynUNet gradients by 1e-2 .. 1e-1 (and the reference's own
fp32 CPU path is itself that far from its fp64 evaluation on those tensors), so no two fp32 implementations can agree to 1e-3
there. `noise_floor` measures that response per parameter; parity tests then require

    err(kernel grad, fp32 oracle grad) <= tol   OR   err(kernel grad, fp64 oracle grad) <= max(tol, k * noise_floor)

(tests/op_cases.py: grad_parity; logits and loss always at tol): well-conditioned tensors must meet the north-star tolerance,
ill-conditioned ones must be no worse than what fp32-roundoff-sized perturbations of the reference's own arithmetic produce.
"""
import torch
import torch.nn.functional as F


class _Noisy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, rel, gen):
        ctx.rel, ctx.gen = rel, gen
        return t * (1 + rel * torch.randn(t.shape, generator=gen, dtype=t.dtype))

    @staticmethod
    def backward(ctx, g):
        return g * (1 + ctx.rel * torch.randn(g.shape, generator=ctx.gen, dtype=g.dtype)), None, None


def _patched(module, rel, gen):
    """Context manager: module.F.conv3d / conv_transpose3d outputs (and their incoming gradients) get relative noise."""
    class _Ctx:
        def __enter__(self):
            self.f = module.F
            orig_c, orig_t = F.conv3d, F.conv_transpose3d

            class _F:
                def __getattr__(self, n):
                    return getattr(F, n)

                @staticmethod
                def conv3d(*a, **k):
                    return _Noisy.apply(orig_c(*a, **k), rel, gen)

                @staticmethod
                def conv_transpose3d(*a, **k):
                    return _Noisy.apply(orig_t(*a, **k), rel, gen)
            module.F = _F()

        def __exit__(self, *exc):
            module.F = self.f
    return _Ctx()


def noise_floor(module, run, rel=1e-7, seeds=(1, 2)):
    """run() -> dict name -> gradient, evaluating the fp32 oracle graph of `module` (oracle.unet3d_ref / oracle.dynunet_ref).
    Returns dict name -> max-norm relative spread of the gradient between evaluations with independent `rel` noise."""
    outs = []
    for s in seeds:
        gen = torch.Generator().manual_seed(s)
        with _patched(module, rel, gen):
            outs.append(run())
    a, b = outs
    return {k: float((a[k].double() - b[k].double()).abs().max() / max(float(b[k].double().abs().max()), 1e-30)) for k in a}
