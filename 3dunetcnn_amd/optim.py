"""HipAdam: torch.optim.Adam semantics (lr, betas, eps, weight_decay as L2, no amsgrad) on the fused HIP kernel.

The reference builds its optimizer with getattr(torch.optim, name)(model.parameters(), **kwargs)
(unet3d/scripts/script_utils.py:80-81; brats2020_config.json:108-111 -> Adam lr=1e-3) and calls
optimizer.zero_grad() / optimizer.step() at unet3d/train/training_utils.py:59,72.

When every parameter (and its .grad) is a slice of one flat buffer -- which is how HipUNet3D stores them -- a step is a
single kernel over the whole buffer (28 B/param of HBM traffic); otherwise one launch per tensor. `grad_scale`
(1/world_size) lets the RCCL gradient SUM be averaged for free.
"""
import torch

from . import ops as _ops


def _contiguous_run(tensors):
    """(base_tensor_ptr, total_elems) if the tensors tile one address range in order (4-element aligned gaps allowed)."""
    ptr = tensors[0].data_ptr()
    base = ptr
    for t in tensors:
        if t.data_ptr() != ptr or not t.is_contiguous():
            return None
        ptr += (t.numel() + 3) // 4 * 4 * 4
    return base, (ptr - base) // 4


class HipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("HipAdam: amsgrad is not implemented")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.grad_scale = 1.0
        self._be = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            be = self._be or _ops.default_backend(ps[0].device)
            # state of the fused path lives under a string key (kept verbatim by Optimizer.state_dict / load_state_dict), indexed
            # by the group's position so that a checkpoint restores into a freshly built optimizer
            st = self.state.setdefault("_flat_group%d" % gi, {})
            st["step"] = st.get("step", 0) + 1
            b1, b2 = group["betas"]
            args = (float(group["lr"]), b1, b2, group["eps"], group["weight_decay"], st["step"], self.grad_scale)
            prun = _contiguous_run([p.data for p in ps]) if len(ps) == len(group["params"]) else None
            grun = _contiguous_run([p.grad for p in ps]) if prun else None
            if prun and grun and prun[1] == grun[1]:
                n = prun[1]
                if "flat_m" in st and st["flat_m"].device != ps[0].device:
                    st["flat_m"], st["flat_v"] = st["flat_m"].to(ps[0].device), st["flat_v"].to(ps[0].device)
                if "flat_m" not in st or st["flat_m"].numel() != n:
                    st["flat_m"] = torch.zeros(n, dtype=torch.float32, device=ps[0].device)
                    st["flat_v"] = torch.zeros(n, dtype=torch.float32, device=ps[0].device)
                # views spanning the whole run (storage is one flat tensor, so as_strided from the first slice is valid)
                pflat = ps[0].data.as_strided((n,), (1,))
                gflat = ps[0].grad.as_strided((n,), (1,))
                be.adam_step(pflat, gflat, st["flat_m"], st["flat_v"], *args)
            else:
                for p in ps:
                    s = self.state[p]
                    if "m" not in s:
                        s["m"] = torch.zeros_like(p.data, memory_format=torch.contiguous_format)
                        s["v"] = torch.zeros_like(p.data, memory_format=torch.contiguous_format)
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    if (p.data_ptr() | g.data_ptr()) & 15 or not p.data.is_contiguous():
                        raise RuntimeError("HipAdam needs 16-byte aligned contiguous parameters")
                    be.adam_step(p.data.view(-1), g.view(-1), s["m"].view(-1), s["v"].view(-1), *args)
            for p in ps:
                torch.autograd.graph.increment_version(p)   # raw-pointer update: keep torch's version counter honest
        return loss
