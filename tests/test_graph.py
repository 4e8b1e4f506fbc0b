"""HipGraphedTrainStep (3dunetcnn_amd/graph.py): the step replayed as one HIP graph must be the eager step, bit for bit."""
import importlib

import pytest
import torch

from oracle import unet3d_ref as R          # synthetic inputs only

graph = importlib.import_module("3dunetcnn_amd.graph")
unet = importlib.import_module("3dunetcnn_amd.unet")
dynunet = importlib.import_module("3dunetcnn_amd.dynunet")
losses = importlib.import_module("3dunetcnn_amd.losses")
optim = importlib.import_module("3dunetcnn_amd.optim")


def test_graphed_step_refuses_cpu_tensors():
    m = unet.HipUNet3D(n_features=4, n_outputs=3, base_width=8, encoder_blocks=[1, 1])
    x, y = R.synthetic_case(1, 4, (16, 16, 16), 3)
    with pytest.raises(RuntimeError, match="MI355X"):
        graph.HipGraphedTrainStep(m, losses.HipDiceLoss(sigmoid=True), optim.HipAdam(m.parameters()), x, y)


def _build(kind):
    torch.manual_seed(3)
    if kind == "unet3d":
        return unet.HipUNet3D(n_features=4, n_outputs=3, base_width=16, encoder_blocks=[1, 2, 2]).cuda().eval()
    return dynunet.HipDynUNet(spatial_dims=3, in_channels=4, out_channels=3, kernel_size=[3] * 4, strides=[1, 2, 2, 2],
                              upsample_kernel_size=[2] * 3, filters=[32, 64, 96, 128]).cuda().eval()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["unet3d", "dynunet"])
def test_graphed_step_equals_eager_step(hip_backend, kind):
    batches = [tuple(t.cuda() for t in R.synthetic_case(2, 4, (32, 32, 32), 3, seed=s)) for s in range(3)]
    # eager reference run (eval(): no Dropout3d mask, so both runs see the same arithmetic)
    m0 = _build(kind)
    crit0, opt0 = losses.HipDiceLoss(sigmoid=True), optim.HipAdam(m0.parameters(), lr=1e-3)
    want = []
    for x, y in batches:
        opt0.zero_grad(set_to_none=True)
        loss = crit0(m0(x), y)
        loss.backward()
        opt0.step()
        want.append(float(loss.detach()))
    # graphed run from the same initial weights
    m1 = _build(kind)
    for (k0, v0), (k1, v1) in zip(m0.state_dict().items(), m1.state_dict().items()):
        assert k0 == k1
    crit1, opt1 = losses.HipDiceLoss(sigmoid=True), optim.HipAdam(m1.parameters(), lr=1e-3)
    before = {k: v.detach().clone() for k, v in m1.state_dict().items()}
    step = graph.HipGraphedTrainStep(m1, crit1, opt1, *batches[0])
    for k, v in m1.state_dict().items():
        assert torch.equal(v, before[k]), f"building the graph changed {k}"
    got = []
    for x, y in batches:
        opt1.zero_grad(set_to_none=True)        # the reference's loop calls it; the graphed step must survive it
        got.append(float(step(x, y).item()))
    assert got == want, (got, want)
    for (k, a), (_, b) in zip(m0.state_dict().items(), m1.state_dict().items()):
        assert torch.equal(a, b), k


@pytest.mark.gpu
def test_graphed_step_train_mode_dropout_draws_a_new_mask_per_replay(hip_backend):
    torch.manual_seed(0)
    m = unet.HipUNet3D(n_features=4, n_outputs=3, base_width=16, encoder_blocks=[1, 1]).cuda().train()
    x, y = (t.cuda() for t in R.synthetic_case(1, 4, (16, 16, 16), 3))
    opt = optim.HipAdam(m.parameters(), lr=0.0)              # lr 0: the weights stay put, only the Dropout3d mask varies
    step = graph.HipGraphedTrainStep(m, losses.HipDiceLoss(sigmoid=True), opt, x, y)
    outs = []
    for _ in range(4):
        step(x, y)
        outs.append(step.logits.clone())
    assert any(not torch.equal(outs[0], o) for o in outs[1:]), "every replay reused one Dropout3d mask"


@pytest.mark.gpu
def test_graphed_step_accepts_ddp_model(hip_backend):
    """A model with a GradientBucketReducer attached is captured WITHOUT the reducer's per-bucket callbacks (a replay cannot call back into
    Python) and exchanges the flat gradient buffer after the replay (graph.py; world size 1 here: the exchange is a no-op, the
    world-size-2 form is tests/test_ddp_gloo.py). The reducer must be attached again afterwards: an eager backward still uses it."""
    ddp = importlib.import_module("3dunetcnn_amd.ddp")
    m = unet.HipUNet3D(n_features=4, n_outputs=3, base_width=8, encoder_blocks=[1, 1]).cuda().eval()
    m.flatten_parameters()
    red = ddp.GradientBucketReducer(m)
    x, y = (t.cuda() for t in R.synthetic_case(1, 4, (16, 16, 16), 3))
    crit, opt = losses.HipDiceLoss(sigmoid=True), optim.HipAdam(m.parameters(), lr=1e-3)
    before = m._flat.detach().clone()
    step = graph.HipGraphedTrainStep(m, crit, opt, x, y)
    assert step.reducer is red
    assert m.grad_ready_callback is not None and m.grad_sync_callback is not None
    assert torch.equal(m._flat, before)                     # construction does not step
    l0 = float(step(x, y))
    l1 = float(step(x, y))
    assert l0 > 0 and l1 != l0 and not torch.equal(m._flat, before)
    opt.zero_grad(set_to_none=True)
    crit(m(x), y).backward()                                # eager backward through the re-attached reducer
    assert red.n_launched == 0                              # (world size 1: no collective is issued)


def _tiny(be, seed):
    torch.manual_seed(seed)
    m = unet.HipUNet3D(n_features=4, n_outputs=3, base_width=8, encoder_blocks=[1, 1]).eval()
    m._be = be
    crit = losses.HipDiceLoss(sigmoid=True)
    crit._be = be
    opt = optim.HipAdam(m.parameters(), lr=1e-3)
    opt._be = be
    return m, crit, opt


def test_uncaptured_graphed_step_with_16bit_activation_storage(emu_backend):
    """HipGraphedTrainStep's host logic (capture=False on the CPU emulator) around a HipAutocastUNet that stores activations as bf16: the
    stepper's step equals the eager step bit for bit (same kernels, same order), storage type restored on the backend afterwards."""
    x, y = R.synthetic_case(1, 4, (16, 16, 16), 3, seed=0)
    res = []
    for stepper_kind in ("eager", "graphed"):
        torch.manual_seed(3)
        m = unet.HipAutocastUNet(n_features=4, n_outputs=3, base_width=8, encoder_blocks=[1, 1]).eval()
        crit, opt = losses.HipDiceLoss(sigmoid=True), optim.HipAdam(m.parameters(), lr=1e-3)
        m._be = crit._be = opt._be = emu_backend
        if stepper_kind == "eager":
            opt.zero_grad(set_to_none=True)
            loss = crit(m(x), y)
            loss.backward()
            opt.step()
        else:
            step = graph.HipGraphedTrainStep(m, crit, opt, x, y, capture=False)
            loss = step(x, y)
        res.append((float(loss.detach()), {k: v.detach().clone() for k, v in m.state_dict().items()}))
        assert emu_backend.act_dtype == torch.float32 and emu_backend.precision == 0
    assert res[0][0] == res[1][0]
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k


def test_pack_tables_are_per_network(emu_backend):
    """Two networks on ONE backend (training model + validation / EMA twin): each keeps the device task table of its one-launch weight
    repack across the other's steps -- the table a captured graph has baked in is never replaced or freed (round-3 advisor finding:
    the cache was a single slot on the shared Backend)."""
    be = emu_backend
    x, y = R.synthetic_case(1, 4, (16, 16, 16), 3)
    a, ca, oa = _tiny(be, 1)
    b, cb, ob = _tiny(be, 2)

    def step(m, c, o):
        o.zero_grad(set_to_none=True)
        c(m(x), y).backward()
        o.step()

    step(a, ca, oa)
    step(a, ca, oa)                       # second forward of A: its packs are refreshed through A's table
    assert len(a._pack_tables) == 1
    ta = next(iter(a._pack_tables.values()))
    assert be.last_pack_table is ta
    step(b, cb, ob)
    step(b, cb, ob)
    tb = next(iter(b._pack_tables.values()))
    assert tb is not ta and be.last_pack_table is tb
    for _ in range(2):                    # alternate: no rebuild, no eviction
        step(a, ca, oa)
        assert be.last_pack_table is ta and len(a._pack_tables) == 1
        step(b, cb, ob)
        assert be.last_pack_table is tb and len(b._pack_tables) == 1
    assert "_pack_table" not in vars(be) and "_pack_table_key" not in vars(be)       # the old single slot is gone
    # a pinned table (what HipGraphedTrainStep marks) survives any number of routing changes of its owner
    ta._mi355_pinned = True
    for k in range(12):
        a._pack_tables[("fake", k)] = torch.zeros(1)
        if len(a._pack_tables) > 8:
            for key in list(a._pack_tables):
                if not getattr(a._pack_tables[key], "_mi355_pinned", False):
                    del a._pack_tables[key]
                    break
    assert any(v is ta for v in a._pack_tables.values())


@pytest.mark.gpu
def test_graph_survives_a_second_model_on_the_same_device(hip_backend):
    """The captured graph replays correctly after ANOTHER model on the device has run forwards and optimizer steps in between (its
    repack must not free or overwrite the task table whose address the graph holds)."""
    batches = [tuple(t.cuda() for t in R.synthetic_case(2, 4, (32, 32, 32), 3, seed=s)) for s in range(3)]
    m0 = _build("unet3d")
    crit0, opt0 = losses.HipDiceLoss(sigmoid=True), optim.HipAdam(m0.parameters(), lr=1e-3)
    want = []
    for x, y in batches:
        opt0.zero_grad(set_to_none=True)
        loss = crit0(m0(x), y)
        loss.backward()
        opt0.step()
        want.append(float(loss.detach()))
    m1 = _build("unet3d")
    crit1, opt1 = losses.HipDiceLoss(sigmoid=True), optim.HipAdam(m1.parameters(), lr=1e-3)
    step = graph.HipGraphedTrainStep(m1, crit1, opt1, *batches[0])
    assert step._pack_table is not None and step._pack_table._mi355_pinned
    torch.manual_seed(9)
    other = unet.HipUNet3D(n_features=4, n_outputs=3, base_width=8, encoder_blocks=[1, 1, 1]).cuda().eval()
    co, oo = losses.HipDiceLoss(sigmoid=True), optim.HipAdam(other.parameters(), lr=1e-3)
    got = []
    for x, y in batches:
        for _ in range(2):                # the twin trains in between: its own repack table, fresh allocations
            oo.zero_grad(set_to_none=True)
            co(other(x), y).backward()
            oo.step()
        torch.cuda.empty_cache()
        got.append(float(step(x, y).item()))
    assert got == want, (got, want)
