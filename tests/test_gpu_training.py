"""GPU parity of the training path (through the C ABI): loss, gradients, AdamW and the full
train_NN trajectory against golden vectors captured from the live reference."""
import os
import shutil

import numpy as np
import pytest

import cases
from linna_amd import _lib
import synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def make_engine(name, steps_as_dataset=True):
    from linna_amd import nn, util, predictor_gpu, trainer
    p = cases.training_problem(name)
    cls = {"ChtoModelv2": nn.ChtoModelv2, "MLP": nn.MLP, "ChtoModelv2_linear": nn.ChtoModelv2_linear,
           "ChtoModelsimple": nn.ChtoModelsimple}[p["kind"]]
    model = cls(p["nin"], p["nout"], None, **p["kw"])
    model.load_state_dict(p["weights"])
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32))
    Xt = util.X_transform_class(t(p["X_mean"]), t(p["X_std"]), "cpu", None)
    Yt = util.Y_transform_class(t(p["y_mean"]), t(p["y_std"]), "cpu")
    pred = predictor_gpu.Predictor(p["nin"], p["nout"], model=model, X_transform=Xt, y_transform=Yt, device="cuda")
    ytd = util.Y_transform_data(p["sigma"], "cpu")
    yinv = util.Y_invtransform_class(t(p["y_mean"]), t(p["y_std"]), t(p["data"]), "cpu")
    lf = util.Loss_fn(t(p["data"]), torch.tensor(p["cov"], dtype=torch.float64),
                      torch.tensor(np.linalg.inv(p["cov"]), dtype=torch.float64), ytd, yinv, "cpu")
    B = p["X"].shape[1]
    X = p["X"].reshape(3 * B, -1)
    Y = p["Y"].reshape(3 * B, -1)
    loader = predictor_gpu.BatchLoader(util.ArrayDataset(X, Y), B, shuffle=False, drop_last=True)
    eng = trainer.TrainEngine(pred, loader, lf, loader, use_graph=False)
    return p, model, pred, eng, B


@pytest.mark.parametrize("name", [c[0] for c in cases.TRAIN])
def test_loss_gradients_and_adamw_match_reference(name):
    from linna_amd.predictor_gpu import _AdamWState
    g = cases.golden(name)
    p, model, pred, eng, B = make_engine(name)
    opt = _AdamWState(model, float(g["lr"]), weight_decay=1e-4)
    losses = []
    for s in range(3):
        rows = torch.arange(s * B, (s + 1) * B, dtype=torch.int32, device="cuda")
        eng.rows.copy_(rows)
        eng._forward_loss_backward()
        if s == 0:
            pred0 = eng.predb[:, :p["nout"]].cpu().numpy()
            np.testing.assert_allclose(pred0, g["pred0"], rtol=8e-5, atol=8e-6 * np.abs(g["pred0"]).max())
            np.testing.assert_allclose(eng.loss_rows.cpu().numpy(), g["loss_rows0"], rtol=3e-5)
            dp = eng.dpred[:, :p["nout"]].cpu().numpy()
            np.testing.assert_allclose(dp, g["dpred0"], rtol=5e-4, atol=5e-6 * np.abs(g["dpred0"]).max())
            for k, gk in model.grad_dict().items():
                ref = g["grad0/" + k]
                got = gk.cpu().numpy() if p["full"] else synth.tensor_digest(gk.cpu().numpy())
                np.testing.assert_allclose(got, ref, rtol=6e-4, atol=6e-6 * np.abs(ref).max() + 1e-9, err_msg=k)
        losses.append(float(eng.loss_mean.item()))
        opt.apply()
        for k, v in model.state_dict().items():
            ref = g["param%d/%s" % (s + 1, k)]
            got = v.cpu().numpy() if p["full"] else synth.tensor_digest(v.cpu().numpy())
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=3e-5 * np.abs(ref).max() + 1e-7, err_msg=k)
    np.testing.assert_allclose(losses, g["losses"], rtol=3e-6)
    # validation metric pieces on minibatch 0 (util.py:1124-1127), with the ORIGINAL weights
    model.load_state_dict(p["weights"])
    vm = eng.validate()
    assert vm.shape == (3,) and np.all(np.isfinite(vm))


def test_chi2_denominator_and_masks():
    from oracle import training
    name = "train_v2_5_3"
    g = cases.golden(name)
    p, model, pred, eng, B = make_engine(name)
    den = eng.den.cpu().numpy()[:B]
    np.testing.assert_allclose(den, g["chisqMd0"], rtol=3e-6)
    eng.rows.copy_(torch.arange(B, dtype=torch.int32, device="cuda"))
    eng._forward_loss_backward()
    dp = eng.dpred[:, :p["nout"]].cpu().numpy()
    assert dp[1, 0] == 0.0                 # masked sentinel entry (1e10) carries no gradient (util.py:1072-1084)


@pytest.mark.parametrize("name", ["train_v2_33_33", "train_v2_26_457", "train_v2_5_3", "train_mlp_7_5"])
def test_adamw_writing_the_weight_streams_equals_update_plus_relayout(name):
    """``linna_net_adamw_step`` (the update AND both weight streams of the next step in one launch) against
    ``linna_adamw_step`` followed by the lazy re-layouts: parameters, moments and every step's loss bit for bit over six
    steps (a wrong slot in either stream would show in the next step's loss or gradients), and the validation pass after
    them, which reads a stream of its own."""
    from linna_amd.predictor_gpu import _AdamWState
    res = []
    for fused in (True, False, "gemm"):
        p, model, pred, eng, B = make_engine(name)
        opt = _AdamWState(model, 2e-3)
        if fused is False:
            opt._streams = False
        if fused != "gemm":
            eng.one_update = False                   # "gemm": linna_net_train_step_update, AdamW in the gradient tiles' epilogue
        losses = []
        for s in range(6):
            out = torch.zeros(1, device="cuda")
            eng.step(opt, torch.arange((s % 3) * B, (s % 3 + 1) * B, dtype=torch.int32, device="cuda"), loss_out=out)
            losses.append(out)
        torch.cuda.synchronize()
        eng.validate()
        res.append((model._flat.cpu().numpy().copy(), opt.m.cpu().numpy().copy(), opt.v.cpu().numpy().copy(),
                    torch.cat(losses).cpu().numpy(), eng.val["loss_rows"].cpu().numpy().copy(), opt._streams, eng.one_update))
    assert np.isfinite(res[0][3]).all()
    for other in (res[1], res[2]):
        for a, b in zip(res[0][:5], other[:5]):
            np.testing.assert_array_equal(a, b)
    if name in ("train_v2_33_33", "train_v2_26_457"):
        assert res[0][5] is True                    # the reference's network trains through the one-launch update
        assert res[2][6] is True                    # ... and through the update in the gradient launch
        # ... with forward + loss + dX chain as ONE launch of the whole-network kernel (two launches per optimiser step)
        assert _lib.load().linna_net_train_launches(model.net_handle(with_grads=True), B) == 2


@pytest.mark.parametrize("name,engine", [("train_v2_33_33", 0), ("train_v2_26_457", 0), ("train_v2_26_457", 8), ("train_v2_33_33", 16)])
def test_train_step_entry_equals_forward_loss_plus_backward(name, engine):
    """``linna_net_train_step`` (forward + loss + backward in one call, the batch mean and AdamW's step constants riding
    in the dX-chain launch) against ``linna_net_forward_loss`` + ``linna_net_backward``: loss rows, batch mean, every
    parameter gradient, the step counter and the bias corrections, bit for bit.  On the engine the batch size picks (4 rows
    per workgroup: forward + loss + dX chain in ONE launch) and on the 8- and 16-row engines of larger batches (the entry
    runs forward + loss and the dX chain as two launches there)."""
    import ctypes as C
    from linna_amd import _lib
    from linna_amd.predictor_gpu import _AdamWState
    got = []
    _lib.engine_rows(engine)
    for one_call in (True, False):
        p, model, pred, eng, B = make_engine(name)
        opt = _AdamWState(model, 1e-3)
        rows = torch.arange(B, 2 * B, dtype=torch.int32, device="cuda")
        out = torch.zeros(1, device="cuda")
        for _ in range(2):                                   # (twice: the second step finds the counter at 1)
            if one_call:
                eng._forward_loss_backward(rows, out, opt)
                assert eng.one_launch is True
                assert _lib.load().linna_net_train_launches(model.net_handle(with_grads=True), B) == (3 if engine else 2)
            else:
                k, m = eng.k, model
                _lib.call("linna_net_forward_loss", m.net_handle(with_grads=True), C.byref(eng.desc), _lib.ptr(eng.X), eng.X.stride(0),
                          _lib.iptr(rows), B, _lib.iptr(k["lg"]) if k["lg"] is not None else None, _lib.ptr(k["xmean"]),
                          _lib.ptr(k["xstd"]), _lib.ptr(eng.xb), eng.xb.stride(0), _lib.ptr(m.workspace(B)), _lib.ptr(eng.predb),
                          eng.predb.stride(0), _lib.ptr(eng._targets()), eng.YN.stride(0), _lib.ptr(eng.den), eng.inv_batch,
                          _lib.ptr(eng.loss_rows), _lib.ptr(out), _lib.ptr(eng.dpred), eng.dpred.stride(0), _lib.ptr(opt.hyper),
                          _lib.iptr(opt.step_dev), opt.betas[0], opt.betas[1], _lib.stream())
                m._last_input = eng.xb
                m.backward(eng.dpred[:, :p["nout"]], param_grads=True)
        torch.cuda.synchronize()
        got.append([eng.loss_rows.cpu().numpy().copy(), out.cpu().numpy().copy(), model.flat_grads().cpu().numpy().copy(),
                    opt.step_dev.cpu().numpy().copy(), opt.hyper.cpu().numpy().copy()])
    _lib.engine_rows(0)
    assert int(got[0][3][0]) == 2 and np.isfinite(got[0][1]).all() and np.abs(got[0][2]).max() > 0
    for a, b in zip(*got):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("B", [1200, 2304])
def test_update_forms_agree_on_every_stream_layout(B):
    """Batch 1200 trains on the 8-row engine, 2304 on the 16-row engine, whose weight streams are laid out differently
    (lane = column against 16-column tiles): AdamW in the gradient tiles' epilogue, AdamW writing the streams and plain AdamW
    with the lazy re-layouts leave the same parameters, moments and losses, bit for bit."""
    from linna_amd import nn, util, predictor_gpu, trainer
    p = cases.training_problem("train_v2_12_40")
    rs = np.random.RandomState(9)
    base_x, base_y = p["X"].reshape(-1, p["nin"]), p["Y"].reshape(-1, p["nout"])
    idx = rs.randint(0, len(base_x), 3 * B)
    X = (base_x[idx] + 0.05 * rs.standard_normal((3 * B, p["nin"]))).astype(np.float32)
    Y = (base_y[idx] * (1 + 0.01 * rs.standard_normal((3 * B, p["nout"])))).astype(np.float32)
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32))
    res = []
    for mode in ("gemm", "streams", "plain"):
        model = nn.ChtoModelv2(p["nin"], p["nout"], None)
        model.load_state_dict(p["weights"])
        pred = predictor_gpu.Predictor(p["nin"], p["nout"], model=model, device="cuda",
                                       X_transform=util.X_transform_class(t(p["X_mean"]), t(p["X_std"]), "cpu", None),
                                       y_transform=util.Y_transform_class(t(p["y_mean"]), t(p["y_std"]), "cpu"))
        ytd = util.Y_transform_data(p["sigma"], "cpu")
        yinv = util.Y_invtransform_class(t(p["y_mean"]), t(p["y_std"]), t(p["data"]), "cpu")
        lf = util.Loss_fn(t(p["data"]), torch.tensor(p["cov"], dtype=torch.float64),
                          torch.tensor(np.linalg.inv(p["cov"]), dtype=torch.float64), ytd, yinv, "cpu")
        loader = predictor_gpu.BatchLoader(util.ArrayDataset(X, Y), B, shuffle=False, drop_last=True)
        eng = trainer.TrainEngine(pred, loader, lf, None, use_graph=False)
        opt = predictor_gpu._AdamWState(model, 1e-3)
        if mode != "gemm":
            eng.one_update = False
        if mode == "plain":
            opt._streams = False
        losses = []
        for s_ in range(4):
            out = torch.zeros(1, device="cuda")
            eng.step(opt, torch.arange((s_ % 3) * B, (s_ % 3 + 1) * B, dtype=torch.int32, device="cuda"), loss_out=out)
            losses.append(out)
        res.append((model._flat.cpu().numpy().copy(), opt.m.cpu().numpy().copy(), opt.v.cpu().numpy().copy(),
                    torch.cat(losses).cpu().numpy(), eng.one_update, opt._streams))
    assert res[0][4] is True and res[1][5] is True and np.isfinite(res[0][3]).all()
    for other in res[1:]:
        for a, b in zip(res[0][:4], other[:4]):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("name", ["train_mlp_7_5", "train_v2_33_33"])
def test_graph_replay_equals_direct_launches(name):
    from linna_amd.predictor_gpu import _AdamWState
    from linna_amd import trainer
    res = []
    for use_graph in (False, True):
        p, model, pred, eng, B = make_engine(name)
        eng.use_graph = use_graph
        opt = _AdamWState(model, 1e-3)
        if use_graph:
            eng.prepare_graph(opt)
        for s in range(3):
            eng.step(opt, torch.arange(s * B, (s + 1) * B, dtype=torch.int32, device="cuda"))
        torch.cuda.synchronize()
        res.append(model.flat_params().cpu().numpy().copy())
    np.testing.assert_array_equal(res[0], res[1])


def test_train_NN_trajectory_matches_reference(tmp_path):
    """The whole util.train_NN -> Predictor.train run of the reference (6 epochs, batch 50,
    lr.npy = 2e-3, same initial weights through the nnmodel_in plug-in, same sample order from
    torch.manual_seed(1234)): per-step training losses and per-epoch validation metrics."""
    from linna_amd import util, nn
    g = cases.golden("train_nn_run")
    out = str(tmp_path) + "/"
    np.savetxt(out + "train_samples_x.txt", g["train_x"]); np.save(out + "train_samples_y.npy", g["train_y"])
    np.savetxt(out + "val_samples_x.txt", g["val_x"]); np.save(out + "val_samples_y.npy", g["val_y"])
    np.save(out + "lr.npy", float(g["lr"]))
    w0 = synth.weights("ChtoModelv2", 5, 3, 301)

    def factory(in_size, out_size, linearmodel, docpu=False):
        m = nn.ChtoModelv2(in_size, out_size, linearmodel, docpu=docpu)
        m.load_state_dict(w0)
        return m

    cov = g["cov"]
    model = util.train_NN(None, cov, np.linalg.inv(cov), np.sqrt(np.diag(cov)), out, [out], g["data"], None, False, True, 2,
                          1.0, False, None, 1, factory, {"num_epochs": int(g["num_epochs"]), "batch_size": int(g["batch_size"])},
                          False)
    train_losses, val_metrics = model.train_history
    np.testing.assert_allclose(model.X_transform.X_mean.numpy(), g["X_mean"], rtol=2e-6, atol=2e-8)
    np.testing.assert_allclose(model.y_transform.y_std.numpy(), g["y_std"], rtol=2e-6)
    assert len(train_losses) == len(g["train_losses"])
    np.testing.assert_allclose(train_losses, g["train_losses"], rtol=4e-6)
    np.testing.assert_allclose(val_metrics, g["val_metrics"], rtol=3e-5)
    # artefacts in the reference's on-disk layout
    for f in ("best.pth.tar", "last.pth.tar", "X_transform.pkl", "y_transform.pkl", "y_invtransform.pkl",
              "y_transform_data.pkl", "y_invtransform_data.pkl"):
        assert os.path.isfile(out + f), f
    ck = torch.load(out + "best.pth.tar", weights_only=True)
    assert int(ck["epoch"]) == int(g["best_epoch"])
    for k, v in ck["state_dict"].items():
        ref = g["best/" + k]
        np.testing.assert_allclose(v.numpy(), ref, rtol=1.5e-4, atol=1.5e-5 * np.abs(ref).max() + 1e-6, err_msg=k)
    # and the freshly written directory serves through retrieve_model + Log_prob
    pm, yinv = util.retrieve_model(out, 5, 3, nn.ChtoModelv2)
    priors = [{"param": "p%d" % i, "dist": "flat", "arg1": -1.0, "arg2": 1.0} for i in range(5)]
    lp = util.Log_prob(g["data"], np.linalg.inv(cov), pm, yinv, util.Transform(priors), 1.0)
    assert np.all(np.isfinite(lp(np.zeros((4, 5), np.float32), returntorch=False)))


def test_train_NN_ypositive_trajectory_matches_reference(tmp_path):
    """``ypositive=True`` training (util.py:1410-1431, 1444-1447, 567-586): positive targets emulated in log
    space.  The reference's run on the same files: rows it drops, log-space statistics, per-step losses and per-epoch
    validation metrics of 4 epochs (the target's logarithm is taken in the loss kernels: linna_loss_desc_t::ylog), and the
    trained directory serves through the exp output map."""
    from linna_amd import util, nn
    g = cases.golden("train_nn_ypos")
    out = str(tmp_path) + "/"
    np.savetxt(out + "train_samples_x.txt", g["train_x"]); np.save(out + "train_samples_y.npy", g["train_y"])
    np.savetxt(out + "val_samples_x.txt", g["val_x"]); np.save(out + "val_samples_y.npy", g["val_y"])
    np.save(out + "lr.npy", float(g["lr"]))
    w0 = synth.weights("ChtoModelv2", 5, 4, 311)

    def factory(in_size, out_size, linearmodel, docpu=False):
        m = nn.ChtoModelv2(in_size, out_size, linearmodel, docpu=docpu)
        m.load_state_dict(w0)
        return m

    cov = g["cov"]
    model = util.train_NN(None, cov, np.linalg.inv(cov), np.sqrt(np.diag(cov)), out, [out], g["data"], None, True, True, 2,
                          1.0, False, None, 1, factory, {"num_epochs": int(g["num_epochs"]), "batch_size": int(g["batch_size"])},
                          False)
    train_losses, val_metrics = model.train_history
    assert model.y_transform.ypositive is True
    np.testing.assert_allclose(model.X_transform.X_mean.numpy(), g["X_mean"], rtol=2e-6, atol=2e-8)
    np.testing.assert_allclose(model.y_transform.y_mean.numpy(), g["y_mean"], rtol=2e-6)
    np.testing.assert_allclose(model.y_transform.y_std.numpy(), g["y_std"], rtol=2e-6)
    assert len(train_losses) == len(g["train_losses"]) == 16                      # 200 rows (one dropped) in batches of 50, 4 epochs
    np.testing.assert_allclose(train_losses, g["train_losses"], rtol=2e-5)       # (logf on the device against torch.log on the host)
    np.testing.assert_allclose(val_metrics, g["val_metrics"], rtol=1e-4)
    ck = torch.load(out + "best.pth.tar", weights_only=True)
    assert int(ck["epoch"]) == int(g["best_epoch"])
    for k, v in model.model.state_dict().items():
        ref = g["final/" + k]
        np.testing.assert_allclose(v.cpu().numpy(), ref, rtol=3e-4, atol=3e-5 * np.abs(ref).max() + 1e-6, err_msg=k)
    pm, yinv = util.retrieve_model(out, 5, 4, nn.ChtoModelv2)
    assert pm.y_transform.ypositive is True
    priors = [{"param": "p%d" % i, "dist": "flat", "arg1": -1.0, "arg2": 1.0} for i in range(5)]
    lp = util.Log_prob(g["data"], np.linalg.inv(cov), pm, yinv, util.Transform(priors), 1.0)
    z = np.random.RandomState(0).standard_normal((16, 5)).astype(np.float32)
    got = lp(z, returntorch=False)
    # the oracle on the trained weights: exp output map, same constants
    from oracle import likelihood
    emu = likelihood.Emulator("ChtoModelv2", 5, 4, {k: v.cpu().numpy() for k, v in pm.model.state_dict().items()}, g["X_mean"], g["X_std"],
                              g["y_mean"], g["y_std"], np.sqrt(np.diag(cov)), ypositive=True)
    ref = likelihood.log_prob(z, emu, priors, g["data"], np.linalg.inv(cov), 1.0)
    np.testing.assert_allclose(got, ref, rtol=2e-4)


def test_train_NN_usebest_matches_reference(tmp_path):
    """``usebest=True`` (util.py:1375-1409; the 18th entry of model_args.pkl when `nbest` is given, main.py:197): the
    optimizer-seeded samples go in front of the designed ones, training and validation."""
    from linna_amd import util, nn
    g = cases.golden("train_nn_usebest")
    out = str(tmp_path) + "/"
    for tag, fx, fy in (("train", "train_samples_x.txt", "train_samples_y.npy"), ("val", "val_samples_x.txt", "val_samples_y.npy"),
                        ("best", "best_samples_x.txt", "best_samples_y.npy"), ("best_val", "best_samples_x_val.txt", "best_samples_y_val.npy")):
        np.savetxt(out + fx, g["x_" + tag]); np.save(out + fy, g["y_" + tag])
    np.save(out + "lr.npy", float(g["lr"]))
    w0 = synth.weights("ChtoModelv2", 5, 3, 321)

    def factory(in_size, out_size, linearmodel, docpu=False):
        m = nn.ChtoModelv2(in_size, out_size, linearmodel, docpu=docpu)
        m.load_state_dict(w0)
        return m

    cov = g["cov"]
    model = util.train_NN(None, cov, np.linalg.inv(cov), np.sqrt(np.diag(cov)), out, [out], g["data"], None, False, True, 2,
                          1.0, False, None, 1, factory, {"num_epochs": int(g["num_epochs"]), "batch_size": int(g["batch_size"])},
                          True)
    train_losses, val_metrics = model.train_history
    assert len(train_losses) == len(g["train_losses"]) == 12                      # (150 + 50) rows in batches of 50, 3 epochs
    np.testing.assert_allclose(model.X_transform.X_mean.numpy(), g["X_mean"], rtol=2e-6, atol=2e-8)
    np.testing.assert_allclose(model.y_transform.y_std.numpy(), g["y_std"], rtol=2e-6)
    np.testing.assert_allclose(train_losses, g["train_losses"], rtol=4e-6)
    np.testing.assert_allclose(val_metrics, g["val_metrics"], rtol=3e-5)


def test_lr_range_test_runs_and_restores_weights(tmp_path):
    from linna_amd import lrfinder
    p, model, pred, eng, B = make_engine("train_mlp_7_5")
    before = model.flat_params().clone()
    lr = lrfinder.range_test(pred, eng, num_iter=20)
    assert 1e-4 <= lr <= 5e-3
    assert torch.equal(before, model.flat_params())


@pytest.mark.parametrize("name", ["train_v2_5_3", "train_v2_12_40"])
def test_lr_range_test_matches_the_oracle_curve(name):
    """The range test on the HIP training step against the oracle's restatement of it (oracle/training.lr_range_test: the
    published algorithm of the third-party finder the reference calls at predictor_gpu.py:222-238, on the oracle's numpy
    training step): the same learning-rate schedule, the same recorded loss curve within float32 training noise, the same
    selected learning rate, and the model untouched afterwards."""
    from oracle import training
    from linna_amd import lrfinder
    g = cases.golden(name)
    p, model, pred, eng, B = make_engine(name)
    before = model.flat_params().clone()
    hist = {}
    lr = lrfinder.range_test(pred, eng, num_iter=30, history=hist)
    assert torch.equal(before, model.flat_params())
    stats = dict(X_mean=p["X_mean"], X_std=p["X_std"], y_mean=p["y_mean"], y_std=p["y_std"],
                 sigma=p["sigma"].astype(np.float32), data_norm=g["data_norm"].reshape(-1), icov_norm=g["icov_norm"])
    batches = [(p["X"][s], p["Y"][s]) for s in range(3)]          # the loader of make_engine: unshuffled, three batches
    lr_ref, lrs, losses = training.lr_range_test(p["weights"], batches, batches, stats, p["kind"], p["nin"], p["nout"], num_iter=30, **p["kw"])
    np.testing.assert_allclose(hist["lr"], lrs, rtol=1e-12)
    assert len(hist["loss"]) == len(losses)
    np.testing.assert_allclose(hist["loss"], losses, rtol=4e-6)
    assert lr == lr_ref and 1e-4 <= lr <= 5e-3


def test_first_training_step_can_be_captured():
    """A hipGraph capture of the VERY FIRST optimiser step (nothing launched before it on this network): the library
    allocates at create / prepare time only, and the descriptor table of the grouped parameter-gradient launch is
    uploaded as kernel arguments.  Replays give the parameters of the same steps launched directly."""
    from linna_amd import nn, util, predictor_gpu, trainer
    from linna_amd.predictor_gpu import _AdamWState
    p = cases.training_problem("train_v2_12_40")
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32))
    X = p["X"].reshape(-1, p["nin"]); Y = p["Y"].reshape(-1, p["nout"])
    B = 50

    def engine(use_graph):
        model = nn.ChtoModelv2(p["nin"], p["nout"], None)
        model.load_state_dict(p["weights"])
        pred = predictor_gpu.Predictor(p["nin"], p["nout"], model=model, device="cuda",
                                       X_transform=util.X_transform_class(t(p["X_mean"]), t(p["X_std"]), "cpu", None),
                                       y_transform=util.Y_transform_class(t(p["y_mean"]), t(p["y_std"]), "cpu"))
        ytd = util.Y_transform_data(p["sigma"], "cpu")
        yinv = util.Y_invtransform_class(t(p["y_mean"]), t(p["y_std"]), t(p["data"]), "cpu")
        lf = util.Loss_fn(t(p["data"]), torch.tensor(p["cov"], dtype=torch.float64),
                          torch.tensor(np.linalg.inv(p["cov"]), dtype=torch.float64), ytd, yinv, "cpu")
        loader = predictor_gpu.BatchLoader(util.ArrayDataset(X, Y), B, shuffle=False, drop_last=True)
        eng = trainer.TrainEngine(pred, loader, lf, None, use_graph=use_graph)
        return model, eng, _AdamWState(model, 1e-3, weight_decay=1e-4)

    model_g, eng_g, opt_g = engine(True)
    eng_g.prepare_graph(opt_g)                                   # capture: the first thing this network ever does
    assert eng_g.graph is not None
    assert model_g.stream_state()[0] == 1 and model_g.stream_state()[1] == 1      # the one-launch paths are in the graph
    model_d, eng_d, opt_d = engine(False)
    for s in range(3):
        rows = torch.arange(s * B, (s + 1) * B, dtype=torch.int32, device="cuda")
        eng_g.step(opt_g, rows)
        eng_d.step(opt_d, rows)
    torch.cuda.synchronize()
    np.testing.assert_allclose(model_g.flat_params().cpu().numpy(), model_d.flat_params().cpu().numpy(), rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(float(eng_g.loss_mean), float(eng_d.loss_mean), rtol=1e-6)


def test_train_gpu_process_shim(tmp_path):
    """linna/train_gpu.py:24-38: `python train_gpu.py <outdir> cuda` reads model_args.pkl, trains,
    writes finish.pkl -- here `python -m linna_amd.train_gpu`."""
    import pickle
    import subprocess
    import sys
    g = cases.golden("train_nn_run")
    out = str(tmp_path) + "/"
    np.savetxt(out + "train_samples_x.txt", g["train_x"]); np.save(out + "train_samples_y.npy", g["train_y"])
    np.savetxt(out + "val_samples_x.txt", g["val_x"]); np.save(out + "val_samples_y.npy", g["val_y"])
    np.save(out + "lr.npy", 2e-3)
    cov = g["cov"]
    args = [None, cov, np.linalg.inv(cov), np.sqrt(np.diag(cov)), out, [out], g["data"], None, False, False, 2, 1.0, True, None,
            1, None, {"num_epochs": 2, "batch_size": 50}, False]
    with open(out + "model_args.pkl", "wb") as f:
        pickle.dump(args, f)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "linna_amd.train_gpu", out, "cuda"], cwd=root, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert os.path.isfile(out + "finish.pkl") and os.path.isfile(out + "best.pth.tar")


@pytest.mark.parametrize("kind,nin,nout,kw", [("ChtoModelv2", 33, 33, {}), ("ChtoModelv2", 26, 457, {}), ("ChtoModelsimple", 6, 4, {}),
                                              ("MLP", 33, 33, {"width": 512, "depth": 4})])
def test_one_launch_dx_chain_equals_gemm_chain(kind, nin, nout, kw, monkeypatch):
    """The backward's dX chain as ONE launch of the whole-network kernel over the transposed weights
    (net_stream.hip, STORE == 2) against the GEMM-per-op chain it replaces: every parameter gradient and the
    gradient with respect to the input, with each of the kernel's three engines and a ragged batch."""
    from linna_amd import nn
    cls = {"ChtoModelv2": nn.ChtoModelv2, "ChtoModelsimple": nn.ChtoModelsimple, "MLP": nn.MLP}[kind]
    torch.manual_seed(5)
    fused = cls(nin, nout, None, **kw)
    fused.init_weight()
    # give the zero-initialised skip weights of the residual blocks something to propagate
    sd = {k: (v if "skip" not in k else 0.05 * torch.randn_like(v)) for k, v in fused.state_dict().items()}
    fused.load_state_dict(sd)
    chain = cls(nin, nout, None, **kw)
    chain.load_state_dict(sd)
    fused.cuda(); chain.cuda()
    B = 301
    x = torch.randn(B, nin, device="cuda")
    dout = torch.randn(B, nout, device="cuda") / B
    monkeypatch.setenv("LINNA_BWD_STREAM", "0")           # read at the network's first backward
    chain.forward(x)
    dx_ref = chain.backward(dout, param_grads=True, need_dx=True)[:, :nin].clone()
    g_ref = {k: v.clone() for k, v in chain.grad_dict().items()}
    monkeypatch.delenv("LINNA_BWD_STREAM")
    for rows in (None, "4", "8", "16"):
        if rows is None:
            _lib.engine_rows(0)
        else:
            _lib.engine_rows(int(rows))
        for need_dx in (True, False):
            fused.flat_grads().zero_()
            fused.forward(x)
            dx = fused.backward(dout, param_grads=True, need_dx=need_dx)
            if need_dx:
                np.testing.assert_allclose(dx[:, :nin].cpu().numpy(), dx_ref.cpu().numpy(), rtol=1e-3,
                                           atol=2e-5 * float(dx_ref.abs().max()), err_msg="dX rows %s" % rows)
            for k, gk in fused.grad_dict().items():
                ref = g_ref[k].cpu().numpy()
                np.testing.assert_allclose(gk.cpu().numpy(), ref, rtol=1e-3, atol=1e-5 * np.abs(ref).max() + 1e-9,
                                           err_msg="%s rows %s" % (k, rows))
    _lib.engine_rows(0)
    # the two objects took different routes: the fused one holds a weight stream for the dX chain
    assert fused.uses_dx_stream() and not chain.uses_dx_stream()
