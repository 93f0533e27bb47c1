"""CPU: host-side logic of the product (no GPU compute): EarlyStopping traces, batch order,
checkpoint formats, reference-artefact loading, parameter layout."""
import os
import sys
import pickle

import numpy as np
import pytest
import torch

import cases
import synth


def test_early_stopping_matches_reference_traces():
    from linna_amd.predictor_gpu import EarlyStopping
    g = cases.golden("early_stopping")
    names = sorted({k.split("/")[0] for k in g.files if k.endswith("/val")})
    assert names
    for name in names:
        es = EarlyStopping(patience=500)
        codes = []
        for a, b in zip(g[name + "/val"], g[name + "/train"]):
            c = es.step(float(a), float(b))
            codes.append(int(c))
            if c == 2:
                break
        np.testing.assert_array_equal(codes, g[name + "/codes"], err_msg=name)
    es = EarlyStopping(patience=40, nqueue=20)
    codes = []
    for a, b in zip(g["overfit/val"], g["overfit/train"]):
        c = es.step(float(a), float(b))
        codes.append(int(c))
        if c == 2:
            break
    np.testing.assert_array_equal(codes, g["overfit_p40/codes"])
    assert 2 in codes and 1 in codes


def test_batch_order_matches_torch_dataloader_with_seed_1234():
    from linna_amd.predictor_gpu import BatchLoader
    from linna_amd.util import ArrayDataset
    g = cases.golden("loader_order")
    n, batch = int(g["n"]), int(g["batch"])
    X = np.zeros((n, 2), np.float32)
    loader = BatchLoader(ArrayDataset(X, X), batch, shuffle=True, drop_last=True)
    torch.manual_seed(1234)
    for ep in range(g["order"].shape[0]):
        got = torch.cat(loader.epoch_batches()).numpy()
        np.testing.assert_array_equal(got, g["order"][ep])


def test_parameter_layout_and_state_dict_roundtrip():
    from linna_amd import nn
    m = nn.ChtoModelv2(33, 33, None)
    assert m.nparams == 843892 and m.macs_per_eval() == 841339
    assert nn.MLP(33, 33, None).nparams == 822305
    keys = list(m.state_dict().keys())
    assert keys[:2] == ["layer1.weight", "layer1.bias"] and "layer2.skip_layer.weight" in keys
    assert "layer8.bias" == keys[-1]
    w = synth.weights("ChtoModelv2", 33, 33, 7)
    m.load_state_dict(w)
    for k, v in m.state_dict().items():
        np.testing.assert_array_equal(v.numpy(), w[k])
    # init: bias 0.01; the skip weights come out Xavier-uniform (nn.py:95-101 re-draws what nn.py:43 zeroed)
    m.init_weight()
    sd = m.state_dict()
    bound = np.sqrt(6.0 / (250 + 500))
    assert 0.9 * bound < float(sd["layer3.skip_layer.weight"].abs().max()) <= bound * (1 + 1e-6)
    assert np.allclose(sd["layer6.bias"].numpy(), 1e-2)
    lin = nn.ChtoModelv2_linear(5, 3, None).state_dict()
    assert np.allclose(lin["linearlayer.weight"].numpy(), 1e-5) and float(lin["linearlayer.bias"].abs().max()) == 0
    with pytest.raises(KeyError):
        m.load_state_dict({"layer1.weight": w["layer1.weight"]})


def _digest_equal(a, ref):
    np.testing.assert_array_equal(synth.tensor_digest(np.asarray(a)), ref)


@pytest.mark.parametrize("kind,nin,nout,seed", [("ChtoModelv2", 33, 33, 11), ("ChtoModelv2", 26, 457, 12),
                                                ("ChtoModelsimple", 6, 4, 13), ("ChtoModelv2_linear", 5, 3, 14)])
def test_initial_weights_are_the_references_bit_for_bit(kind, nin, nout, seed):
    """``torch.manual_seed(s); Model(nin, nout, None)`` and a later ``init_weight()`` give the live reference's
    tensors exactly (digests in tests/golden/init_parity.npz) and leave torch's generator at the same position --
    for the product's network classes and for the oracle's numpy restatement of torch's CPU generator."""
    from linna_amd import nn
    from oracle import emulator as E
    g = cases.golden("init_parity")
    tag = "%s_%d_%d" % (kind, nin, nout)
    torch.manual_seed(seed)
    m = getattr(nn, kind)(nin, nout, None)
    np.testing.assert_array_equal(torch.rand(3).numpy(), g[tag + "/after_construct"])
    gen = E.TorchCPUGenerator(seed)
    p = E.init_params(kind, nin, nout, gen)
    assert list(p) == list(m.state_dict())
    for k, v in m.state_dict().items():
        _digest_equal(v.numpy(), g[tag + "/construct/" + k])
        _digest_equal(p[k], g[tag + "/construct/" + k])
    torch.manual_seed(seed + 100)
    m.init_weight()
    np.testing.assert_array_equal(torch.rand(3).numpy(), g[tag + "/after_reinit"])
    p = E.reinit_params(kind, nin, nout, E.TorchCPUGenerator(seed + 100))
    for k, v in m.state_dict().items():
        _digest_equal(v.numpy(), g[tag + "/reinit/" + k])
        _digest_equal(p[k], g[tag + "/reinit/" + k])
    assert float(m.state_dict()["layer2.skip_layer.weight"].abs().max()) > 0


def test_reference_fixture_artefacts_load_on_cpu():
    """The reference's committed checkpoint + transform pickles (linna.util.* classes) load
    through the product's readers."""
    from linna_amd import util, nn, nnutils
    outdir = os.path.join(cases.GOLDEN, "2dgaussian_Fulltconn/iter_0/")
    g = cases.golden("fixture2d")
    ck = nnutils.read_checkpoint(os.path.join(outdir, "best.pth.tar"))
    assert set(ck) >= {"epoch", "state_dict", "optim_dict"}
    model, yinv = util.retrieve_model(outdir, 2, 2, nn.ChtoModelv2, device="cpu")
    assert model.model.nparams == 8016
    np.testing.assert_allclose(model.X_transform.X_std.numpy(), g["X_std"])
    np.testing.assert_allclose(model.y_transform.y_mean.numpy(), g["y_mean"])
    np.testing.assert_allclose(yinv.sigma.detach().numpy(), g["sigma"])
    for k, v in ck["state_dict"].items():
        np.testing.assert_array_equal(model.model.state_dict()[k].numpy(), v.numpy())


def test_latin_hypercube_reproduces_the_reference_fixture_designs():
    """``NN_samplerv1.gensample_flat`` (util.py:775-814 over pyDOE2's centred Latin hypercube, restated) against the
    designs the reference's own run directory holds: 20 training and 5 validation points in [-2, 2]^2, written by the
    reference with pyDOE2 and its seed 123456 -- bit for bit, row order included."""
    from linna_amd import util
    ns = util.NN_samplerv1("/nonexistent/", [[-2.0, 2.0], [-2.0, 2.0]])
    for n, name in ((20, "train_samples_x.txt"), (5, "val_samples_x.txt")):
        ref = np.loadtxt(os.path.join(cases.GOLDEN, "2dgaussian_Fulltconn/iter_0", name))
        got = ns.gensample_flat(n)
        assert got.shape == ref.shape == (n, 2)
        np.testing.assert_array_equal(got, ref)
    # the design family: every column visits every one of the n cells exactly once; cuts make it grow by 1000 and keep n rows
    x = util.NN_samplerv1("/nonexistent/", [[0.0, 1.0]] * 5).gensample_flat(333)
    for j in range(5):
        np.testing.assert_array_equal(np.sort(np.floor(x[:, j] * 333).astype(int)), np.arange(333))
    cut = util.NN_samplerv1("/nonexistent/", [[0.0, 1.0]] * 3).gensample_flat(200, omegab2cut=[0, 1, 0.01, 0.5])
    assert cut.shape == (200, 3) and np.all((cut[:, 0] * cut[:, 1] ** 2 > 0.01) & (cut[:, 0] * cut[:, 1] ** 2 < 0.5))
    # A_s (second parameter, upper limit < 1e-5) is spread uniformly in its logarithm (util.py:795-803)
    a = util.NN_samplerv1("/nonexistent/", [[0.0, 1.0], [1e-9, 5e-9]]).gensample_flat(100)[:, 1]
    la = np.sort(np.log(a))
    np.testing.assert_allclose(np.diff(la), np.diff(la)[0], rtol=1e-9)


def test_chain_resampling_matches_the_live_reference():
    """``NN_samplerv1.gensample_chain_randomsample`` (util.py:864-897: training points of iterations >= 1 are drawn from
    the previous chain, inside the prior box and the omega_b h^2 cuts, seed 123456) against the live reference on the
    same synthetic chain (tests/golden/host_designs.npz), bit for bit; the 7-element cut the reference cannot read
    fails here in the same way."""
    from linna_amd import util
    g = cases.golden("host_designs")
    rs = np.random.RandomState(77)                                   # make_golden.host_design_inputs
    chain = rs.standard_normal((6000, 4)) * np.array([0.6, 0.5, 1.4, 0.8]) + np.array([0.3, 0.6, 0.0, 0.1])
    prior = [[-0.8, 1.4], [-0.4, 1.6], [-2.0, 2.0], [-1.0, 1.2]]
    cuts = {"none": None, "ombh2": [0, 1, 0.01, 0.9], "ombh2_p2_p3": [0, 1, 0.01, 0.9, 2, -1.5, 1.5, 3, -0.5, 1.0]}
    ns = util.NN_samplerv1("/nonexistent/", prior)
    keep = chain.copy()
    for tag, cut in cuts.items():
        for n in (300, 17):
            got = ns.gensample_chain_randomsample(n, chain, None, omegab2cut=cut)
            np.testing.assert_array_equal(got, g["%s/%d" % (tag, n)])
    np.testing.assert_array_equal(chain, keep)                       # the caller's chain is not modified
    with pytest.raises(IndexError):
        ns.gensample_chain_randomsample(10, chain, None, omegab2cut=[0, 1, 0.01, 0.9, 2, -1.5, 1.5])


def test_checkmeanstd_matches_the_live_reference(capsys):
    """sampler.py:370-387 on four synthetic chains (stationary, drifting mean, growing spread, odd length): the two
    drift statistics the reference prints and its verdicts for four threshold pairs (tests/golden/host_designs.npz)."""
    from linna_amd import sampler
    g = cases.golden("host_designs")
    rs = np.random.RandomState(55)                                   # make_golden.meanstd_inputs
    a = rs.standard_normal((400, 6, 5))
    chains = [a, a + np.linspace(0, 0.6, 400)[:, None, None], a * np.linspace(0.7, 1.4, 400)[:, None, None],
              rs.standard_normal((301, 4, 3)) * np.array([1.0, 2.0, 0.5])]
    for i, c in enumerate(chains):
        ref = g["checkmeanstd/%d" % i]
        capsys.readouterr()
        verdicts = [bool(sampler.checkmeanstd(c, x, y)) for x, y in [(0.1, 0.1), (0.02, 0.1), (0.1, 0.01), (1.0, 1.0)]]
        printed = [float(v) for v in capsys.readouterr().out.split()[:2]]
        np.testing.assert_allclose(printed, ref[:2], rtol=1e-12, atol=1e-15)
        assert verdicts == [bool(v) for v in ref[2:]]


def test_logprior_and_mad_match_the_live_reference():
    """``LogPrior`` (util.py:1129-1157) and ``median_absolute_deviation`` (:1308-1313) against the live reference
    (tests/golden/importance_helpers.npz)."""
    from linna_amd import util
    g = cases.golden("importance_helpers")
    rs = np.random.RandomState(88)                                   # make_golden.importance_inputs
    nout, ndim, n = 7, 4, 60
    A = rs.standard_normal((nout, nout))
    cov = A @ A.T / nout + 0.3 * np.eye(nout)
    data = rs.uniform(0.5, 1.5, nout)
    theory = np.concatenate([data, [9.0, 9.0]])[None, :] + rs.standard_normal((n, nout + 2)) * 0.4
    samples = rs.uniform(-1.5, 1.5, (n, ndim))
    priors = [{"param": "a", "dist": "flat", "arg1": -1.0, "arg2": 1.2}, {"param": "b", "dist": "gauss", "arg1": 0.2, "arg2": 0.7},
              {"param": "c", "dist": "flat", "arg1": -1.4, "arg2": 1.4}, {"param": "d", "dist": "gauss", "arg1": -0.3, "arg2": 1.1}]
    lpr = util.LogPrior(priors)
    got = np.array([lpr(s_) for s_ in samples], np.float64)
    np.testing.assert_array_equal(np.isinf(got), np.isinf(g["logprior"]))
    ok = np.isfinite(got)
    np.testing.assert_allclose(got[ok], g["logprior"][ok], rtol=1e-14)
    t = torch.tensor(theory[:, :nout], dtype=torch.float32)
    np.testing.assert_array_equal(util.median_absolute_deviation(t, t.median(axis=0).values, 0).numpy(), g["mad"])


def test_artefact_readers_execute_nothing(tmp_path):
    """Transform pickles and checkpoints of a run directory go through closed allow-lists: a file naming any
    other global (here os.system / builtins.eval) is refused before anything is imported or called."""
    import pickle
    from linna_amd import util, nnutils

    class Evil(object):
        def __reduce__(self):
            return (os.system, ("echo pwned > %s" % os.path.join(str(tmp_path), "pwned"),))
    bad = os.path.join(str(tmp_path), "X_transform.pkl")
    with open(bad, "wb") as f:
        pickle.dump(Evil(), f)
    with open(bad, "rb") as f, pytest.raises(pickle.UnpicklingError):
        util.CPU_Unpickler(f).load()
    with open(bad, "wb") as f:
        f.write(b"cbuiltins\neval\n(V1+1\ntR.")
    with open(bad, "rb") as f, pytest.raises(pickle.UnpicklingError):
        util.CPU_Unpickler(f).load()
    ck = os.path.join(str(tmp_path), "best.pth.tar")
    torch.save({"epoch": 1, "state_dict": {}, "optim_dict": Evil()}, ck)
    with pytest.raises(Exception):
        nnutils.read_checkpoint(ck)
    assert not os.path.exists(os.path.join(str(tmp_path), "pwned"))
    # a checkpoint whose optimiser state holds numpy scalars (lr from np.load(lr.npy)) still loads
    torch.save({"epoch": 2, "state_dict": {"w": torch.ones(2)},
                "optim_dict": {"param_groups": [{"lr": np.float64(1e-3), "weight_decay": np.load(_npy(tmp_path, 1e-4))}]}}, ck)
    got = nnutils.read_checkpoint(ck)
    assert float(got["optim_dict"]["param_groups"][0]["lr"]) == 1e-3
    # our own transform pickles round-trip through the same reader
    xt = util.X_transform_class(torch.zeros(3), torch.ones(3), "cpu", [0, 2])
    xt.pickle(bad)
    with open(bad, "rb") as f:
        back = util.CPU_Unpickler(f).load()
    assert back.dolog10index == [0, 2] and torch.equal(back.X_std, torch.ones(3))
    # the reference documents dolog10index as an "int array" (train_NN forwards it as given): a numpy array round-trips too
    xt = util.X_transform_class(torch.zeros(3), torch.ones(3), "cpu", np.array([0, 2]))
    xt.pickle(bad)
    with open(bad, "rb") as f:
        back = util.CPU_Unpickler(f).load()
    assert back.dolog10index == [0, 2] and all(type(i) is int for i in back.dolog10index)
    # ... and a transform pickle that does hold numpy arrays (one the reference wrote from such arguments) is data, not code
    xt.dolog10index = np.array([0, 2])
    xt.pickle(bad)
    with open(bad, "rb") as f:
        back = util.CPU_Unpickler(f).load()
    np.testing.assert_array_equal(back.dolog10index, [0, 2])
    x = torch.tensor([[10.0, 1.0, 100.0]])
    np.testing.assert_allclose(back(x).numpy(), [[1.0, 1.0, 2.0]])


def test_checkpoint_written_under_the_other_numpy_generation_loads(tmp_path):
    """The reference's ``optim_dict`` holds ``lr`` as a numpy scalar (``np.load(lr.npy) * size``); a file written under
    numpy 1.x names ``numpy.core.multiarray.scalar``, one written under numpy 2 ``numpy._core...`` -- torch matches
    allowed globals by that text.  Both load (weights_only=True throughout), zip and legacy stream formats."""
    import io
    import zipfile
    from linna_amd import nnutils
    old, new = nnutils._numpy_module_names()
    state = {"epoch": 3, "state_dict": {"w": torch.arange(6.0).reshape(2, 3)},
             "optim_dict": {"param_groups": [{"lr": np.float64(1e-3) * 2, "weight_decay": np.float32(1e-4)}]}}
    for legacy in (False, True):
        p = os.path.join(str(tmp_path), "ck%d.pth.tar" % legacy)
        torch.save(state, p, _use_new_zipfile_serialization=not legacy)
        if legacy:
            raw = open(p, "rb").read()
            assert new + b"multiarray" in raw
            open(p, "wb").write(raw.replace(new, old))
        else:
            buf = io.BytesIO()
            with zipfile.ZipFile(p) as zin, zipfile.ZipFile(buf, "w") as zout:
                for info in zin.infolist():
                    raw = zin.read(info.filename)
                    if info.filename.endswith("data.pkl"):
                        assert new + b"multiarray" in raw
                        raw = raw.replace(new, old)
                    zout.writestr(info.filename, raw)
            open(p, "wb").write(buf.getvalue())
        with torch.serialization.safe_globals(nnutils._numpy_scalar_globals()), pytest.raises(Exception):
            torch.load(p, weights_only=True)                          # (the other generation's name is not on torch's list)
        ck = nnutils.read_checkpoint(p)
        assert float(ck["optim_dict"]["param_groups"][0]["lr"]) == 2e-3 and torch.equal(ck["state_dict"]["w"], state["state_dict"]["w"])
    # a pickle that names anything else is still refused after the rename
    class Evil(object):
        def __reduce__(self):
            return (os.system, ("true",))
    p = os.path.join(str(tmp_path), "evil.pth.tar")
    torch.save({"epoch": 1, "state_dict": {}, "optim_dict": Evil()}, p)
    with pytest.raises(Exception):
        nnutils.read_checkpoint(p)


def test_renaming_pickled_numpy_globals_keeps_framed_streams_loadable(monkeypatch):
    """ADVICE r3 (nnutils.py:52): protocol >= 4 pickles are FRAMED, and a renamed module string changes the byte length a
    FRAME opcode promised.  The copy drops the FRAME opcodes (frames are optional) and handles BINUNICODE8: a protocol-4 and
    a protocol-5 stream of our own, renamed towards the OTHER numpy generation's module name, still unpickle to the same
    value (the stub module ``numpy.core`` resolves it)."""
    import pickle
    import pickletools
    import warnings
    from linna_amd import nnutils
    old, new = nnutils._numpy_module_names()
    monkeypatch.setattr(nnutils, "_numpy_module_names", lambda: (new, old))       # rename THIS numpy's name to the other one
    obj = {"lr": np.float64(2e-3), "wd": np.float32(1e-4), "pad": "x" * 300}
    for proto in (2, 4, 5):
        raw = pickle.dumps(obj, protocol=proto)
        names = [op.name for op, _, _ in pickletools.genops(raw)]
        assert ("FRAME" in names) == (proto >= 4)
        out, used = nnutils._rename_numpy_globals(raw + b"trailing")
        assert used == len(raw) and new not in out and old in out
        assert "FRAME" not in [op.name for op, _, _ in pickletools.genops(out)]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                back = pickle.loads(out)                                          # (bytes this test made itself)
            except (ImportError, AttributeError):
                continue                                                          # a numpy without the forwarding stub: the walk above is the check
        assert back == obj and type(back["lr"]) is np.float64
    # the stale-frame failure this replaces: the same rename with the FRAME opcodes left in place does not load
    raw = pickle.dumps(obj, protocol=4)
    with pytest.raises(Exception):
        pickle.loads(raw.replace(new, old))


def test_quadratic_forms_of_the_post_steps_are_float64():
    """``logp_theory_data`` / ``chisqcut_all`` (util.py:1506-1517, 1260-1270) feed the importance weights
    ``w = exp(logp - lp)``: float64 on the host as in the reference.  With an inverse covariance of condition number
    1e10 the fp32 error of d^T S d would be percent-level; the float64 value is reproduced to 1e-10."""
    from linna_amd import util
    rs = np.random.RandomState(4)
    nout, n = 120, 500
    q, _ = np.linalg.qr(rs.standard_normal((nout, nout)))
    S = (q * np.logspace(0, 10, nout)[None, :]) @ q.T
    S = 0.5 * (S + S.T)
    assert np.linalg.cond(S) > 1e8
    th = rs.standard_normal((n, nout + 3))                            # theory rows longer than the data vector are cut
    data = rs.standard_normal(nout)
    d = th[:, :nout] - data
    ref = np.array([di @ S @ di for di in d])
    np.testing.assert_allclose(util.chi2_rows(d, S), ref, rtol=1e-10)
    lp = util.logp_theory_data(np.zeros((n, 2)), th, data, S, lambda s: 0.0)
    np.testing.assert_allclose(lp, -0.5 * ref, rtol=1e-10)
    d32 = d.astype(np.float32)
    fp32 = np.einsum("bi,ij,bj->b", d32, S.astype(np.float32), d32)
    assert np.max(np.abs(fp32 - ref) / ref) > 1e-6                    # (what the fp32 form would have cost)


def _npy(tmp_path, v):
    p = os.path.join(str(tmp_path), "v.npy")
    np.save(p, v)
    return p


def test_transform_pickles_name_the_references_classes(tmp_path):
    """``X_transform.pkl`` & co. written here name ``linna.util.<Class>`` (identical attribute sets), so a run directory
    continues under the reference without this package; this package's own reader maps the name back.  Where the
    reference tree is present (the build container), its own unpickler loads the files as ITS classes and they transform
    like ours."""
    from linna_amd import util
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32))
    objs = {"X_transform": util.X_transform_class(t(np.arange(4.0)), t(np.full(4, 2.0)), "cpu", [0, 2]),
            "y_transform": util.Y_transform_class(t(np.zeros(3)), t(np.ones(3)), "cpu"),
            "y_invtransform": util.Y_invtransform_class(t(np.zeros(3)), t(np.ones(3)), t(np.ones(3)), "cpu"),
            "y_transform_data": util.Y_transform_data(np.array([1.0, 2.0, 3.0]), "cpu"),
            "y_invtransform_data": util.Y_invtransform_data(np.array([1.0, 2.0, 3.0]), "cpu")}
    for k, o in objs.items():
        path = os.path.join(str(tmp_path), k + ".pkl")
        o.pickle(path)
        raw = open(path, "rb").read()
        assert b"clinna.util\n" + type(o).__name__.encode() + b"\n" in raw and b"linna_amd" not in raw
        with open(path, "rb") as f:
            back = util.CPU_Unpickler(f).load()
        assert type(back) is type(o)
    if not os.path.isdir("/root/reference/linna"):
        pytest.skip("reference tree not present: the reference-side half of this check runs in the build container")
    # in a process of its own: the reference import installs stand-ins for its absent third-party modules
    x, y = torch.rand(5, 4) + 0.5, torch.rand(5, 3)
    torch.save({"x": x, "y": y, "want": {k: o(x if k == "X_transform" else y) for k, o in objs.items()}},
               os.path.join(str(tmp_path), "io.pt"))
    code = (
        "import sys, os, torch\n"
        "sys.path.insert(0, %r)\n"
        "import _ref_import\n"
        "_, rutil, _, _ = _ref_import.import_reference()\n"
        "d = %r\n"
        "io = torch.load(os.path.join(d, 'io.pt'))\n"
        "for k, want in io['want'].items():\n"
        "    with open(os.path.join(d, k + '.pkl'), 'rb') as f:\n"
        "        r = rutil.CPU_Unpickler(f).load()\n"
        "    assert type(r).__module__ == 'linna.util', type(r)\n"
        "    assert torch.equal(r(io['x'] if k == 'X_transform' else io['y']), want), k\n"
        "print('REFERENCE_READS_OK')\n") % (cases.GOLDEN, str(tmp_path))
    import subprocess
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "REFERENCE_READS_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_checkpoint_written_in_reference_layout(tmp_path):
    from linna_amd import nn, nnutils
    from linna_amd.predictor_gpu import _AdamWState
    m = nn.ChtoModelv2(4, 2, None)
    opt = _AdamWState.__new__(_AdamWState)
    opt.model, opt.lr, opt.weight_decay, opt.betas, opt.eps = m, 1e-3, 1e-4, (0.9, 0.999), 1e-8
    opt.m, opt.v = torch.zeros_like(m.flat_params()), torch.ones_like(m.flat_params())
    opt.step_dev, opt.hyper = torch.tensor([7], dtype=torch.int32), torch.zeros(4)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    nnutils.save_checkpoint({"epoch": 3, "state_dict": sd, "optim_dict": opt.state_dict()}, True, str(tmp_path))
    ck = torch.load(os.path.join(str(tmp_path), "best.pth.tar"), weights_only=True)
    assert ck["epoch"] == 3 and list(ck["state_dict"]) == list(m.state_dict())
    pg = ck["optim_dict"]["param_groups"][0]
    assert pg["lr"] == 1e-3 and pg["weight_decay"] == 1e-4 and len(pg["params"]) == len(sd)
    assert float(ck["optim_dict"]["state"][0]["step"]) == 7.0
    # and a torch.optim.AdamW over same-shaped tensors accepts it (what the reference would do)
    params = [torch.nn.Parameter(v.clone()) for v in sd.values()]
    topt = torch.optim.AdamW(params, lr=1.0)
    topt.load_state_dict(ck["optim_dict"])
    assert topt.param_groups[0]["lr"] == 1e-3


def test_transform_python_path_matches_reference_theta():
    from linna_amd import util
    name = "v2lin_5_3_log10"
    g = cases.golden(name)
    prob = cases.serving_problem(name)
    t = util.Transform(prob["priors"])
    theta = np.stack([t(z) for z in g["z"][:8]])
    np.testing.assert_allclose(theta, g["theta"][:8], rtol=2e-6, atol=2e-6)
    back = util.invTransform(prob["priors"])(theta[0].astype(np.float64))
    np.testing.assert_allclose(back, g["z"][0], rtol=2e-3, atol=2e-3)


def test_loss_constants_match_reference():
    from linna_amd import util
    g = cases.golden("train_v2_12_40")
    p = cases.training_problem("train_v2_12_40")
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32))
    ytd = util.Y_transform_data(p["sigma"], "cpu")
    yinv = util.Y_invtransform_class(t(p["y_mean"]), t(p["y_std"]), t(p["data"]), "cpu")
    lf = util.Loss_fn(t(p["data"]), torch.tensor(p["cov"], dtype=torch.float64),
                      torch.tensor(np.linalg.inv(p["cov"]), dtype=torch.float64), ytd, yinv, "cpu")
    sigma, ymean, ystd, data_norm, cinv = lf.auxileryfunction.arrays()
    np.testing.assert_allclose(cinv, g["icov_norm"], rtol=1e-5, atol=1e-6 * np.abs(g["icov_norm"]).max())
    np.testing.assert_allclose(data_norm, g["data_norm"].reshape(-1), rtol=1e-5, atol=1e-6)
    assert np.array_equal(cinv, cinv.T)


def test_missing_library_is_a_loud_error(monkeypatch):
    from linna_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/liblinna_hip.so")
    with pytest.raises(_lib.LinnaHipError):
        _lib.load()


def test_cpu_model_refuses_to_compute():
    from linna_amd import nn, _lib
    m = nn.MLP(7, 5, None, width=48, depth=3)
    with pytest.raises(_lib.LinnaHipError):
        m.forward(torch.zeros(2, 7))


def test_chainstore_grows_the_hdf5_file_in_place(tmp_path):
    """Incremental flushes append to ``<name>.h5`` (one chunk per dataset and block, written by the background
    thread): the file is current after every flush; a resumed store continues it without duplicating what it
    loaded; blocks of different lengths and a legacy part file of an earlier version are taken in."""
    from linna_amd.sampler import ChainStore
    from linna_amd import h5lite
    rs = np.random.RandomState(0)
    name = str(tmp_path / "chemcee_256.h5")
    blocks = [(rs.standard_normal((n, 4, 3)), rs.standard_normal((n, 4, 3)), rs.standard_normal((n, 4))) for n in (5, 5, 3, 5)]
    cat = lambda j, upto: np.concatenate([b[j] for b in blocks[:upto]])
    st = ChainStore(name)
    assert not st.exists()
    for i, (z, th, lp) in enumerate(blocks[:2]):
        st.append(z, th, lp, np.full(4, i + 1.0))
        st.flush(final=False)
        st.drain()                                          # written by a background thread
        d = ChainStore.load(name)
        np.testing.assert_array_equal(d["chain"], cat(0, i + 1))
        np.testing.assert_array_equal(d["accepted"], np.full(4, i + 1.0))
    assert st.exists() and os.path.isfile(st.h5) and not ChainStore._parts(st.base)
    with h5lite.File(st.h5) as f:                           # emcee's layout, extensible along the step axis
        assert f["mcmc/chain"].maxshape == (h5lite.UNDEF, 4, 3) and f["mcmc"].attrs["iteration"] == 10
    # resume in a new store: load, append as one block, continue; a short block, then a full one
    d = ChainStore.load(name)
    st2 = ChainStore(name, write_txt=True)
    st2.append(d["chain"], d["chain_transformed"], d["log_prob"], d["accepted"])
    for i in (2, 3):
        st2.append(*blocks[i], np.full(4, i + 1.0))
        st2.flush(final=False)
    st2.drain()
    d2 = ChainStore.load(name)
    np.testing.assert_array_equal(d2["chain"], cat(0, 4))
    np.testing.assert_array_equal(d2["chain_transformed"], cat(1, 4))
    np.testing.assert_array_equal(d2["log_prob"], cat(2, 4))
    assert d2["iteration"] == 18
    st2.flush()
    d3 = ChainStore.load(name)
    np.testing.assert_array_equal(d3["chain"], d2["chain"])
    assert os.path.isfile(st2.base + ".txt")
    # chunks hold CHUNK_ROWS steps whatever the first block's length (a resumed chain arrives as ONE long block), the
    # resumed file kept growing IN PLACE (same inode, no rewrite, nothing to lose), and it grew by what was appended
    with h5lite.File(st2.h5) as f:
        assert f["mcmc/chain"].chunks[0] == ChainStore.CHUNK_ROWS == 100
    ino, size = os.stat(name).st_ino, os.path.getsize(name)
    d = ChainStore.load(name)
    st4 = ChainStore(name)
    st4.append(d["chain"], d["chain_transformed"], d["log_prob"], d["accepted"])
    st4.append(*blocks[0], np.full(4, 9.0))
    st4.flush(final=False); st4.drain()
    assert os.stat(name).st_ino == ino and not os.path.exists(name + ".tmp")
    assert os.path.getsize(name) - size < 8192
    d5 = ChainStore.load(name)
    assert d5["iteration"] == 23
    np.testing.assert_array_equal(d5["chain"][:18], d2["chain"])
    np.testing.assert_array_equal(d5["chain"][18:], blocks[0][0])
    st4.flush()
    # a chunk of 100 steps must stay below 4 GiB (32-bit chunk sizes): refused up front, not in the writer thread
    big = ChainStore(str(tmp_path / "big" / "chemcee_256.h5"))
    fake = np.lib.stride_tricks.as_strided(np.zeros(1), shape=(1, 1 << 22, 16), strides=(0, 0, 0))
    big.chain.append(fake); big.chain_transformed.append(fake); big.log_prob.append(fake[:, :, 0])
    with pytest.raises(h5lite.H5Error):
        big._spec(0)
    # a store that is only flushed at the end writes contiguous datasets in one pass
    st3 = ChainStore(str(tmp_path / "zeus_256.h5"))
    for z, th, lp in blocks:
        st3.append(z.astype(np.float32), th.astype(np.float32), lp.astype(np.float32), np.zeros(4))
    st3.flush()
    d4 = ChainStore.load(st3.h5)
    np.testing.assert_array_equal(d4["chain"], cat(0, 4).astype(np.float32))
    assert d4["chain"].dtype == np.float32


def test_bench_identity_emulators_are_the_identity():
    """bench.identity_state (the `hmc` object of the bench line: an emulator of the README problem that IS theory = identity,
    with random weights in every unit that does not carry x): the oracle's forward pass returns x to fp32 rounding for the
    4 x 512 MLP and for ChtoModelv2(33,33) on the whole prior range (|x| < 1.74 in normalised units), and most weights are
    non-trivial (the arithmetic of a real weight set, not a sparse toy)."""
    import bench
    from oracle import emulator
    x = np.random.RandomState(0).uniform(-1.74, 1.74, (64, 33)).astype(np.float32)
    for kind, kw in (("MLP", dict(width=512, depth=4)), ("ChtoModelv2", {})):
        sd = bench.identity_state(kind)
        h = emulator.forward(sd, x, kind, 33, 33, **kw)
        assert np.abs(h - x).max() <= 2.5e-7, kind
        nz = sum(int(np.count_nonzero(v)) for v in sd.values()) / float(sum(v.size for v in sd.values()))
        assert nz > 0.8, (kind, nz)


def test_nbest_points_are_designed_around_the_best_fit(tmp_path):
    """``nbest`` (main.py:140-152, util.py:1235-1252): a Nelder-Mead fit of the true theory, then draws from
    N(best fit, inverse Hessian) with the theory evaluated at them, as ``best_samples_*`` next to the designed points.
    numdifftools (the reference's Hessian) is absent: ``numerical_hessian`` is checked against the analytic one; the draws
    are unseeded in the reference as well, so the check is on the files and on where the points lie."""
    from linna_amd import util
    rs = np.random.RandomState(3)
    nd, nout = 3, 5
    A = rs.standard_normal((nout, nd))
    truth = np.array([0.3, -0.2, 0.1])
    data = A @ truth
    invcov = np.diag(1.0 / rs.uniform(0.01, 0.02, nout))

    def theory(x, outdir=None):
        return A @ np.asarray(x[1])

    def negloglike(x):
        d = data - theory([-1, x], None)
        return d.dot(invcov.dot(d))

    H = util.numerical_hessian(negloglike, truth)
    np.testing.assert_allclose(H, 2 * A.T @ invcov @ A, rtol=1e-5, atol=1e-6 * np.abs(H).max())
    out = str(tmp_path) + "/"
    sampler_ = util.NN_samplerv1(out, [[-1, 1]] * nd)
    util.generate_training_point(theory, sampler_, None, out, 40, 10, data, invcov, None, negloglike=negloglike, nbest_in=20)
    bx, by = np.loadtxt(out + "best_samples_x.txt"), np.load(out + "best_samples_y.npy")
    vx, vy = np.loadtxt(out + "best_samples_x_val.txt"), np.load(out + "best_samples_y_val.npy")
    assert bx.shape == (20, nd) and by.shape == (20, nout) and vx.shape == (5, nd) and vy.shape == (5, nout)
    np.testing.assert_allclose(by, bx @ A.T, rtol=1e-12)
    assert np.all(np.abs(bx.mean(0) - truth) < 0.2)                  # around the best fit, with the fit's own (tiny) errors
    # a second call finds the files and changes nothing (stage-level idempotence, util.py:1191-1233)
    util.generate_training_point(theory, sampler_, None, out, 40, 10, data, invcov, None, negloglike=negloglike, nbest_in=20)
    np.testing.assert_array_equal(bx, np.loadtxt(out + "best_samples_x.txt"))
    # and train_NN's loader puts them in front (util.py:1375-1409)
    tx, ty, vx2, vy2, _ = util._load_samples([out], usebest=True)
    assert len(tx) == 60 and len(vx2) == 15
    np.testing.assert_array_equal(tx[:20], bx)
    np.testing.assert_array_equal(vy2[:5], vy)


def test_numerical_hessian_with_a_noisy_theory_and_a_saddle():
    """the chi^2 of an external theory code carries its own noise, which a second difference amplifies by 1 / h^2.
    With ~1e-7 relative noise the step-doubling choice keeps the curvatures within a few per cent (a fixed 1e-4 step is off by
    factors); a best-fit point that is no minimum warns before makepositivedefinite reshapes the spectrum."""
    import warnings
    from linna_amd import util
    rs = np.random.RandomState(5)
    nd = 4
    A = rs.standard_normal((8, nd))
    x0 = np.array([0.4, -1.2, 0.05, 2.0])
    Q = A.T @ A
    noise = np.random.RandomState(9)

    def f(x):
        v = 50.0 + (x - x0) @ Q @ (x - x0)
        return v * (1.0 + 1e-7 * noise.standard_normal())
    H = util.numerical_hessian(f, x0, scale=np.full(nd, 4.0))             # (parameter scales: the widths of the priors)
    assert np.abs(np.diag(H) - 2 * np.diag(Q)).max() < 0.05 * np.abs(np.diag(Q)).max()
    assert np.abs(H - 2 * Q).max() < 0.1 * np.abs(Q).max()
    Hfixed = util.numerical_hessian(f, x0, rel_step=1e-5)                 # a small fixed step drowns in the same noise
    assert np.abs(np.diag(Hfixed) - 2 * np.diag(Q)).max() > np.abs(np.diag(H) - 2 * np.diag(Q)).max()
    saddle = lambda x: x[0] ** 2 - x[1] ** 2
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        util.numerical_hessian(saddle, np.zeros(2))
    assert any("not positive" in str(m.message) for m in w)


def test_slice_round_expectation_from_usage_counters():
    """``SliceEnsembleSampler.expected_points``: the engine hint of the later slice rounds (linna_slice_half_step's
    ``expect_rows``) from the device's usage counters -- mean walkers still active behind a round x the next round's points per
    walker x 1.25 + 8; the first round of each kind unused (1); a round that practically never runs 2^20; differences since
    the last look, not the whole history; nothing from fewer than 8 new calls."""
    import torch  # noqa: F401  (the module imports it)
    from linna_amd.sampler import SliceEnsembleSampler as S
    m_sched, nt_sched, half = [2, 4, 8], [4, 8, 16, 32], 2048
    nr = 7
    c = np.zeros(5 + 2 * nr)
    calls = 201                                          # (the latest call's counts are rolled by the next one: 200 counted)
    c[4 + 2 * nr] = calls
    c[4 + nr:4 + 2 * nr] = np.array([0.0037, 0.0, 0.0, 0.0846, 6e-5, 0.0, 0.0]) * 200 * half
    rows, base = S.expected_points(c, m_sched, nt_sched, half)
    assert rows[0] == 1 and rows[3] == 1
    assert rows[1] == int(8 * 0.0037 * half * 1.25) + 8 == 83                   # second stepping-out round: 2 x 4 ends per walker still active
    assert rows[2] == 1 << 20 and rows[6] == 1 << 20                            # nobody has ever been behind those
    assert rows[4] == int(8 * 0.0846 * half * 1.25) + 8 and 1400 < rows[4] < 2048   # second shrinking round: the 8-row engine
    assert rows[5] == int(16 * 6e-5 * half * 1.25) + 8 == 10
    assert base[0] == 200 and np.allclose(base[1], c[4 + nr:4 + 2 * nr])
    # the next look sees only what happened since: the walkers now finish earlier
    c2 = c.copy()
    c2[4 + 2 * nr] = 401
    c2[4 + nr:4 + 2 * nr] += np.array([0.0, 0.0, 0.0, 0.01, 0.0, 0.0, 0.0]) * 200 * half
    rows2, base2 = S.expected_points(c2, m_sched, nt_sched, half, base)
    assert rows2[1] == 1 << 20 and rows2[4] == int(8 * 0.01 * half * 1.25) + 8 and base2[0] == 400
    # too few new calls: no change
    c3 = c2.copy(); c3[4 + 2 * nr] = 405
    assert S.expected_points(c3, m_sched, nt_sched, half, base2) == (None, base2)
    # one stepping-out round, two shrinking rounds (the small ensembles): entry 1 is a first round, entry 2 the rescue
    c4 = np.zeros(5 + 2 * 3); c4[4 + 2 * 3] = 101
    rows4, _ = S.expected_points(c4, [8], [16, 16], 64)
    assert rows4 == [1, 1, 1 << 20]



def test_sample_text_files_are_parsed_once_and_reparsed_when_they_change(tmp_path):
    """``util._loadtxt_cached`` (train_NN re-reads the ``*_samples_x.txt`` of every earlier iteration, util.py:1346-1373): the
    parse is cached per (path, size, mtime); a rewritten file is parsed again; the array handed out is a private copy."""
    import time
    from linna_amd import util
    f = str(tmp_path / "train_samples_x.txt")
    a = np.arange(12.0).reshape(4, 3)
    np.savetxt(f, a)
    x1 = util._loadtxt_cached(f)
    x1[0, 0] = 99.0                                            # the caller may do what it likes with it
    np.testing.assert_array_equal(util._loadtxt_cached(f), a)
    time.sleep(0.01)
    np.savetxt(f, a[:2] + 0.5)                                 # another size and mtime
    np.testing.assert_array_equal(util._loadtxt_cached(f), a[:2] + 0.5)
