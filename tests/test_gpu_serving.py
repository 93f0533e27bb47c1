"""GPU parity (through the C ABI): fused GEMM, network forward, Log_prob and its gradient
against the numpy oracle and the golden vectors captured from the live reference."""
import ctypes as C
import os

import numpy as np
import pytest

import cases
from linna_amd import _lib

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a, dtype=None):
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda", dtype=dtype)


def run_gemm(**kw):
    from linna_amd import _lib
    g = _lib.Gemm()
    g.npairs = kw.get("npairs", 1)
    g.alpha0 = kw.get("alpha0", 1.0)
    for k, v in kw.items():
        if k.startswith("p0") or k.startswith("p1"):
            setattr(g.p[int(k[1])], k[3:], v)
        elif k not in ("npairs", "alpha0"):
            setattr(g, k, v)
    _lib.call("linna_gemm_f32", _lib.ctx(), C.byref(g), _lib.stream())
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (4096, 512, 512), (500, 1000, 33), (37, 33, 125), (1, 5, 7),
                                   (300, 16, 1000), (4096, 33, 512), (129, 65, 40)])
def test_gemm_nt_bias_relu(M, N, K):
    from linna_amd import _lib
    rs = np.random.RandomState(M + N + K)
    A = rs.standard_normal((M, K)).astype(np.float32)
    W = rs.standard_normal((N, K)).astype(np.float32)
    b = rs.standard_normal(N).astype(np.float32)
    dA, dW, db = dev(A), dev(W), dev(b)
    ldc = _lib.ld4(N)
    out = torch.full((M, ldc), 7.0, device="cuda")
    run_gemm(p0_A=dA.data_ptr(), p0_lda=K, p0_alay=0, p0_B=dW.data_ptr(), p0_ldb=K, p0_blay=0, p0_K=K, M=M, N=N,
             C=out.data_ptr(), ldc=ldc, bias0=db.data_ptr(), relu=1)
    ref = np.maximum(A.astype(np.float64) @ W.T.astype(np.float64) + b, 0)
    got = out.cpu().numpy()
    scale = np.abs(A).astype(np.float64) @ np.abs(W.T).astype(np.float64) + np.abs(b)
    assert np.all(np.abs(got[:, :N] - ref) <= 4e-7 * scale + 1e-6)
    if ldc > N:
        assert np.all(got[:, N:] == 7.0)          # padding columns untouched


def test_gemm_identity_asymmetric_layout_check():
    """A = I with an ASYMMETRIC B catches a transposed C write (cdna guide, section 3)."""
    n = 96
    A = np.eye(n, dtype=np.float32)
    Bm = (np.arange(n)[:, None] * 1000 + np.arange(n)[None, :]).astype(np.float32)   # B[n][k]
    out = torch.zeros((n, n), device="cuda")
    dA, dB = dev(A), dev(Bm)
    run_gemm(p0_A=dA.data_ptr(), p0_lda=n, p0_alay=0, p0_B=dB.data_ptr(), p0_ldb=n, p0_blay=0, p0_K=n, M=n, N=n,
             C=out.data_ptr(), ldc=n)
    np.testing.assert_array_equal(out.cpu().numpy(), Bm.T)
    # k-major B: C = A . B[k][n]
    run_gemm(p0_A=dA.data_ptr(), p0_lda=n, p0_alay=0, p0_B=dB.data_ptr(), p0_ldb=n, p0_blay=1, p0_K=n, M=n, N=n,
             C=out.data_ptr(), ldc=n)
    np.testing.assert_array_equal(out.cpu().numpy(), Bm)
    # k-major A as well: C = A^T-stored . B
    At = (np.arange(n)[:, None] * 3 + np.arange(n)[None, :] * 5).astype(np.float32) / 64.0     # stored [K][M]
    I = np.eye(n, dtype=np.float32)
    dAt, dI = dev(At), dev(I)
    run_gemm(p0_A=dAt.data_ptr(), p0_lda=n, p0_alay=1, p0_B=dI.data_ptr(), p0_ldb=n, p0_blay=1, p0_K=n, M=n, N=n,
             C=out.data_ptr(), ldc=n)
    np.testing.assert_array_equal(out.cpu().numpy(), At.T)


@pytest.mark.parametrize("M,N,K0,K1", [(500, 250, 32, 500), (130, 70, 16, 90), (4096, 500, 16, 1000)])
def test_gemm_dual_pair_resblock_epilogue(M, N, K0, K1):
    rs = np.random.RandomState(1)
    T = rs.standard_normal((M, K0)).astype(np.float32); W2 = rs.standard_normal((N, K0)).astype(np.float32)
    X = rs.standard_normal((M, K1)).astype(np.float32); Ws = (rs.standard_normal((N, K1)) / np.sqrt(K1)).astype(np.float32)
    b2 = rs.standard_normal(N).astype(np.float32)
    dT, dW2, dX, dWs, db2 = dev(T), dev(W2), dev(X), dev(Ws), dev(b2)
    out = torch.zeros((M, N), device="cuda")
    run_gemm(npairs=2, alpha0=0.1, p0_A=dT.data_ptr(), p0_lda=K0, p0_alay=0, p0_B=dW2.data_ptr(), p0_ldb=K0, p0_blay=0,
             p0_K=K0, p1_A=dX.data_ptr(), p1_lda=K1, p1_alay=0, p1_B=dWs.data_ptr(), p1_ldb=K1, p1_blay=0, p1_K=K1,
             M=M, N=N, C=out.data_ptr(), ldc=N, bias0=db2.data_ptr(), relu=1)
    ref = np.maximum(0.1 * (T.astype(np.float64) @ W2.T + b2) + X.astype(np.float64) @ Ws.T, 0)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("M,N,K", [(500, 33, 33), (4096, 457, 457), (77, 1000, 1000)])
def test_gemm_rowdot_matches_quadratic_form(M, N, K):
    from linna_amd import _lib
    rs = np.random.RandomState(2)
    D = rs.standard_normal((M, K)).astype(np.float32)
    S = rs.standard_normal((K, N)).astype(np.float32)
    slots = _lib.load().linna_gemm_dot_slots(M, N)
    part = torch.zeros((M, slots), device="cuda")
    dD, dS = dev(D), dev(S)
    run_gemm(p0_A=dD.data_ptr(), p0_lda=K, p0_alay=0, p0_B=dS.data_ptr(), p0_ldb=N, p0_blay=1, p0_K=K, M=M, N=N,
             dotwith=dD.data_ptr(), lddot=K, dot_partial=part.data_ptr(), dot_slots=slots)
    ref = ((D.astype(np.float64) @ S) * D).sum(-1)
    scale = (np.abs(D.astype(np.float64)) @ np.abs(S) * np.abs(D)).sum(-1)
    got = part.cpu().numpy().astype(np.float64).sum(-1)
    assert np.all(np.abs(got - ref) <= 1e-6 * scale)


def build_logprob(name, temperature=1.0, prob=None):
    from linna_amd import nn, util, predictor_gpu
    prob = cases.serving_problem(name) if prob is None else prob
    cls = {"ChtoModelv2": nn.ChtoModelv2, "ChtoModelsimple": nn.ChtoModelsimple,
           "ChtoModelv2_linear": nn.ChtoModelv2_linear, "MLP": nn.MLP}[prob["kind"]]
    model = cls(prob["nin"], prob["nout"], None, **prob["kw"])
    model.load_state_dict(prob["weights"])
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32))
    Xt = util.X_transform_class(t(prob["X_mean"]), t(prob["X_std"]), "cpu", prob["dolog10"])
    Yt = util.Y_transform_class(t(prob["y_mean"]), t(prob["y_std"]), "cpu", ypositive=prob["ypositive"])
    pred = predictor_gpu.Predictor(prob["nin"], prob["nout"], model=model, X_transform=Xt, y_transform=Yt, device="cuda")
    yinv = util.Y_invtransform_data(prob["sigma"], "cpu")
    lp = util.Log_prob(t(prob["data"]), t(prob["invcov"]), pred, yinv, util.Transform(prob["priors"]), temperature,
                       util.gaussianlogliklihood, nograd=True)
    return lp, pred, yinv, prob


@pytest.mark.parametrize("name", [c[0] for c in cases.SERVING])
def test_predict_and_logprob_match_reference(name):
    g = cases.golden(name)
    lp, pred, yinv, prob = build_logprob(name)
    m = yinv(pred.predict(torch.as_tensor(g["theta"]))).cpu().numpy()
    scale = np.abs(g["m"]).max()
    np.testing.assert_allclose(m, g["m"], rtol=8e-5, atol=8e-6 * scale)
    for j, T in enumerate(g["temps"]):
        lpT = build_logprob(name, float(T))[0]
        got = lpT(g["z"], returntorch=False)
        np.testing.assert_allclose(got, g["loglike"][:, j], rtol=2e-5)
        one = lpT(g["z"][3])                      # single-walker call: reference semantics (scalar)
        assert one.dim() == 0
        np.testing.assert_allclose(float(one), g["loglike"][3, j], rtol=1e-5)


@pytest.mark.parametrize("name", [c[0] for c in cases.SERVING if not c[8]])
def test_logprob_gradient_matches_autograd(name):
    g = cases.golden(name)
    lp = build_logprob(name)[0]
    z, _ = lp._to_device(g["z"])
    lnp, grad = lp.evaluate_with_grad(z)
    np.testing.assert_allclose(lnp.cpu().numpy(), g["loglike"][:, 0], rtol=1.5e-5)
    import parity
    parity.rowmax_close(grad.cpu().numpy(), g["grad"], 4e-5, 1.5e-7)


def test_oracle_agrees_at_large_batch():
    """Full-size batch (4096 walkers, BASELINE config 2 shape) against the numpy oracle."""
    from oracle import likelihood
    lp, pred, yinv, prob = build_logprob("mlp_33_33")
    emu = cases.oracle_emulator(prob)
    z = np.random.RandomState(3).standard_normal((4096, 33)).astype(np.float32)
    got = lp(z, returntorch=False)
    ref = likelihood.log_prob(z, emu, prob["priors"], prob["data"], prob["invcov"], 1.0, dtype=np.float64)
    np.testing.assert_allclose(got, ref, rtol=1.5e-5)


def test_very_large_ragged_batch():
    """70 001 walkers in one call (4376 workgroups of the 16-row engine, the last one a single row; row indices past
    2^16): every row against the oracle on a sample, and the rows of the first and last workgroups bit for bit against
    the same rows evaluated as a batch of their own on the same engine."""
    from oracle import likelihood
    lp, pred, yinv, prob = build_logprob("v2_33_33")
    emu = cases.oracle_emulator(prob)
    n = 70001
    z = np.random.RandomState(11).standard_normal((n, 33)).astype(np.float32)
    zd = torch.as_tensor(z, device="cuda")
    out = torch.empty(n, device="cuda")
    lp.evaluate(zd, out=out)
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    idx = np.r_[0:64, 65500:65600, n - 64:n]
    ref = likelihood.log_prob(z[idx], emu, prob["priors"], prob["data"], prob["invcov"], 1.0, dtype=np.float64)
    np.testing.assert_allclose(got[idx], ref, rtol=4e-6)
    # a batch of 8192 rows runs the same 16-row engine: same rows, same bits, wherever they sit in the big batch
    lo = 61440                                                    # a multiple of 16: the same row <-> lane map
    part = torch.empty(8192, device="cuda")
    lp.evaluate(zd[lo:lo + 8192].contiguous(), out=part)
    np.testing.assert_array_equal(part.cpu().numpy(), got[lo:lo + 8192])


def test_reference_fixture_known_answers():
    """The reference's own test fixture (tests/test_main.py:47-51 reads it): checkpoint and
    transform pickles load through the product's retrieve_model; log-probabilities match."""
    from linna_amd import util, nn
    g = cases.golden("fixture2d")
    outdir = os.path.join(cases.GOLDEN, "2dgaussian_Fulltconn/iter_0/")
    model, yinv = util.retrieve_model(outdir, 2, 2, nn.ChtoModelv2)
    priors = [{"param": "test_%d" % i, "dist": "flat", "arg1": -2.0, "arg2": 2.0} for i in range(2)]
    lp = util.Log_prob(np.array([0.1, 1.0]), np.linalg.inv(np.diag([0.5, 0.2])), model, yinv, util.Transform(priors),
                       1.0, util.gaussianlogliklihood, nograd=True)
    got = lp(g["z"], returntorch=False)
    np.testing.assert_allclose(got, g["loglike"], rtol=5e-6, atol=5e-7)
    np.testing.assert_allclose(got[:4], [-2.9208457, -3.2061472, -4.2610073, -4.2800837], rtol=3e-6)


def test_nan_maps_to_minus_inf_and_empty_edge():
    lp = build_logprob("simple_6_4")[0]
    z = np.zeros((3, 6), np.float32)
    z[1, 2] = np.nan
    out = lp(z, returntorch=False)
    assert np.isfinite(out[0]) and out[1] == -np.inf and np.isfinite(out[2])
    with pytest.raises(ValueError):
        lp(np.zeros((2, 5), np.float32))
    assert lp(np.zeros((0, 6), np.float32), returntorch=False).shape == (0,)      # empty batch
    from linna_amd import _lib
    with pytest.raises(_lib.LinnaHipError):                                         # the C ABI rejects B < 1 loudly
        lp.evaluate(torch.zeros((0, 6), device="cuda"))
    # device entry points take float32 rows with unit column stride (any row stride): anything else is refused, not misread
    zt = torch.randn(6, 8, device="cuda")
    with pytest.raises(_lib.LinnaHipError):
        lp.evaluate(zt.t())                                                         # a transposed view
    with pytest.raises(_lib.LinnaHipError):
        lp.evaluate(torch.zeros((4, 6), device="cuda", dtype=torch.float64))
    with pytest.raises(_lib.LinnaHipError):
        lp.evaluate_with_grad(zt.t())
    wide = torch.randn(5, 12, device="cuda")                                        # rows of a wider buffer: row stride 12
    np.testing.assert_array_equal(lp.evaluate(wide[:, :6]).cpu().numpy(), lp.evaluate(wide[:, :6].contiguous()).cpu().numpy())


def test_whole_network_kernel_edges_and_agreement_with_layered_path():
    """The whole-network kernel (used by evaluate for eligible MLPs) against the layer-by-layer
    path (used by evaluate_with_grad) and the oracle: ragged batch, NaN row, theta output."""
    from oracle import likelihood
    lp, pred, yinv, prob = build_logprob("mlp_33_33")
    emu = cases.oracle_emulator(prob)
    for B in (1, 15, 16, 37, 1000):
        z = np.random.RandomState(B).standard_normal((B, 33)).astype(np.float32)
        if B > 2:
            z[2, 5] = np.nan
        zd = torch.as_tensor(z, device="cuda")
        theta = torch.empty_like(zd)
        fused = lp.evaluate(zd, theta=theta).cpu().numpy()
        layered, _ = lp.evaluate_with_grad(zd)
        layered = layered.cpu().numpy()
        ref = likelihood.log_prob(z, emu, prob["priors"], prob["data"], prob["invcov"], 1.0)
        ok = np.isfinite(ref)
        np.testing.assert_allclose(fused[ok], ref[ok], rtol=1.5e-5)
        np.testing.assert_allclose(fused[ok], layered[ok], rtol=2e-6)
        assert np.all(fused[~ok] == -np.inf)
        th_ref = likelihood.prior_map(z, prob["priors"])
        np.testing.assert_allclose(theta.cpu().numpy()[ok], th_ref[ok], rtol=1e-5, atol=1e-5)
    # dense inverse covariance goes through the fused network + MFMA row-dot
    lpd, _, _, probd = build_logprob("mlp_33_33_dense")
    z = np.random.RandomState(9).standard_normal((300, 33)).astype(np.float32)
    got = lpd(z, returntorch=False)
    ref = likelihood.log_prob(z, cases.oracle_emulator(probd), probd["priors"], probd["data"], probd["invcov"], 1.0,
                              dtype=np.float64)
    np.testing.assert_allclose(got, ref, rtol=1e-5)


def test_ypositive_output_map_runs_in_the_whole_network_kernel(monkeypatch):
    """``Y_transform_class(ypositive=True)`` (util.py:532-542: exp of the affine map) is evaluated in the finish of the
    whole-network kernel, so such emulators keep the one-launch evaluation and the one-launch sampler moves: against the
    reference's golden values, the layer-by-layer path and the three-launch half step (bit for bit)."""
    from linna_amd import sampler
    lp, pred, yinv, prob = build_logprob("v2_4_2_ypos")
    g = cases.golden("v2_4_2_ypos")
    got = lp(g["z"], returntorch=False)
    np.testing.assert_allclose(got, g["loglike"][:, 0], rtol=3e-6, atol=5e-8)
    z = torch.as_tensor(np.random.RandomState(4).standard_normal((777, 4)).astype(np.float32) * 0.4, device="cuda")
    fused = lp.evaluate(z).cpu().numpy()
    monkeypatch.setenv("LINNA_DISABLE_FUSED", "1")
    lp2 = build_logprob("v2_4_2_ypos")[0]
    layered = lp2.evaluate(z).cpu().numpy()
    monkeypatch.delenv("LINNA_DISABLE_FUSED")
    np.testing.assert_allclose(fused, layered, rtol=2e-6, atol=8e-8)
    x0 = np.random.RandomState(1).standard_normal((64, 4)).astype(np.float32) * 0.3
    a = sampler.EnsembleSampler(64, 4, lp, seed=3, randomize_split=False)
    b = sampler.EnsembleSampler(64, 4, lp, seed=3, randomize_split=False, fused=False)
    a.set_state(x0); b.set_state(x0)
    for _ in range(5):
        a.step(); b.step()
    assert a.fused is True and b.fused is False
    assert torch.equal(a.coords, b.coords) and torch.equal(a.logp, b.logp)


@pytest.mark.parametrize("nin,nout", [(5, 3), (26, 40), (33, 33)])
def test_input_skip_network_runs_in_the_whole_network_kernel(nin, nout, monkeypatch):
    """``ChtoModelv2_linear`` (nn.py:136-198: ``layer8(h) + 1e-3 linearlayer(x)``, the class cosmolike_run.py:193 can
    select by name): the input skip is the second K part of the last layer's GEMM -- the network input rows are kept in
    LDS and copied behind h -- so the model keeps the one-launch evaluation and sampler moves.  Against the oracle, the
    layer-by-layer path, and the three-launch half step (bit for bit); the reference's golden values (dense covariance:
    whole-network kernel + row-dot GEMM) are checked by test_predict_and_logprob_match_reference[v2lin_5_3_log10]."""
    import synth
    from oracle import likelihood
    from linna_amd import sampler
    seed = 300 + nin
    data, cov, priors = synth.gaussian_problem(nin, nout, seed, dense=False)
    X_mean, X_std, y_mean, y_std = synth.transform_constants(nin, nout, seed)
    w = synth.weights("ChtoModelv2_linear", nin, nout, seed)
    w["linearlayer.weight"] = (w["linearlayer.weight"] * 30).astype(np.float32)        # make the 1e-3 branch count
    prob = dict(kind="ChtoModelv2_linear", nin=nin, nout=nout, kw={}, weights=w, priors=priors, data=data, cov=cov,
                invcov=np.linalg.inv(cov), sigma=np.sqrt(np.diag(cov)), X_mean=X_mean, X_std=X_std, y_mean=y_mean, y_std=y_std,
                dolog10=None, ypositive=False)
    lp = build_logprob(None, 1.0, prob)[0]
    z = np.random.RandomState(2).standard_normal((333, nin)).astype(np.float32) * 0.5
    zd = torch.as_tensor(z, device="cuda")
    fused = lp.evaluate(zd).cpu().numpy()
    ref = likelihood.log_prob(z, cases.oracle_emulator(prob), priors, data, prob["invcov"], 1.0)
    np.testing.assert_allclose(fused, ref, rtol=2e-5, atol=3e-6)
    monkeypatch.setenv("LINNA_DISABLE_FUSED", "1")
    layered = build_logprob(None, 1.0, prob)[0].evaluate(zd).cpu().numpy()
    monkeypatch.delenv("LINNA_DISABLE_FUSED")
    np.testing.assert_allclose(fused, layered, rtol=2e-6, atol=8e-7)
    # without the skip the values differ: the branch is really in the kernel
    w0 = dict(w); w0["linearlayer.weight"] = np.zeros_like(w["linearlayer.weight"]); w0["linearlayer.bias"] = np.zeros_like(w["linearlayer.bias"])
    noskip = build_logprob(None, 1.0, dict(prob, weights=w0))[0].evaluate(zd).cpu().numpy()
    assert np.abs(noskip - fused).max() > 1e-3 * np.abs(fused).max()
    x0 = np.random.RandomState(1).standard_normal((64, nin)).astype(np.float32) * 0.3
    a = sampler.EnsembleSampler(64, nin, lp, seed=3, randomize_split=False)
    b = sampler.EnsembleSampler(64, nin, lp, seed=3, randomize_split=False, fused=False)
    a.set_state(x0); b.set_state(x0)
    for _ in range(4):
        a.step(); b.step()
    assert a.fused is True and b.fused is False
    assert torch.equal(a.coords, b.coords) and torch.equal(a.logp, b.logp)


def _custom_problem(nin, nout, seed, width, depth, dense=False):
    """A serving problem outside cases.SERVING (no golden file: checked against the oracle)."""
    import synth
    data, cov, priors = synth.gaussian_problem(nin, nout, seed, dense=dense)
    X_mean, X_std, y_mean, y_std = synth.transform_constants(nin, nout, seed)
    kw = {"width": width, "depth": depth}
    return dict(kind="MLP", nin=nin, nout=nout, kw=kw, weights=synth.weights("MLP", nin, nout, seed, **kw),
                priors=priors, data=data, cov=cov, invcov=np.linalg.inv(cov), sigma=np.sqrt(np.diag(cov)),
                X_mean=X_mean, X_std=X_std, y_mean=y_mean, y_std=y_std, dolog10=None, ypositive=False)


@pytest.mark.parametrize("nin,nout,width,depth,which", [
    (33, 33, 512, 4, "big"),         # bench shape: WIDE x4 + SPLIT(8-way K) last layer
    (64, 64, 512, 2, "big"),         # 64 inputs / outputs, 3 linear layers
    (3, 1, 512, 1, "big"),           # two linear layers, one output column
    (65, 40, 512, 3, "small"),       # 65 inputs: the prologue's wide-input loop
    (200, 33, 512, 1, "small"),      # 200 inputs, theta recomputed at the end
    (20, 33, 256, 3, "small"),       # hidden width 256: SPLIT with 4 column groups x 2 K parts
    (12, 100, 128, 2, "small"),      # width 128 and 100 outputs: SPLIT with 2 column groups x 4 K parts
    (9, 700, 1000, 2, "small"),      # widths > 512: two-pass WIDE segments, wide final layer (d via LDS)
])
def test_whole_network_kernels_against_oracle(nin, nout, width, depth, which, monkeypatch):
    """The whole-network kernel (net_stream.hip) over the segment shapes its program builder can emit,
    against the oracle and the layer-by-layer path, ragged batches included, with each of its three
    engines (16 rows per workgroup on 16x16x4 MFMAs; 8 and 4 rows on 4x4x1) forced in turn and with the
    engine the batch size selects."""
    from oracle import likelihood
    from linna_amd import _lib
    prob = _custom_problem(nin, nout, 900 + nin + width, width, depth)
    lp, pred, yinv, _ = build_logprob(None, prob=prob)
    emu = cases.oracle_emulator(prob)
    for B in (1, 17, 4096 if which == "big" else 300):
        z = (0.7 * np.random.RandomState(B).standard_normal((B, nin))).astype(np.float32)
        zd = torch.as_tensor(z, device="cuda")
        ref = likelihood.log_prob(z, emu, prob["priors"], prob["data"], prob["invcov"], 1.0)
        monkeypatch.setenv("LINNA_DISABLE_FUSED_GRAD", "1")
        layered = build_logprob(None, prob=prob)[0].evaluate_with_grad(zd)[0].cpu().numpy()
        monkeypatch.delenv("LINNA_DISABLE_FUSED_GRAD")
        for rows in (None, 4, 8, 16):
            if rows is None:
                _lib.engine_rows(0)
            else:
                _lib.engine_rows(int(rows))
            theta = torch.empty_like(zd)
            got = lp.evaluate(zd, theta=theta).cpu().numpy()
            np.testing.assert_allclose(got, ref, rtol=2e-5, atol=4e-5, err_msg="rows %s B %d" % (rows, B))
            np.testing.assert_allclose(got, layered, rtol=1.5e-5, atol=8e-5, err_msg="rows %s B %d" % (rows, B))
            np.testing.assert_allclose(theta.cpu().numpy(), likelihood.prior_map(z, prob["priors"]), rtol=1e-5, atol=1e-5)
        _lib.engine_rows(0)


def test_stream_kernel_follows_weight_updates():
    """The whole-network kernel reads a fragment-order COPY of the weights: the copy must follow an
    AdamW step (C entry), a torch-side write through flat_params()/state_dict()/load_state_dict(),
    and a graph replay that holds an AdamW step."""
    from oracle import likelihood
    from linna_amd import _lib
    prob = _custom_problem(33, 33, 77, 512, 4)
    lp, pred, yinv, _ = build_logprob(None, prob=prob)
    model = pred.model
    z = (0.5 * np.random.RandomState(3).standard_normal((64, 33))).astype(np.float32)
    zd = torch.as_tensor(z, device="cuda")

    def check():
        sd = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
        p2 = dict(prob, weights=sd)
        ref = likelihood.log_prob(z, cases.oracle_emulator(p2), prob["priors"], prob["data"], prob["invcov"], 1.0)
        got = lp.evaluate(zd).cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=1.5e-5, atol=3e-5)
        return got

    a = check()
    model.flat_params().mul_(1.01)                       # torch-side write through the accessor
    b = check()
    assert np.abs(a - b).max() > 1e-3
    sd = {k: 0.98 * v.detach().cpu() for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    c = check()
    assert np.abs(c - b).max() > 1e-3
    # AdamW through the C entry (lr large enough to move the output)
    n = model.flat_params().numel()
    g = torch.ones(n, device="cuda"); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    hyper = torch.tensor([1e-3, 0.0, 0.0, 0.0], device="cuda"); step = torch.zeros(1, dtype=torch.int32, device="cuda")
    flat = model.flat_params()
    lp.evaluate(zd)                                      # copy is fresh here
    _lib.call("linna_adamw_step", _lib.ctx(0), _lib.ptr(flat), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v),
              C.c_size_t(n), _lib.ptr(hyper), _lib.iptr(step), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-8), 0,
              _lib.stream())
    d = check()
    assert np.abs(d - c).max() > 1e-3


@pytest.mark.parametrize("nin,nout,width,depth", [(33, 33, 512, 4), (5, 3, 300, 2), (64, 64, 1024, 1), (12, 40, 700, 3), (10, 5, 600, 2)])
def test_fused_gradient_against_layered_path_and_oracle(nin, nout, width, depth, monkeypatch):
    """lnP and d lnP / d z in one launch (forward segments, turnaround, backward segments over W^T) against the
    layer-by-layer forward + dX chain and the oracle's analytic gradient; ragged batches, log10 inputs."""
    from oracle import likelihood
    prob = _custom_problem(nin, nout, 500 + nin + width, width, depth)
    prob["dolog10"] = [0] if nin > 4 else None
    if prob["dolog10"]:
        prob["priors"][0] = {"param": "p0", "dist": "flat", "arg1": 0.1, "arg2": 2.0}
    fused = build_logprob(None, 2.0, prob=prob)[0]
    fused._ensure()                                   # linna_logprob_create reads the switch: create inside the window
    monkeypatch.setenv("LINNA_DISABLE_FUSED_GRAD", "1")
    layered = build_logprob(None, 2.0, prob=prob)[0]
    layered._ensure()
    monkeypatch.delenv("LINNA_DISABLE_FUSED_GRAD")
    emu = cases.oracle_emulator(prob)
    for B, rows in ((1, None), (17, None), (1000, None), (1000, 8), (300, 16), (4100, 4)):
        if rows is None:
            _lib.engine_rows(0)     # the engine the batch size selects
        else:
            _lib.engine_rows(int(rows))
        z = (0.6 * np.random.RandomState(B).standard_normal((B, nin))).astype(np.float32)
        zd = torch.as_tensor(z, device="cuda")
        lf, gf = fused.evaluate_with_grad(zd)
        ll, gl = layered.evaluate_with_grad(zd)
        lf, gf, ll, gl = lf.cpu().numpy(), gf.cpu().numpy(), ll.cpu().numpy(), gl.cpu().numpy()
        scale = np.abs(gl).max()
        np.testing.assert_allclose(lf, ll, rtol=5e-6, atol=2e-5)
        np.testing.assert_allclose(gf, gl, rtol=2e-6, atol=2e-7 * scale)
        _, gref = likelihood.grad_log_prob(z.astype(np.float64), emu, prob["priors"], prob["data"], prob["invcov"], 2.0,
                                           dtype=np.float64)
        np.testing.assert_allclose(gf, gref, rtol=1e-4, atol=1e-5 * scale)
    _lib.engine_rows(0)
    # the two objects really took different routes (meaningful on the big shape only; best of several
    # short runs: a stray hipFree from garbage collection in the middle of a run costs milliseconds)
    if width == 512 and depth == 4:
        import gc, time
        gc.collect()
        zd = torch.randn(4096, nin, device="cuda") * 0.5
        out = torch.empty(4096, device="cuda"); grad = torch.empty(4096, nin, device="cuda")

        def t(lp):
            best = 1e9
            for _ in range(6):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(5):
                    lp.evaluate_with_grad(zd, out=out, grad=grad)
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            return best
        assert t(fused) < 0.9 * t(layered)        # (one launch against forward-with-stores + dX chain + prior-map kernels)


@pytest.mark.parametrize("name", ["mlp_33_33", "v2_33_33", "v2_26_457", "v2_40_1000"])
def test_full_size_properties(name):
    """BASELINE sizes (4096 walkers; configs 2, 3, 4 shapes), checked through properties that do not
    need the oracle at that size: walkers are independent (a permutation of the rows permutes the
    outputs bit for bit, a walker's value does not depend on its neighbours or on the batch size), the
    temperature enters as -chi2/(2T) - |z|^2/2 exactly, and a sample of rows agrees with the oracle."""
    from oracle import likelihood
    B = 4096
    lp1, pred, yinv, prob = build_logprob(name, 1.0)
    nin = prob["nin"]
    rs = np.random.RandomState(11)
    z = rs.standard_normal((B, nin)).astype(np.float32)
    base = lp1(z, returntorch=False)
    assert base.shape == (B,) and np.all(np.isfinite(base))
    perm = rs.permutation(B)
    np.testing.assert_array_equal(lp1(z[perm], returntorch=False), base[perm])
    # ragged batch sizes around the row tiles.  The same rows give the same bits while the same engine runs;
    # smaller batches take the 8- and 4-row engines of the whole-network kernel, which sum k in another order
    # (and with a dense inverse covariance the row-dot GEMM picks its tiling and K split by batch size)
    for n in (1, 15, 17, 1000, 4095):
        got = lp1(z[:n], returntorch=False).reshape(-1)
        if name in ("mlp_33_33", "v2_33_33") and n > 2048:              # same engine as the full batch: same bits
            np.testing.assert_array_equal(got, base[:n])
        else:
            np.testing.assert_allclose(got, base[:n], rtol=1.5e-5)
    # one walker repeated in every row
    rep = lp1(np.repeat(z[7:8], 257, axis=0), returntorch=False)
    assert np.all(rep == rep[0])
    np.testing.assert_allclose(rep[0], base[7], rtol=5e-6)
    # temperature: lnP_T + |z|^2/2 = (lnP_1 + |z|^2/2) / T
    half = 0.5 * np.sum(z.astype(np.float64) ** 2, axis=1)
    for T in (4.0, 16.0):
        lpT = build_logprob(name, T)[0](z, returntorch=False)
        np.testing.assert_allclose((lpT + half) * T, base + half, rtol=8e-6, atol=8e-5)
    # a sample of rows against the oracle (float64 accumulation)
    idx = rs.choice(B, 96, replace=False)
    ref = likelihood.log_prob(z[idx], cases.oracle_emulator(prob), prob["priors"], prob["data"], prob["invcov"], 1.0,
                              dtype=np.float64)
    np.testing.assert_allclose(base[idx], ref, rtol=8e-6)


@pytest.mark.parametrize("name", ["mlp_33_33", "v2_33_33"])
def test_full_size_gradient_properties(name):
    """Config 5 shape (4096 chains): the gradient entry returns the same lnP as the evaluation entry, rows
    are independent, and a sample of rows agrees with the oracle's reverse pass."""
    B = 4096
    lp = build_logprob(name)[0]
    nin = 33
    rs = np.random.RandomState(12)
    z = (0.5 * rs.standard_normal((B, nin))).astype(np.float32)
    zd, _ = lp._to_device(z)
    lnp, g = lp.evaluate_with_grad(zd)
    lnp, g = lnp.cpu().numpy().copy(), g.cpu().numpy()[:, :nin].copy()
    np.testing.assert_allclose(lnp, lp(z, returntorch=False), rtol=4e-6, atol=4e-6)
    perm = rs.permutation(B)
    zp, _ = lp._to_device(z[perm])
    lnp_p, g_p = lp.evaluate_with_grad(zp)
    np.testing.assert_array_equal(lnp_p.cpu().numpy(), lnp[perm])
    np.testing.assert_array_equal(g_p.cpu().numpy()[:, :nin], g[perm])
    # a sample of rows against the oracle's reverse pass
    from oracle import likelihood
    prob = cases.serving_problem(name)
    idx = rs.choice(B, 64, replace=False)
    lref, gref = likelihood.grad_log_prob(z[idx], cases.oracle_emulator(prob), prob["priors"], prob["data"],
                                          prob["invcov"], 1.0)
    np.testing.assert_allclose(lnp[idx], lref, rtol=1e-5)
    np.testing.assert_allclose(g[idx], gref, rtol=0, atol=2e-5 * np.abs(gref).max())


@pytest.mark.parametrize("nin,nout,width,depth", [
    (33, 33, 512, 4),        # U = d S as a SPLIT segment behind d (nout <= 64)
    (12, 100, 128, 2),       # two column groups
    (20, 250, 256, 3),       # four column groups
    (9, 457, 512, 2),        # WIDE, one pass: d stays in the other buffer
    (9, 700, 1000, 2),       # WIDE, two passes
])
def test_dense_covariance_as_last_segment(nin, nout, width, depth, monkeypatch):
    """Dense inverse covariance: the output map folded into the stream's last layer and S appended as the last
    segment of the whole-network kernel (lnP in one launch), against the oracle and against the network launch +
    row-dot GEMM it replaces, with each engine and ragged batches; the fused stretch move runs on it too."""
    from oracle import likelihood
    from linna_amd import sampler
    prob = _custom_problem(nin, nout, 300 + nin + nout, width, depth, dense=True)
    fused = build_logprob(None, 4.0, prob=prob)[0]
    fused._ensure()
    monkeypatch.setenv("LINNA_DENSE_FUSED", "0")          # read when the log-probability object is created
    unfused = build_logprob(None, 4.0, prob=prob)[0]
    unfused._ensure()
    monkeypatch.delenv("LINNA_DENSE_FUSED")
    emu = cases.oracle_emulator(prob)
    for B in (1, 17, 300):
        z = (0.7 * np.random.RandomState(B).standard_normal((B, nin))).astype(np.float32)
        zd = torch.as_tensor(z, device="cuda")
        ref = likelihood.log_prob(z, emu, prob["priors"], prob["data"], prob["invcov"], 4.0, dtype=np.float64)
        base = unfused.evaluate(zd).cpu().numpy()
        np.testing.assert_allclose(base, ref, rtol=2e-5)
        for rows in (None, 4, 8, 16):
            if rows is None:
                _lib.engine_rows(0)
            else:
                _lib.engine_rows(int(rows))
            theta = torch.empty_like(zd)
            got = fused.evaluate(zd, theta=theta).cpu().numpy()
            np.testing.assert_allclose(got, ref, rtol=2e-5, err_msg="rows %s B %d" % (rows, B))
            np.testing.assert_allclose(got, base, rtol=2e-5, err_msg="rows %s B %d" % (rows, B))
            np.testing.assert_allclose(theta.cpu().numpy(), likelihood.prior_map(z, prob["priors"]), rtol=1e-5, atol=1e-5)
    _lib.engine_rows(0)
    # one-launch half step on the dense problem, bit-identical to propose / evaluate / accept
    nw = 70
    x0 = (0.3 * np.random.RandomState(5).standard_normal((nw, nin))).astype(np.float32)
    a = sampler.EnsembleSampler(nw, nin, fused, seed=21)
    b = sampler.EnsembleSampler(nw, nin, fused, seed=21, fused=False)
    a.set_state(x0); b.set_state(x0)
    for _ in range(4):
        a.step(); b.step()
    torch.cuda.synchronize()
    assert a.fused is True
    assert torch.equal(a.coords, b.coords) and torch.equal(a.logp, b.logp) and torch.equal(a.naccept, b.naccept)


@pytest.mark.parametrize("nout,how", [(16, "direct"), (33, "direct"), (64, "direct"), (33, "singular")])
def test_dense_covariance_in_the_direct_form_as_a_side_segment(nout, how, monkeypatch):
    """chi^2 = d S d^T with S itself as the program's last segment (not its Cholesky factor): what a caller gets with
    LINNA_DENSE_FACTORED=0, and what a singular / indefinite inverse covariance gets by itself (the factorisation fails,
    util.py:953-955 has no such requirement).  For nout <= 64 that segment is a SIDE segment of the 16-row serving engine
    (outside the weight stream, net_stream.hip `last_ok`): every engine against the float64 oracle and the layered path."""
    from oracle import likelihood
    nin = 8
    prob = _custom_problem(nin, nout, 900 + nout, 64, 2, dense=True)
    if how == "singular":
        w, v = np.linalg.eigh(prob["invcov"])
        w[:3] = 0.0                                                   # rank-deficient: Cholesky fails, the direct form serves
        prob["invcov"] = (v * w[None, :]) @ v.T
    else:
        monkeypatch.setenv("LINNA_DENSE_FACTORED", "0")
    fused = build_logprob(None, 1.0, prob=prob)[0]
    p = fused._ensure()
    assert "Sfac" not in p["keep"]
    monkeypatch.setenv("LINNA_DENSE_FUSED", "0")
    unfused = build_logprob(None, 1.0, prob=prob)[0]
    unfused._ensure()
    monkeypatch.delenv("LINNA_DENSE_FUSED")
    from linna_amd import nn
    n, txt = nn.describe_program(fused.model.model, 16, dense_nout=nout)
    assert txt.strip().splitlines()[n].startswith("SIDE"), txt                   # the covariance segment of the 16-row program
    emu = cases.oracle_emulator(prob)
    for B in (1, 17, 300):
        z = (0.7 * np.random.RandomState(B).standard_normal((B, nin))).astype(np.float32)
        zd = torch.as_tensor(z, device="cuda")
        ref = likelihood.log_prob(z, emu, prob["priors"], prob["data"], prob["invcov"], 1.0, dtype=np.float64)
        base = unfused.evaluate(zd).cpu().numpy()
        np.testing.assert_allclose(base, ref, rtol=2e-5)
        for rows in (0, 4, 8, 16):
            _lib.engine_rows(rows)
            got = fused.evaluate(zd).cpu().numpy()
            np.testing.assert_allclose(got, ref, rtol=2e-5, err_msg="rows %s B %d" % (rows, B))
            np.testing.assert_allclose(got, base, rtol=2e-5, err_msg="rows %s B %d" % (rows, B))
    _lib.engine_rows(0)


@pytest.mark.parametrize("nout,dense", [(1100, False), (1100, True)])
def test_networks_wider_than_the_whole_network_kernel(nout, dense):
    """``ChtoModelv2(nin, nout > 1024)`` -- layer8 is nout x nout -- is outside the whole-network kernel (layers <= 1024 wide):
    the layer-by-layer GEMM path serves evaluation and gradient.  lnP of 1000 walkers against the float64 oracle; the
    gradient row-wise, with THE ReLU-kink exception (tests/parity.py near_relu_kink: with 3000 hidden units about one row
    in a thousand has a unit within fp32 rounding of zero) -- every row beyond the tolerance must be such a row, and there
    are at most 0.5 % of them."""
    import synth
    import parity
    from oracle import likelihood
    nin, seed = 12, 900 + nout
    data, cov, priors = synth.gaussian_problem(nin, nout, seed, dense=dense, cond=1e2)
    X_mean, X_std, y_mean, y_std = synth.transform_constants(nin, nout, seed)
    prob = dict(kind="ChtoModelv2", nin=nin, nout=nout, kw={}, weights=synth.weights("ChtoModelv2", nin, nout, seed), priors=priors,
                data=data, cov=cov, invcov=np.linalg.inv(cov), sigma=np.sqrt(np.diag(cov)), X_mean=X_mean, X_std=X_std, y_mean=y_mean,
                y_std=y_std, dolog10=None, ypositive=False)
    lp = build_logprob(None, 1.0, prob)[0]
    emu = cases.oracle_emulator(prob)
    B = 1000
    z = np.random.RandomState(B).standard_normal((B, nin)).astype(np.float32) * 0.5
    got = lp(z, returntorch=False)
    ref, gref = likelihood.grad_log_prob(z, emu, priors, data, prob["invcov"], 1.0, dtype=np.float64)
    np.testing.assert_allclose(got, ref, rtol=1e-5)
    zd, _ = lp._to_device(z)
    lnp, g = lp.evaluate_with_grad(zd)
    np.testing.assert_allclose(lnp.cpu().numpy(), ref, rtol=1e-5)
    rowerr = (np.abs(g.cpu().numpy() - gref) / np.abs(gref).max(axis=1, keepdims=True)).max(axis=1)
    beyond = np.where(rowerr > 5e-5)[0]
    assert len(beyond) <= B // 200, (len(beyond), rowerr.max())
    for r in beyond:
        assert parity.near_relu_kink(z[r], emu, priors), "row %d differs by %.2e of its maximum away from any ReLU kink" % (r, rowerr[r])


@pytest.mark.parametrize("nout", [513, 640, 961, 1000, 1024])
def test_dense_factor_skips_its_zero_triangle(nout):
    """Dense inverse covariances wider than 512: the log-likelihood segment multiplies by the lower-triangular Cholesky
    factor.  Mode 1: the second column pass (columns >= 512) starts at row 512 -- ``nout = 513`` leaves that pass ONE step,
    1024 the full 32.  Mode 2 (default), 960 < nout <= 1024: the balanced assignment -- wave w multiplies the 64-column
    blocks w and 15 - w, each from its first non-zero row (per-wave run lengths, 2 steps - 60 steps in every wave).
    lnP against the float64 oracle on every engine, and BIT-IDENTICAL between the three modes (the skipped products are
    zeros; same sums in the same order)."""
    from oracle import likelihood
    from linna_amd import _lib
    prob = _custom_problem(10, nout, 700 + nout, 64, 1, dense=True)
    emu = cases.oracle_emulator(prob)
    z = np.random.RandomState(nout).standard_normal((300, 10)).astype(np.float32) * 0.5
    ref = likelihood.log_prob(z, emu, prob["priors"], prob["data"], prob["invcov"], 1.0, dtype=np.float64)
    lib = _lib.load()
    mode0 = lib.linna_dense_tri(-1)
    assert mode0 == 2
    try:
        for rows in (16, 8, 4):
            prev = _lib.engine_rows(rows)
            got = {}
            try:
                for mode in (0, 1, 2):
                    lib.linna_dense_tri(mode)                       # (applies to objects created afterwards)
                    lp = build_logprob(None, 1.0, prob)[0]
                    got[mode] = lp(z, returntorch=False)
            finally:
                _lib.engine_rows(prev)
            np.testing.assert_allclose(got[2], ref, rtol=2e-5, err_msg="engine %d" % rows)
            np.testing.assert_array_equal(got[1], got[0], err_msg="engine %d" % rows)
            np.testing.assert_array_equal(got[2], got[0], err_msg="engine %d" % rows)
    finally:
        lib.linna_dense_tri(mode0)
