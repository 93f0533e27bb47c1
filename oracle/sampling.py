"""Oracle: counter RNG, ensemble stretch move, leapfrog HMC (numpy).

TEST INFRASTRUCTURE ONLY.

* ``hmc_chain`` restates linna/HMCSampler.py:19-68 (PINNED by tests/golden/hmc_trace.npz).
* ``hmc_batched_step`` is the same leapfrog applied independently per walker (the
  per-walker semantics of sampler.py:67-98): PINNED by tests/golden/hmc_move.npz,
  the output of the reference's own ``_hmc_wrapper`` called directly.
* ``stretch_half_step`` restates emcee 3.0.2 ``RedBlueMove.propose`` /
  ``StretchMove.get_proposal`` (Goodman & Weare 2010, a = 2): PARITY UNPINNED -- emcee is
  a third-party dependency (requirements.txt:14) absent from the reference tree and from
  this image; anchored on the reference call sites sampler.py:493-495, 519-530, its geometry pinned by
  the reference-held emcee chain (tests/test_stretch_fixture.py).
* ``integrated_time`` restates emcee 3.0.2 ``autocorr.integrated_time`` as the reference calls it
  (``get_autocorr_time(tol=0)``, sampler.py:538; zeus' ``AutoCorrTime`` callback with ``discard=0.2``,
  sampler.py:684): FFT autocorrelation per series, averaged over walkers, Sokal window c = 5.  PARITY UNPINNED for the
  same reason (emcee absent); exercised on the reference-held emcee chain ``chemcee_256.h5``.
* ``checkmeanstd`` restates sampler.py:370-387.
* ``slice_half_step`` / ``slice_iteration`` / ``slice_tune_mu`` restate zeus-mcmc's ``EnsembleSampler.sample`` loop
  with its default ``DifferentialMove`` (Karamanis & Beutler 2021, "Ensemble slice sampling", Stat. Comput. 31:61,
  Algorithms 2-4; zeus 2.x ``ensemble.py`` / ``moves.py``) as the reference drives it
  (``zeus.EnsembleSampler(nwalkers, ndim, lnp, pool=pool, maxiter=1E5)`` at sampler.py:728, ``run_mcmc`` at :735, the
  reference's DEFAULT method: main.py:22): PARITY UNPINNED -- zeus-mcmc is an unpinned third-party dependency
  (setup.py:13) absent from the reference tree and from this image, and the reference holds no zeus chain to anchor on.
  The random numbers are the Philox draws the HIP kernels make (same counters), so a half step can be replayed
  decision by decision (tests/test_gpu_slice_replay.py).
* ``philox4x32`` is the published Philox4x32-10 generator (Salmon et al. 2011); the HIP
  sampler kernels use the same counter layout so draws can be replayed here.
"""
import numpy as np

# --------------------------------------------------------------------------- Philox
_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32(counter, key, rounds=10):
    """counter: uint32[..., 4], key: uint32[..., 2] -> uint32[..., 4]."""
    c = np.array(counter, dtype=np.uint32, copy=True)
    k = np.array(np.broadcast_to(key, c.shape[:-1] + (2,)), dtype=np.uint32, copy=True)
    with np.errstate(over="ignore"):
        for _ in range(rounds):
            p0 = _M0 * c[..., 0].astype(np.uint64)
            p1 = _M1 * c[..., 2].astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & _MASK).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & _MASK).astype(np.uint32)
            n0 = hi1 ^ c[..., 1] ^ k[..., 0]
            n2 = hi0 ^ c[..., 3] ^ k[..., 1]
            c = np.stack([n0, lo1, n2, lo0], axis=-1)
            k = np.stack([k[..., 0] + _W0, k[..., 1] + _W1], axis=-1)
    return c


def u01(bits):
    """uint32 -> float32 uniform in (0, 1): (bits >> 8 + 0.5) * 2^-24."""
    return ((bits >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)


def walker_bits(seed, walkers, step, stream, sub=0):
    """uint32[n, 4] for the walkers ``walkers`` (ids) at (seed, step, stream, sub): counter = (walker, step, stream, sub),
    key = (seed_lo, seed_hi) -- ``walker_bits`` of csrc/common.h."""
    walkers = np.asarray(walkers)
    ctr = np.zeros((len(walkers), 4), np.uint32)
    ctr[:, 0] = walkers.astype(np.uint32)
    ctr[:, 1] = np.uint32(step & 0xFFFFFFFF)
    ctr[:, 2] = np.uint32(stream)
    ctr[:, 3] = np.uint32(sub)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], np.uint32)
    return philox4x32(ctr, key)


def walker_draws(seed, step, stream, nwalkers, sub=0):
    """4 uniforms per walker for (seed, step, stream): counter = (walker, step, stream, sub),
    key = (seed_lo, seed_hi)."""
    return u01(walker_bits(seed, np.arange(nwalkers), step, stream, sub))


def scaled_index(bits, n):
    """(bits * n) >> 32: an index in [0, n) from 32 random bits, as the kernels form it."""
    return ((bits.astype(np.uint64) * np.uint64(n)) >> np.uint64(32)).astype(np.int64)


def normal_from_uniform(u1, u2):
    """Box-Muller, float32."""
    r = np.sqrt(np.float32(-2.0) * np.log(u1))
    return (r * np.cos(np.float32(2 * np.pi) * u2)).astype(np.float32)


# --------------------------------------------------------------------------- stretch move
def stretch_propose(s, c, u_z, rint, a=2.0):
    """emcee StretchMove.get_proposal: zz = ((a-1)u+1)^2/a; q = c[r] - (c[r]-s) zz;
    factor = (ndim-1) log zz."""
    dt = s.dtype
    zz = ((dt.type(a) - dt.type(1)) * u_z.astype(dt) + dt.type(1)) ** 2 / dt.type(a)
    cr = c[rint]
    q = cr - (cr - s) * zz[:, None]
    factors = dt.type(s.shape[1] - 1.0) * np.log(zz)
    return q.astype(dt), factors.astype(dt)


def stretch_accept(logp_old, logp_new, factors, u_acc):
    """emcee RedBlueMove.propose: accept iff factor + new - old > log(u)."""
    lnpdiff = factors + logp_new - logp_old
    with np.errstate(invalid="ignore"):
        return lnpdiff > np.log(u_acc)


def stretch_half_step(coords, logp, S, C, u_z, rint, u_acc, logprob_fn, a=2.0):
    """Advance the walkers ``S`` (index array) against the complementary set ``C``."""
    q, f = stretch_propose(coords[S], coords[C], u_z, rint, a)
    new_lp = logprob_fn(q)
    acc = stretch_accept(logp[S], new_lp, f, u_acc)
    coords = coords.copy()
    logp = logp.copy()
    coords[S[acc]] = q[acc]
    logp[S[acc]] = new_lp[acc]
    return coords, logp, acc


# --------------------------------------------------------------------------- ensemble slice sampling (zeus)
class SliceLimit(RuntimeError):
    """zeus: 'Number of expansions / contractions exceeded maximum limit!' (``maxiter``)."""


def slice_half_step(coords, logp, S, C, mu, seed, step, half, logprob_fn, maxsteps=10000, maxiter=100000, trace=None):
    """One half of a zeus iteration: the walkers ``S`` (the "active" set, index array into ``coords``) each take one slice
    update along a direction drawn from the complementary ("inactive") set ``C``.  Karamanis & Beutler 2021, Algorithm 2
    (differential move), 3 (stepping out, Neal 2003 with the J / K budget) and 4 (shrinking); zeus ``ensemble.py`` loop body
    and ``moves.DifferentialMove.get_direction``; driven by the reference at sampler.py:728-735.

      direction_k = 2 mu (c_a - c_b),  a != b drawn from the inactive set            (zeus: ``2.0 * mu * (X[pairs[0]] - X[pairs[1]])``)
      Z0_k        = lnP(x_k) - Exp(1)   (= lnP + log u)                               (zeus: ``Z[active] - np.random.exponential``)
      L_k = -U(0,1), R_k = L_k + 1;  J_k = floor(maxsteps U(0,1)), K_k = maxsteps - 1 - J_k
      stepping out:  while J >= 1 and Z0 < lnP(x + L d):  L -= 1, J -= 1, nexp += 1   (and the same with R / K / +1); all the
                     walkers' open ends of a pass are evaluated in ONE ``logprob_fn`` call
      shrinking:     x' = x + W d, W = L + U(0,1) (R - L); accept iff Z0 < lnP(x'), else W < 0: L = W, W > 0: R = W, ncon += 1

    Random numbers: the Philox draws of the HIP kernels, counter (walker id, ``step``, stream, sub) -- stream ``half`` sub 0:
    (a, b, height, bracket), sub 1: J; stream ``2 + half`` sub t: shrinking trial t = 1, 2, ....  Deviation from zeus, stated:
    zeus draws the ``ns`` direction pairs WITHOUT replacement among the ordered pairs (``random.sample(permutations)``), a
    counter generator draws each walker's pair independently -- every walker's pair has the same uniform law, two walkers may
    share one.  float32 arithmetic in the kernels' order (mul then add, no contraction).
    Returns (coords, logp, nexp, ncon); ``trace`` (a dict) receives the per-walker record the replay test compares.
    """
    f = np.float32
    coords = np.array(coords, f, copy=True)
    logp = np.array(logp, f, copy=True)
    S = np.asarray(S)
    C = np.asarray(C)
    ns, nc = len(S), len(C)
    if nc < 2:
        raise ValueError("the differential move needs two complementary walkers")
    b0 = walker_bits(seed, S, step, half, 0)
    ia = scaled_index(b0[:, 0], nc)
    ib = scaled_index(b0[:, 1], nc - 1)
    ib = ib + (ib >= ia)
    scale = f(2.0 * mu)
    D = (scale * (coords[C[ia]] - coords[C[ib]])).astype(f)
    X = coords[S]
    Z0 = (logp[S] + np.log(u01(b0[:, 2]))).astype(f)
    L = (-u01(b0[:, 3])).astype(f)
    R = (L + f(1)).astype(f)
    J = np.floor(f(maxsteps) * u01(walker_bits(seed, S, step, half, 1)[:, 0])).astype(np.int64)
    J = np.minimum(J, int(maxsteps) - 1)             # (the float32 product may round up to maxsteps itself)
    K = (int(maxsteps) - 1) - J
    margin = np.full(ns, np.inf)                       # smallest |lnP - Z0| any decision of this walker rested on
    nexp_w = np.zeros(ns, np.int64)
    ncon_w = np.zeros(ns, np.int64)
    neval = 0
    mJ = np.ones(ns, bool)
    mK = np.ones(ns, bool)
    cnt = 0
    with np.errstate(invalid="ignore"):
        while mJ.any() or mK.any():
            cnt += int(mJ.any()) + int(mK.any())
            if cnt > maxiter:
                raise SliceLimit("Number of expansions exceeded maximum limit!")
            mJ &= J >= 1
            mK &= K >= 1
            jl, jr = np.flatnonzero(mJ), np.flatnonzero(mK)
            if len(jl) + len(jr) == 0:
                break
            pts = np.concatenate([X[jl] + L[jl, None] * D[jl], X[jr] + R[jr, None] * D[jr]]).astype(f)
            Zp = np.asarray(logprob_fn(pts), f)
            neval += len(pts)
            ZL, ZR = Zp[:len(jl)], Zp[len(jl):]
            margin[jl] = np.fmin(margin[jl], np.abs(ZL.astype(np.float64) - Z0[jl]))
            margin[jr] = np.fmin(margin[jr], np.abs(ZR.astype(np.float64) - Z0[jr]))
            out = Z0[jl] < ZL
            L[jl[out]] -= f(1); J[jl[out]] -= 1; nexp_w[jl[out]] += 1
            mJ[jl[~out]] = False
            out = Z0[jr] < ZR
            R[jr[out]] += f(1); K[jr[out]] -= 1; nexp_w[jr[out]] += 1
            mK[jr[~out]] = False
        Lx, Rx = L.copy(), R.copy()
        W = np.zeros(ns, f)
        Xp = X.copy()
        Zacc = np.zeros(ns, f)
        m = np.ones(ns, bool)
        t = 0
        while m.any():
            t += 1
            j = np.flatnonzero(m)
            u = u01(walker_bits(seed, S[j], step, 2 + half, t)[:, 0])
            W[j] = L[j] + u * (R[j] - L[j])
            Xp[j] = X[j] + W[j, None] * D[j]
            Zp = np.asarray(logprob_fn(Xp[j]), f)
            neval += len(j)
            margin[j] = np.fmin(margin[j], np.abs(Zp.astype(np.float64) - Z0[j]))
            ok = Z0[j] < Zp
            Zacc[j[ok]] = Zp[ok]
            m[j[ok]] = False
            rej = j[~ok]
            lo, hi = rej[W[rej] < 0], rej[W[rej] > 0]
            L[lo] = W[lo]; R[hi] = W[hi]
            ncon_w[lo] += 1; ncon_w[hi] += 1
            if t > maxiter:
                raise SliceLimit("Number of contractions exceeded maximum limit!")
    coords[S] = Xp
    logp[S] = Zacc
    if trace is not None:
        trace.update(ia=ia, ib=ib, D=D, Z0=Z0, L_out=Lx, R_out=Rx, L=L, R=R, W=W, Zacc=Zacc, margin=margin, nexp=nexp_w,
                     ncon=ncon_w, neval=neval, J=J, K=K, ntrials=t)
    return coords, logp, int(nexp_w.sum()), int(ncon_w.sum())


def slice_iteration(coords, logp, halves, mu, seed, step, logprob_fn, maxsteps=10000, maxiter=100000, traces=None):
    """One zeus iteration: the two half ensembles in turn, each against the other's CURRENT positions (zeus: ``for ensembles
    in [[0, 1], [1, 0]]``).  ``halves`` = the iteration's random split (two index arrays).  Returns (coords, logp, nexp, ncon)."""
    nexp = ncon = 0
    for h in (0, 1):
        tr = {} if traces is not None else None
        coords, logp, e, c = slice_half_step(coords, logp, halves[h], halves[1 - h], mu, seed, step, h, logprob_fn, maxsteps,
                                             maxiter, tr)
        nexp += e
        ncon += c
        if traces is not None:
            traces.append(tr)
    return coords, logp, nexp, ncon


def slice_tune_mu(mu, nexp, ncon, count, tolerance=0.05, patience=5):
    """zeus' Robbins-Monro rule behind every tuning iteration: ``nexp = max(1, nexp); mu *= 2 nexp / (nexp + ncon)``; the
    expansion fraction within ``tolerance`` of 1/2 ``patience`` + 1 times in a row ends tuning.
    Returns (mu, count, still_tuning)."""
    nexp = max(1, nexp)
    mu = mu * (2.0 * nexp / (nexp + ncon))            # (zeus: ``self.mu *= 2.0 * nexp / (nexp + ncon)`` -- the factor first)
    count = count + 1 if abs(nexp / (nexp + ncon) - 0.5) < tolerance else 0
    return mu, count, not count > patience


# --------------------------------------------------------------------------- HMC
def hmc_chain(lnp_and_grad, x0, mass, num_samps, num_steps, step_size, momenta, uniforms):
    """linna/HMCSampler.py:19-68 with the random draws passed in explicitly.

    ``lnp_and_grad(x[nin]) -> (lnP, grad[nin])``; ``momenta[num_samps, nin]`` are the
    standard-normal draws of :26 (before ``* sqrt(m)``), ``uniforms[num_samps]`` those of :59.
    Returns (xs[num_samps, nin], lnPs[num_samps], accepted[num_samps]).
    """
    f = np.float32
    x_cur = np.asarray(x0, f).copy()
    mass = np.asarray(mass, f)
    xs, lnps, accs = [], [], []
    for i in range(num_samps):
        x = x_cur.copy()
        p = momenta[i].astype(f) * np.sqrt(mass)                      # :26
        lnP, grad = lnp_and_grad(x)                                   # :29,32
        prev = lnP
        H_init = f(0.5) * np.sum(p * p / mass) - lnP                  # :31
        p = p + f(0.5) * grad * f(step_size)                          # :35
        x = x + (p / mass) * f(step_size)                             # :36
        lnP, grad = lnp_and_grad(x)                                   # :39-40
        for _ in range(1, num_steps):                                 # :43-48
            p = p + grad * f(step_size)
            x = x + (p / mass) * f(step_size)
            lnP, grad = lnp_and_grad(x)
        p = p + f(0.5) * grad * f(step_size)                          # :51
        H_prime = f(0.5) * np.sum(p * p / mass) - lnP                 # :54
        ratio = np.exp(np.minimum(H_init - H_prime, 0))               # :57
        if uniforms[i] < min(ratio, 1):                               # :58-59
            x_cur = x
            xs.append(x.copy()); lnps.append(lnP); accs.append(True)
        else:
            xs.append(x_cur.copy()); lnps.append(prev); accs.append(False)
    return np.array(xs, f), np.array(lnps, f), np.array(accs)


def hmc_batched_step(lnp_and_grad_rows, x, lnp, grad, mass, num_steps, step_size, p0, u, details=None):
    """One HMC transition for every row of ``x[B, nin]`` independently.

    ``lnp_and_grad_rows(x[B,nin]) -> (lnP[B], grad[B,nin])``; ``p0`` standard-normal draws,
    ``u`` uniforms.  (lnp, grad) are the cached values at ``x``.  Returns the new
    (x, lnp, grad, accepted).
    """
    f = np.float32
    mass = np.asarray(mass, f)[None, :]
    eps = f(step_size)
    p = p0.astype(f) * np.sqrt(mass)
    H0 = f(0.5) * np.sum(p * p / mass, -1) - lnp
    q = x.copy()
    g = grad
    p = p + f(0.5) * eps * g
    for i in range(num_steps):
        q = q + eps * (p / mass)
        l, g = lnp_and_grad_rows(q)
        if i < num_steps - 1:
            p = p + eps * g
    p = p + f(0.5) * eps * g
    H1 = f(0.5) * np.sum(p * p / mass, -1) - l
    if details is not None:      # the proposal itself and the kinetic-energy factor of sampler.py:95-97 (K0 - K1)
        details.update(q=q.copy(), lnp_new=np.asarray(l).copy(), factor=(H0 + lnp) - (H1 + l))
    with np.errstate(invalid="ignore", over="ignore"):
        acc = u < np.exp(np.minimum(H0 - H1, 0))
    acc &= np.isfinite(l)
    xn = np.where(acc[:, None], q, x)
    ln = np.where(acc, l, lnp)
    gn = np.where(acc[:, None], g, grad)
    return xn.astype(f), ln.astype(f), gn.astype(f), acc


# --------------------------------------------------------------------------- convergence statistics
def _next_pow_two(n):
    i = 1
    while i < n:
        i = i << 1
    return i


def function_1d(x):
    """emcee.autocorr.function_1d: normalised autocorrelation function of one series through a zero-padded FFT."""
    x = np.atleast_1d(np.asarray(x, np.float64))
    n = _next_pow_two(len(x))
    f = np.fft.fft(x - np.mean(x), n=2 * n)
    acf = np.fft.ifft(f * np.conjugate(f))[:len(x)].real
    with np.errstate(invalid="ignore", divide="ignore"):
        return acf / acf[0]


def auto_window(taus, c):
    """emcee.autocorr.auto_window."""
    m = np.arange(len(taus)) < c * taus
    if np.any(m):
        return int(np.argmin(m))
    return len(taus) - 1


def integrated_time(x, c=5.0):
    """emcee.autocorr.integrated_time(x, c=5, tol=0) for a chain ``x[nstep, nwalker, ndim]`` -> tau[ndim]
    (what ``sampler.get_autocorr_time(tol=0)`` returns at sampler.py:538)."""
    x = np.atleast_1d(np.asarray(x, np.float64))
    if x.ndim == 1:
        x = x[:, None, None]
    if x.ndim == 2:
        x = x[:, :, None]
    nt, nw, nd = x.shape
    tau = np.empty(nd)
    windows = np.empty(nd, int)
    for d in range(nd):
        f = np.zeros(nt)
        for k in range(nw):
            f += function_1d(x[:, k, d])
        f /= nw
        taus = 2.0 * np.cumsum(f) - 1.0
        windows[d] = auto_window(taus, c)
        tau[d] = taus[windows[d]]
    return tau


def checkmeanstd_stats(samples):
    """The two numbers sampler.py:370-387 compares with its thresholds: median over parameters of the first-half /
    second-half shift of the mean (in units of the second half's standard deviation) and of the standard deviation."""
    samples = np.asarray(samples, np.float64)
    half = int(len(samples) / 2)
    a = samples[:half].reshape(-1, samples.shape[-1])
    b = samples[half:].reshape(-1, samples.shape[-1])
    meanshifte = np.median(np.abs(np.mean(a, axis=0) - np.mean(b, axis=0)) / np.std(b, axis=0))
    stdshifte = np.median((np.std(a, axis=0) - np.std(b, axis=0)) / np.std(b, axis=0))
    return meanshifte, stdshifte
