"""Ensemble and HMC walkers on MI355X.

Mirrors the sampling layer of the reference: ``linna/sampler.py`` (``HMCSampler`` -- despite
its name the emcee driver, :389-554; ``checkmeanstd`` :370-387; ``ZeusSampler`` :699-737) and
the walker loop they delegate to emcee.  Here the ensemble lives on the GPU:

* ``EnsembleSampler``: affine-invariant stretch move (emcee ``StretchMove`` inside
  ``RedBlueMove``, a = 2, random split of the walkers in two halves every step).  Per half
  step: ``linna_stretch_propose`` -> fused ``Log_prob`` pipeline on the whole half ->
  ``linna_stretch_accept``; Philox draws keyed (seed; walker, step, stream).
* ``BatchedHMC``: the leapfrog of ``linna/HMCSampler.py`` run independently per walker with
  the emulator's reverse-mode gradient (``linna_logprob_grad``).
* walkers shard across ranks (one process per GPU): every rank advances its own
  sub-ensemble, chain state is gathered with one RCCL all-gather per flush.

The emcee and zeus arithmetic itself is third-party code absent from the reference tree: both moves are
restated from the published algorithms; the test suite replays the kernels against its own numpy
restatements of them (``stretch_half_step`` / ``slice_half_step`` in the repo's oracle/sampling.py --
test infrastructure, never imported here) and checks the posteriors statistically.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib, h5lite

__all__ = ["EnsembleSampler", "SliceEnsembleSampler", "BatchedHMC", "HMCSampler", "ZeusSampler", "checkmeanstd", "integrated_time",
           "ChainStore", "DeviceChain", "read_chain_and_cut", "get_good_walker_list"]


# ------------------------------------------------------------------ convergence statistics (host)
def _next_pow_two(n):
    i = 1
    while i < n:
        i <<= 1
    return i


def _autocorr_1d(x):
    x = np.asarray(x, np.float64)
    n = _next_pow_two(len(x))
    f = np.fft.fft(x - np.mean(x), n=2 * n)
    acf = np.fft.ifft(f * np.conjugate(f))[:len(x)].real
    if acf[0] == 0:
        return np.full(len(x), np.nan)
    return acf / acf[0]


def integrated_time(x, c=5.0):
    """Integrated autocorrelation time per parameter of a chain ``x[nstep, nwalker, ndim]``
    (emcee's estimator as called at sampler.py:538 with tol=0: FFT autocorrelation averaged over
    walkers, Sokal window with c = 5)."""
    x = np.atleast_1d(x)
    if x.ndim == 1:
        x = x[:, None, None]
    if x.ndim == 2:
        x = x[:, :, None]
    nt, nw, nd = x.shape
    tau = np.empty(nd)
    for d in range(nd):
        f = np.zeros(nt)
        for k in range(nw):
            f += _autocorr_1d(x[:, k, d])
        f /= nw
        taus = 2.0 * np.cumsum(f) - 1.0
        m = np.arange(len(taus)) < c * taus
        win = int(np.argmin(m)) if np.any(m) else len(taus) - 1      # emcee's auto_window, edge cases included
        tau[d] = taus[win]
    return tau


class DeviceChain(object):
    """The chain of a run kept on the GPU for the convergence statistics, which are computed there INCREMENTALLY.

    The reference recomputes emcee's integrated autocorrelation time of the WHOLE chain every 100 iterations on the
    host (sampler.py:538, 684) -- quadratic in the chain length over a run.  Only the lags up to Sokal's window enter
    that estimate, so this class keeps per (walker, parameter) series the running lagged products ``S_k`` for ``k < K``
    and the running sum in float64 (``linna_acorr_update``: each check adds the products of the ~100 new rows, and takes
    out those of the rows zeus' moving discard drops), and ``linna_acorr_tau`` turns them into emcee's estimate: mean
    removal through prefix sums, normalisation, walker average, cumulative sum, window -- kernels of this library
    (csrc/autocorr.hip), no FFT.  ``K`` starts at 512 lags and grows (new lags computed from the stored chain) whenever
    the window is not found below it or comes within a third of it.  The statistics' copy of the chain is
    ``ct[row][parameter][walker]`` (walkers padded to a multiple of 64).  All work is enqueued on the current stream;
    ``tau_begin`` / ``tau_end`` let a driver collect the result one block later without stalling the sampler."""

    MAX_WALKERS = 1024      # the running sums (8 bytes x lags per series, read and written at every check) cover an evenly spaced
                            # subset of at most so many walkers; the drivers confirm a positive check with ALL walkers (sums
                            # of their own, from the stored chain) before they stop.  None: every walker, always.
    LAGS0 = 512             # initial lag capacity (a multiple of 32)

    def __init__(self, max_walkers=-1):
        self.ct = None           # [capacity, nd, nwp] fp32 on the device, grown by doubling
        self.n = 0
        self.nw = self.nd = self.nws = self.nwp = self.nwc = self.wstride = None
        self.max_walkers = self.MAX_WALKERS if max_walkers == -1 else max_walkers
        self._S = self._T = None     # running sums of the window [_lo, _hi) over the subset's lanes
        self._lo = self._hi = 0
        self._scratch = None
        self.lag_growths = 0

    # -- storage
    def _setup(self, z):
        self.nw, self.nd = int(z.shape[1]), int(z.shape[2])
        self.wstride = 1 if not self.max_walkers else max(1, -(-self.nw // int(self.max_walkers)))
        self.nws = len(range(0, self.nw, self.wstride))
        self.nwp = (self.nw + 63) & ~63              # lanes per parameter in ct: the subset's walkers first, then the others
        self.nwc = (self.nws + 63) & ~63             # lanes the running sums cover
        self.dev = z.device
        self.ctx = _lib.ctx(self.dev.index)

    @property
    def subset(self):
        """True when the routine estimate averages over fewer walkers than the ensemble has."""
        return self.nws < self.nw

    def append(self, z_block):
        z = torch.as_tensor(z_block)
        z = (z if z.is_cuda else z.cuda()).to(torch.float32).contiguous()
        if z.ndim != 3 or len(z) == 0:
            raise ValueError("chain blocks are [nsteps, nwalkers, ndim]")
        if self.ct is None:
            self._setup(z)
        elif (int(z.shape[1]), int(z.shape[2])) != (self.nw, self.nd):
            raise ValueError("chain block of another ensemble")
        need = self.n + len(z)
        if self.ct is None or need > len(self.ct):
            cap = max(need, 2 * (0 if self.ct is None else len(self.ct)), 1024)
            buf = torch.empty((cap, self.nd, self.nwp), dtype=torch.float32, device=self.dev)
            if self.n:
                buf[:self.n] = self.ct[:self.n]
                # the old buffer may belong to ANOTHER stream's pool (a resumed run fills the chain on the default stream, the
                # in-loop appends run on the statistics stream): the caching allocator must not hand it out while this copy reads it
                self.ct.record_stream(torch.cuda.current_stream(self.dev))
            self.ct = buf
        for i0 in range(0, len(z), 32768):                   # (grid.y of the transposing kernel is the step count)
            blk = z[i0:i0 + 32768]
            _lib.call("linna_chain_append_t", self.ctx, _lib.ptr(blk), self.nd, len(blk), self.nw, self.nd, self.wstride,
                      _lib.ptr(self.ct), self.nwp, self.n + i0, _lib.stream())
        self.n = need

    def __len__(self):
        return self.n

    def last(self, n):
        """The last ``n`` steps as one device tensor [n, nw, nd] (walkers in the ensemble's order)."""
        w = np.arange(self.nw)
        lane = np.where(w % self.wstride == 0, w // self.wstride, self.nws + w - w // self.wstride - 1)
        idx = torch.as_tensor(lane, device=self.dev)
        return self.ct[max(0, self.n - int(n)):self.n].index_select(2, idx).permute(0, 2, 1).contiguous()

    # -- running sums
    def _dptr(self, t):
        return _lib.ptr(t, torch.float64)

    def _update(self, S, T, a0, a1, lo, hi, k0, k1, remove):
        _lib.call("linna_acorr_update", self.ctx, _lib.ptr(self.ct), self.nd, self.nwp, int(S.shape[2]), int(a0), int(a1), int(lo),
                  int(hi), int(k0), int(k1), self._dptr(S), self._dptr(T) if T is not None else None, 1 if remove else 0, _lib.stream())

    def _fresh(self, lags, full=False):
        z = lambda *sh: torch.zeros(sh, dtype=torch.float64, device=self.dev)
        nwc = self.nwp if full else self.nwc
        return z(lags, self.nd, nwc), z(self.nd, nwc)

    def _advance(self, lo, hi):
        """Bring the running sums to the window [lo, hi)."""
        if self._S is None or lo < self._lo or hi < self._hi or lo > self._hi:
            lags = self.LAGS0 if self._S is None else len(self._S)
            self._S, self._T = self._fresh(lags)
            self._lo = self._hi = lo
        if hi > self._hi:
            self._update(self._S, self._T, self._hi, hi, self._lo, hi, 0, len(self._S), False)
            self._hi = hi
        if lo > self._lo:
            self._update(self._S, self._T, self._lo, lo, self._lo, self._hi, 0, len(self._S), True)
            self._lo = lo

    def _grow(self, S, lo, hi, lags):
        """More lags: the old rows are kept, the new ones computed from the stored chain."""
        lags = min((int(lags) + 31) & ~31, (hi - lo + 31) & ~31)
        if lags <= len(S):
            return S
        big = torch.zeros((lags,) + tuple(S.shape[1:]), dtype=torch.float64, device=self.dev)
        big[:len(S)] = S
        self._update(big, None, lo, hi, lo, hi, len(S), lags, False)
        self.lag_growths += 1
        return big

    def _estimate(self, S, T, lo, hi, c, nlive):
        """Enqueue emcee's estimate from the sums; returns the device result [3 nd] (tau | window | more lags needed)."""
        kuse = min(len(S), hi - lo) - 1
        nwc = int(S.shape[2])
        nb = int(_lib.load().linna_acorr_scratch_bytes(self.nd, nwc, kuse))
        if self._scratch is None or self._scratch.numel() * 8 < nb:
            self._scratch = torch.empty((nb + 7) // 8, dtype=torch.float64, device=self.dev)
        out = torch.empty(3 * self.nd, dtype=torch.float64, device=self.dev)
        _lib.call("linna_acorr_tau", self.ctx, _lib.ptr(self.ct), self.nd, self.nwp, nwc, int(nlive),
                  int(lo), int(hi), int(kuse), self._dptr(S), self._dptr(T), float(c), self._dptr(self._scratch), self._dptr(out), _lib.stream())
        return out

    def tau_begin(self, discard=0, c=5.0, upto=None, all_walkers=False):
        """Enqueue everything a check needs on the current stream -- the sums' update, the estimate, its copy into pinned
        memory -- and return a token for ``tau_end``; nothing waits.  ``all_walkers``: the estimate over every walker of
        the ensemble (sums of their own, computed from the stored chain) instead of the routine subset."""
        hi = self.n if upto is None else int(upto)
        lo = int(discard)
        if not 0 <= lo < hi <= self.n:
            raise ValueError("empty chain window")
        full = bool(all_walkers) and self.subset
        live = (upto is None or hi == self.n) and not full
        if live:
            self._advance(lo, hi)
            S, T = self._S, self._T
        else:                                              # another window, or every walker: sums of their own, from scratch
            S, T = self._fresh(min(len(self._S) if self._S is not None else self.LAGS0, (hi - lo + 31) & ~31), full)
            self._update(S, T, lo, hi, lo, hi, 0, len(S), False)
        nlive = self.nw if full else self.nws
        out = self._estimate(S, T, lo, hi, c, nlive)
        pin = torch.empty(3 * self.nd, dtype=torch.float64).pin_memory()
        pin.copy_(out, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        return dict(pin=pin, ev=ev, lo=lo, hi=hi, c=c, live=live, S=S, T=T, nlive=nlive)

    def tau_end(self, tok):
        """The estimate of ``tau_begin`` as numpy [nd].  When Sokal's window lies beyond the lags kept so far they are doubled
        (new lags from the stored chain) and the estimate repeated -- early in a run, while the estimate still grows with
        the chain; afterwards the capacity stays ahead of the window (1.5 x)."""
        nd = self.nd
        while True:
            tok["ev"].synchronize()
            r = tok["pin"].numpy()
            tau, win, more = r[:nd].copy(), r[nd:2 * nd].copy(), r[2 * nd:]
            lo, hi, S = tok["lo"], tok["hi"], tok["S"]
            if not np.any(more > 0):
                break
            S = self._grow(S, lo, hi, 2 * len(S))
            if tok["live"]:
                self._S = S
            tok["S"] = S
            out = self._estimate(S, tok["T"], lo, hi, tok["c"], tok["nlive"])
            tok["pin"].copy_(out, non_blocking=True)
            tok["ev"] = torch.cuda.Event()
            tok["ev"].record(torch.cuda.current_stream(self.dev))
        if tok["live"] and np.all(np.isfinite(win)):
            want = int(1.5 * float(np.max(win))) + 64
            if want > len(self._S) and len(self._S) < ((hi - lo + 31) & ~31):
                self._S = self._grow(self._S, lo, hi, max(want, 2 * len(self._S)))
        self.last_window = win
        return tau

    def integrated_time(self, discard=0, c=5.0, upto=None, all_walkers=False):
        """emcee's estimator (autocovariance of the mean-removed series normalised at lag 0, averaged over walkers, Sokal
        window, tol=0) per parameter -> numpy [nd]; ``discard`` leading steps are dropped (zeus: 20 %); ``upto``: only the
        first ``upto`` steps (the chain as it was at an earlier check); ``all_walkers``: see ``tau_begin``."""
        return self.tau_end(self.tau_begin(discard, c, upto, all_walkers))

    def checkmeanstd(self, nlast, meanshift, stdshift):
        """sampler.py:370-387 on the last ``nlast`` steps: first-half / second-half drift (moments: linna_chain_meanstd)."""
        t0 = max(0, self.n - int(nlast))
        half = int((self.n - t0) / 2)
        out = torch.empty((self.nd, 2, 2), dtype=torch.float64, device=self.dev)
        _lib.call("linna_chain_meanstd", self.ctx, _lib.ptr(self.ct), self.nd, self.nwp, self.nw, t0, t0 + half, self.n,
                  self._dptr(out), _lib.stream())
        m = out.cpu().numpy()
        with np.errstate(invalid="ignore", divide="ignore"):
            meanshifte = np.median(np.abs(m[:, 0, 0] - m[:, 1, 0]) / m[:, 1, 1])
            stdshifte = np.median((m[:, 0, 1] - m[:, 1, 1]) / m[:, 1, 1])
        print(meanshifte, stdshifte, flush=True)
        return bool((meanshifte < meanshift) and (stdshifte < stdshift))


def checkmeanstd(samples, meanshift, stdshift):
    """sampler.py:370-387: first-half / second-half drift of mean and standard deviation."""
    half = int(len(samples) / 2)
    a = samples[:half].reshape(-1, samples.shape[-1])
    b = samples[half:].reshape(-1, samples.shape[-1])
    meanshifte = np.median(np.abs(np.mean(a, axis=0) - np.mean(b, axis=0)) / np.std(b, axis=0))
    stdshifte = np.median((np.std(a, axis=0) - np.std(b, axis=0)) / np.std(b, axis=0))
    print(meanshifte, stdshifte, flush=True)
    return (meanshifte < meanshift) & (stdshifte < stdshift)


# ------------------------------------------------------------------ chain storage
class _OnDisk(object):
    """Placeholder for a chain block that lives in the HDF5 file only (ChainStore drops big blocks once written)."""
    __slots__ = ("n",)

    def __init__(self, n):
        self.n = int(n)

    def __len__(self):
        return self.n


class ChainStore(object):
    """Chain backend in the reference's file formats: ``chemcee_256.h5`` as emcee's ``HDFBackend`` /
    ``Transformbackend`` lay it out (group ``mcmc``: ``chain``, ``chain_transformed``, ``log_prob``,
    ``accepted`` + attributes ``nwalkers``, ``ndim``, ``iteration``, ``has_blobs``; sampler.py:322-368)
    and ``zeus_256.h5`` as ``ZeusTransformCallback`` does (root datasets ``samples``,
    ``chain_transformed``, ``logprob``; sampler.py:556-577), through ``h5lite`` (no h5py in this
    image), plus the ``<name>.txt`` layout the reference's reader accepts as a fallback
    (main.py:166-167, 293-295: rows of ``theta..., log_prob``; ``write_txt=True``).  Files written by the reference load
    the same way, so a run directory started there resumes here."""

    GZIP_LIMIT = 64 << 20           # zeus layout: gzip chunks like the reference below this many bytes per dataset
    MAX_QUEUED = 8                  # device blocks waiting for the writer: a disk slower than the sampler stalls the sampler
                                    # (put blocks) instead of piling chain blocks up in HBM (108 MB each at 4096 walkers)

    def __init__(self, filename, transform=None, write_txt=False):
        self.base = filename[:-3] if filename.endswith(".h5") else filename
        self.layout = "zeus" if os.path.basename(self.base).startswith("zeus") else "emcee"
        self.transform = transform
        self.write_txt = write_txt          # also keep <name>.txt (theta..., log_prob rows) next to the HDF5 file
        self.chain, self.chain_transformed, self.log_prob = [], [], []
        self.accepted = None
        self._flushed = 0
        self._writer, self._queue, self._error = None, None, None
        self._events, self._copy_stream = {}, None
        self._appender, self._appended = None, 0
        self.busy = {"fetch": 0.0, "append": 0.0}      # seconds the two background threads spent working (driver profiles)

    # Incremental flushes run on two background threads in a pipeline: the first brings a block from the device to the
    # host (copy stream of its own, after the event recorded when the block was appended), the second appends it to the
    # HDF5 file (whose bulk writes fan out over a few more threads, h5lite.Appender).  Zipping / writing 7 MB per
    # convergence check (128 walkers) on the sampling thread took as long as the 100 iterations between two checks; at
    # 4096 walkers a block is 2 x 54 MB and one thread doing copy + write in turn sustained a third of the sampling
    # rate.  numpy's file I/O, os.pwrite and the device copies release the GIL; the sampling thread spends its time
    # inside ctypes calls, which release it too.
    def _enqueue(self, path, arrays):
        import queue, threading, time
        if self._writer is None:
            self._queue, self._wqueue = queue.Queue(maxsize=self.MAX_QUEUED), queue.Queue(maxsize=8)

            def fetch():
                while True:
                    item = self._queue.get()
                    try:
                        if item is not None and self._error is None:
                            t0 = time.perf_counter()
                            self._to_host(item[1])                    # device blocks come to the host HERE, off the sampling thread
                            if torch.is_tensor(item[2]):
                                item = (item[0], item[1], self._acc_host(item[2]))
                            self.busy["fetch"] += time.perf_counter() - t0
                    except Exception as e:          # surfaced by the next drain()
                        self._error = e
                    self._wqueue.put(item)
                    if item is None:
                        return

            def work():
                while True:
                    item = self._wqueue.get()
                    try:
                        if item is None:
                            return
                        if self._error is None:
                            t0 = time.perf_counter()
                            self._append_block(item[1], item[2])
                            self.busy["append"] += time.perf_counter() - t0
                    except Exception as e:
                        self._error = e
                    finally:
                        self._queue.task_done()
            self._fetcher = threading.Thread(target=fetch, name="linna-chain-fetch", daemon=True)
            self._writer = threading.Thread(target=work, name="linna-chain-writer", daemon=True)
            self._fetcher.start()
            self._writer.start()
        self._queue.put((path, arrays[0], arrays[1]))

    def _acc_host(self, a):
        if torch.is_tensor(a):
            if a.is_cuda and self._copy_stream is not None:
                with torch.cuda.stream(self._copy_stream):
                    return a.to("cpu", torch.float64).numpy()
            return a.to("cpu", torch.float64).numpy()
        return np.asarray(a, np.float64)

    def _to_host(self, k):
        """Block k as numpy arrays (in place).  Blocks appended as device tensors are copied on a stream of their own,
        after the event recorded when they were appended: the copy neither waits for nor delays the sampling stream.
        The host side is PINNED memory from torch's caching host allocator (54 MB in 1 ms against 6-11 ms into pageable
        memory; the buffers are recycled once a written block has been dropped, see _append_block)."""
        if not torch.is_tensor(self.chain[k]):
            return
        ev = self._events.pop(k, None)
        dev = self.chain[k].device
        if dev.type == "cuda":
            if self._copy_stream is None:
                self._copy_stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(self._copy_stream):
                if ev is not None:
                    self._copy_stream.wait_event(ev)
                host = []
                for t in (self.chain[k], self.chain_transformed[k], self.log_prob[k]):
                    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                    h.copy_(t, non_blocking=True)
                    host.append(h)
                self._copy_stream.synchronize()
            host = [h.numpy() for h in host]
        else:
            host = [t.numpy() for t in (self.chain[k], self.chain_transformed[k], self.log_prob[k])]
        self.chain[k], self.chain_transformed[k], self.log_prob[k] = host           # (frees the device copies)

    def drain(self):
        """Wait until every part queued so far is on disk."""
        if self._queue is not None:
            self._queue.join()
        if self._error is not None:
            e, self._error = self._error, None
            raise e

    @property
    def h5(self):
        return self.base + ".h5"

    @property
    def npz(self):                                          # consolidated file of earlier versions of this package
        return self.base + ".npz"

    def exists(self):
        return os.path.isfile(self.h5) or os.path.isfile(self.npz) or bool(self._parts(self.base))

    def remove(self):
        self.drain()
        for f in [self.h5, self.npz, self.base + ".txt"] + self._parts(self.base):
            if os.path.isfile(f):
                os.remove(f)

    def append(self, z_block, theta_block, logp_block, accepted):
        # float32 blocks stay float32 (what the device produced; emcee / h5py read either width): half the
        # bytes of every part file and of the final HDF5 file -- at 4096 walkers the chain is 2 x 54 MB per 100 steps
        # Device tensors are kept as they are (no copy on the sampling thread: at 4096 walkers a block is 2 x 54 MB);
        # they come to the host in the writer thread, or at the latest when the arrays are asked for.
        if torch.is_tensor(z_block) and z_block.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(z_block.device))
            self._events[len(self.chain)] = ev
            self.chain.append(z_block); self.chain_transformed.append(theta_block); self.log_prob.append(logp_block)
            self.accepted = accepted.detach().clone() if torch.is_tensor(accepted) else np.asarray(accepted, np.float64)
            return
        keep = lambda a: np.asarray(a) if np.asarray(a).dtype == np.float32 else np.asarray(a, np.float64)
        self.chain.append(keep(z_block))
        self.chain_transformed.append(keep(theta_block))
        self.log_prob.append(keep(logp_block))
        self.accepted = np.asarray(accepted, np.float64)

    # -- the chain file grows in place (h5lite.Appender): one chunk of every dataset per flushed block ------------
    def _names(self):
        return (("samples", "chain_transformed", "logprob") if self.layout == "zeus"
                else ("mcmc/chain", "mcmc/chain_transformed", "mcmc/log_prob"))

    DROP_BYTES = 16 << 20           # blocks of at least this size are dropped from host memory once they are in the file
    CHUNK_ROWS = 100                # steps per HDF5 chunk: the flush cadence (emcee's backend grows its file likewise),
                                    # whatever the length of the first block -- a resumed chain arrives as ONE long block

    def _spec(self, k):
        nw, nd = self.chain[k].shape[1], self.chain[k].shape[2]
        dt = lambda blocks: np.result_type(*[np.asarray(b).dtype for b in blocks if not torch.is_tensor(b)] or [np.float32])
        tails = dict(zip(self._names(), (((nw, nd), dt(self.chain[:k + 1])), ((nw, nd), dt(self.chain_transformed[:k + 1])),
                                         ((nw,), dt(self.log_prob[:k + 1])))))
        for name, (tail, d) in tails.items():
            if self.CHUNK_ROWS * int(np.prod(tail)) * np.dtype(d).itemsize >= 2 ** 32:
                raise h5lite.H5Error("%s: a chunk of %d steps would reach 4 GiB (HDF5 chunk sizes are 32-bit)" % (name, self.CHUNK_ROWS))
        return nw, nd, tails

    def _adopt_existing(self, k):
        """A resumed run: blocks 0..j of this store are the rows the chain file already holds (sampler drivers load the
        file and append it as one block).  When the file is one of ours (extensible datasets) it simply keeps growing:
        nothing is rewritten, nothing can be lost.  Returns True when adopted."""
        if not os.path.isfile(self.h5):
            return False
        try:
            ap = h5lite.Appender.open(self.h5)
        except Exception:
            return False
        names = self._names()
        try:
            if not all(n in ap.ds for n in names) or any(ap.ds[n]["chunk_rows"] != self.CHUNK_ROWS for n in names):
                raise ValueError
            have = ap.nrows(names[0])
            rows, j = 0, 0
            while j <= k and rows < have:
                rows += len(self.chain[j]); j += 1
            if rows != have or have == 0:
                raise ValueError
            self._to_host(j - 1)
            with h5lite.File(self.h5) as f:                      # the block really is what the file ends with
                last = f[names[0]][have - 1:have]
            if not np.array_equal(np.asarray(last)[0], np.asarray(self.chain[j - 1])[-1]):
                raise ValueError
        except Exception:
            ap.close()
            return False
        self._appender, self._appended = ap, j
        return True

    def _start_appender(self, k):
        """The extensible chain file.  A file of ours that already holds the leading blocks is continued in place;
        otherwise a NEW file is built under ``<name>.h5.tmp`` -- the blocks before k (a resumed run: the old rows)
        included -- and takes the old file's place only when it is complete, so a kill at any moment leaves a readable
        chain behind."""
        self._to_host(k)
        if self._adopt_existing(k):
            return
        nw, nd, spec = self._spec(k)
        names = self._names()
        tmp = self.h5 + ".tmp"
        if self.layout == "zeus":
            ap = h5lite.Appender.create(tmp, spec, chunk_rows=self.CHUNK_ROWS)
        else:
            ap = h5lite.Appender.create(tmp, {n.split("/")[1]: v for n, v in spec.items()}, group="mcmc",
                                        group_attrs=dict(version="3.0.2", nwalkers=np.int64(nw), ndim=np.int64(nd), has_blobs=False,
                                                         iteration=np.int64(0)),
                                        fixed={"accepted": np.zeros(nw)}, chunk_rows=self.CHUNK_ROWS)
        for j in range(k):                                          # what came before block k goes in before the swap
            self._to_host(j)
            ap.append({names[0]: self.chain[j], names[1]: self.chain_transformed[j], names[2]: self.log_prob[j]})
        if self.layout != "zeus" and k > 0:
            ap.set_attr("mcmc", "iteration", ap.nrows(names[0]))
        ap.close()
        os.replace(tmp, self.h5)
        self._appender = h5lite.Appender.open(self.h5)
        self._appended = k

    def _append_block(self, k, accepted):
        if self._appender is None:
            self._start_appender(k)
        names = self._names()
        while self._appended <= k:                                  # (blocks of a resumed run first, then block k)
            j = self._appended
            self._to_host(j)
            z, th, lp = self.chain[j], self.chain_transformed[j], self.log_prob[j]
            self._appender.append({names[0]: z, names[1]: th, names[2]: lp})
            self._appended += 1
            if z.nbytes >= self.DROP_BYTES:
                # a big block that is in the file does not stay in host memory as well (at 4096 walkers the chain grows
                # by 1.1 MB per iteration): only its length remains here, readers go to the file; its pinned buffers
                # return to the allocator for the next block
                self.chain[j] = self.chain_transformed[j] = self.log_prob[j] = _OnDisk(len(z))
        if self.layout != "zeus":
            self._appender.set_attr("mcmc", "iteration", self._appender.nrows(names[0]))
            if accepted is not None:
                self._appender.set_data("mcmc/accepted", self._acc_host(accepted))

    def arrays(self):
        self.drain()
        for k in range(len(self.chain)):
            self._to_host(k)
        if any(isinstance(b, _OnDisk) for b in self.chain):     # written blocks were dropped: the file has them (and
            if self._appender is not None:                      # whatever is not in it yet goes there first)
                if len(self.chain) > self._appended:
                    self._append_block(len(self.chain) - 1, None)
                self._appender.fh.flush()
            d = self.read_h5(self.h5)
            return d["chain"], d["chain_transformed"], d["log_prob"]
        return (np.concatenate(self.chain), np.concatenate(self.chain_transformed), np.concatenate(self.log_prob))

    def flush(self, final=True):
        """``final=False`` (the incremental flush after every convergence check, sampler.py:359/720): the blocks that
        are not on disk yet are appended to ``<name>.h5`` by the writer thread, one chunk per dataset and block, as the
        reference's HDF5 backend does -- the file is current after every flush and a killed run resumes from it.
        ``final=True``: whatever is left, then the file is closed (a store that was never flushed incrementally writes
        contiguous datasets in one pass); ``<name>.txt`` on request."""
        if not final:
            for k in range(self._flushed, len(self.chain)):
                self._enqueue(None, (k, self.accepted.clone() if torch.is_tensor(self.accepted) else np.array(self.accepted)))
            self._flushed = len(self.chain)
            return
        self.drain()
        for k in range(len(self.chain)):
            self._to_host(k)
        self.accepted = self._acc_host(self.accepted) if self.accepted is not None else None
        if self._appender is not None:                      # the file has grown with the run: add what is left, done
            if len(self.chain) > self._appended:
                self._append_block(len(self.chain) - 1, self.accepted)
            self._appender.close()
            self._appender = None
        else:
            # never flushed incrementally: the blocks go to the file one after the other (contiguous datasets), without a
            # concatenated copy of the chain in memory
            self.write_h5(self.h5, list(self.chain), list(self.chain_transformed), list(self.log_prob), self.accepted, self.layout)
        if self.write_txt:
            z, th, lp = self.arrays()
            flat = np.concatenate([th.reshape(-1, th.shape[-1]), lp.reshape(-1, 1)], axis=1)
            np.savetxt(self.base + ".txt", flat[-100000:])
        for f in self._parts(self.base) + [self.npz]:       # superseded by the consolidated file
            if os.path.isfile(f):
                os.remove(f)
        self._flushed = len(self.chain)

    @staticmethod
    def write_h5(path, z, th, lp, accepted, layout="emcee"):
        w = h5lite.Writer()
        nbytes = lambda a: sum(b.nbytes for b in a) if isinstance(a, list) else a.nbytes
        first = z[0] if isinstance(z, list) else z
        nsteps = sum(len(b) for b in z) if isinstance(z, list) else len(z)
        if layout == "zeus":
            for name, a in (("samples", z), ("chain_transformed", th), ("logprob", lp)):
                w.dataset(None, name, a, compression="gzip" if nbytes(a) <= ChainStore.GZIP_LIMIT else None)
        else:
            nw, nd = first.shape[1], first.shape[2]
            g = w.group("mcmc", attrs=dict(version="3.0.2",                  # the emcee release whose layout this is
                                           nwalkers=np.int64(nw), ndim=np.int64(nd), has_blobs=False,
                                           iteration=np.int64(nsteps)))
            w.dataset(g, "accepted", np.zeros(nw) if accepted is None else np.asarray(accepted, np.float64))
            w.dataset(g, "chain", z)
            w.dataset(g, "chain_transformed", th)
            w.dataset(g, "log_prob", lp)
        tmp = path + ".tmp"
        w.save(tmp)
        os.replace(tmp, path)

    @staticmethod
    def read_h5(path):
        """Either layout, written here or by the reference (h5py)."""
        with h5lite.File(path) as f:
            if "mcmc" in f:
                g = f["mcmc"]
                n = int(g.attrs["iteration"])
                out = {"chain": g["chain"].read(nrows=n), "log_prob": g["log_prob"].read(nrows=n),
                       "accepted": g["accepted"].read()}
                out["chain_transformed"] = g["chain_transformed"].read(nrows=n) if "chain_transformed" in g else None
            else:
                out = {"chain": f["samples"].read(), "chain_transformed": f["chain_transformed"].read(),
                       "log_prob": f["logprob"].read()}
                out["accepted"] = np.zeros(out["chain"].shape[1])
            out["iteration"] = len(out["chain"])
            return out

    def _part(self, k):
        return "%s.part%05d.npz" % (self.base, k)

    @staticmethod
    def _parts(base):
        import glob
        return sorted(glob.glob(base + ".part*.npz"))

    @staticmethod
    def _load_consolidated(base):
        if os.path.isfile(base + ".h5"):
            return ChainStore.read_h5(base + ".h5")
        if os.path.isfile(base + ".npz"):
            d = np.load(base + ".npz")
            return {k: d[k] for k in d.files}
        return None

    @staticmethod
    def load(filename):
        base = filename[:-3] if filename.endswith(".h5") else filename
        parts = ChainStore._parts(base)
        d = ChainStore._load_consolidated(base)
        if parts:                                           # an interrupted run: (consolidated file, if any) + parts
            blocks = [np.load(f) for f in parts]
            out = {k: np.concatenate([b[k] for b in blocks]) for k in ("chain", "chain_transformed", "log_prob")}
            out["accepted"] = blocks[-1]["accepted"]
            if d is not None:
                for k in ("chain", "chain_transformed", "log_prob"):
                    out[k] = np.concatenate([d[k], out[k]])
            out["iteration"] = len(out["chain"])
            return out
        if d is None:
            raise FileNotFoundError(base + ".h5")
        return d

    def get_last_sample(self):
        self.drain()
        return self.load(self.base)["chain"][-1]


def get_good_walker_list(log_prob_samples):
    """util.py:57-66 (``walkercut=True``; no caller in the reference passes it): the walkers whose mean log-probability over
    the last 10000 steps, truncated to an integer, falls into the highest of sklearn's ``KMeans()`` clusters (8 clusters,
    unseeded, as there; ``np.int`` of the reference's numpy is today's ``int``)."""
    from sklearn.cluster import KMeans
    x = np.mean(log_prob_samples[-10000:, :], axis=0)
    X = np.array(list(zip(x, np.zeros(len(x)))), dtype=int)
    ms = KMeans()
    ms.fit(X)
    best = ms.labels_[np.argmax(ms.cluster_centers_[:, 0])]
    print(np.where(ms.labels_ == best)[0], ms.cluster_centers_, ms.labels_)
    return np.where(ms.labels_ == best)[0]


def read_chain_and_cut(chainname, nk, ntimes=20, walkercut=False, method="emcee", flat=False):
    """util.py:68-94: last ``nk`` autocorrelation times of the stored chain (theta space)."""
    d = ChainStore.load(chainname)
    if nk > ntimes:
        print("Error: keep number greater then chain samples. nk: {0}, ntimes: {1}. This will lead to inclusion of all "
              "burn in step".format(nk, ntimes))
    if torch.cuda.is_available():                   # the same estimator, batched on the device (see DeviceChain)
        dc = DeviceChain()
        dc.append(np.asarray(d["chain"], np.float32))
        tau = dc.integrated_time(all_walkers=True)              # util.py:78-80 uses every walker; a one-shot call
    else:
        tau = integrated_time(d["chain"])
    nkeep = int(np.nanmedian(tau) * nk)
    chain = d["chain_transformed"]
    lp = d["log_prob"]
    good = get_good_walker_list(np.asarray(lp)) if walkercut else slice(None)       # util.py:86-89
    chain = np.asarray(chain[-nkeep:][:, good].reshape(-1, chain.shape[-1]), np.float64)
    lp = np.asarray(lp[-nkeep:][:, good], np.float64)
    if flat:
        lp = lp.reshape(-1, 1)
    return chain, lp, d


# ------------------------------------------------------------------ stretch-move ensemble on the device
class EnsembleSampler(object):
    """Affine-invariant ensemble sampler with every walker evaluated in one batched GPU call.

    ``log_prob`` is a ``linna_amd.util.Log_prob`` (device pipeline).  ``nwalkers`` is the number
    of walkers OWNED BY THIS RANK; with ``exchange='allgather'`` and a process group the
    complementary set of a half step is drawn from the walkers of all ranks (one all-gather of
    ``[nwalkers, ndim]`` per half step), otherwise ranks run independent sub-ensembles and only
    chain state is gathered.
    """

    def __init__(self, nwalkers, ndim, log_prob, a=2.0, seed=0, randomize_split=True, dist_group=None,
                 exchange="none", fused=True):
        self.nw, self.ndim, self.lp, self.a, self.seed = int(nwalkers), int(ndim), log_prob, float(a), int(seed)
        if self.nw < 2 or self.nw % 2:
            raise ValueError("need an even number of walkers >= 2")
        self.randomize_split = randomize_split
        self.group, self.exchange = dist_group, exchange
        self.fused = None if fused else False        # None: try linna_stretch_half_step on the first half step
        # a user loglikelihoodfunc / externalloglike is host code: proposals and the Metropolis test stay kernels,
        # the log-probability of a half ensemble goes through Log_prob.evaluate_any (emulator on the GPU + callbacks)
        self.host_lp = not getattr(log_prob, "device_only", True)
        if self.host_lp:
            self.fused = False
        p = log_prob._ensure()
        self.dev = p["dev"]
        self.ctx = _lib.ctx(self.dev.index)
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=self.dev)
        self.ld = _lib.ld4(self.ndim)
        self.coords, self.logp = z(self.nw, self.ld), z(self.nw)
        self.half = self.nw // 2
        self.Q, self.factors, self.lp_new = z(self.half, self.ld), z(self.half), z(self.half)
        self.naccept = torch.zeros(self.nw, dtype=torch.int32, device=self.dev)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.iteration = 0
        self._dev_steps = 0                          # value of the device step counter (== iteration unless fused)
        self._rs = np.random.RandomState(self.seed ^ 0x5EED)
        self._split_pos, self._split_host, self._split_dev, self._split_evt = 0, None, None, None
        self._split, self._draws, self._split_idx = None, 0, np.arange(self.nw)
        self.block_run = None                        # None: try linna_stretch_run (a whole block of iterations per C call)
        from . import dist as ldist
        self.rank, self.world = ldist.rank(dist_group), ldist.world_size(dist_group)
        self._gathered = None

    # -- state
    def set_state(self, x0):
        x0 = torch.as_tensor(np.asarray(x0, np.float32), device=self.dev)
        if x0.shape != (self.nw, self.ndim):
            raise ValueError("x0 must be [nwalkers, ndim]")
        self.coords.zero_()
        self.coords[:, :self.ndim].copy_(x0)
        self._lnp(self.coords, self.logp)
        if not bool(torch.isfinite(self.logp).all()):
            raise ValueError("initial state has non-finite log-probability")   # emcee raises the same way

    def _lnp(self, X, out):
        """lnP of the rows of ``X[B, ld]`` into ``out[B]``."""
        if self.host_lp:
            out.copy_(self.lp.evaluate_any(X))
        else:
            self.lp.evaluate(X, out=out)

    _SPLIT_CHUNK = 64

    def _draw_splits(self, n):
        """Device int32 [n, 2, nw/2]: the random equal splits of the next ``n`` iterations (== the shuffled ``arange % 2``
        of emcee's RedBlueMove), drawn on the host in one go and shipped in ONE asynchronous copy from pinned memory (a
        per-iteration blocking copy left the GPU idle for ~25 us of every 150 us iteration)."""
        slot = self._draws & 1
        self._draws += 1
        if self._split_host is None:
            self._split_host, self._split_evt = [None, None], [None, None]
        if self._split_evt[slot] is not None:
            self._split_evt[slot].synchronize()                 # the copy that last read this pinned buffer is done
        if self._split_host[slot] is None or len(self._split_host[slot]) < n:
            self._split_host[slot] = torch.empty((max(n, self._SPLIT_CHUNK), 2, self.half), dtype=torch.int32).pin_memory()
        host = self._split_host[slot].numpy()
        idx = self._split_idx
        for i in range(n):
            self._rs.shuffle(idx)
            host[i] = idx.reshape(2, self.half)
        dev = torch.empty((n, 2, self.half), dtype=torch.int32, device=self.dev)
        dev.copy_(self._split_host[slot][:n], non_blocking=True)
        self._split_evt[slot] = torch.cuda.Event()
        self._split_evt[slot].record()
        return dev

    def _splits(self):
        """This iteration's split, device int32 [2, nw/2] (drawn 64 iterations at a time)."""
        if not self.randomize_split:
            if self._split is None:
                self._split = torch.as_tensor(np.arange(self.nw, dtype=np.int32).reshape(2, self.half), device=self.dev)
            return self._split
        if self._split_dev is None or self._split_pos >= len(self._split_dev):
            self._split_dev, self._split_pos = self._draw_splits(self._SPLIT_CHUNK), 0
        halves = self._split_dev[self._split_pos]
        self._split_pos += 1
        return halves

    def _splits_block(self, n):
        """The splits of the next ``n`` iterations as one device tensor [n, 2, nw/2] (what is left of the chunk `_splits`
        was handing out first: both routes consume the same sequence of permutations)."""
        rem = None
        if self._split_dev is not None and self._split_pos < len(self._split_dev):
            rem = self._split_dev[self._split_pos:self._split_pos + n]
            self._split_pos += len(rem)
            if len(rem) == n:
                return rem
        new = self._draw_splits(n - (0 if rem is None else len(rem)))
        return new if rem is None else torch.cat([rem, new])

    def step(self):
        """One ensemble iteration (both halves).  Everything is enqueued on the current stream."""
        st = _lib.stream()
        halves = self._splits()
        lib_seed = C.c_uint64(self.seed + 0x9E3779B97F4A7C15 * (self.rank + 1) & 0xFFFFFFFFFFFFFFFF)
        for h in (0, 1):
            S, Cc = halves[h], halves[1 - h]
            comp, ldc, cidx, nc = self.coords, self.ld, Cc, self.half
            if self.exchange == "allgather" and self.world > 1:
                comp, cidx, nc = self._allgather_complement(Cc)
            if self.fused is not False:
                # propose + log-probability + accept in ONE launch (bit-identical to the three below)
                rc = _lib.load().linna_stretch_half_step(
                    self.lp._ensure()["handle"], _lib.ptr(self.coords), self.ld, self.ndim, _lib.ptr(self.logp),
                    _lib.iptr(S), self.half, _lib.ptr(comp), ldc, _lib.iptr(cidx), nc, lib_seed,
                    _lib.iptr(self.step_dev), self.iteration - self._dev_steps, h, self.a, _lib.iptr(self.naccept), st)
                if rc == 0:
                    self.fused = True
                    continue
                if rc != _lib.ERR_UNSUPPORTED or self.fused is True:
                    _lib.check(rc)
                self.fused = False                   # this log-probability cannot: three launches from now on
            _lib.call("linna_stretch_propose", self.ctx, _lib.ptr(self.coords), self.ld, self.ndim, _lib.iptr(S), self.half,
                      _lib.ptr(comp), ldc, _lib.iptr(cidx), nc, lib_seed, _lib.iptr(self.step_dev), h, self.a,
                      _lib.ptr(self.Q), self.ld, _lib.ptr(self.factors), st)
            self._lnp(self.Q, self.lp_new)
            _lib.call("linna_stretch_accept", self.ctx, _lib.ptr(self.coords), self.ld, self.ndim, _lib.ptr(self.logp),
                      _lib.iptr(S), self.half, _lib.ptr(self.Q), self.ld, _lib.ptr(self.lp_new), _lib.ptr(self.factors),
                      lib_seed, _lib.iptr(self.step_dev), h, _lib.iptr(self.naccept), st)
        if self.fused is not True:                   # the fused entry takes the iteration as an offset instead
            _lib.call("linna_step_increment", self.ctx, _lib.iptr(self.step_dev), st)
            self._dev_steps += 1
        self.iteration += 1

    def _allgather_complement(self, Cc):
        """Complementary walkers of ALL ranks: gather this rank's complementary half."""
        from . import dist as ldist
        gathered = ldist.gather_rows(self.coords[Cc.long()], self.group)
        if self._gathered is None:
            self._gidx = torch.arange(gathered.shape[0], dtype=torch.int32, device=self.dev)
        self._gathered = gathered
        return gathered, self._gidx, gathered.shape[0]

    def run(self, nsteps, store=True):
        """Advance ``nsteps``; returns (chain[nsteps, nw, ndim], logp[nsteps, nw]) as device tensors.  One rank, fused half
        steps: ONE C call enqueues the whole block (linna_stretch_run: 2 launches per iteration whose finish also writes the
        chain rows -- the per-iteration host work of the loop below, two ctypes calls and two device copies, cost as much
        as the 60 us an iteration of 128 walkers takes on the GPU); bit-identical to that loop."""
        chain = torch.empty((nsteps, self.nw, self.ndim), dtype=torch.float32, device=self.dev) if store else None
        lps = torch.empty((nsteps, self.nw), dtype=torch.float32, device=self.dev) if store else None
        if (nsteps > 0 and self.block_run is not False and self.fused is not False and type(self).step is EnsembleSampler.step
                and not (self.exchange == "allgather" and self.world > 1)):
            # the splits are host draws (41 us each at 4096 walkers): big ensembles go out in pieces of a few iterations, so
            # that the GPU starts on the first piece while the host shuffles the next
            piece = nsteps if not self.randomize_split else max(8, min(nsteps, 65536 // self.nw))
            lib_seed = C.c_uint64(self.seed + 0x9E3779B97F4A7C15 * (self.rank + 1) & 0xFFFFFFFFFFFFFFFF)
            i0 = 0
            while i0 < nsteps:
                n = min(piece, nsteps - i0)
                if self.randomize_split:
                    sp, stride = self._splits_block(n), self.nw
                else:
                    sp, stride = self._splits(), 0
                rc = _lib.load().linna_stretch_run(
                    self.lp._ensure()["handle"], _lib.ptr(self.coords), self.ld, self.ndim, _lib.ptr(self.logp), self.nw, _lib.iptr(sp),
                    stride, n, lib_seed, _lib.iptr(self.step_dev), self.iteration - self._dev_steps, self.a, _lib.iptr(self.naccept),
                    _lib.ptr(chain[i0:]) if store else None, _lib.ptr(lps[i0:]) if store else None, _lib.stream())
                if rc != 0:
                    break
                self.block_run = self.fused = True
                self.iteration += n
                i0 += n
            if i0 == nsteps:
                return chain, lps
            if rc != _lib.ERR_UNSUPPORTED or self.block_run is True:
                _lib.check(rc)
            self.block_run = False
            if self.randomize_split:                 # hand the drawn splits back to the per-iteration route
                self._split_dev, self._split_pos = sp, 0
        for i in range(nsteps):
            self.step()
            if store:
                chain[i].copy_(self.coords[:, :self.ndim])
                lps[i].copy_(self.logp)
        return chain, lps

    def theta_of(self, z):
        """Physical parameters of latent points ``z[..., ndim]`` (device), via the prior-map kernel."""
        p = self.lp._ensure()
        flat = z.reshape(-1, self.ndim).contiguous()
        n = flat.shape[0]
        th = torch.empty_like(flat)
        x = torch.empty((n, self.ld), dtype=torch.float32, device=self.dev)
        k, d = p["keep"], p["desc"]
        _lib.call("linna_prior_map_fwd", self.ctx, _lib.ptr(flat), self.ndim, n, self.ndim, _lib.iptr(k["is_flat"]),
                  _lib.ptr(k["a1"]), _lib.ptr(k["a2"]), None, _lib.ptr(k["xmean"]), _lib.ptr(k["xstd"]), _lib.ptr(x), self.ld,
                  _lib.ptr(th), self.ndim, _lib.stream())
        return th.reshape(z.shape)

    def gather_chain(self, chain, lps):
        """RCCL all-gather of this rank's chain block: [nsteps, world*nw, ndim] on every rank."""
        from . import dist as ldist
        return ldist.gather_chain(chain, lps, self.group)


# ------------------------------------------------------------------ ensemble slice sampling (zeus)
class _FastOverflow(Exception):
    """A walker of the one-call slice half step needed more rounds than the call holds (SliceEnsembleSampler._guard)."""


class SliceEnsembleSampler(EnsembleSampler):
    """zeus' ensemble slice sampler (Karamanis & Beutler 2021; driven by sampler.py:728-735) with
    every trial point of a half ensemble evaluated in one batched GPU call.

    Per half step: differential-move directions ``mu (c_a - c_b)``, slice height
    ``Z0 = logp + log u``, stepping out (both bracket ends evaluated together, 2 x nw/2 points per
    round) and shrinking (nw/2 points per round) until every walker has accepted.  ``mu`` is tuned
    during the first iterations as zeus does: ``mu *= 2 nexp / (nexp + ncon)`` until the expansion
    fraction stays within ``tolerance`` of 1/2 for ``patience`` iterations.  Third-party algorithm
    restated from the publication (zeus-mcmc is not in the reference tree): parity unpinned; every half step is replayed
    decision by decision against ``oracle.sampling.slice_half_step`` (tests/test_gpu_slice_replay.py) and the posterior is
    checked statistically.  ``mu`` has zeus' meaning (direction ``2 mu (c_a - c_b)``), ``maxsteps`` is zeus' stepping-out
    budget (``J = floor(maxsteps u)`` steps to the left at most, ``maxsteps - 1 - J`` to the right).
    """

    FAST_EXPANSIONS = 12            # at least so many per side and half step on the one-call path (a tuned mu needs about one)
    FAST_TRIALS = 32                # at least so many shrinking trials per walker and half step (each halves the bracket)
    FAST_FIRST = None               # bracket ends per side in the first stepping-out round (None: by ensemble size, below)
    FAST_CAP = (8, 32)              # most ends per side / trials a later round looks ahead

    def __init__(self, nwalkers, ndim, log_prob, mu=1.0, seed=0, tune=True, tolerance=0.05, patience=5, maxsteps=10000,
                 maxiter=100000, dist_group=None, exchange="none", fast=None):
        EnsembleSampler.__init__(self, nwalkers, ndim, log_prob, seed=seed, dist_group=dist_group, exchange=exchange)
        z = lambda *sh: torch.zeros(sh, dtype=torch.float32, device=self.dev)
        ns = self.half
        self.mu = float(mu)
        # zeus' direction is 2 mu (c_a - c_b) (moves.DifferentialMove.get_direction); the kernels multiply by mu_dev[0]
        self.mu_dev = torch.full((1,), 2.0 * self.mu, dtype=torch.float32, device=self.dev)
        self.tune, self.tolerance, self.patience, self.maxsteps, self.maxiter = tune, tolerance, patience, maxsteps, maxiter
        self._tune_count = 0
        self.DIR, self.Q2 = z(ns, self.ld), z(2 * ns, self.ld)
        self.Z0, self.Wacc, self.Zacc = z(ns), z(ns), z(ns)
        self.LR = z(2 * ns)                              # bracket ends, contiguous: both evaluated in one launch
        self.L, self.R = self.LR[:ns], self.LR[ns:]
        self.fused_points = None                         # None: try linna_logprob_eval_slice_points first
        self.ntrial = 2                                  # shrink trials per round (2 x nw/2 points = one full launch)
        self.W = z(self.ntrial * ns)
        self.Z2 = z(2 * ns)
        self.flags = torch.zeros(3 * ns, dtype=torch.int32, device=self.dev)
        self.counters = torch.zeros(4, dtype=torch.int32, device=self.dev)   # expansions, contractions, active (ping-pong)
        self._neval_host = 0
        self._cs, self._pin = None, None
        # One C call per half step (linna_slice_half_step) once mu is tuned: speculative rounds -- `m` bracket ends per side
        # per stepping-out round, `nt` trials per shrinking round -- and a fixed number of rounds, gated off on the device
        # behind the one that finished the last walker.  The host then never waits inside an iteration (with 4-128 walkers
        # the round-by-round loop below was bound by its own launches and count read-backs, not by the GPU).  The first
        # round of each kind evaluates every walker and is sized by the ensemble (a launch of up to ~2000 points costs what
        # one point costs; beyond ~4000 the evaluation is compute bound at 14 ns a point); the later rounds evaluate only
        # the walkers still active, so they look twice as far ahead each time for next to nothing and few rounds cover
        # FAST_EXPANSIONS stepping-out steps per side and FAST_TRIALS shrinking trials.  A walker that needs more is
        # counted (and kept in place), and `run` / `step` then take the sampler back to the state they started from and
        # redo their iterations on the unbounded round loop: the chain is always the round loop's.
        self.fast = fast
        self._fast_ok = None                              # None: try the entry on the first tuned iteration
        first = self.FAST_FIRST or (8 if ns <= 64 else 4 if ns <= 256 else 2 if ns <= 2048 else 1)   # measured: tools/slice_probe.py
        if self.FAST_FIRST is None and ns <= 64:
            # small ensembles (the reference's 128 walkers): every round costs ~30 us of launches whatever it evaluates, and the
            # usage counters (linna_slice_half_step, `round_usage`) show nothing left behind 8 bracket ends per side or behind 16
            # trials in 58 000 walker half steps (R5, tools/slice_probe.py): ONE stepping-out round of 8, the second shrinking
            # round kept as the rescue (16 + 16 trials); a walker beyond that sends the run to the round loop as before
            self.set_schedule([8], [16, 16])
        else:
            self.set_schedule(self._schedule(first, self.FAST_EXPANSIONS, self.FAST_CAP[0]),
                              self._schedule(2 * first, self.FAST_TRIALS, self.FAST_CAP[1]))
        self._last_nexp = None                            # expansions of the last tuning iteration (whole ensemble)
        self._fast_after = 0                              # no one-call steps before this iteration (set after an overflow)
        self._guarded = False
        self.noverflow = 0                                # runs redone on the round loop
        # test hook: called behind every half step with (half index, S, dict of the device arrays Z0 / L / R / Wacc / Zacc of the
        # half ensemble) -- tests/test_gpu_slice_replay.py replays each half step against oracle.sampling.slice_half_step
        self.probe = None

    def set_schedule(self, m_sched, nt_sched):
        """Bracket ends per side of each stepping-out round and trials of each shrinking round of the one-call path."""
        self.m_sched, self.nt_sched = [int(v) for v in m_sched], [int(v) for v in nt_sched]
        self.m, self.nt_fast = max(self.m_sched), max(self.nt_sched)         # (scratch sizes)
        self._expect, self._expect_base, self._expect_seen, self.expected_rows = None, (0.0, np.zeros(0)), -1, None
        self.nexp_rounds, self.nshr_rounds = len(self.m_sched), len(self.nt_sched)
        self._m_arr = (C.c_int * self.nexp_rounds)(*self.m_sched)
        self._nt_arr = (C.c_int * self.nshr_rounds)(*self.nt_sched)
        if getattr(self, "_fast_bufs", None) is not None:              # (a schedule change mid-run: keep the evaluation count)
            self._neval_host += int(self._fast_bufs["counters"][3].item())
        self._fast_bufs = None

    @staticmethod
    def _schedule(first, total, cap):
        """first, 2 first, 4 first ... (at most `cap` each) until at least `total` are covered; two rounds at least."""
        out = [min(first, cap)]
        while sum(out) < total or len(out) < 2:
            out.append(min(2 * out[-1], cap))
        return out

    # -- data-dependent rounds with one round of lookahead ------------------------------------------
    # A round = a few small kernels + one evaluation + a kernel that counts the walkers still active.
    # Reading that count on the host every round left the GPU idle while the host came back and queued
    # the next round (~35 us of every ~110 us).  Instead round r+1 is queued BEFORE the count of round r
    # is known: its evaluation is gated on the device-side count (linna_logprob_eval_if), so when round
    # r turns out to have been the last, round r+1 costs a few empty launches.  The count travels over
    # a second stream into pinned memory, so waiting for it does not wait for round r+1.
    def _run_rounds(self, enqueue, maxrounds, what="expansions"):
        if self._cs is None:
            self._cs = torch.cuda.Stream(device=self.dev)
            self._pin = torch.zeros(2, dtype=torch.int32).pin_memory()
        main = torch.cuda.current_stream(self.dev)
        pending = None
        for r in range(maxrounds):
            slot = 2 + (r & 1)
            self.counters[slot:slot + 1].zero_()
            gate = C.c_void_p(self.counters.data_ptr() + 4 * (2 + ((r - 1) & 1))) if r > 0 else None
            enqueue(r, slot, gate)
            done = torch.cuda.Event()
            done.record(main)
            with torch.cuda.stream(self._cs):
                self._cs.wait_event(done)
                self._pin[r & 1:(r & 1) + 1].copy_(self.counters[slot:slot + 1], non_blocking=True)
                landed = torch.cuda.Event()
                landed.record(self._cs)
            if pending is not None:
                pending[0].synchronize()
                if int(self._pin[pending[1]]) == 0:
                    return                              # round r-1 finished every walker; round r (queued) is gated off
            pending = (landed, r & 1)
        pending[0].synchronize()
        if int(self._pin[pending[1]]) != 0:              # zeus: RuntimeError behind `maxiter` passes (sampler.py:728: maxiter=1E5)
            raise RuntimeError("Number of %s exceeded maximum limit! Make sure that the pdf is well-defined." % what)

    def _eval_if(self, Q, Z, gate):
        if self.host_lp:                                # host callbacks: evaluated whatever the gate says (results of a
            Z.copy_(self.lp.evaluate_any(Q))            # gated-off round are never used)
            return
        p = self.lp._ensure()
        B = Q.shape[0]
        _lib.call("linna_logprob_eval_if", p["handle"], _lib.ptr(Q), Q.stride(0), B, _lib.ptr(self.lp._workspace(B, False)),
                  _lib.ptr(Z), None, 0, gate, _lib.stream())

    def _eval_points(self, S, w, nrep, gate):
        """lnP at coords[S[k]] + w[j*ns + k] DIR[k] into Z2[:nrep*ns]: one launch that never writes the points when
        the whole-network kernel serves this log-probability, else linna_slice_points + the gated evaluation."""
        ns, st, P = self.half, _lib.stream(), _lib.ptr
        if self.host_lp:
            self.fused_points = False
        if self.fused_points is not False:
            rc = _lib.load().linna_logprob_eval_slice_points(
                self.lp._ensure()["handle"], P(self.coords), self.ld, self.ndim, _lib.iptr(S), ns, P(self.DIR), self.ld,
                P(w), nrep, P(self.Z2), gate, st)
            if rc == 0:
                self.fused_points = True
                return
            if rc != _lib.ERR_UNSUPPORTED or self.fused_points is True:
                _lib.check(rc)
            self.fused_points = False
        _lib.call("linna_slice_points", self.ctx, P(self.coords), self.ld, self.ndim, _lib.iptr(S), ns, P(self.DIR), self.ld,
                  P(w), P(self.Q2), self.ld, nrep, st)
        self._eval_if(self.Q2[:nrep * ns], self.Z2[:nrep * ns], gate)

    @property
    def neval(self):
        """Log-probability evaluations so far (points, both paths; reads a device counter)."""
        dev = int(self._fast_bufs["counters"][3].item()) if self._fast_bufs is not None else 0
        return self._neval_host + dev

    def round_usage(self):
        """How much of the one-call path's look-ahead the run has used: for every stepping-out and shrinking round, the mean
        fraction of a half ensemble's walkers still active BEHIND it (device counters of linna_slice_half_step; one read)."""
        if self._fast_bufs is None:
            return None
        nr = self.nexp_rounds + self.nshr_rounds
        c = self._fast_bufs["counters"].cpu().numpy().astype(np.float64)
        calls = c[4 + 2 * nr] - 1                       # (the counts of the latest call are added by the next one)
        if calls < 1:
            return None
        frac = c[4 + nr:4 + 2 * nr] / (calls * self.half)
        return {"half_steps": int(calls), "active_after_expand_round": frac[:self.nexp_rounds].tolist(),
                "active_after_shrink_round": frac[self.nexp_rounds:].tolist()}

    def _use_fast(self):
        """The one-call half step serves every ensemble size (measured: 4.8x the round loop at 128 walkers, 1.3x at 4096).
        While mu is still being tuned it is used only once an iteration has needed few expansions: the walkers of a run
        start in a 1e-3 ball (util.py:937) and the first iterations step out hundreds of times per side, which the
        unbounded round loop handles and the fixed rounds of the one-call path would not."""
        if self.fast is False or self.host_lp or self._fast_ok is False or self.iteration < self._fast_after:
            return False
        # (_last_nexp counts the WHOLE ensemble's expansions when the walkers are sharded over ranks, see _tune_mu)
        if self.tune and not (self._last_nexp is not None and self._last_nexp < 2.0 * self.nw * (self.world if self._shared() else 1)):
            return False
        return True

    def _step_fast(self, halves, seed):
        """Both half steps through linna_slice_half_step; False when the entry does not serve this log-probability."""
        ns, P, I = self.half, _lib.ptr, _lib.iptr
        if self._fast_bufs is None:
            z = lambda *sh: torch.zeros(sh, dtype=torch.float32, device=self.dev)
            nrep = max(2 * self.m, self.nt_fast)
            self._fast_bufs = dict(state=z(5 * ns), W=z(2 * self.m * ns), Wd=z(self.nt_fast * ns), Zt=z(nrep * ns),
                                   list=torch.zeros(nrep * ns, dtype=torch.int32, device=self.dev),
                                   counters=torch.zeros(5 + 2 * (self.nexp_rounds + self.nshr_rounds), dtype=torch.int32, device=self.dev))
        b, st = self._fast_bufs, _lib.stream()
        self._refresh_expectation()
        for h in (0, 1):
            S, Cc = halves[h], halves[1 - h]
            comp, ldc, cidx, nc = self.coords, self.ld, Cc, self.half
            if self.exchange == "allgather" and self.world > 1:
                comp, cidx, nc = self._allgather_complement(Cc)
            rc = _lib.load().linna_slice_half_step(
                self.lp._ensure()["handle"], P(self.coords), self.ld, self.ndim, P(self.logp), I(S), ns, P(comp), ldc, I(cidx), nc,
                P(self.mu_dev), seed, I(self.step_dev), h, self._m_arr, self.nexp_rounds, self._nt_arr, self.nshr_rounds, P(self.DIR), self.ld,
                P(b["state"]), I(self.flags), P(b["W"]), P(b["Wd"]), P(b["Zt"]), I(b["list"]), I(b["counters"]), 1 if h == 0 else 0,
                1 if h == 1 else 0, self._expect, int(self.maxsteps), st)    # (the second half step's last kernel advances the device step counter)
            if rc != 0:
                if rc == _lib.ERR_UNSUPPORTED and h == 0 and self._fast_ok is None:
                    self._fast_ok = False
                    return False
                _lib.check(rc)
            if self.probe is not None:
                s5 = b["state"]
                self.probe(h, S, dict(Z0=s5[:ns], L=s5[ns:2 * ns], R=s5[2 * ns:3 * ns], Wacc=s5[3 * ns:4 * ns], Zacc=s5[4 * ns:],
                                      counters=b["counters"], fast=True))
        self._fast_ok = True
        self.iteration += 1
        self._fast_steps = getattr(self, "_fast_steps", 0) + 1
        if self.tune:
            c = b["counters"][:3].cpu().numpy()          # one read per iteration while mu is tuned, as the round loop
            if c[2]:
                if self._shared():
                    self._overflow_seen = True       # (ranks exchange walkers every half step: no rank may leave the loop alone)
                else:
                    raise _FastOverflow()
            self._tune_mu(int(c[0]), int(c[1]))
        return True

    EXPECT_AT = (16, 64, 256)       # one-call iterations after which the usage counters are read (then every 1024)
    USE_EXPECT = True               # (False: the later rounds' engines by the fixed rule, a quarter of the previous round's -- A/B)

    def _refresh_expectation(self):
        """The rounds after the first evaluate only the walkers still active; their launch is sized for all but runs the
        engine (4 / 8 / 16 rows per workgroup) that suits the EXPECTED number of trial points -- 60 points cost 56 us on the
        16-row engine and 29 on the 4-row one.  The expectation is what the usage counters have shown since the last look
        (mean active fraction behind each round x the next round's points per walker x 1.25 + 8; a round that has
        practically never run gets the 16-row engine, whose launch has the fewest workgroups to dismiss), read at fixed iteration
        counts so that a run is reproducible: 16, 64, 256, then every 1024 one-call iterations (one small device read each)."""
        n = getattr(self, "_fast_steps", 0)
        if not self.USE_EXPECT:
            return
        if not (n in self.EXPECT_AT or (n and n % 1024 == 0)) or n == getattr(self, "_expect_seen", -1):
            return
        self._expect_seen = n
        nr = self.nexp_rounds + self.nshr_rounds
        rows, base = self.expected_points(self._fast_bufs["counters"].cpu().numpy(), self.m_sched, self.nt_sched, self.half, self._expect_base)
        if rows is None:
            return
        self._expect_base = base
        self._expect = (C.c_int * nr)(*rows)
        self.expected_rows = rows

    @staticmethod
    def expected_points(counters, m_sched, nt_sched, half, base=None):
        """From the usage counters of linna_slice_half_step (include/linna_hip.h: ``[4 + nr + r]`` walkers still active behind
        round r summed over the earlier calls, ``[4 + 2 nr]`` the calls) and the counters' state at the last look
        (``base`` = (calls, sums)): the trial points each round is expected to evaluate -- entry r of the stepping-out rounds,
        ``len(m_sched) + r`` of the shrinking ones; 1 for the first round of each kind (it evaluates every walker), 2^20 for a
        round whose mean is below half a point -- and the new ``base``.  (None, base) when fewer than 8 calls are new."""
        nexp, nr = len(m_sched), len(m_sched) + len(nt_sched)
        c = np.asarray(counters, np.float64)
        calls, cum = c[4 + 2 * nr] - 1, c[4 + nr:4 + 2 * nr]
        last_calls, last_cum = base if base is not None and len(base[1]) == nr else (0.0, np.zeros(nr))
        if calls - last_calls < 8:
            return None, base
        frac = (cum - last_cum) / ((calls - last_calls) * half)
        pts = [2 * m for m in m_sched] + list(nt_sched)
        rows = [1] * nr
        for i in range(nr):
            if i and i != nexp:
                mean = pts[i] * frac[i - 1] * half
                # (a round that practically never runs: the engine with the fewest workgroups to launch and dismiss)
                rows[i] = int(mean * 1.25) + 8 if mean >= 0.5 else 1 << 20
        return rows, (calls, cum.copy())

    def _tune_mu(self, nexp, ncon):
        """zeus: mu *= 2 nexp / (nexp + ncon) until the expansion fraction stays within ``tolerance`` of 1/2 for
        ``patience`` iterations.  One ensemble sharded over ranks tunes ONE mu from the WHOLE ensemble's counts (zeus and
        the reference do: sampler.py:728-735): the two counts are summed over the ranks first, so every rank carries the
        same mu, leaves tuning at the same iteration and the chain is comparable to the single-rank one."""
        if self._shared():
            from . import dist as ldist
            t = torch.tensor([float(nexp), float(ncon)], dtype=torch.float32, device=self.mu_dev.device)
            ldist.allreduce_grads(t, None, self.group)
            nexp, ncon = (int(round(v)) for v in t.tolist())
        self._last_nexp = nexp
        nexp = max(1, nexp)
        self.mu *= 2.0 * nexp / (nexp + ncon)
        self.mu_dev.fill_(2.0 * self.mu)
        self._tune_count = self._tune_count + 1 if abs(nexp / (nexp + ncon) - 0.5) < self.tolerance else 0
        if self._tune_count > self.patience:
            self.tune = False
            self.tune_off_iteration = self.iteration

    def _shared(self):
        return self.world > 1 and self.exchange == "allgather"

    def _overflowed(self):
        """A walker needed more stepping-out steps or shrinking trials than the one-call path's rounds hold (one device
        read); with walkers exchanged between ranks: on ANY rank -- every rank then redoes the run."""
        mine = bool(self._fast_bufs is not None and getattr(self, "_fast_steps", 0)
                    and int(self._fast_bufs["counters"][2].item()) > 0) or getattr(self, "_overflow_seen", False)
        self._overflow_seen = False
        if self._shared():
            from . import dist as ldist
            return ldist.any_rank(mine, self.group)
        return mine

    def _snapshot(self):
        return dict(coords=self.coords.clone(), logp=self.logp.clone(), naccept=self.naccept.clone(), step_dev=self.step_dev.clone(),
                    mu=self.mu, tune=self.tune, tune_count=self._tune_count, last_nexp=self._last_nexp, iteration=self.iteration,
                    dev_steps=self._dev_steps, rs=self._rs.get_state(), split_pos=self._split_pos, split_dev=self._split_dev,
                    split_idx=self._split_idx.copy(), neval=self._neval_host)      # (a drawn chunk of splits is never written again)

    def _restore(self, k):
        torch.cuda.current_stream(self.dev).synchronize()
        self.coords.copy_(k["coords"]); self.logp.copy_(k["logp"]); self.naccept.copy_(k["naccept"]); self.step_dev.copy_(k["step_dev"])
        self.mu, self.tune, self._tune_count, self._last_nexp = k["mu"], k["tune"], k["tune_count"], k["last_nexp"]
        self.mu_dev.fill_(2.0 * self.mu)
        self.iteration, self._dev_steps, self._neval_host = k["iteration"], k["dev_steps"], k["neval"]
        self._rs.set_state(k["rs"]); self._split_pos, self._split_dev = k["split_pos"], k["split_dev"]
        self._split_idx[:] = k["split_idx"]
        if self._fast_bufs is not None:
            self._fast_bufs["counters"][2:3].zero_()

    def _guard(self, body):
        """Run ``body`` (iterations that may take the one-call path); if a walker overflowed its rounds, go back to the state
        at entry and run ``body`` again on the round loop -- what comes out is the round loop's chain either way."""
        if self._guarded or not self._use_fast_possible():
            return body()
        snap = self._snapshot()
        self._guarded = True
        try:
            try:
                out = body()
                redo = self._overflowed()
            except _FastOverflow:
                redo = True
            if redo:
                self._restore(snap)
                self.noverflow += 1
                self._fast_after = snap["iteration"] + 1 << 30       # (this attempt: round loop only)
                out = body()
                # a posterior whose brackets have a heavy tail (few walkers, directions between near neighbours: the 2-D
                # notebook problem at 4 walkers overflowed every few iterations) would otherwise spend its run on the round
                # loop: the schedule looks further ahead from now on -- the added rounds evaluate only the walkers still
                # active and cost a few gated-off launches -- and the one-call path stays in use; at the deepest level the
                # old rule applies (200 quiet iterations on the round loop)
                self._fast_after = self.iteration if self._escalate() else self.iteration + 200
            return out
        finally:
            self._guarded = False

    ESCALATIONS = 2                 # times an overflow may deepen the schedule (x4 stepping-out steps, x2 trials each time)

    def _escalate(self):
        """Deepen the one-call path's look-ahead after an overflow: four times the stepping-out steps per side, twice the
        trials, in rounds that double (at most 32 bracket ends per side / 64 trials per round: a wave's lanes in the logic
        kernels).  False at the deepest level."""
        lvl = getattr(self, "_esc_level", 0)
        if lvl >= self.ESCALATIONS:
            return False
        self._esc_level = lvl + 1
        self.set_schedule(self._schedule(self.m_sched[0], 4 * sum(self.m_sched), 32),
                          self._schedule(self.nt_sched[0], 2 * sum(self.nt_sched), 64))
        return True

    def _use_fast_possible(self):
        return not (self.fast is False or self.host_lp or self._fast_ok is False)

    def run(self, nsteps, store=True):
        return self._guard(lambda: EnsembleSampler.run(self, nsteps, store))

    def step(self):
        if not self._guarded:
            return self._guard(self._step)
        return self._step()

    def _step(self):
        st, ns, ndim = _lib.stream(), self.half, self.ndim
        halves = self._splits()
        seed = C.c_uint64((self.seed + 0x9E3779B97F4A7C15 * (self.rank + 1)) & 0xFFFFFFFFFFFFFFFF)
        if self._use_fast() and self._step_fast(halves, seed):
            return
        self.counters.zero_()
        P = _lib.ptr
        nt = self.ntrial
        for h in (0, 1):
            S, Cc = halves[h], halves[1 - h]
            comp, ldc, cidx, nc = self.coords, self.ld, Cc, self.half
            if self.exchange == "allgather" and self.world > 1:
                comp, cidx, nc = self._allgather_complement(Cc)
            _lib.call("linna_slice_init", self.ctx, P(self.logp), _lib.iptr(S), ns, P(comp), ldc, _lib.iptr(cidx), nc, ndim,
                      P(self.mu_dev), seed, _lib.iptr(self.step_dev), h, P(self.DIR), self.ld, P(self.Z0), P(self.L),
                      P(self.R), _lib.iptr(self.flags), int(self.maxsteps), st)

            def expand_round(r, slot, gate):            # stepping out, both ends per round
                self._eval_points(S, self.LR, 2, gate)
                self._neval_host += 2 * ns
                _lib.call("linna_slice_expand", self.ctx, P(self.Z0), P(self.Z2), C.c_void_p(self.Z2.data_ptr() + 4 * ns),
                          P(self.L), P(self.R), _lib.iptr(self.flags), ns, _lib.iptr(self.counters), slot, st)

            # shrinking, TWO trials per round: the second is placed as if the first were rejected (the bracket
            # after a rejection depends on where the trial fell, not on its density), so one launch evaluates
            # 2 x nw/2 points -- the whole GPU instead of half of it -- and the rounds halve; the accepted
            # point is the one the one-trial-per-round procedure accepts (same Philox sub-counters).
            def shrink_round(r, slot, gate):
                _lib.call("linna_slice_draw", self.ctx, P(self.L), P(self.R), _lib.iptr(S), P(self.W), _lib.iptr(self.flags),
                          ns, seed, _lib.iptr(self.step_dev), 2 + h, r * nt, nt, st)
                self._eval_points(S, self.W, nt, gate)
                self._neval_host += nt * ns
                _lib.call("linna_slice_shrink", self.ctx, P(self.Z0), P(self.Z2), P(self.L), P(self.R), P(self.W),
                          _lib.iptr(self.flags), P(self.Wacc), P(self.Zacc), ns, _lib.iptr(self.counters), slot, nt, st)

            # (a side steps out at most maxsteps - 1 times: its budget; shrinking is bounded by zeus' `maxiter` passes only)
            self._run_rounds(expand_round, min(int(self.maxsteps), int(self.maxiter)) + 1)
            self._run_rounds(shrink_round, (int(self.maxiter) + nt - 1) // nt, "contractions")
            _lib.call("linna_slice_commit", self.ctx, P(self.coords), self.ld, ndim, P(self.logp), _lib.iptr(S), ns,
                      P(self.DIR), self.ld, P(self.Wacc), P(self.Zacc), st)
            if self.probe is not None:
                self.probe(h, S, dict(Z0=self.Z0, L=self.L, R=self.R, Wacc=self.Wacc, Zacc=self.Zacc, counters=self.counters, fast=False))
        _lib.call("linna_step_increment", self.ctx, _lib.iptr(self.step_dev), st)
        self.iteration += 1
        if self.tune:                                   # zeus: mu *= 2 nexp / (nexp + ncon)
            c = self.counters.cpu().numpy()
            self._tune_mu(int(c[0]), int(c[1]))


# ------------------------------------------------------------------ batched per-walker HMC
class BatchedHMC(object):
    """``linna/HMCSampler.py:19-68`` for B independent chains: p ~ N(0, m); half kick; ``num_steps``
    x (drift, gradient, kick); final half kick; Metropolis test on H = p^2/2m - lnP."""

    def __init__(self, log_prob, x0, mass=None, seed=0, fused=True):
        self.fused = fused                      # kick + drift in the gradient launch's finish (False: the separate entries)
        if not getattr(log_prob, "device_only", True):
            raise NotImplementedError("HMC needs the gradient of the log-probability: a user loglikelihoodfunc / "
                                      "externalloglike is host code without one")
        self.lp = log_prob
        p = log_prob._ensure()
        self.dev, self.ctx = p["dev"], _lib.ctx(p["dev"].index)
        x0 = torch.as_tensor(np.asarray(x0, np.float32), device=self.dev)
        self.B, self.ndim = x0.shape
        self.ld = _lib.ld4(self.ndim)
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=self.dev)
        self.x, self.q, self.p = z(self.B, self.ld), z(self.B, self.ld), z(self.B, self.ld)
        self.x[:, :self.ndim].copy_(x0)
        self.mass = torch.ones(self.ndim, dtype=torch.float32, device=self.dev) if mass is None else \
            torch.as_tensor(np.asarray(mass, np.float32), device=self.dev)
        self.lnp, self.lnp_new, self.H0 = z(self.B), z(self.B), z(self.B)
        self.g, self.g_new = z(self.B, self.ld), z(self.B, self.ld)
        self.naccept = torch.zeros(self.B, dtype=torch.int32, device=self.dev)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.seed = int(seed)
        self.lp.evaluate_with_grad(self.x, out=self.lnp, grad=self.g)

    def step(self, num_steps, step_size, p0=None, u=None):
        """One HMC transition per chain.  ``p0[B, ndim]`` (standard-normal draws) and ``u[B]`` replace
        the Philox draws when given (replaying HMCSampler.py:26 / :59 streams)."""
        st, eps = _lib.stream(), float(step_size)
        seed = C.c_uint64(self.seed)
        args = (self.ctx, self.B, self.ndim, _lib.ptr(self.mass))
        p0d = None if p0 is None else torch.as_tensor(np.ascontiguousarray(p0, np.float32), device=self.dev)
        ud = None if u is None else torch.as_tensor(np.ascontiguousarray(u, np.float32), device=self.dev)
        if self.fused and num_steps >= 1:
            # 3 + num_steps launches: momentum draw + first half kick + first drift; per leapfrog step the gradient with the
            # next kick (a half one behind the last step, :51) and drift in its finish; Metropolis test; counter
            _lib.call("linna_hmc_start", *args, seed, _lib.iptr(self.step_dev), _lib.ptr(self.lnp),
                      _lib.ptr(p0d) if p0d is not None else None, self.ndim, _lib.ptr(self.g), self.ld, 0.5 * eps, eps,
                      _lib.ptr(self.x), self.ld, _lib.ptr(self.p), self.ld, _lib.ptr(self.q), self.ld, _lib.ptr(self.H0), st)
            h, ws = self.lp._ensure()["handle"], _lib.ptr(self.lp._workspace(self.B, True))
            for i in range(num_steps):
                last = i == num_steps - 1
                _lib.call("linna_logprob_grad_leapfrog", h, _lib.ptr(self.q), self.ld, self.B, ws, _lib.ptr(self.lnp_new),
                          _lib.ptr(self.g_new), self.ld, _lib.ptr(self.p), self.ld, _lib.ptr(self.mass),
                          0.5 * eps if last else eps, 0.0 if last else eps, st)
            self._accept(args, seed, ud, st)
            return
        _lib.call("linna_hmc_init", *args, seed, _lib.iptr(self.step_dev), _lib.ptr(self.lnp),
                  _lib.ptr(p0d) if p0d is not None else None, self.ndim, _lib.ptr(self.p), self.ld, _lib.ptr(self.H0), st)
        self.q.copy_(self.x)
        g = self.g
        for i in range(num_steps):
            ek = 0.5 * eps if i == 0 else eps                     # :35 half kick first, full kicks after
            _lib.call("linna_hmc_kick_drift", *args, ek, eps, _lib.ptr(g), self.ld, _lib.ptr(self.p), self.ld,
                      _lib.ptr(self.q), self.ld, st)
            self.lp.evaluate_with_grad(self.q, out=self.lnp_new, grad=self.g_new)
            g = self.g_new
        _lib.call("linna_hmc_kick_drift", *args, 0.5 * eps, 0.0, _lib.ptr(g), self.ld, _lib.ptr(self.p), self.ld,
                  _lib.ptr(self.q), self.ld, st)                                             # :51
        self._accept(args, seed, ud, st)

    def _accept(self, args, seed, ud, st):
        _lib.call("linna_hmc_accept", *args, seed, _lib.iptr(self.step_dev), _lib.ptr(self.H0), _lib.ptr(self.p), self.ld,
                  _lib.ptr(self.q), self.ld, _lib.ptr(self.lnp_new), _lib.ptr(self.g_new), self.ld,
                  _lib.ptr(ud) if ud is not None else None, _lib.ptr(self.x), self.ld, _lib.ptr(self.lnp), _lib.ptr(self.g),
                  _lib.iptr(self.naccept), st)
        _lib.call("linna_step_increment", self.ctx, _lib.iptr(self.step_dev), st)

    def sample(self, num_samps, num_steps, step_size):
        chain = torch.empty((num_samps, self.B, self.ndim), dtype=torch.float32, device=self.dev)
        lnps = torch.empty((num_samps, self.B), dtype=torch.float32, device=self.dev)
        for i in range(num_samps):
            self.step(num_steps, step_size)
            chain[i].copy_(self.x[:, :self.ndim])
            lnps[i].copy_(self.lnp)
        return chain, lnps


# ------------------------------------------------------------------ drivers with the reference's names
class _Ranks(object):
    """How a driver's ensemble is split over the ranks of a run (``linna_amd.dist.init()``; the reference farms the walkers'
    log-probability calls out over its MPI pool, util.py:99-256).  Three modes (``exchange``; ``LINNA_ENSEMBLE_EXCHANGE``):

    * ``"local"`` -- the default whenever every rank's share holds a valid ensemble (``nwalkers / world >= 2 ndim``, emcee's and
      zeus' own floor): ``nwalkers / world`` walkers per GPU as a SUB-ENSEMBLE of its own -- stretch / slice partners from the
      rank's LOCAL complementary half, the slice sampler's mu tuned per sub-ensemble -- no collective inside an iteration;
      chain blocks are gathered once per convergence check ("an RCCL gather for chain state"), rank 0 alone keeps the chain
      file ([iterations, nwalkers, ndim], walker blocks in rank order) and the statistics over ALL walkers and tells the
      others when to stop.  Sub-ensembles are valid ensemble samplers of the same posterior, so the merged chain is too.
    * ``"root"`` -- shares too small for that: the whole ensemble runs on rank 0, the other ranks wait at the end of the call.
    * ``"allgather"`` -- on request only: ONE ensemble sharded, partners from the complementary walkers of ALL ranks, i.e. one
      all-gather per half step.  A small collective's latency is of the order of the 30 us half step it follows: this mode is
      slower than one GPU at every ensemble size measured and exists for runs that need the one-ensemble
      chain (its mu is tuned from the all-reduced counts)."""

    def __init__(self, nwalkers, group, ndim=None, exchange=None):
        from . import dist as ldist
        self.group, self.real_world, self.rank = group, ldist.world_size(group), ldist.rank(group)
        want = exchange or os.environ.get("LINNA_ENSEMBLE_EXCHANGE") or "auto"
        if want not in ("auto", "local", "root", "allgather"):
            raise ValueError("exchange = %r (auto, local, root or allgather)" % (want,))
        even = nwalkers % (2 * self.real_world) == 0
        if self.real_world == 1:
            mode = "single"
        elif want == "allgather" or want == "local":
            if not even:
                raise ValueError("nwalkers = %d cannot be split into even halves over %d ranks" % (nwalkers, self.real_world))
            mode = want
        elif want == "root":
            mode = "root"
        else:
            mode = "local" if even and (ndim is None or nwalkers // self.real_world >= 2 * ndim) else "root"
        self.mode = mode
        self.active = mode != "root" or self.rank == 0               # (root: ranks > 0 do not sample)
        self.world = 1 if mode in ("single", "root") else self.real_world   # ranks that exchange data inside the sampling loop
        self.nw = nwalkers // self.world
        self.exchange = "allgather" if mode == "allgather" else "none"
        self.ens_group = group                                       # the sampler's group (chain gathers)

    def mine(self, x0):
        if self.world == 1:
            return np.asarray(x0)
        return np.asarray(x0)[self.rank * self.nw:(self.rank + 1) * self.nw]

    def bcast(self, obj):
        """A host object of rank 0 on every rank (control plane)."""
        if self.world == 1:
            return obj
        import torch.distributed as tdist
        box = [obj]
        tdist.broadcast_object_list(box, src=0 if self.group is None else tdist.get_global_rank(self.group, 0), group=self.group)
        return box[0]

    def gather(self, ens, c, l):
        """(chain block, log-probabilities, acceptance counts) of the whole ensemble on every rank."""
        if self.world == 1:
            return c, l, ens.naccept
        from . import dist as ldist
        c, l = ens.gather_chain(c, l)
        return c, l, ldist.gather_rows(ens.naccept.float(), self.group)

    def barrier(self):
        if self.real_world > 1:
            from . import dist as ldist
            ldist.barrier(self.group)


class _Prof(object):
    """Where a driver run spends its time (``sample(..., profile={})``): host seconds per phase and, from event pairs on
    the streams the work was enqueued on, device seconds per phase.  Without a dict every hook is a no-op."""

    def __init__(self, out):
        import collections
        self.out, self.h, self.ev = out, collections.defaultdict(float), collections.defaultdict(list)

    class _Span(object):
        __slots__ = ("p", "key", "stream", "t0", "e0")

        def __init__(self, p, key, stream):
            self.p, self.key, self.stream = p, key, stream

        def __enter__(self):
            import time
            if self.p.out is not None:
                if self.stream is not False:
                    self.e0 = torch.cuda.Event(enable_timing=True)
                    self.e0.record(self.stream if self.stream is not None else torch.cuda.current_stream())
                self.t0 = time.perf_counter()

        def __exit__(self, *a):
            import time
            if self.p.out is not None:
                self.p.h[self.key] += time.perf_counter() - self.t0
                if self.stream is not False:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record(self.stream if self.stream is not None else torch.cuda.current_stream())
                    self.p.ev[self.key].append((self.e0, e1))

    def host(self, key):
        return self._Span(self, key, False)

    def both(self, key, stream=None):
        """Host seconds inside the block and device seconds between two events recorded on ``stream`` around it."""
        return self._Span(self, key, stream)

    def finish(self, **extra):
        if self.out is None:
            return
        torch.cuda.synchronize()
        for k, v in self.h.items():
            self.out["host_%s_s" % k] = v
        for k, pairs in self.ev.items():
            self.out["gpu_%s_s" % k] = 1e-3 * sum(a.elapsed_time(b) for a, b in pairs)
        self.out.update(extra)


def _run_blocks(ens, rk, store, dchain, done, nsamp, ncheck, incremental, begin_check, decide, prof):
    """The sampling loop of both drivers (sampler.py:530-552, :728-735): blocks of ``ncheck`` iterations, each followed by
    the chain-file append and a convergence check.  Software-pipelined: the statistics of block i run on a stream of
    their own WHILE the sampler already advances block i + 1, and the host reads their verdict (33 numbers in pinned
    memory) only after it has enqueued that block -- neither the GPU nor the host waits for a check.  When the verdict is
    "stop" the block sampled meanwhile is dropped, so the chain ends exactly where the reference's criterion ends it."""
    dev = ens.dev
    stats = torch.cuda.Stream(device=dev) if rk.rank == 0 else None
    pending = None
    with _lib.quiet_gc():                 # (no full pass of the cyclic collector while the launch queue is a few ms deep)
        while done < nsamp:
            with prof.both("sampling"):
                c, l = ens.run(ncheck)
            c, l, acc = rk.gather(ens, c, l)
            stop = False
            if rk.rank == 0 and pending is not None:
                with prof.host("wait_check"), torch.cuda.stream(stats):
                    stop = decide(pending, done)
                pending = None
            if rk.bcast(stop):
                break
            done += ncheck
            if rk.rank == 0:
                with prof.both("theta"):
                    th = ens.theta_of(c)
                with prof.host("store_append"):
                    store.append(c, th, l, acc)                               # device tensors: copied off this thread
                    if incremental:
                        store.flush(final=False)
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(dev))
                stats.wait_event(ready)
                with torch.cuda.stream(stats), prof.both("stats", stats):
                    c.record_stream(stats)
                    dchain.append(c)
                    pending = begin_check(done)
    if rk.rank == 0 and pending is not None:                              # the last block's check: printed, not acted upon
        with prof.host("wait_check"), torch.cuda.stream(stats):
            decide(pending, done)
    if stats is not None:
        stats.synchronize()
    return done


class HMCSampler(object):
    """The reference's emcee driver (sampler.py:389-554): burn-in, restart from the best region,
    sample until the integrated autocorrelation time and the mean/std drift have converged.
    Multi-rank runs give every rank a sub-ensemble of ``nwalkers / world`` walkers and gather the chain once per convergence
    check (``_Ranks``); every rank of the run must make the call (``dist.enter`` raises within two minutes otherwise)."""

    def __init__(self, lnp, dlnp, ddlnp, ndim, nwalkers, x0=None, m=None, transform=None, torchspeed=False, seed=0,
                 dist_group=None, exchange=None):
        self.lnp, self.dlnp, self.ddlnp = lnp, dlnp, ddlnp
        self.exchange = exchange                             # how a multi-rank run splits the ensemble (_Ranks); None: automatic
        self.transform, self.x0, self.nparams, self.nwalkers = transform, x0, ndim, nwalkers
        self.m = np.ones(ndim) if m is None else m
        self.sampler = None
        self.seed, self.group = seed, dist_group

    def sample(self, pool, nsamp, samp_steps=0, samp_eps=0, Madapt=1000, outdir="./", progress=False, overwrite=False,
               ntimes=10, tautol=0.01, method="emcee", incremental=True, meanshift=0.1, stdshift=0.1, nk=2, ncheck=100,
               burnin=100, profile=None):
        if method != "emcee":
            # sampler.py's "hmc"/"nuts" branches are unreachable in the reference
            raise NotImplementedError(method)
        import time
        t_start = time.perf_counter()
        prof = _Prof(profile)
        from . import dist as ldist
        ldist.enter("sampler.HMCSampler.sample", self.group)
        rk = _Ranks(self.nwalkers, self.group, self.nparams, self.exchange)
        filename = os.path.join(outdir, "chemcee_256.h5")
        store = ChainStore(filename, self.transform)
        if not rk.active:                                    # ("root": the whole ensemble runs on rank 0)
            rk.barrier()
            return store
        x0 = self.x0
        resume = False
        if rk.rank == 0 and store.exists():
            if overwrite:
                store.remove()
            else:
                print("init from previous")
                prev = ChainStore.load(filename)
                x0, resume = prev["chain"][-1], True
                store.append(prev["chain"], prev["chain_transformed"], prev["log_prob"], prev["accepted"])   # re-chunked into the new file at the first flush
        x0, resume = rk.bcast((x0, resume))
        ens = EnsembleSampler(rk.nw, self.nparams, self.lnp, seed=self.seed, dist_group=self.group, exchange=rk.exchange)
        self.sampler = ens
        print("start", flush=True)
        if not resume:
            print("burnin...", flush=True)                                   # sampler.py:519-529
            with prof.host("burnin"), _lib.stage("run_mcmc.burnin"):
                ens.set_state(rk.mine(x0))
                c, l = ens.run(burnin)
                c, l, _ = rk.gather(ens, c, l)
                if rk.rank == 0:
                    # the 50 nwalkers best burn-in samples, nwalkers of them drawn with numpy's generator as the reference
                    # does (sampler.py:524-528); ranked on the device, only the drawn rows come to the host (at 4096
                    # walkers the host's argsort of 409600 log-probabilities took as long as 400 iterations)
                    order = torch.argsort(l.reshape(-1), descending=True, stable=True)[:int(50 * self.nwalkers)]
                    pick = torch.as_tensor(np.random.randint(0, len(order), self.nwalkers), device=order.device)
                    x0 = c.reshape(-1, self.nparams)[order[pick]].cpu().numpy()
                x0 = rk.bcast(x0)
            print("burnin done...", flush=True)
            ens.naccept.zero_()
        ens.set_state(rk.mine(x0))
        st = {"old_tau": np.inf}
        done = 0 if not resume else sum(len(c) for c in store.chain)
        done = rk.bcast(done)
        dchain = DeviceChain()                                               # convergence statistics stay on the GPU
        for blk in store.chain:
            dchain.append(np.asarray(blk, np.float32))

        def decide(tok, done_now):
            """True = stop (sampler.py:532-552, at every `ncheck` iterations as there: tau of the whole chain now against
            tau `ncheck` iterations earlier).  The estimate is incremental (DeviceChain), so a check costs the same however
            long the chain has grown -- rounds 1-4 thinned the checks out because each was a batch of FFTs over the chain."""
            n = tok["hi"]
            tau = dchain.tau_end(tok)                                         # sampler.py:538
            old_tau = st["old_tau"]
            st["old_tau"] = tau
            if np.isnan(np.sum(tau)) and n > 10:
                return True
            with np.errstate(invalid="ignore", divide="ignore"):
                converged = np.all(tau * ntimes < n)                          # :545-547
                converged &= np.all(np.abs(old_tau - tau) / tau < tautol)
                if converged and dchain.subset:
                    # the estimate above averaged over a subset of the walkers: the decision is taken on all of them
                    tau = dchain.integrated_time(upto=n, all_walkers=True)
                    old_tau = dchain.integrated_time(upto=n - ncheck, all_walkers=True) if n > ncheck else np.inf
                    converged = np.all(tau * ntimes < n) and np.all(np.abs(old_tau - tau) / tau < tautol)
                converged = converged and dchain.checkmeanstd(max(2, int(nk * np.mean(tau))), meanshift, stdshift)
                print("max, min tau diff, max tau, ninter: {0}, {1}, {2}, {3}\n".format(
                    np.max(np.abs(old_tau - tau) / tau), np.min(np.abs(old_tau - tau) / tau), np.max(tau), n), flush=True)
            return bool(converged)

        with _lib.stage("run_mcmc.blocks"):
            done = _run_blocks(ens, rk, store, dchain, done, nsamp, ncheck, incremental, lambda n: dchain.tau_begin(), decide, prof)
        if rk.rank == 0:
            with prof.host("final_flush"), _lib.stage("run_mcmc.final_flush"):
                store.flush()
        rk.barrier()                                                          # the file is complete before any rank reads it
        prof.finish(total_s=time.perf_counter() - t_start, iterations=done, writer_fetch_s=store.busy["fetch"],
                    writer_append_s=store.busy["append"], lag_capacity=0 if dchain._S is None else len(dchain._S),
                    lag_growths=dchain.lag_growths)
        self.sampler = None
        return store


class ZeusSampler(object):
    """sampler.py:699-737: zeus' ensemble slice sampler (``SliceEnsembleSampler``) with the reference's
    convergence callback (IAT on the last 80 %, sampler.py:684,729; mean/std drift) and file names.
    Multi-rank runs give every rank a sub-ensemble with a mu of its own and gather the chain once per convergence check
    (``_Ranks``); every rank of the run must make the call (``dist.enter``)."""

    def __init__(self, lnp, ndim, nwalkers, x0=None, transform=None, seed=0, dist_group=None, exchange=None):
        self.lnp, self.transform, self.x0, self.nparams, self.nwalkers = lnp, transform, x0, ndim, nwalkers
        self.sampler = None
        self.seed, self.group = seed, dist_group
        self.exchange = exchange                             # how a multi-rank run splits the ensemble (_Ranks); None: automatic

    def sample(self, pool, nsamp, outdir="./", progress=False, overwrite=False, ntimes=10, tautol=0.01, incremental=True,
               meanshift=0.1, stdshift=0.1, nk=2, ncheck=100, profile=None):
        import time
        t_start = time.perf_counter()
        prof = _Prof(profile)
        from . import dist as ldist
        ldist.enter("sampler.ZeusSampler.sample", self.group)
        rk = _Ranks(self.nwalkers, self.group, self.nparams, self.exchange)
        store = ChainStore(os.path.join(outdir, "zeus_256.h5"), self.transform)
        if not rk.active:                                    # ("root": the whole ensemble runs on rank 0)
            rk.barrier()
            return store
        x0 = self.x0
        if rk.rank == 0:
            if store.exists() and overwrite:
                store.remove()
            if store.exists():
                print("init from previous")
                prev = ChainStore.load(store.base)
                x0 = prev["chain"][-1]
                store.append(prev["chain"], prev["chain_transformed"], prev["log_prob"], prev["accepted"])   # re-chunked into the new file at the first flush
        x0 = rk.bcast(x0)
        ens = SliceEnsembleSampler(rk.nw, self.nparams, self.lnp, seed=self.seed, dist_group=self.group, exchange=rk.exchange)
        self.sampler = ens
        ens.set_state(rk.mine(x0))
        st = {"old_tau": np.inf}
        done = rk.bcast(sum(len(c) for c in store.chain))
        dchain = DeviceChain()
        for blk in store.chain:
            dchain.append(np.asarray(blk, np.float32))

        def decide(tok, done_now):                          # sampler.py:667-696 (zeus' callback: every `ncheck` iterations)
            n = tok["hi"]
            tau = float(np.mean(dchain.tau_end(tok)))       # discard=0.2, sampler.py:684,729
            old_tau = st["old_tau"]
            st["old_tau"] = tau
            with np.errstate(invalid="ignore", divide="ignore"):
                converged = tau * ntimes < n
                converged &= abs(old_tau - tau) / tau < tautol
                if converged and dchain.subset:               # decide on all walkers (the routine checks use a subset)
                    tau = float(np.mean(dchain.integrated_time(discard=int(n * 0.2), upto=n, all_walkers=True)))
                    prev = n - ncheck
                    old_tau = float(np.mean(dchain.integrated_time(discard=int(prev * 0.2), upto=prev, all_walkers=True))) if prev > 0 else np.inf
                    converged = tau * ntimes < n and abs(old_tau - tau) / tau < tautol
                converged = converged and bool(dchain.checkmeanstd(max(2, int(nk * tau)), meanshift, stdshift))
            return bool(converged)

        with _lib.stage("run_mcmc.blocks"):
            done = _run_blocks(ens, rk, store, dchain, done, min(nsamp, 100000), ncheck, incremental,
                               lambda n: dchain.tau_begin(discard=int(n * 0.2)), decide, prof)
        if rk.rank == 0:
            with prof.host("final_flush"), _lib.stage("run_mcmc.final_flush"):
                store.flush()
        rk.barrier()
        prof.finish(total_s=time.perf_counter() - t_start, iterations=done, writer_fetch_s=store.busy["fetch"],
                    writer_append_s=store.busy["append"], lag_capacity=0 if dchain._S is None else len(dchain._S),
                    lag_growths=dchain.lag_growths)
        return store
