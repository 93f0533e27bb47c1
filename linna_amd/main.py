"""LINNA top-level entry points on MI355X: ``ml_sampler`` / ``ml_sampler_core`` with the
reference's signatures (linna/main.py:22, 77) and the ``theory()`` / ``priors`` callback surface.

Per iteration: design training points -> evaluate the user's theory -> train the emulator
(HIP kernels, in process; the reference's ``train_gpu.py`` subprocess + ``finish.pkl``
rendezvous is kept as on-disk artefacts) -> load it back -> run the ensemble sampler on the
GPU with the fused Log_prob pipeline -> feed the chain to the next iteration.
"""
import gc
import os
import pickle

import numpy as np
import torch

from .nn import *  # noqa: F401,F403
from .util import (limit_threads_to_quota, Transform, invTransform, NN_samplerv1, generate_training_point, train_NN, retrieve_model, Log_prob,
                   gaussianlogliklihood, run_mcmc, read_chain_and_cut, LogPrior, logp_theory_data)
from . import nn as lnn
from ._lib import stage


def ml_sampler(outdir, theory, priors, data, cov, init, pool, nwalkers, gpunode, omegab2cut=None, nepoch=4500,
               method="zeus", nbest=None, chisqcut=None, loglikelihoodfunc=None):
    """main.py:22-75: the hyper-parameter schedule of To et al. 2022."""
    ntrainArr = [10000, 10000, 10000, 10000]
    nvalArr = [500, 500, 500, 500]
    if method == "emcee":
        nkeepArr, ntimesArr = [2, 2, 5, 4], [5, 5, 10, 15]
    elif method == "zeus":
        nkeepArr, ntimesArr = [2, 2, 5, 5], [5, 5, 10, 50]
    else:
        raise NotImplementedError(method)
    ntautolArr = [0.03, 0.03, 0.02, 0.01]
    temperatureArr = [4.0, 2.0, 1.0, 1.0]
    meanshiftArr = [0.2, 0.2, 0.2, 0.2]
    stdshiftArr = [0.15, 0.15, 0.15, 0.15]
    params = {"trainingoption": 1, "num_epochs": nepoch, "batch_size": 500}
    return ml_sampler_core(ntrainArr, nvalArr, nkeepArr, ntimesArr, ntautolArr, meanshiftArr, stdshiftArr, outdir, theory,
                           priors, data, cov, init, pool, nwalkers, "cuda", None, False, temperatureArr, omegab2cut, False, 1,
                           gpunode, lnn.ChtoModelv2, params, method, nbest=nbest, chisqcut=chisqcut,
                           loglikelihoodfunc=loglikelihoodfunc)


def ml_sampler_core(ntrainArr, nvalArr, nkeepArr, ntimesArr, ntautolArr, meanshiftArr, stdshiftArr, outdir, theory, priors,
                    data, cov, init, pool, nwalkers, device, dolog10index, ypositive, temperatureArr, omegab2cut=None,
                    docuda=False, tsize=1, gpunode=None, nnmodel_in=None, params=None, method="emcee", nbest=None,
                    chisqcut=None, loglikelihoodfunc=None, nsigma=3, externalloglike=None):
    """main.py:77-335.  Returns ``(chain[nsamp, ndim] in theta space, log_prob)``.
    ``gpunode`` / ``docuda`` / ``device`` (main.py:193-245: which Slurm node runs ``train_gpu.py`` under srun) keep their
    places in the signature; the emulator always trains in this process on the local GPU."""
    if method == "emcee":
        filename = "chemcee_256.h5"
    elif method == "zeus":
        filename = "zeus_256.h5"
    else:
        raise NotImplementedError(method)
    if nnmodel_in is None:
        nnmodel_in = lnn.ChtoModelv2
    limit_threads_to_quota()
    # one process per GPU under torchrun (WORLD_SIZE > 1): rendezvous + the library's RCCL communicator; the emulator then
    # trains data-parallel (predictor_gpu.py:246,265-266: `tsize` ranks, lr * size) and the walkers of the ONE ensemble
    # shard over the ranks (the reference's MPI pool, util.py:99-256).  Rank 0 designs the training points, calls the
    # user's theory and owns every file; the others wait at the barriers below.
    from . import dist as ldist
    world = ldist.init()
    rank = ldist.rank() if world > 1 else 0
    if world > 1:
        print("rank %d of %d: %s" % (rank, world, ldist.collectives()), flush=True)
    params = dict(params or {})
    ndim = len(init)
    data, cov = np.asarray(data, np.float64), np.asarray(cov, np.float64)
    sigma = np.sqrt(np.diag(cov))
    inv_cov = np.linalg.inv(cov)
    prior_range = []
    for item in priors:
        if item["dist"] == "flat":
            prior_range.append([item["arg1"], item["arg2"]])
        elif item["dist"] == "gauss":
            prior_range.append([item["arg1"] - 5 * item["arg2"], item["arg1"] + 5 * item["arg2"]])
        else:
            raise ValueError("not implement dist : {0}".format(item["dist"]))
    transform = Transform(priors)
    init = invTransform(priors)(np.asarray(init, np.float64))
    master = (pool is None or pool.is_master()) and rank == 0
    store = None
    nk = ntimes = None
    for i, (nt, nv, nk, ntimes, tautol, temperature, meanshift, stdshift) in enumerate(
            zip(ntrainArr, nvalArr, nkeepArr, ntimesArr, ntautolArr, temperatureArr, meanshiftArr, stdshiftArr)):
        temperature = temperature ** 2                                           # main.py:153
        print("#" * 100)
        print("iteration: {0}".format(i), flush=True)
        print("#" * 100)
        outdir_in = os.path.join(outdir, "iter_{0}/".format(i))
        chain = None
        if i > 0:
            prev = os.path.join(outdir, "iter_{0}/".format(i - 1), filename[:-3])
            with stage("read_chain_and_cut"):
                chain, _, _ = read_chain_and_cut(prev, nk, ntimes, method=method)
        nnsampler = NN_samplerv1(outdir_in, prior_range)
        nbest_in = nbest                                                         # main.py:140-145: only a LIST entry <= 0 means "none"
        if isinstance(nbest, list):
            nbest_in = nbest[i] if nbest[i] > 0 else None
        negloglike = None
        if nbest_in is not None:
            import tempfile
            tempdir = tempfile.TemporaryDirectory()

            def negloglike(x, tempdir=tempdir):
                d = data - theory([-1, x], tempdir)
                return d.dot(inv_cov.dot(d))
        if rank == 0:
            with stage("training_points"):
                generate_training_point(theory, nnsampler, pool, outdir_in, nt, nv, data, inv_cov, chain, nsigma=nsigma,
                                        omegab2cut=omegab2cut, options=params.get("trainingoption", 0), negloglike=negloglike,
                                        nbest_in=nbest_in, chisqcut=chisqcut)
        chain = None
        gc.collect()
        ldist.barrier()                                                          # the training points are on disk
        outdir_list = [os.path.join(outdir, "iter_{0}/".format(m)) for m in range(i + 1)]
        args = [None, cov, inv_cov, sigma, outdir_in, outdir_list, data, dolog10index, ypositive, False, 2, temperature,
                True, None, world, None, params, nbest_in is not None]
        if master:
            with open(os.path.join(outdir_in, "model_args.pkl"), "wb") as f:     # main.py:192-198 (artefact parity)
                pickle.dump(args, f)
        if (master or world > 1) and not ldist.agree(os.path.isfile(os.path.join(outdir_in, "finish.pkl"))):
            args[15] = nnmodel_in
            with stage("train_NN"):
                train_NN(*args, device=device if str(device).startswith("cuda") else "cuda", rank=rank)   # every rank: data parallel
            if master:
                with open(os.path.join(outdir_in, "finish.pkl"), "wb") as f:     # train_gpu.py:36-38
                    pickle.dump([True], f)
        ldist.barrier()                                                          # checkpoints and transform pickles are on disk
        with stage("retrieve_model"):
            model, y_invtransform_data = retrieve_model(outdir_in, len(init), len(data), nnmodel_in)
        if ldist.agree(any(os.path.isfile(os.path.join(outdir_in, filename[:-3] + ext)) for ext in (".h5", ".npz"))):   # main.py:273-274
            continue
        log_prob = Log_prob(data.astype(np.float32), inv_cov.astype(np.float32), model, y_invtransform_data, transform,
                            temperature, nograd=True, loglikelihoodfunc=loglikelihoodfunc or gaussianlogliklihood,
                            externalloglike=externalloglike)
        if pool is not None:
            pool.noduplicate = True                                              # main.py:282-283
        with stage("run_mcmc"):
            store = run_mcmc(nnsampler, outdir_in, method, ndim, nwalkers, init, log_prob, pool=pool, transform=transform,
                             ntimes=ntimes, tautol=tautol, meanshift=meanshift, stdshift=stdshift, nk=nk)
        if pool is not None:
            pool.noduplicate_close()                                             # main.py:285-286
    last = os.path.join(outdir, "iter_{0}/".format(len(ntrainArr) - 1), filename[:-3])
    with stage("read_chain_and_cut"):
        chain, _, d = read_chain_and_cut(last, nk, ntimes, method=method)
    log_prob_samples_x = d["log_prob"].reshape(-1)                               # main.py:291
    if "nimp" in params and rank == 0:                                           # main.py:297-334 (rank 0 owns the files)
        f_samples, f_lp = os.path.join(outdir, "samples_im.npy"), os.path.join(outdir, "log_prob_samples_x.npy")
        if not os.path.isfile(f_samples):
            chain, lp_flat, _ = read_chain_and_cut(last, nk, ntimes, method=method, flat=True)
            print(chain.shape, lp_flat.shape)
            select = np.random.randint(0, len(chain), params["nimp"])
            chain, log_prob_samples_x = chain[select], lp_flat[select]
            np.save(f_samples, chain)
            np.save(f_lp, log_prob_samples_x)
        else:
            chain, log_prob_samples_x = np.load(f_samples), np.load(f_lp)
        outimp = os.path.join(outdir, "imp/")
        nns = NN_samplerv1(outimp, prior_range)
        os.makedirs(outimp, exist_ok=True)
        f_theory = os.path.join(outdir, "theory.npy")
        if not os.path.isfile(f_theory):
            th = nns.generate_training_data(zip(range(len(chain)), chain), theory, pool=pool, args=[outimp])
            np.save(f_theory, th)
        else:
            th = np.load(f_theory)
        log_prob_samples_x = np.asarray(log_prob_samples_x).flatten()
        logp = np.array(logp_theory_data(chain, th, data, inv_cov, LogPrior(priors)))   # chi^2 of all rows: one GPU pass
        w = np.exp(logp - log_prob_samples_x)
        with np.errstate(divide="ignore", invalid="ignore"):
            lw = np.log(w)
            w[np.abs(lw - np.mean(lw)) > 2 * np.std(lw)] = 0
        w = w / np.sum(w)
        np.save(os.path.join(outdir, "weight_im.npy"), [log_prob_samples_x.flatten(), logp, w])
    if "nimp" in params and world > 1:
        # (control plane: the other ranks wait here while rank 0 runs the nimp theory evaluations -- dist.broadcast_object
        # goes over the gloo side group with the long timeout, not over the data path's NCCL group)
        chain, log_prob_samples_x = ldist.broadcast_object((chain, log_prob_samples_x) if rank == 0 else None)
    return chain, log_prob_samples_x
