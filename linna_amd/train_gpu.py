"""Process-boundary shim with the reference's command line (linna/train_gpu.py:24-38):

    python -m linna_amd.train_gpu <outdir> cuda

reads ``model_args.pkl`` (the positional arguments of ``train_NN``, main.py:197), trains on
the GPU and writes ``finish.pkl``.  ``ml_sampler_core`` calls ``train_NN`` in process; this
entry exists for job scripts that launch training separately (jobscript/example_sampler.job).
"""
import pickle
import sys

if __name__ == "__main__":
    from linna_amd import util, nn
    outdir = sys.argv[1]
    with open(outdir + "/model_args.pkl", "rb") as f:
        args = util.ArgsUnpickler(f).load()
    if args[15] is None:
        args[15] = nn.ChtoModelv2
    util.train_NN(*args)
    with open(outdir + "/finish.pkl", "wb") as f:
        pickle.dump([True], f)
