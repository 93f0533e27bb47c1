"""Leapfrog HMC with the emulator gradient (mirrors linna/HMCSampler.py:6-68).

``HMCSampler(lnP, x0, m, transform=None)``: ``lnP`` is a ``linna_amd.util.Log_prob`` (its
gradient comes from the HIP reverse pass instead of torch autograd); ``x0`` is one chain
``[ndim]`` as in the reference, or ``[B, ndim]`` for B independent chains advanced together.
``sample(num_samps, num_steps, step_size)`` returns the reference's list of dicts for a single
chain.  ``momenta`` / ``uniforms`` may be supplied to replay a given random stream.
"""
import numpy as np
import torch

from .sampler import BatchedHMC


class HMCSampler(object):
    def __init__(self, lnP, x0, m, transform=None, device="cuda", seed=0):
        x0 = np.asarray(x0.detach().cpu() if torch.is_tensor(x0) else x0, np.float32)
        self.single = x0.ndim == 1
        m = np.asarray(m.detach().cpu() if torch.is_tensor(m) else m, np.float32).reshape(-1)
        self.hmc = BatchedHMC(lnP, x0.reshape(1, -1) if self.single else x0, mass=m, seed=seed)
        self.transform = transform if transform is not None else (lambda x: x)

    def sample(self, num_samps, num_steps, step_size, momenta=None, uniforms=None):
        chain = []
        h = self.hmc
        for i in range(num_samps):
            before = h.naccept.clone()
            p0 = None if momenta is None else np.asarray(momenta[i], np.float32).reshape(h.B, h.ndim)
            u = None if uniforms is None else np.asarray(uniforms[i], np.float32).reshape(h.B)
            h.step(num_steps, step_size, p0=p0, u=u)
            acc = (h.naccept - before).cpu().numpy().astype(bool)
            x = h.x[:, :h.ndim].cpu()
            lnp = h.lnp.cpu().numpy()
            if self.single:
                xt = self.transform(x[0])
                chain.append({"x": np.asarray(xt.detach().cpu() if torch.is_tensor(xt) else xt), "lnP": lnp[0],
                              "accepted": bool(acc[0])})
            else:
                chain.append({"x": x.numpy(), "lnP": lnp, "accepted": acc})
        return chain
