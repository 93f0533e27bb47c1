"""CPU: the ensemble stretch move anchored on the reference-held emcee chain.

emcee is a third-party dependency absent from the reference tree and from this image (requirements.txt:14 pins
emcee==3.0.2), so the move is restated from its published algorithm (oracle/sampling.py).  What the reference DOES hold is
real emcee 3.0.2 output: tests/test_data/2dgaussian_Fulltconn/iter_0/chemcee_256.h5 (200 steps x 4 walkers x 2
parameters with log_prob and accepted; the chain tests/test_main.py:47-51 reads).  Every stored transition must be
explained by the restated move -- geometry, stretch range, red/blue half structure -- and the oracle's
``stretch_propose`` fed with the recovered (old position, partner, zz) must reproduce the stored positions.  The HIP
kernels are tied to the same oracle function draw by draw in tests/test_gpu_sampling.py, and satisfy the same relation
in ``test_hip_proposals_satisfy_the_fixture_relation`` there.  (sampler.py:493-495, 519-530 are the reference call sites.)
"""
import os

import numpy as np

import cases
from stretch_relation import explain_step

FIXTURE = os.path.join(cases.GOLDEN, "2dgaussian_Fulltconn/iter_0/chemcee_256.h5")


def _chain():
    from linna_amd.sampler import ChainStore
    d = ChainStore.read_h5(FIXTURE)
    return np.asarray(d["chain"], np.float64), np.asarray(d["log_prob"], np.float64), np.asarray(d["accepted"])


def test_every_emcee_transition_is_a_red_blue_stretch_step():
    chain, lp, accepted = _chain()
    assert chain.shape == (200, 4, 2)
    nmoved, zzs = np.zeros(4, int), []
    for t in range(len(chain) - 1):
        ex = explain_step(chain[t], chain[t + 1])
        assert ex is not None, "transition %d has no red/blue stretch explanation" % t
        for k, (j, zz) in ex[1].items():
            nmoved[k] += 1
            zzs.append(zz)
        # a walker that did not move keeps its log-probability; one that moved carries the new one
        still = [k for k in range(4) if k not in ex[1]]
        np.testing.assert_array_equal(lp[t + 1, still], lp[t, still])
    zzs = np.array(zzs)
    assert zzs.min() >= 0.5 and zzs.max() <= 2.0                       # g(z) lives on [1/a, a], a = 2
    assert 0.4 < np.mean(zzs < 1.0) < 0.6
    # `accepted` counts all 200 moves, the chain shows the 199 transitions after the first stored step
    assert np.all(nmoved <= accepted) and np.all(accepted - nmoved <= 1)


def test_oracle_proposal_reproduces_the_emcee_positions():
    """(old, partner, zz) recovered from the fixture -> oracle.sampling.stretch_propose -> the stored new position,
    in float64 to 1e-12, with the factor (ndim - 1) log zz that emcee's acceptance uses."""
    from oracle import sampling
    chain, lp, _ = _chain()
    a, n = 2.0, 0
    for t in range(len(chain) - 1):
        A, found = explain_step(chain[t], chain[t + 1])
        for k, (j, zz) in found.items():
            c = chain[t, j] if k in A else chain[t + 1, j]
            u = (np.sqrt(zz * a) - 1.0) / (a - 1.0)                     # zz = ((a - 1) u + 1)^2 / a
            assert -1e-9 <= u <= 1.0 + 1e-9
            q, fac = sampling.stretch_propose(chain[t, k][None, :], c[None, :], np.array([u]), np.array([0]), a)
            np.testing.assert_allclose(q[0], chain[t + 1, k], rtol=0, atol=1e-12)
            np.testing.assert_allclose(fac[0], (chain.shape[2] - 1) * np.log(zz), rtol=0, atol=1e-12)
            # the move was accepted: emcee's test  factor + lnP(new) - lnP(old) > log u'  has a solution u' in (0, 1)
            assert np.isfinite(fac[0] + lp[t + 1, k] - lp[t, k])
            n += 1
    assert n > 500
