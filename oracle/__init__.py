"""CPU oracle for the LINNA emulator hot path -- TEST INFRASTRUCTURE ONLY.

A plain-numpy restatement of the reference's algorithm (chto/linna, pure Python/torch) for
the path named in BASELINE.json: emulator forward/backward, transforms, Gaussian
log-likelihood, training loss, AdamW, HMC leapfrog, the ensemble stretch move and the ensemble slice move.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this package, and only as the checker.  The product (``linna_amd``) never
imports it: the product path fails loudly when the HIP library is missing.

Parity status
-------------
* emulator / transforms / likelihood / loss / gradients / AdamW / HMCSampler.py: PINNED
  against the live reference imported in the build container
  (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``) and against the reference's
  own committed fixture ``tests/test_data/2dgaussian_Fulltconn/iter_0`` (copied as data
  under ``tests/golden/2dgaussian_Fulltconn``).
* ensemble stretch move (``oracle.sampling.stretch_*``): PARITY UNPINNED.  The arithmetic
  lives in emcee (requirements.txt:14 pins emcee==3.0.2), which is neither vendored in the
  reference tree nor installed here; it is restated from emcee's published algorithm
  (Goodman & Weare 2010; emcee ``RedBlueMove``/``StretchMove``) and anchored on the
  reference's call sites sampler.py:493-495, 519-530.  Its GEOMETRY (red/blue halves, the line
  through the complementary walker, the stretch range, the (ndim - 1) ln z factor) is pinned by the
  reference-held emcee 3.0.2 chain ``chemcee_256.h5`` (tests/test_stretch_fixture.py); emcee's random
  stream is not.
* ensemble slice move (``oracle.sampling.slice_half_step`` / ``slice_iteration`` / ``slice_tune_mu``; the reference's DEFAULT
  sampler, main.py:22): PARITY UNPINNED.  The arithmetic lives in zeus-mcmc (setup.py:13, unpinned; neither vendored nor
  installed), and the reference holds no zeus chain or test vector: restated from the published algorithm (Karamanis & Beutler
  2021, Algorithms 2-4; Neal 2003, Fig. 3 and 5; zeus ``ensemble.py`` / ``moves.DifferentialMove``) and anchored on the reference's
  call site sampler.py:728-735 (``maxiter=1E5``, every other argument zeus' default).  Checked on the CPU against a per-walker
  re-derivation and the move's invariants (tests/test_oracle_slice.py); the HIP kernels are replayed against it half step by
  half step (tests/test_gpu_slice_replay.py).
* ``oracle.training.lr_range_test``: PARITY UNPINNED (torch_lr_finder, third party, absent), restated
  from its published algorithm; anchored on predictor_gpu.py:222-238.

Every function cites the reference file:line it follows (paths relative to the
reference root).
"""
from . import emulator, likelihood, training, sampling  # noqa: F401
