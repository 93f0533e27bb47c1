"""Randomised network shapes through the whole-network kernel (tools/fuzz_net_stream.py): plain MLPs and the
reference's residual architectures, every engine, diagonal and dense covariance, evaluation, gradient, training
forward and the one-launch dX chain, against the numpy oracle and the GEMM-chain paths."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_random_shapes():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_net_stream
    assert fuzz_net_stream.run(48, 9000) == 0
