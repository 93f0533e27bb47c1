"""Randomised shapes (tools/fuzz_net_stream.py, tools/fuzz_moves_loss.py): plain MLPs and the reference's residual
architectures through the whole-network kernel -- every engine, diagonal and dense covariance, evaluation,
gradient, training forward and one-launch dX chain against the numpy oracle and the GEMM-chain paths; the
one-launch stretch half step bit for bit against its three-launch form; the one-launch loss against the
five-launch path; random shapes, layouts and epilogues through linna_gemm_f32 against numpy; random training steps against the oracle and against the run with every one-launch path off."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_random_network_shapes():
    import fuzz_net_stream
    assert fuzz_net_stream.run(48, 9000) == 0


def test_random_moves_and_losses():
    import fuzz_moves_loss
    assert fuzz_moves_loss.moves(16, 9100) == 0
    assert fuzz_moves_loss.slices(10, 9150) == 0
    assert fuzz_moves_loss.loss(24, 9200) == 0


def test_random_gemms():
    import fuzz_gemm
    assert fuzz_gemm.run(80, 9300) == 0


def test_random_training_steps():
    import fuzz_train
    assert fuzz_train.run(24, 9400) == 0
