"""linna_amd -- MI355X-native (gfx950) implementation of the LINNA emulator hot path.

Same Python surface as the reference package (``linna``): ``main.ml_sampler``,
``nn.ChtoModelv2``, ``predictor_gpu.Predictor``, ``util.Log_prob`` ... with the arithmetic in
hand-written HIP kernels behind a C ABI (``include/linna_hip.h``).
"""
__version__ = "0.1.0"
