"""Oracle: emulator network forward / backward (numpy).  TEST INFRASTRUCTURE ONLY.

Restates linna/nn.py.  Parameters are held in a dict keyed exactly like the reference's
``state_dict`` (nn.py:77-86), weights row-major ``[out, in]``.
"""
import numpy as np

KINDS = ("ChtoModelv2", "ChtoModelsimple", "ChtoModelv2_linear", "MLP")


# --------------------------------------------------------------------------- topology
def hidden_size_for(out_size):
    """nn.py:74-76: hidden = max(32, 32*out), forced to 1000 when out > 30."""
    h = max(32, int(out_size * 32))
    if out_size > 30:
        h = 1000
    return h


def topology(kind, in_size, out_size, width=512, depth=4):
    """Return the ordered list of (name, op, shapes) the network is made of.

    op is one of
      ("linear", key, K, N, relu:bool)
      ("resblock", key, K, C, N)          nn.py:11-56
      ("inskip", key, K, N, scale)        nn.py:160-163,195 (ChtoModelv2_linear only)
    """
    ops = []
    if kind == "MLP":
        # the synthetic plain MLP of BASELINE configs 2/5
        # (in -> width x depth -> out), ReLU between layers, identity on the last.
        k = in_size
        for i in range(depth):
            ops.append(("linear", "layer%d" % (i + 1), k, width, True))
            k = width
        ops.append(("linear", "layer%d" % (depth + 1), k, out_size, False))
        return ops
    if kind not in KINDS:
        raise ValueError(kind)
    channel = 4 if kind == "ChtoModelsimple" else 16        # nn.py:73, 314
    h = hidden_size_for(out_size)
    ops.append(("linear", "layer1", in_size, h, True))      # nn.py:77,121
    ops.append(("resblock", "layer2", h, channel, h // 2))  # nn.py:78
    h //= 2
    ops.append(("resblock", "layer3", h, channel * 2, h // 2))  # nn.py:80
    h //= 2
    ops.append(("resblock", "layer4", h, channel * 4, h // 2))  # nn.py:82
    h //= 2
    h6 = h if kind == "ChtoModelsimple" else h * 4          # nn.py:84 vs 325
    ops.append(("linear", "layer6", h, h6, True))
    ops.append(("linear", "layer7", h6, out_size, True))    # nn.py:85,126
    ops.append(("linear", "layer8", out_size, out_size, False))  # nn.py:86,130
    if kind == "ChtoModelv2_linear":
        ops.append(("inskip", "linearlayer", in_size, out_size, 1e-3))  # nn.py:160,195
    return ops


def param_shapes(kind, in_size, out_size, **kw):
    """Ordered {state_dict key: shape}, same order as torch's ``state_dict()``."""
    shapes = {}
    for op in topology(kind, in_size, out_size, **kw):
        if op[0] == "linear" or op[0] == "inskip":
            _, key, K, N = op[:4]
            shapes[key + ".weight"] = (N, K)
            shapes[key + ".bias"] = (N,)
        else:
            _, key, K, C, N = op
            shapes[key + ".layer1.weight"] = (C, K)
            shapes[key + ".layer1.bias"] = (C,)
            shapes[key + ".layer2.weight"] = (N, C)
            shapes[key + ".layer2.bias"] = (N,)
            if K != N:                                       # nn.py:28-31
                shapes[key + ".skip_layer.weight"] = (N, K)
    return shapes


def macs_per_eval(kind, in_size, out_size, **kw):
    return sum(int(np.prod(s)) for k, s in param_shapes(kind, in_size, out_size, **kw).items()
               if k.endswith("weight"))


class TorchCPUGenerator(object):
    """torch's default CPU generator for float32 ``uniform_`` restated on numpy: ``torch.manual_seed(s)`` seeds an
    mt19937 by ``init_genrand(s)`` exactly as ``np.random.RandomState(s)`` does, and ``uniform_(lo, hi)`` maps one
    32-bit draw per element as ``(x & (2**24 - 1)) * 2**-24 * (hi - lo) + lo`` in double (lo, hi rounded to
    float32 first), then rounds to float32.  Checked bit for bit against the reference's initial weights
    (tests/golden/init_parity.npz)."""

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)

    def uniform(self, shape, lo, hi):
        n = int(np.prod(shape))
        x = self.rs.randint(0, 2 ** 32, size=n, dtype=np.uint32)
        u = (x & np.uint32((1 << 24) - 1)).astype(np.float64) * 2.0 ** -24
        lo, hi = float(np.float32(lo)), float(np.float32(hi))
        return (u * (hi - lo) + lo).astype(np.float32).reshape(shape)

    def consume(self, n):
        if n > 0:
            self.rs.randint(0, 2 ** 32, size=int(n), dtype=np.uint32)


def _xavier(gen, shape):
    """nn.init.xavier_uniform_ (gain 1): U(-a, a), a = sqrt(3) * sqrt(2 / (fan_in + fan_out))."""
    a = np.sqrt(3.0) * np.sqrt(2.0 / float(shape[0] + shape[1]))
    return gen.uniform(shape, -a, a)


def reinit_params(kind, in_size, out_size, gen, **kw):
    """``model.init_weight()`` (nn.py:91-108 / :169-183 / :332-349): ``self.modules()`` walks the module tree in
    pre-order, so every residual block first runs its own ``init_weight`` (nn.py:34-43: Xavier on layer1, layer2,
    skip_layer, then skip zeroed) and is then visited child by child by the outer loop, which draws all three
    weights AGAIN -- the skip weights of a trained-from-scratch network are Xavier-uniform, not zero.  Biases 1e-2.
    ``gen`` is a ``TorchCPUGenerator``: same stream and draw order as torch, so the result is the reference's bits."""
    p = {}
    shapes = param_shapes(kind, in_size, out_size, **kw)
    for op in topology(kind, in_size, out_size, **kw):
        key = op[1]
        if op[0] == "resblock":
            names = [k for k in (key + ".layer1.weight", key + ".layer2.weight", key + ".skip_layer.weight") if k in shapes]
            for k in names:
                _xavier(gen, shapes[k])                      # the block's own init_weight, overwritten
            for k in names:
                p[k] = _xavier(gen, shapes[k])
            p[key + ".layer1.bias"] = np.full(shapes[key + ".layer1.bias"], 1e-2, np.float32)
            p[key + ".layer2.bias"] = np.full(shapes[key + ".layer2.bias"], 1e-2, np.float32)
        else:
            p[key + ".weight"] = _xavier(gen, shapes[key + ".weight"])
            p[key + ".bias"] = np.full(shapes[key + ".bias"], 1e-2, np.float32)
    return {k: p[k] for k in shapes}


def init_params(kind, in_size, out_size, gen, **kw):
    """The weights of a freshly CONSTRUCTED reference network (``torch.manual_seed(s); ChtoModelv2(...)``,
    nn.py:63-89): every ``nn.Linear`` constructor draws kaiming-uniform weights and a uniform bias (one draw per
    element, overwritten later), every residual block runs its ``init_weight`` once in its constructor (nn.py:33),
    then the network's ``init_weight()`` (``reinit_params``); ``ChtoModelv2_linear`` finally sets its input skip to
    weight 1e-5 / bias 0 (nn.py:162-163)."""
    for op in topology(kind, in_size, out_size, **kw):
        if op[0] == "resblock":
            _, key, K, C, N = op
            gen.consume(C * K + C)
            gen.consume(N * C + N)
            if K != N:
                gen.consume(N * K)
            gen.consume(C * K)
            gen.consume(N * C)
            if K != N:
                gen.consume(N * K)
        else:
            _, key, K, N = op[:4]
            gen.consume(N * K + N)
    p = reinit_params(kind, in_size, out_size, gen, **kw)
    if kind == "ChtoModelv2_linear":                         # nn.py:162-163
        p["linearlayer.bias"][...] = 0
        p["linearlayer.weight"][...] = 1e-5
    return p


# --------------------------------------------------------------------------- forward
def _relu(x):
    return np.maximum(x, 0)


def forward(params, x, kind, in_size, out_size, keep=False, **kw):
    """Network forward in the dtype of ``x`` (nn.py:110-133, 185-198, 351-374).

    With ``keep=True`` also returns the list of per-op caches needed by ``backward``.
    """
    x = np.atleast_2d(x)
    dt = x.dtype
    s0 = x
    h = x
    caches = []
    out = None
    for op in topology(kind, in_size, out_size, **kw):
        if op[0] == "linear":
            _, key, K, N, relu = op
            W = params[key + ".weight"].astype(dt)
            b = params[key + ".bias"].astype(dt)
            y = h @ W.T + b
            if relu:
                y = _relu(y)
            caches.append((h, y))
            h = y
        elif op[0] == "resblock":
            _, key, K, C, N = op
            W1 = params[key + ".layer1.weight"].astype(dt)
            b1 = params[key + ".layer1.bias"].astype(dt)
            W2 = params[key + ".layer2.weight"].astype(dt)
            b2 = params[key + ".layer2.bias"].astype(dt)
            t = _relu(h @ W1.T + b1)                          # nn.py:53
            if K != N:
                skip = h @ params[key + ".skip_layer.weight"].astype(dt).T
            else:
                skip = h
            y = _relu((t @ W2.T + b2) * dt.type(0.1) + skip)  # nn.py:54
            caches.append((h, t, y))
            h = y
        else:  # inskip: out = layer8(...) + 1e-3 * linearlayer(s)   nn.py:195
            _, key, K, N, scale = op
            W = params[key + ".weight"].astype(dt)
            b = params[key + ".bias"].astype(dt)
            h = h + dt.type(scale) * (s0 @ W.T + b)
            caches.append((s0,))
    out = h
    if keep:
        return out, caches
    return out


def backward(params, caches, dout, kind, in_size, out_size, need_param_grads=True, **kw):
    """Reverse-mode through ``forward``: returns (dx, grads dict).

    Mirrors what torch autograd computes for predictor_gpu.py:285 (all grads) and for
    HMCSampler.py:32 (input gradient only, ``need_param_grads=False``).
    """
    dt = dout.dtype
    grads = {}
    ops = topology(kind, in_size, out_size, **kw)
    dh = dout
    dx_extra = None
    for op, cache in zip(reversed(ops), reversed(caches)):
        if op[0] == "inskip":
            _, key, K, N, scale = op
            (s0,) = cache
            W = params[key + ".weight"].astype(dt)
            g = dh * dt.type(scale)
            if need_param_grads:
                grads[key + ".weight"] = g.T @ s0
                grads[key + ".bias"] = g.sum(0)
            dx_extra = g @ W
        elif op[0] == "linear":
            _, key, K, N, relu = op
            hin, y = cache
            W = params[key + ".weight"].astype(dt)
            dz = dh * (y > 0) if relu else dh
            if need_param_grads:
                grads[key + ".weight"] = dz.T @ hin
                grads[key + ".bias"] = dz.sum(0)
            dh = dz @ W
        else:
            _, key, K, C, N = op
            hin, t, y = cache
            W1 = params[key + ".layer1.weight"].astype(dt)
            W2 = params[key + ".layer2.weight"].astype(dt)
            dz = dh * (y > 0)
            db = dz * dt.type(0.1)
            dt_ = (db @ W2) * (t > 0)
            if need_param_grads:
                grads[key + ".layer2.weight"] = db.T @ t
                grads[key + ".layer2.bias"] = db.sum(0)
                grads[key + ".layer1.weight"] = dt_.T @ hin
                grads[key + ".layer1.bias"] = dt_.sum(0)
            dh_new = dt_ @ W1
            if K != N:
                Ws = params[key + ".skip_layer.weight"].astype(dt)
                if need_param_grads:
                    grads[key + ".skip_layer.weight"] = dz.T @ hin
                dh_new = dh_new + dz @ Ws
            else:
                dh_new = dh_new + dz
            dh = dh_new
    if dx_extra is not None:
        dh = dh + dx_extra
    return dh, grads
