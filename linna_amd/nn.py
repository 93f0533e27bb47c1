"""Emulator networks of LINNA on MI355X.

Mirrors the plug-in surface of the reference's ``linna/nn.py``: a network class is
constructed as ``nnmodel_in(in_size, out_size, linearmodel, docpu=False)`` (util.py:636,
1468), exposes ``forward``, ``init_weight``, ``state_dict``/``load_state_dict`` with the
reference's key names (``layer1.weight``, ``layer2.skip_layer.weight`` ... nn.py:77-86), and
can be moved with ``.to(device)``.

Unlike the reference these are not ``torch.nn.Module``s: all parameters live in ONE flat
fp32 device buffer (one fused AdamW launch, one gradient all-reduce) and ``forward`` /
``backward`` run hand-written HIP kernels through ``liblinna_hip.so``.  torch only owns the
memory.
"""
import collections
import ctypes as C
import math

import numpy as np
import torch

from . import _lib

__all__ = ["ChtoModelv2", "ChtoModelv2_linear", "ChtoModelsimple", "MLP", "MLP4x512", "ResBlock_batchnorm"]


def _hidden(out_size):
    # nn.py:74-76
    return 1000 if out_size > 30 else max(32, int(out_size * 32))


def _xavier_uniform(shape):
    """``nn.init.xavier_uniform_`` on a fresh contiguous ``[N, K]`` tensor (torch's global CPU generator)."""
    bound = math.sqrt(3.0) * math.sqrt(2.0 / float(shape[0] + shape[1]))
    return torch.empty(shape, dtype=torch.float32).uniform_(-bound, bound)


def _consume(n):
    """Advance torch's global CPU generator by ``n`` float32 uniform draws."""
    if n > 0:
        torch.empty(int(n), dtype=torch.float32).uniform_()


class _Op(object):
    __slots__ = ("op", "key", "K", "C", "N", "relu", "alpha")

    def __init__(self, op, key, K, N, C=0, relu=0, alpha=1.0):
        self.op, self.key, self.K, self.C, self.N, self.relu, self.alpha = op, key, K, C, N, relu, alpha

    def tensors(self):
        """(state_dict key, shape) in torch's registration order."""
        if self.op == _lib.OP_RESBLOCK:
            t = [(self.key + ".layer1.weight", (self.C, self.K)), (self.key + ".layer1.bias", (self.C,)),
                 (self.key + ".layer2.weight", (self.N, self.C)), (self.key + ".layer2.bias", (self.N,))]
            if self.K != self.N:                       # nn.py:28-31: Identity when in == out
                t.append((self.key + ".skip_layer.weight", (self.N, self.K)))
            return t
        return [(self.key + ".weight", (self.N, self.K)), (self.key + ".bias", (self.N,))]


class _Emulator(object):
    """Flat-buffer network executed by liblinna_hip.so."""

    def __init__(self, in_size, out_size, linearmodel=None, docpu=False):
        if linearmodel is not None:
            # the reference always passes None (util.py:634, 1464); the PCA baseline is dead code there
            raise NotImplementedError("linearmodel is not supported (always None in the reference)")
        self.in_size, self.out_size = int(in_size), int(out_size)
        self.linearmodel = None
        self.docpu = docpu           # kept for signature parity; there is no CPU path here
        self.ops = self._build_ops()
        self._index = collections.OrderedDict()
        off = 0
        for op in self.ops:
            for key, shp in op.tensors():
                # packed convention of include/linna_hip.h: weight rows padded to a multiple of 4
                # floats (16-byte rows for LDS-DMA), every tensor 16-byte aligned; pads stay zero
                n = shp[0] * _lib.ld4(shp[1]) if len(shp) == 2 else _lib.ld4(shp[0])
                self._index[key] = (off, shp)
                off += n
        self.nflat = off
        self.nparams = sum(int(np.prod(s)) for _, s in self._index.values())
        self._flat = torch.zeros(self.nflat, dtype=torch.float32)
        self._grad = None
        self._net = None
        self._ws = {}
        self.training = False
        self._constructor_draws()
        self.init_weight()
        self._post_init()

    # ------------------------------------------------------------------ structure
    def _build_ops(self):
        raise NotImplementedError

    @property
    def device(self):
        return self._flat.device

    def macs_per_eval(self):
        return sum(int(np.prod(s)) for k, (_, s) in self._index.items() if k.endswith("weight"))

    # ------------------------------------------------------------------ parameters
    def _view(self, buf, key):
        off, shp = self._index[key]
        if len(shp) == 2:
            ld = _lib.ld4(shp[1])
            return buf[off:off + shp[0] * ld].view(shp[0], ld)[:, :shp[1]]
        return buf[off:off + shp[0]]

    def init_weight(self):
        """``init_weight()`` of the reference's network classes (nn.py:91-108 with the block's nn.py:34-43), bit for
        bit: same torch global RNG, same draw order.  ``self.modules()`` there walks the tree in pre-order, so a
        residual block is initialised by its own ``init_weight`` (Xavier-uniform on its three Linear weights, skip
        weights zeroed) and then its three Linear children are visited AGAIN by the outer loop -- the skip weights
        end up Xavier-uniform, not zero, and the first three draws of every block are discarded.  Biases 1e-2."""
        host = torch.zeros(self.nflat, dtype=torch.float32)
        for op in self.ops:
            if op.op == _lib.OP_RESBLOCK:
                names = [op.key + ".layer1.weight", op.key + ".layer2.weight"]
                if op.K != op.N:
                    names.append(op.key + ".skip_layer.weight")
                for key in names:                                   # the block's own init_weight: overwritten below
                    _xavier_uniform(self._index[key][1])
                for key in names:
                    self._view(host, key).copy_(_xavier_uniform(self._index[key][1]))
                self._view(host, op.key + ".layer1.bias").fill_(1e-2)
                self._view(host, op.key + ".layer2.bias").fill_(1e-2)
            else:
                self._view(host, op.key + ".weight").copy_(_xavier_uniform(self._index[op.key + ".weight"][1]))
                self._view(host, op.key + ".bias").fill_(1e-2)
        self._flat.copy_(host)
        self.weights_changed()

    def _constructor_draws(self):
        """The RNG draws the reference's constructor makes before its ``init_weight()``: every ``nn.Linear`` is born
        with kaiming-uniform weights and a uniform bias (one draw per element, in registration order), and every
        residual block runs its own ``init_weight`` once (nn.py:33).  All of it is overwritten; consuming the same
        number of draws makes ``torch.manual_seed(s); Model(...)`` yield the reference's initial weights."""
        for op in self.ops:
            if op.op == _lib.OP_RESBLOCK:
                shapes = [(op.C, op.K), (op.N, op.C)] + ([(op.N, op.K)] if op.K != op.N else [])
                _consume(op.C * op.K + op.C)
                _consume(op.N * op.C + op.N)
                if op.K != op.N:
                    _consume(op.N * op.K)                           # skip_layer: bias=False
                for shp in shapes:
                    _consume(shp[0] * shp[1])
            else:
                _consume(op.N * op.K + op.N)

    def _post_init(self):
        pass

    def weights_changed(self):
        """Tell the HIP library that parameter memory may have been written from the torch side
        (`linna_weights_changed`): serving objects re-lay their fragment-order weight copy before
        their next evaluation.  Every accessor that hands out a WRITABLE view of the parameters
        calls this, so `model.flat_params().copy_(...)` / `state_dict()[k].copy_(...)` followed by
        an evaluation is safe; a caller that keeps such a view and writes through it later calls
        it again after the write."""
        if self._flat.is_cuda:
            _lib.call("linna_weights_changed", _lib.ctx(self._flat.device.index))

    def state_dict(self):
        self.weights_changed()
        return collections.OrderedDict((k, self._view(self._flat, k)) for k in self._index)

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self._index if k not in sd]
        extra = [k for k in sd if k not in self._index]
        if strict and (missing or extra):
            raise KeyError("state_dict mismatch: missing %s, unexpected %s" % (missing, extra))
        for k in self._index:
            if k in sd:
                src = sd[k]
                src = torch.as_tensor(np.asarray(src)) if not torch.is_tensor(src) else src
                dst = self._view(self._flat, k)
                if tuple(src.shape) != tuple(dst.shape):
                    raise ValueError("%s: checkpoint shape %s, model shape %s" % (k, tuple(src.shape), tuple(dst.shape)))
                dst.copy_(src.detach().to(torch.float32))
        self.weights_changed()
        return self

    def parameters(self):
        self.weights_changed()
        return [self._flat]

    def named_parameters(self):
        return list(self.state_dict().items())

    def flat_params(self):
        self.weights_changed()
        return self._flat

    def flat_grads(self):
        if self._grad is None or self._grad.device != self._flat.device:
            # four spare floats behind the gradients: the step's scalar loss rides in the same all-reduce (grad_tail)
            self._grad_buf = torch.zeros(self.nflat + 4, dtype=torch.float32, device=self._flat.device)
            self._grad = self._grad_buf[:self.nflat]
            self._net = None           # layer table carries gradient pointers
        return self._grad

    def grad_tail(self):
        """One float right behind the flat gradient buffer (data-parallel training sums gradient and loss in one call)."""
        self.flat_grads()
        return self._grad_buf[self.nflat:self.nflat + 1]

    def grad_dict(self):
        g = self.flat_grads()
        return collections.OrderedDict((k, self._view(g, k)) for k in self._index)

    def to(self, device=None, **kwargs):
        if device is None:             # e.g. .to(memory_format=...) in main.py:267: nothing to do
            return self
        device = torch.device(device)
        if device != self._flat.device:
            self._flat = self._flat.to(device)
            self._grad = None
            self._destroy_net()
            self._ws = {}
        return self

    def cuda(self):
        return self.to("cuda")

    def cpu(self):
        return self.to("cpu")

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def __deepcopy__(self, memo):
        new = self.__class__.__new__(self.__class__)
        new.__dict__.update({k: v for k, v in self.__dict__.items() if k not in ("_flat", "_grad", "_grad_buf", "_net", "_ws")})
        new._flat = self._flat.clone()
        new._grad = None
        new._net = None
        new._ws = {}
        return new

    # ------------------------------------------------------------------ HIP network handle
    def _destroy_net(self):
        if self._net is not None:
            _lib.load().linna_net_destroy(self._net)
            self._net = None

    def __del__(self):
        try:
            self._destroy_net()
        except Exception:
            pass

    def _ptr(self, buf, key):
        off, _ = self._index[key]
        return C.c_void_p(buf.data_ptr() + 4 * off)

    def net_handle(self, with_grads=False):
        """linna_net_t* over the CURRENT flat buffers (rebuilt if they moved)."""
        if not self._flat.is_cuda:
            raise _lib.LinnaHipError("the emulator runs on the GPU only: call .to('cuda') first (no CPU fallback)")
        g = self.flat_grads() if with_grads else self._grad
        sig = (self._flat.data_ptr(), g.data_ptr() if g is not None else 0)
        if self._net is not None and self._net_sig == sig:
            return self._net
        self._destroy_net()
        arr = _lib.sized_array(_lib.Layer, len(self.ops))
        for i, op in enumerate(self.ops):
            L = arr[i]
            L.op, L.K, L.C, L.N, L.relu, L.alpha = op.op, op.K, op.C, op.N, op.relu, op.alpha
            P = lambda k: self._ptr(self._flat, k)
            G = (lambda k: self._ptr(g, k)) if g is not None else (lambda k: None)
            if op.op == _lib.OP_RESBLOCK:
                L.W1, L.b1 = P(op.key + ".layer1.weight"), P(op.key + ".layer1.bias")
                L.W2, L.b2 = P(op.key + ".layer2.weight"), P(op.key + ".layer2.bias")
                L.gW1, L.gb1 = G(op.key + ".layer1.weight"), G(op.key + ".layer1.bias")
                L.gW2, L.gb2 = G(op.key + ".layer2.weight"), G(op.key + ".layer2.bias")
                if op.K != op.N:
                    L.Ws, L.gWs = P(op.key + ".skip_layer.weight"), G(op.key + ".skip_layer.weight")
            else:
                L.W, L.b = P(op.key + ".weight"), P(op.key + ".bias")
                L.gW, L.gb = G(op.key + ".weight"), G(op.key + ".bias")
        h = C.c_void_p()
        _lib.call("linna_net_create", _lib.ctx(self._flat.device.index), arr, len(self.ops), self.in_size, C.byref(h))
        self._net, self._net_sig = h, sig
        return h

    def workspace(self, B, kind="fwd"):
        key = (kind, int(B))
        ws = self._ws.get(key)
        if ws is None:
            lib = _lib.load()
            fn = lib.linna_net_fwd_ws_bytes if kind == "fwd" else lib.linna_net_bwd_ws_bytes
            nbytes = fn(self.net_handle(), int(B))
            ws = torch.empty(nbytes // 4 + 4, dtype=torch.float32, device=self._flat.device)
            self._ws[key] = ws
        return ws

    # ------------------------------------------------------------------ compute
    def forward(self, s, colmap=None, out=None):
        """Network output for ``s[B, in_size]`` (or ``[in_size]``) on the device; with
        ``colmap`` (a ``_lib.ColMap``) the output-side affine/exp is fused into the last GEMM."""
        one = s.dim() == 1
        x = s.view(1, -1) if one else s
        if x.shape[1] != self.in_size:
            raise ValueError("expected %d inputs, got %d" % (self.in_size, x.shape[1]))
        x = x.detach().to(device=self._flat.device, dtype=torch.float32).contiguous()
        y = self.forward_buffer(x, x.shape[0], colmap=colmap, out=out)
        return y.reshape(-1) if one else y

    def forward_buffer(self, xbuf, B, colmap=None, out=None):
        """Same, for an input already on the device as ``xbuf[B, ld >= in_size]`` (padded rows
        are fine: the row stride is passed to the kernels)."""
        ldo = _lib.ld4(self.out_size)
        if out is None:
            out = torch.empty((B, ldo), dtype=torch.float32, device=xbuf.device)
        _lib.call("linna_net_forward", self.net_handle(), _lib.ptr(xbuf), xbuf.stride(0), B,
                  _lib.ptr(self.workspace(B)), _lib.ptr(out), out.stride(0),
                  C.byref(colmap) if colmap is not None else None, _lib.stream())
        self._last_input = xbuf
        return out[:, :self.out_size]

    __call__ = forward

    def stream_state(self):
        """(forward, dX chain, dX chain down to the input): 1 where that part runs as one launch of the
        whole-network kernel, 0 where it runs as one GEMM per op, -1 before its first call."""
        f, d, di = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        _lib.call("linna_net_stream_state", self.net_handle(), C.byref(f), C.byref(d), C.byref(di))
        return f.value, d.value, di.value

    def uses_dx_stream(self):
        return 1 in self.stream_state()[1:]

    def backward(self, dout, param_grads=True, need_dx=False):
        """Reverse pass for the most recent ``forward`` (same batch): fills ``flat_grads()``
        and/or returns d/d(input)."""
        x = self._last_input
        B = x.shape[0]
        dout = dout.to(torch.float32)
        if dout.stride(-1) != 1:
            dout = dout.contiguous()
        dx = torch.empty((B, _lib.ld4(self.in_size)), dtype=torch.float32, device=x.device) if need_dx else None
        _lib.call("linna_net_backward", self.net_handle(with_grads=param_grads), _lib.ptr(x), x.stride(0), B,
                  _lib.ptr(self.workspace(B)), _lib.ptr(self.workspace(B, "bwd")), C.c_void_p(dout.data_ptr()),
                  dout.stride(0), _lib.ptr(dx) if dx is not None else None, dx.stride(0) if dx is not None else 0,
                  1 if param_grads else 0, _lib.stream())
        return dx[:, :self.in_size] if dx is not None else None


class ChtoModelv2(_Emulator):
    """The network ``ml_sampler`` hard-wires (main.py:70); topology of nn.py:59-133."""
    channel = 16
    wide_layer6 = True

    def _build_ops(self):
        L, R = _lib.OP_LINEAR, _lib.OP_RESBLOCK
        h = _hidden(self.out_size)
        ops = [_Op(L, "layer1", self.in_size, h, relu=1)]
        for i, mult in enumerate((1, 2, 4)):
            ops.append(_Op(R, "layer%d" % (i + 2), h, h // 2, C=self.channel * mult))
            h //= 2
        h6 = 4 * h if self.wide_layer6 else h
        ops.append(_Op(L, "layer6", h, h6, relu=1))
        ops.append(_Op(L, "layer7", h6, self.out_size, relu=1))
        ops.append(_Op(L, "layer8", self.out_size, self.out_size, relu=0))
        return ops


class ChtoModelsimple(ChtoModelv2):
    """nn.py:300-374: channel 4, layer6 h -> h."""
    channel = 4
    wide_layer6 = False


class ChtoModelv2_linear(ChtoModelv2):
    """nn.py:136-198: adds 1e-3 * Linear(in, out)(input) to the output."""

    def _build_ops(self):
        ops = ChtoModelv2._build_ops(self)
        ops.append(_Op(_lib.OP_INSKIP, "linearlayer", self.in_size, self.out_size, alpha=1e-3))
        return ops

    def _post_init(self):
        # nn.py:162-163: only the constructor does this; a later init_weight() leaves the layer Xavier-initialised
        self._view(self._flat, "linearlayer.bias").zero_()
        self._view(self._flat, "linearlayer.weight").fill_(1e-5)


class MLP(_Emulator):
    """Plain ReLU MLP ``in -> width x depth -> out`` (BASELINE configs 2 and 5: 4 x 512);
    not a reference class, constructible through the same plug-in signature."""

    def __init__(self, in_size, out_size, linearmodel=None, docpu=False, width=512, depth=4):
        self.width, self.depth = int(width), int(depth)
        _Emulator.__init__(self, in_size, out_size, linearmodel, docpu)

    def _build_ops(self):
        ops, k = [], self.in_size
        for i in range(self.depth):
            ops.append(_Op(_lib.OP_LINEAR, "layer%d" % (i + 1), k, self.width, relu=1))
            k = self.width
        ops.append(_Op(_lib.OP_LINEAR, "layer%d" % (self.depth + 1), k, self.out_size, relu=0))
        return ops


MLP4x512 = MLP


class ResBlock_batchnorm(object):
    """Shape helper named after nn.py:11-56 (which, despite the name, has no batch-norm).
    The block itself executes inside the network kernels (``linna_resblock_fwd``)."""

    def __init__(self, in_size, channel, out_size):
        self.in_size, self.channel, self.out_size = in_size, channel, out_size
        self.op = _Op(_lib.OP_RESBLOCK, "block", in_size, out_size, C=channel)


def describe_program(model, rows=16, dense_nout=0):
    """The serving program the whole-network kernel would run for ``model`` on the engine of ``rows`` rows per workgroup, as
    text (``linna_program_describe``: host-side planning, no GPU needed; parameter pointers are placeholders)."""
    arr = _lib.sized_array(_lib.Layer, len(model.ops))
    nxt = [4096]

    def fake(n):                                            # distinct, 16-byte aligned, never read
        p = nxt[0]
        nxt[0] += 16 * ((int(n) + 3) // 4 + 1)
        return C.c_void_p(p)
    for i, op in enumerate(model.ops):
        L = arr[i]
        L.op, L.K, L.C, L.N, L.relu, L.alpha = op.op, op.K, op.C, op.N, op.relu, op.alpha
        if op.op == _lib.OP_RESBLOCK:
            L.W1, L.b1, L.W2, L.b2 = fake(op.C * op.K), fake(op.C), fake(op.N * op.C), fake(op.N)
            if op.K != op.N:
                L.Ws = fake(op.N * op.K)
        else:
            L.W, L.b = fake(op.N * op.K), fake(op.N)
    buf = C.create_string_buffer(8192)
    n = _lib.load().linna_program_describe(arr, len(model.ops), model.in_size, int(rows), int(dense_nout), buf, len(buf))
    if n < 0:
        _lib.check(n)
    return n, buf.value.decode()
