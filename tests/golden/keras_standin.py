"""numpy stand-ins for the part of the Keras 2.2.4 surface that /root/reference/src/prednet.py touches.

Used ONLY by tests/golden/make_golden.py in the build container, so that the reference's own `PredNet.build`
(prednet.py:192-233), `get_initial_state` (143-190) and `step` (235-308) EXECUTE and their results can be committed as
ref_prednet.npz.  What this pins is the reference's WIRING: which convolution sees which concatenation in which channel
order, the state list layout, the weight list order `trainable_weights` ends up in, the gate formula, the error split,
the pooling/upsampling placement, the clip of the pixel layer.

What it does NOT pin are the primitives themselves: keras==2.2.4 / tensorflow==1.15 are not under /root/reference and
are not installable here, so the meaning of `Conv2D` (cross-correlation, kernel (kh, kw, Cin, Cout), 'same' = zero pad 1,
bias then activation), `hard_sigmoid` (clip(0.2 x + 0.5, 0, 1)), `UpSampling2D` (x2 nearest), `MaxPooling2D` (2x2,
stride 2), `Layer.set_weights` (pairs trainable_weights in order) and `K.rnn` (t = 0..T-1 from the initial state) is
written below from knowledge of that library.  DESIGN.md §2 says the same.

Arithmetic: tensors are float32 (Keras' floatx); every convolution accumulates in float64 and rounds once, tanh is
evaluated in float64 and rounded once -- a neutral arbiter that all float32 restatements (numpy einsum, torch, the
C oracle's two contracts, the HIP kernels) are held to at 2e-5.
"""
import contextlib
import types

import numpy as np

F32 = np.float32


class Variable:
    """A weight: Keras creates it in Layer.build, `set_weights` assigns into it."""

    def __init__(self, value, name):
        self.value = np.asarray(value, F32)
        self.name = name


# ---- keras.activations ------------------------------------------------------------------------------------------
def relu(x):
    return np.maximum(x, F32(0))


def tanh(x):
    return np.tanh(np.asarray(x, np.float64)).astype(F32)


def hard_sigmoid(x):
    return np.clip(F32(0.2) * x + F32(0.5), F32(0), F32(1)).astype(F32)


def linear(x):
    return x


_ACT = {"relu": relu, "tanh": tanh, "hard_sigmoid": hard_sigmoid, "linear": linear, None: linear}


def activations_get(identifier):
    return identifier if callable(identifier) else _ACT[identifier]


# ---- keras.layers -----------------------------------------------------------------------------------------------
_SCOPE = []


class Conv2D:
    def __init__(self, filters, kernel_size, padding="valid", activation=None, data_format=None):
        assert padding == "same" and data_format == "channels_last"
        self.filters = int(filters)
        self.k = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        assert self.k[0] % 2 == 1 and self.k[1] % 2 == 1
        self.activation = activations_get(activation)
        self.trainable_weights = []

    def build(self, input_shape):
        cin = input_shape[-1]
        scope = "/".join(_SCOPE)
        self.kernel = Variable(np.zeros(self.k + (cin, self.filters)), scope + "/kernel")
        self.bias = Variable(np.zeros((self.filters,)), scope + "/bias")
        self.trainable_weights = [self.kernel, self.bias]

    def call(self, x):
        n, h, w, cin = x.shape
        kh, kw = self.k
        assert cin == self.kernel.value.shape[2], (x.shape, self.kernel.value.shape)
        xp = np.zeros((n, h + kh - 1, w + kw - 1, cin), np.float64)
        xp[:, kh // 2: kh // 2 + h, kw // 2: kw // 2 + w] = x
        win = np.lib.stride_tricks.sliding_window_view(xp, (kh, kw), axis=(1, 2))        # (n, h, w, cin, kh, kw)
        acc = np.einsum("nhwcyx,yxco->nhwo", win, self.kernel.value.astype(np.float64), optimize=True)
        return self.activation((acc + self.bias.value.astype(np.float64)).astype(F32))


class UpSampling2D:
    def __init__(self, data_format=None):
        assert data_format == "channels_last"

    def call(self, x):
        return np.repeat(np.repeat(x, 2, axis=1), 2, axis=2)


class MaxPooling2D:
    def __init__(self, data_format=None):
        assert data_format == "channels_last"

    def call(self, x):
        n, h, w, c = x.shape
        return x[:, : h // 2 * 2, : w // 2 * 2].reshape(n, h // 2, 2, w // 2, 2, c).max(axis=(2, 4))


class InputSpec:
    def __init__(self, ndim=None, shape=None):
        self.ndim, self.shape = ndim, shape


class Recurrent:
    """The base-class behaviour PredNet relies on: constructor keywords (`weights`, `return_sequences`, ...),
    `set_weights`, and `__call__` = build, apply the constructor's weights, then the K.rnn loop over time."""

    def __init__(self, weights=None, return_sequences=False, name=None, trainable=True, **kwargs):
        self._initial_weights = weights
        self.return_sequences = return_sequences
        self.name = name
        self.built = False

    def set_weights(self, weights):
        assert len(weights) == len(self.trainable_weights), (len(weights), len(self.trainable_weights))
        for var, w in zip(self.trainable_weights, weights):
            assert var.value.shape == np.shape(w), (var.name, var.value.shape, np.shape(w))
            var.value = np.asarray(w, F32).copy()

    def get_weights(self):
        return [v.value.copy() for v in self.trainable_weights]

    def get_config(self):
        return {"return_sequences": self.return_sequences, "name": self.name}

    def __call__(self, x, record=None):
        x = np.asarray(x, F32)
        if not self.built:
            self.build((None,) + x.shape[1:])
            self.built = True
            if self._initial_weights is not None:
                self.set_weights(self._initial_weights)
        states = self.get_initial_state(x)
        if record is not None:
            record.append([np.array(s) for s in states])
        outs = []
        for t in range(x.shape[1]):
            out, states = self.step(x[:, t], states)
            outs.append(out)
            if record is not None:
                record.append([np.array(s) for s in states])
        return np.stack(outs, axis=1) if self.return_sequences else outs[-1]


# ---- keras.backend ----------------------------------------------------------------------------------------------
@contextlib.contextmanager
def name_scope(name):
    _SCOPE.append(name)
    try:
        yield
    finally:
        _SCOPE.pop()


def _switch(cond, a, b):
    return a if cond else b


backend = types.SimpleNamespace(
    _BACKEND="tensorflow",
    backend=lambda: "tensorflow",
    image_data_format=lambda: "channels_last",
    zeros_like=lambda x: np.zeros_like(x, dtype=F32),
    zeros=lambda shape: np.zeros(shape, F32),
    sum=lambda x, axis=None: np.sum(x, axis=axis, dtype=F32),
    mean=lambda x, axis=None, keepdims=False: np.mean(x, axis=axis, keepdims=keepdims, dtype=F32),
    dot=lambda a, b: (a @ b).astype(F32),
    reshape=lambda x, shape: np.reshape(x, shape),
    concatenate=lambda xs, axis=-1: np.concatenate(list(xs), axis=axis),
    minimum=lambda a, b: np.minimum(a, F32(b)) if np.isscalar(b) else np.minimum(a, b),
    batch_flatten=lambda x: np.reshape(x, (x.shape[0], -1)),
    name_scope=name_scope,
    switch=_switch,
    variable=lambda v, dtype=None: np.asarray(v),
)


def install(stub):
    """Put the stand-ins into the already-registered stub modules (make_golden._install_stubs)."""
    import sys
    K = sys.modules["keras.backend"]
    K.__dict__.update(backend.__dict__)
    L = sys.modules["keras.layers"]
    L.Recurrent, L.Conv2D, L.UpSampling2D, L.MaxPooling2D = Recurrent, Conv2D, UpSampling2D, MaxPooling2D
    sys.modules["keras.engine"].InputSpec = InputSpec
    sys.modules["keras.activations"].get = activations_get
