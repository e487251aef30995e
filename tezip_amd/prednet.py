"""Host-side description of the PredNet predictor (shapes, weight list order, init).

Mirrors the constructor surface of the reference layer
(/root/reference/src/prednet.py:77-118 `PredNet(stack_sizes, R_stack_sizes, A_filt_sizes,
Ahat_filt_sizes, R_filt_sizes, ...)`) and its `build` (prednet.py:192-233): 23 convolutions
for the default 4-level model, weight list ordered by sorted key -- a, ahat, c, f, i, o --
then level, each as [kernel (3,3,Cin,Cout) HWIO, bias (Cout)].  No arithmetic happens here:
the forward pass is the HIP library's (tezip_amd/csrc).
"""
import numpy as np

DEFAULT_STACK = (3, 48, 96, 192)  # train.py:51-52


class PredNetConfig:
    def __init__(self, stack_sizes=DEFAULT_STACK, R_stack_sizes=None, A_filt_sizes=None,
                 Ahat_filt_sizes=None, R_filt_sizes=None, pixel_max=1.0):
        self.stack_sizes = tuple(int(v) for v in stack_sizes)
        self.R_stack_sizes = tuple(int(v) for v in (R_stack_sizes or stack_sizes))
        L = self.nb_layers = len(self.stack_sizes)
        if len(self.R_stack_sizes) != L:
            raise ValueError("len(R_stack_sizes) must equal len(stack_sizes)")  # prednet.py:85
        self.A_filt_sizes = tuple(A_filt_sizes or (3,) * (L - 1))
        self.Ahat_filt_sizes = tuple(Ahat_filt_sizes or (3,) * L)
        self.R_filt_sizes = tuple(R_filt_sizes or (3,) * L)
        if len(self.A_filt_sizes) != L - 1 or len(self.Ahat_filt_sizes) != L or len(self.R_filt_sizes) != L:
            raise ValueError("filter size lists have the wrong length")  # prednet.py:87-91
        if set(self.A_filt_sizes + self.Ahat_filt_sizes + self.R_filt_sizes) - {3}:
            raise NotImplementedError("only 3x3 filters (train.py:53-55) are supported by the HIP path")
        if float(pixel_max) != 1.0:
            raise NotImplementedError("pixel_max must be 1.0 (prednet.py:79)")
        self.pixel_max = 1.0

    def weight_shapes(self):
        """[(name, shape)] in the Keras weight-list order (prednet.py:210-227)."""
        L, st, rs = self.nb_layers, self.stack_sizes, self.R_stack_sizes
        out = []
        for l in range(L - 1):
            out += [("a%d/kernel" % l, (3, 3, 2 * st[l], st[l + 1])), ("a%d/bias" % l, (st[l + 1],))]
        for l in range(L):
            out += [("ahat%d/kernel" % l, (3, 3, rs[l], st[l])), ("ahat%d/bias" % l, (st[l],))]
        for g in ("c", "f", "i", "o"):
            for l in range(L):
                cin = 2 * st[l] + rs[l] + (rs[l + 1] if l < L - 1 else 0)
                out += [("%s%d/kernel" % (g, l), (3, 3, cin, rs[l])), ("%s%d/bias" % (g, l), (rs[l],))]
        return out

    def n_params(self):
        return sum(int(np.prod(s)) for _, s in self.weight_shapes())

    def init_weights(self, seed=123, bias_scale=0.0):
        """Keras defaults: glorot_uniform kernels, zero biases (train.py:4 seeds with 123).
        bias_scale>0 draws uniform biases instead (tests use it to exercise the bias path)."""
        rng = np.random.default_rng(seed)
        ws = []
        for name, shape in self.weight_shapes():
            if name.endswith("kernel"):
                fan_in, fan_out = 9 * shape[2], 9 * shape[3]
                lim = np.sqrt(6.0 / (fan_in + fan_out))
                ws.append(rng.uniform(-lim, lim, size=shape).astype(np.float32))
            elif bias_scale:
                ws.append(rng.uniform(-bias_scale, bias_scale, size=shape).astype(np.float32))
            else:
                ws.append(np.zeros(shape, np.float32))
        return ws

    def to_json_dict(self):
        return {"class_name": "PredNet", "config": {
            "stack_sizes": list(self.stack_sizes), "R_stack_sizes": list(self.R_stack_sizes),
            "A_filt_sizes": list(self.A_filt_sizes), "Ahat_filt_sizes": list(self.Ahat_filt_sizes),
            "R_filt_sizes": list(self.R_filt_sizes), "pixel_max": float(self.pixel_max), "data_format": "channels_last"}}
