"""Third independent restatement of the PredNet layer: torch conv2d / max_pool2d / interpolate
(test infrastructure; see oracle/prednet_np.py for the Keras-2.2.4 semantics it assumes and why
the predictor's parity is UNPINNED).

Follows /root/reference/src/prednet.py:143-190 (zero initial state), 235-308 (step) LITERALLY:
every `predict` is the two-timestep evaluation from zero state that compress.py:224-229 issues,
the upsampled r_{l+1} is materialised (prednet.py:264) and convolved with the full 3x3 kernel on
the concatenated input [r, e, r_up] -- no folding of constants, no collapsed taps, the library's
own summation order.  It therefore plays the part of a "foreign" decoder (TensorFlow, another
GPU library) when the cross-decoder deviation of the HIP path is measured
(tests/cross_decoder_deviation.py), and it cross-checks the tap-collapse algebra
of the canonical oracle / the kernels.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import prednet_np


class TorchPredNet:
    def __init__(self, weights, stack_sizes, R_stack_sizes, hp, wp, device="cpu", dtype=torch.float32):
        self.st, self.rs, self.L = tuple(stack_sizes), tuple(R_stack_sizes), len(stack_sizes)
        self.hp, self.wp, self.dev, self.dtype = hp, wp, torch.device(device), dtype
        ws = prednet_np.split_weights(weights, self.L)
        self.w = {k: [(torch.from_numpy(np.ascontiguousarray(kk)).permute(3, 2, 0, 1).contiguous().to(self.dev, dtype),
                       torch.from_numpy(np.ascontiguousarray(bb)).to(self.dev, dtype)) for kk, bb in v]
                  for k, v in ws.items()}

    def _conv(self, x, kb):
        return F.conv2d(x[None], kb[0], kb[1], padding=1)[0]

    @staticmethod
    def _hs(x):  # Keras hard_sigmoid
        return torch.clamp(0.2 * x + 0.5, 0, 1)

    def predict2(self, frame):
        """frame (Hp,Wp,3) float32 in [0,1] -> (X_hat[0,0], X_hat[0,1]) as numpy (Hp,Wp,3)."""
        f = np.ascontiguousarray(frame, dtype=np.float32)
        outs = self.predict_seq(np.stack([f, np.zeros_like(f)]))
        return outs[0], outs[1]

    def predict_seq(self, X, return_states=False):
        """Model.predict on one sample: X (T,Hp,Wp,3) -> X_hat (T,Hp,Wp,3); with return_states also the state list
        (r, c, e per level, HWC numpy) after the last step, the layout of prednet.py:298."""
        L, st, rs, hp, wp = self.L, self.st, self.rs, self.hp, self.wp
        z = lambda c, l: torch.zeros(c, hp >> l, wp >> l, device=self.dev, dtype=self.dtype)  # noqa: E731
        r = [z(rs[l], l) for l in range(L)]
        c = [z(rs[l], l) for l in range(L)]
        e = [z(2 * st[l], l) for l in range(L)]
        outs = []
        with torch.no_grad():
            for t in range(len(X)):
                a = torch.from_numpy(np.ascontiguousarray(X[t], dtype=np.float32)).to(self.dev, self.dtype).permute(2, 0, 1)
                rn, cn = [None] * L, [None] * L
                for l in reversed(range(L)):
                    up = [F.interpolate(rn[l + 1][None], scale_factor=2, mode="nearest")[0]] if l < L - 1 else []
                    x = torch.cat([r[l], e[l]] + up)
                    i = self._hs(self._conv(x, self.w["i"][l]))
                    f = self._hs(self._conv(x, self.w["f"][l]))
                    o = self._hs(self._conv(x, self.w["o"][l]))
                    cn[l] = f * c[l] + i * torch.tanh(self._conv(x, self.w["c"][l]))
                    rn[l] = o * torch.tanh(cn[l])
                for l in range(L):
                    ahat = torch.relu(self._conv(rn[l], self.w["ahat"][l]))
                    if l == 0:
                        ahat = torch.clamp(ahat, max=1.0)
                        outs.append(ahat.permute(1, 2, 0).to(torch.float32).cpu().numpy())
                    e[l] = torch.cat([torch.relu(ahat - a), torch.relu(a - ahat)])
                    if l < L - 1:
                        a = F.max_pool2d(torch.relu(self._conv(e[l], self.w["a"][l]))[None], 2)[0]
                r, c = rn, cn
        if return_states:
            return np.stack(outs), [s.permute(1, 2, 0).to(torch.float32).cpu().numpy() for s in r + c + e]
        return np.stack(outs)

    def c0(self, hp=None, wp=None):
        return self.predict2(np.zeros((self.hp, self.wp, 3), np.float32))[0]

    def next(self, frame):
        return self.predict2(frame)[1]


def trunc255(pred):
    """compress.py:307,311: int(pred_f32 * 255.0) on the float32 product."""
    return (np.asarray(pred, np.float32) * np.float32(255.0)).astype(np.int32)


def deviation_by_depth(next_a, next_b, key_frame_f32, depth):
    """Both predictors roll out `depth` frames from the same key frame, each feeding on its own
    predictions (what an encoder with predictor A and a decoder with predictor B do).  Per depth:
    (fraction of samples whose trunc(pred*255) differs, max |difference|, max |pred_a - pred_b|).
    The integer difference IS the pixel error a B-order decoder makes on a lossless stream
    written with A (decompress.py:252-253: recon = trunc(pred_B*255) - (trunc(pred_A*255) - orig))."""
    a = b = np.asarray(key_frame_f32, np.float32)
    rows = []
    for d in range(1, depth + 1):
        a, b = next_a(a), next_b(b)
        diff = trunc255(a) - trunc255(b)
        rows.append((d, float((diff != 0).mean()), int(np.abs(diff).max()), float(np.abs(a - b).max())))
    return rows
