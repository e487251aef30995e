"""Independent numpy restatement of the PredNet layer (test infrastructure).

Follows /root/reference/src/prednet.py: get_initial_state (143-190, zeros), build (192-233,
weight shapes and list order), step (235-308).  Keras-2.2.4 semantics taken from knowledge
of that library (it is not under /root/reference, not installable here => the predictor's
parity is UNPINNED): Conv2D = cross-correlation, kernel (kh,kw,Cin,Cout), padding 'same',
bias then activation; hard_sigmoid = clip(0.2x+0.5,0,1); UpSampling2D = x2 nearest;
MaxPooling2D = 2x2/stride 2; channels_last; K.rnn runs t=0..T-1 from the initial state.

This version convolves the CONCATENATED input with one einsum per conv (float32, numpy's
own summation order) and uses np.tanh, so it agrees with the canonical fmaf-chain oracle
(oracle/tz_oracle.c) only to rounding (~1e-5): it is the cross-check that the canonical
oracle computes the right function, not a bit-exact twin.
"""
import numpy as np


def split_weights(weights, L):
    """Keras weight list (prednet.py:212: sorted keys a, ahat, c, f, i, o; per level kernel, bias)."""
    it = iter(weights)
    w = {k: [] for k in ("a", "ahat", "c", "f", "i", "o")}
    for key, n in (("a", L - 1), ("ahat", L), ("c", L), ("f", L), ("i", L), ("o", L)):
        for _ in range(n):
            w[key].append((np.asarray(next(it), np.float32), np.asarray(next(it), np.float32)))
    return w


def conv_same(x, k, b):
    h, w, _ = x.shape
    xp = np.zeros((h + 2, w + 2, x.shape[2]), np.float32)
    xp[1:-1, 1:-1] = x
    win = np.lib.stride_tricks.sliding_window_view(xp, (3, 3), axis=(0, 1))  # (h,w,C,3,3)
    return (np.einsum("hwcyx,yxco->hwo", win, k, optimize=True) + b).astype(np.float32)


def hard_sigmoid(x):
    return np.clip(np.float32(0.2) * x + np.float32(0.5), 0, 1).astype(np.float32)


def relu(x):
    return np.maximum(x, 0).astype(np.float32)


def upsample(x):
    return np.repeat(np.repeat(x, 2, axis=0), 2, axis=1)


def pool(x):
    h, w, c = x.shape
    return x.reshape(h // 2, 2, w // 2, 2, c).max(axis=(1, 3))


def step(w, L, a, r_tm1, c_tm1, e_tm1):
    """prednet.py:235-308 with output_mode='prediction'. Returns (frame_prediction, r, c, e)."""
    r, c, e = [None] * L, [None] * L, [None] * L
    r_up = None
    for l in reversed(range(L)):
        inputs = [r_tm1[l], e_tm1[l]] + ([r_up] if l < L - 1 else [])
        x = np.concatenate(inputs, axis=-1)
        i = hard_sigmoid(conv_same(x, *w["i"][l]))
        f = hard_sigmoid(conv_same(x, *w["f"][l]))
        o = hard_sigmoid(conv_same(x, *w["o"][l]))
        g = np.tanh(conv_same(x, *w["c"][l])).astype(np.float32)
        c[l] = f * c_tm1[l] + i * g
        r[l] = o * np.tanh(c[l]).astype(np.float32)
        if l > 0:
            r_up = upsample(r[l])
    pred = None
    for l in range(L):
        ahat = relu(conv_same(r[l], *w["ahat"][l]))
        if l == 0:
            ahat = np.minimum(ahat, np.float32(1.0))
            pred = ahat
        e[l] = np.concatenate([relu(ahat - a), relu(a - ahat)], axis=-1)
        if l < L - 1:
            a = pool(relu(conv_same(e[l], *w["a"][l])))
    return pred, r, c, e


def predict(weights, stack, rstack, X, return_states=False):
    """Model.predict on (T, Hp, Wp, C) for one sample -> (T, Hp, Wp, C) (compress.py:227); with return_states also
    the state list r + c + e (prednet.py:298) after the last step."""
    L = len(stack)
    w = split_weights(weights, L)
    T, hp, wp, _ = X.shape
    r = [np.zeros((hp >> l, wp >> l, rstack[l]), np.float32) for l in range(L)]
    c = [z.copy() for z in r]
    e = [np.zeros((hp >> l, wp >> l, 2 * stack[l]), np.float32) for l in range(L)]
    out = []
    for t in range(T):
        pred, r, c, e = step(w, L, X[t].astype(np.float32), r, c, e)
        out.append(pred)
    if return_states:
        return np.stack(out), r + c + e
    return np.stack(out)
