"""ctypes loader for oracle/libtzoracle.so (the C restatement; test infrastructure only)."""
import ctypes as C
import ctypes as C_
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
MODES = {"abs": 0, "rel": 1, "absrel": 2, "pwrel": 3}


def build(force=False):
    so = os.path.join(HERE, "libtzoracle.so")
    src = [os.path.join(HERE, f) for f in ("tz_oracle.c", "tz_math.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", HERE, "-s", "-B", "libtzoracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(HERE, "libtzoracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.tzo_model_create.restype = C.c_void_p
        L.tzo_model_create.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        for f in ("tzo_model_destroy", "tzo_model_prepare"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = None
        L.tzo_model_c0.argtypes = [C.c_void_p, C.c_void_p]
        L.tzo_model_next.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.tzo_model_set_contract.argtypes = [C.c_void_p, C.c_int]
        L.tzo_model_set_contract.restype = None
        L.tzo_predict2_literal.argtypes = [C.c_void_p] * 4
        L.tzo_sse_frame.restype = C.c_double
        L.tzo_sse_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.tzo_error_bound.restype = C.c_int
        L.tzo_error_bound.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_double, C.c_double]
        L.tzo_delta_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.tzo_spatial_delta.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_void_p]
        L.tzo_spatial_undelta.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_void_p]
        L.tzo_histogram.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_int]
        L.tzo_lut_apply.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
        L.tzo_reconstruct_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.tzo_act_probe.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
        L.tzo_u8_to_f32_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class CPredNet:
    """PredNet in the canonical TZ-PA1 arithmetic (see tz_oracle.c).  `weights` is the Keras
    weight list order (prednet.py:212): a, ahat, c, f, i, o per level, kernel then bias."""

    def __init__(self, weights, stack, rstack, hp, wp):
        self.w = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
        self.stack, self.rstack, self.hp, self.wp = list(stack), list(rstack), hp, wp
        L = len(stack)
        ptrs = (C.c_void_p * len(self.w))(*[w.ctypes.data for w in self.w])
        st = np.array(stack, dtype=np.int32)
        rs = np.array(rstack, dtype=np.int32)
        self.h = lib().tzo_model_create(L, _p(st), _p(rs), hp, wp, C.cast(ptrs, C.c_void_p))
        if not self.h:
            raise ValueError("bad model shape")
        self.L = L

    def __del__(self):
        if getattr(self, "h", None):
            lib().tzo_model_destroy(self.h)
            self.h = None

    def set_contract(self, contract):
        """1 = TZ-PA1 (direct chains everywhere), 2 = TZ-PA2 (tz_oracle.c conv3x3_wino for the per-frame convolutions of
        levels >= 1), 0 = by padded frame size (the default of a new model: TZ-PA2 from 256 x 256 pixels on)."""
        lib().tzo_model_set_contract(self.h, int(contract))
        return self

    def c0(self, hp=None, wp=None):
        out = np.empty((self.hp, self.wp, self.stack[0]), np.float32)
        lib().tzo_model_c0(self.h, _p(out))
        return out

    def next(self, frame, debug=False):
        f = np.ascontiguousarray(frame, dtype=np.float32)
        assert f.shape == (self.hp, self.wp, self.stack[0])
        out = np.empty_like(f)
        if not debug:
            lib().tzo_model_next(self.h, _p(f), _p(out), None)
            return out
        L = self.L
        bufs = []
        for kind in range(3):
            for l in range(L):
                ch = 2 * self.stack[l] if kind == 0 else self.rstack[l]
                bufs.append(np.empty((self.hp >> l, self.wp >> l, ch), np.float32))
        arr = (C.c_void_p * (3 * L))(*[b.ctypes.data for b in bufs])
        lib().tzo_model_next(self.h, _p(f), _p(out), C.cast(arr, C.c_void_p))
        return out, dict(e=bufs[:L], r=bufs[L:2 * L], c=bufs[2 * L:])

    def predict2_literal(self, frame):
        f = np.ascontiguousarray(frame, dtype=np.float32)
        o0, o1 = np.empty_like(f), np.empty_like(f)
        lib().tzo_predict2_literal(self.h, _p(f), _p(o0), _p(o1))
        return o0, o1


def u8_to_f32_frame(key, hp, wp):
    key = np.ascontiguousarray(key, np.uint8)
    out = np.empty((hp, wp, 3), np.float32)
    lib().tzo_u8_to_f32_frame(_p(key), key.shape[0], key.shape[1], hp, wp, _p(out))
    return out


def sse_frame(key_u8, pred_pad):
    key_u8 = np.ascontiguousarray(key_u8, np.uint8)
    pred_pad = np.ascontiguousarray(pred_pad, np.float32)
    return lib().tzo_sse_frame(_p(key_u8), _p(pred_pad), key_u8.shape[0], key_u8.shape[1],
                               pred_pad.shape[0], pred_pad.shape[1])


def delta_frame(pred_pad, orig, zero=False):
    pred_pad = np.ascontiguousarray(pred_pad, np.float32)
    orig = np.ascontiguousarray(orig, np.uint8)
    out = np.empty(orig.shape, np.int16)
    lib().tzo_delta_frame(_p(pred_pad), _p(orig), orig.shape[0], orig.shape[1], pred_pad.shape[0],
                          pred_pad.shape[1], int(zero), _p(out))
    return out


def error_bound_frame(orig_hwc, diff_hwc, mode, value):
    """All 3 channel chains of one HWC frame, in place semantics of compress.py:316-319."""
    orig = np.ascontiguousarray(orig_hwc, np.uint8)
    d = np.ascontiguousarray(diff_hwc, np.int16).copy()
    n = orig.shape[0] * orig.shape[1]
    v1 = float(value[1]) if len(value) > 1 else 0.0
    for c in range(3):
        rc = lib().tzo_error_bound(C.c_void_p(orig.ctypes.data + c), C.c_void_p(d.ctypes.data + 2 * c),
                                   n, 3, MODES[mode], float(value[0]), v1)
        if rc:
            raise ValueError("%s bound must be >= 0 (the reference raises on a negative tolerance)" % mode)
    return d


def conv_probe(x, xu, w, bias, winograd):
    """One 3x3 'same' convolution in the oracle's direct statement (TZ-PA1) or its Winograd F(2x2, 3x3) statement (not used
    by the predictor yet).  x: (H, W, C) or None; xu: (H/2, W/2, Cu) nearest-upsampled source or None; w: HWIO."""
    H, W = (x.shape[:2] if x is not None else (2 * xu.shape[0], 2 * xu.shape[1]))
    C = 0 if x is None else x.shape[2]
    Cu = 0 if xu is None else xu.shape[2]
    w = np.ascontiguousarray(w, np.float32)
    bias = np.ascontiguousarray(bias, np.float32)
    assert w.shape[:3] == (3, 3, C + Cu)
    out = np.empty((H, W, w.shape[3]), np.float32)
    xs = None if x is None else np.ascontiguousarray(x, np.float32)
    xus = None if xu is None else np.ascontiguousarray(xu, np.float32)
    L = lib()
    L.tzo_conv_probe.argtypes = [C_.c_int, C_.c_void_p, C_.c_int, C_.c_void_p, C_.c_int, C_.c_int, C_.c_int, C_.c_void_p,
                                 C_.c_void_p, C_.c_int, C_.c_void_p]
    L.tzo_conv_probe.restype = None
    L.tzo_conv_probe(int(bool(winograd)), None if xs is None else xs.ctypes.data, C, None if xus is None else xus.ctypes.data, Cu,
                     H, W, w.ctypes.data, bias.ctypes.data, w.shape[3], out.ctypes.data)
    return out


def spatial_delta(x, offset):
    x = np.ascontiguousarray(x, np.int16)
    out = np.empty_like(x)
    lib().tzo_spatial_delta(_p(x), x.size, int(offset), _p(out))
    return out


def spatial_undelta(x, offset):
    x = np.ascontiguousarray(x, np.int16)
    out = np.empty_like(x)
    lib().tzo_spatial_undelta(_p(x), x.size, int(offset), _p(out))
    return out


def histogram(y, nbins=2111):
    y = np.ascontiguousarray(y, np.int16)
    h = np.zeros(nbins, np.int64)
    lib().tzo_histogram(_p(y), y.size, _p(h), nbins)
    return h


def lut_apply(x, lut_i16):
    x = np.ascontiguousarray(x, np.int16)
    lut = np.ascontiguousarray(lut_i16, np.int16)
    assert lut.size == 65536
    out = np.empty_like(x)
    lib().tzo_lut_apply(_p(x), x.size, _p(lut), _p(out))
    return out


def reconstruct_frame(pred_pad, key_or_none, diff):
    diff = np.ascontiguousarray(diff, np.int16)
    h, w = diff.shape[:2]
    out = np.empty(diff.shape, np.uint8)
    pp = np.ascontiguousarray(pred_pad, np.float32) if pred_pad is not None else None
    kk = np.ascontiguousarray(key_or_none, np.uint8) if key_or_none is not None else None
    lib().tzo_reconstruct_frame(_p(pp) if pp is not None else None, _p(kk) if kk is not None else None,
                                _p(diff), h, w, pp.shape[1] if pp is not None else w, _p(out))
    return out


def act_probe(x):
    x = np.ascontiguousarray(x, np.float32)
    hs, th = np.empty_like(x), np.empty_like(x)
    lib().tzo_act_probe(_p(x), x.size, _p(hs), _p(th))
    return hs, th


def encode_tail(delta_i16, entropy=True):
    """compress.py:329-373 on an int16 delta stack of any size, through the C functions: spatial delta
    over the whole flattened stack (+ the 1600 offset), bincount, table by count descending with ties
    in ascending symbol order (the stable reverse sort of compress.py:356-359), remap.
    Returns (payload int16[N], table int16[T] | None)."""
    flat = np.ascontiguousarray(delta_i16, np.int16).reshape(-1)
    if not entropy:
        return spatial_delta(flat, 0), None
    y = spatial_delta(flat, 1)
    hist = histogram(y)
    syms = np.nonzero(hist)[0]
    order = sorted(zip(syms.tolist(), hist[syms].tolist()), key=lambda e: e[1], reverse=True)
    table = np.array([s for s, _ in order], dtype=np.int16)
    lut = np.arange(65536, dtype=np.int64) - 32768
    lut[table.astype(np.int64) + 32768] = np.arange(len(table))
    return lut_apply(y, lut.astype(np.int16)), table
