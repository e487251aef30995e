"""ctypes binding of libtezip_hip.so (the C ABI declared in include/tezip_hip.h).

This is the whole Python <-> native boundary: plain pointers and sizes.  Arguments may be
numpy arrays (host memory) or torch CUDA tensors (device memory; only `.data_ptr()` is
used).  There is NO CPU fallback: if the library is missing or no GPU is usable the calls
raise.

Streams: work on device buffers is enqueued on the context's stream and is complete only after
`Context.synchronize()`.  When torch produces or consumes those buffers either create the
context on a torch stream (`s = torch.cuda.Stream(); Context(dev, stream=s.cuda_stream)` and run
the torch side under `torch.cuda.stream(s)`) or synchronise both sides explicitly (bench.py and
tezip_amd.dist.HipEngine do the latter).  torch's DEFAULT stream has the handle 0, which the C ABI
reads as "make your own stream": such a context does not share anything with torch."""
import ctypes as C
import os
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libtezip_hip.so")

TZ_OK = 0
TZ_NBINS = 2111
TZ_MAX_TABLE = 1021
MODES = {"abs": 0, "rel": 1, "absrel": 2, "pwrel": 3}

_SIGS = {
    "tz_version": (C.c_int, []),
    "tz_build_info": (C.c_char_p, []),
    "tz_strerror": (C.c_char_p, [C.c_int]),
    "tz_last_error": (C.c_char_p, [C.c_void_p]),
    "tz_ctx_create": (C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "tz_ctx_destroy": (C.c_int, [C.c_void_p]),
    "tz_ctx_synchronize": (C.c_int, [C.c_void_p]),
    "tz_ctx_stream": (C.c_void_p, [C.c_void_p]),
    "tz_model_load": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tz_model_prepare": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "tz_predict_c0": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tz_predict_next": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "tz_predict_tap": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "tz_set_conv_impl": (C.c_int, [C.c_void_p, C.c_int]),
    "tz_scan_fault_inject": (C.c_int, [C.c_void_p, C.c_uint, C.c_uint]),
    "tz_set_contract": (C.c_int, [C.c_void_p, C.c_int]),
    "tz_get_contract": (C.c_int, [C.c_void_p]),
    "tz_rollout_contract": (C.c_int, [C.c_void_p]),
    "tz_act_probe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tz_rollout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                             C.c_void_p, C.c_void_p]),
    "tz_rollout_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "tz_frames_begin": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "tz_frames_put": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "tz_frames_fence": (C.c_int, [C.c_void_p]),
    "tz_frames_get": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "tz_payload_begin": (C.c_int, [C.c_void_p, C.c_size_t]),
    "tz_payload_put": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    "tz_decoded_get": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "tz_payload_get": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    "tz_set_payload_deferred": (C.c_int, [C.c_void_p, C.c_int]),
    "tz_payload_wait": (C.c_int, [C.c_void_p]),
    "tz_get_predictions": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tz_encode": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                            C.POINTER(C.c_int), C.c_void_p]),
    "tz_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]),
    "tz_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "tz_host_free": (C.c_int, [C.c_void_p]),
    "tz_encode_delta": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p]),
    "tz_encode_begin": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p]),
    "tz_encode_finish": (C.c_int, [C.c_void_p, C.c_int, C.c_int16, C.c_void_p, C.c_int, C.c_void_p]),
    "tz_decode_delta": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "tz_delta_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "tz_error_bound": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_double, C.c_double]),
    "tz_spatial_delta": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int16, C.c_int, C.c_void_p, C.c_void_p]),
    "tz_byte_shuffle": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tz_byte_unshuffle": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "tz_build_table": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]),
    "tz_remap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]),
    "tz_unmap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "tz_spatial_undelta": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int16, C.c_void_p]),
    "tz_reconstruct": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p]),
    "tz_window_sse": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "tz_timer_start": (C.c_int, [C.c_void_p]),
    "tz_timer_stop": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "tz_prof_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "tz_prof_count": (C.c_int, []),
    "tz_prof_name": (C.c_char_p, [C.c_int]),
    "tz_prof_get": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "tz_prof_reset": (C.c_int, [C.c_void_p]),
}
EXPORTS = sorted(_SIGS)

_LIB = None


class TezipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("tezip_hip status %d: %s" % (status, message))
        self.status = status


def load():
    """Load libtezip_hip.so; raises if it has not been built (python -m tezip_amd.build)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `python -m tezip_amd.build` "
                              "(there is no CPU fallback)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _LIB = lib
        if diagnostic_defines(lib) and not os.environ.get("TEZIP_ALLOW_DIAGNOSTIC_BUILD"):
            _LIB = None
            raise ImportError("%s is a MEASUREMENT build (%s; compiled with TEZIP_DEFINES): its kernels are ablated or "
                              "instrumented and must not serve jobs, tests or the bench.  Rebuild with `python -m "
                              "tezip_amd.build` (TEZIP_DEFINES unset), or set TEZIP_ALLOW_DIAGNOSTIC_BUILD=1 in a "
                              "measurement script." % (LIB_PATH, " ".join(diagnostic_defines(lib))))
    return _LIB


def build_info(lib=None):
    return (lib or load()).tz_build_info().decode()


def diagnostic_defines(lib=None):
    """The diagnostic switches the loaded library was compiled with (tz_build_info), [] for a product build."""
    return build_info(lib).split("defines:", 1)[1].split()


def pad8(v):
    return (v + 7) // 8 * 8


def _numel(x):
    return int(x.numel()) if hasattr(x, "numel") else int(np.asarray(x).size)


class _Pinned:
    """Owner of one tz_host_alloc block (freed when the last numpy view goes away)."""

    def __init__(self, lib, ptr, nbytes):
        self.lib, self.ptr = lib, ptr
        self.__array_interface__ = {"data": (ptr, False), "shape": (nbytes,), "typestr": "|u1", "version": 3}

    def __del__(self):
        try:
            self.lib.tz_host_free(C.c_void_p(self.ptr))
        except Exception:
            pass


def pinned_empty(shape, dtype):
    """numpy array in page-locked host memory (tz_host_alloc): the library DMAs such buffers
    directly and overlaps the transfers with the predictor.  Needs a GPU (raises otherwise)."""
    lib = load()
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    p = C.c_void_p()
    rc = lib.tz_host_alloc(max(nbytes, 16), C.byref(p))
    if rc != TZ_OK:
        raise TezipError(rc, lib.tz_strerror(rc).decode() + " (tz_host_alloc)")
    owner = _Pinned(lib, p.value, max(nbytes, 16))
    return np.asarray(owner)[:nbytes].view(dtype).reshape(shape)


def pinned_copy(arr):
    out = pinned_empty(arr.shape, arr.dtype)
    out[...] = arr
    return out


class _ResultPool:
    """Recycled host buffers for the big results (payload, decoded frames).  A fresh np.empty costs a
    page fault per 4 KB while the staging threads fill it (126 MB payload of cfg3: 10 ms, as long as
    everything the GPU does in tz_encode); a block that comes back when the caller drops the array
    has its pages already.  (Page-locked blocks would save 1-2 ms more per call but cost 85 ms per
    500 MB the first time; callers who want that pass their own pinned_empty buffer.)  The MAX_FREE /
    MAX_FREE_BYTES most recently returned blocks are kept.  TEZIP_RESULT_POOL=0 turns the pool off."""
    MIN_BYTES, MAX_FREE, MAX_FREE_BYTES = 1 << 20, 6, 3 << 30

    def __init__(self):
        self.free = []                       # [uint8 base arrays], newest last
        self.lock = threading.RLock()        # (a finaliser may run inside empty())
        self.on = os.environ.get("TEZIP_RESULT_POOL", "1") != "0"

    def empty(self, shape, dtype):
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if not self.on or nbytes < self.MIN_BYTES:
            return np.empty(shape, dtype)
        base = None
        with self.lock:
            best = None
            for i, b in enumerate(self.free):
                if nbytes <= b.size <= 2 * nbytes and (best is None or b.size < self.free[best].size):
                    best = i
            if best is not None:
                base = self.free.pop(best)
        if base is None:
            base = np.empty((nbytes + (1 << 21) - 1) >> 21 << 21, np.uint8)
        return np.asarray(_Pooled(self, base))[:nbytes].view(dtype).reshape(shape)

    def give(self, base):
        with self.lock:
            self.free.append(base)
            while len(self.free) > self.MAX_FREE or sum(b.size for b in self.free) > self.MAX_FREE_BYTES:
                self.free.pop(0)


class _Pooled:
    """Owner of a pooled block: the arrays handed out are views of it; when the last one goes, the block
    returns to the pool."""

    def __init__(self, pool, base):
        self.pool, self.base = pool, base
        self.__array_interface__ = {"data": (base.ctypes.data, False), "shape": (base.size,), "typestr": "|u1", "version": 3}

    def __del__(self):
        try:
            self.pool.give(self.base)
        except Exception:
            pass


_RESULTS = _ResultPool()


def _ptr(x, dtype=None):
    """Pointer of a numpy array (host) or torch tensor (device or host)."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        if dtype is not None and x.dtype != dtype:
            raise TypeError("expected %s, got %s" % (dtype, x.dtype))
        if not x.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        return x.ctypes.data
    if hasattr(x, "data_ptr"):
        if not x.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return x.data_ptr()
    raise TypeError("unsupported buffer type %r" % type(x))


class Context:
    """One context per GPU/process (tz_ctx)."""

    def __init__(self, device=0, stream=None):
        """stream: a hipStream_t handle as an int (a torch.cuda.Stream's .cuda_stream), or None / 0 for a stream of the
        context's own (see the module docstring)."""
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.tz_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != TZ_OK:
            raise TezipError(rc, self.lib.tz_strerror(rc).decode() + " (tz_ctx_create; a MI355X is required)")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.tz_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def _ck(self, rc):
        if rc != TZ_OK:
            msg = self.lib.tz_last_error(self.h).decode() or self.lib.tz_strerror(rc).decode()
            raise TezipError(rc, msg)

    def synchronize(self):
        self._ck(self.lib.tz_ctx_synchronize(self.h))

    # ---- model
    def load_model(self, config, weights):
        ws = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
        shapes = config.weight_shapes()
        if len(ws) != len(shapes) or any(tuple(w.shape) != s for w, (_, s) in zip(ws, shapes)):
            raise ValueError("weight list does not match the model (prednet.py:210-227 order)")
        ptrs = (C.c_void_p * len(ws))(*[w.ctypes.data for w in ws])
        st = np.array(config.stack_sizes, dtype=np.int32)
        rs = np.array(config.R_stack_sizes, dtype=np.int32)
        self._ck(self.lib.tz_model_load(self.h, config.nb_layers, st.ctypes.data, rs.ctypes.data,
                                        C.cast(ptrs, C.c_void_p)))
        self.config = config

    def prepare(self, hp, wp, max_batch=1):
        self._ck(self.lib.tz_model_prepare(self.h, hp, wp, max_batch))
        self.hp, self.wp = hp, wp

    def predict_c0(self):
        out = np.empty((self.hp, self.wp, 3), np.float32)
        self._ck(self.lib.tz_predict_c0(self.h, out.ctypes.data))
        return out

    def predict_next(self, frames, out=None):
        n = frames.shape[0]
        if out is None:
            out = np.empty((n, self.hp, self.wp, 3), np.float32)
        self._ck(self.lib.tz_predict_next(self.h, _ptr(frames, np.float32), n, _ptr(out, np.float32)))
        return out

    def predict_tap(self, kind, level):
        st, rs = self.config.stack_sizes, self.config.R_stack_sizes
        ch = 2 * st[level] if kind == 0 else rs[level]
        out = np.empty((self.hp >> level, self.wp >> level, ch), np.float32)
        self._ck(self.lib.tz_predict_tap(self.h, kind, level, out.ctypes.data))
        return out

    def act_probe(self, x, check_reciprocal=False):
        """Diagnostic: (hard_sigmoid(x), tanh(x)) as the kernels compute them; check_reciprocal adds the count of
        float32 d in [4, 2^27] whose division-free 1 - 2/d differs from the division (must be 0)."""
        x = np.ascontiguousarray(x, np.float32).reshape(-1)
        hs, th = np.empty_like(x), np.empty_like(x)
        bad = C.c_ulonglong(0)
        self._ck(self.lib.tz_act_probe(self.h, x.ctypes.data, x.size, hs.ctypes.data, th.ctypes.data,
                                       C.byref(bad) if check_reciprocal else None))
        return (hs, th, int(bad.value)) if check_reciprocal else (hs, th)

    def set_conv_impl(self, lds_dma, lat=None):
        """Diagnostic: 1 = LDS-DMA convolution kernels where they apply (default), 0 = the general kernel.
        lat: None = k_convlat where the cost model picks it (default), "never", "always"."""
        code = {None: 0, "never": 1, "always": 2}[lat]
        self._ck(self.lib.tz_set_conv_impl(self.h, int(bool(lds_dma)) | (code << 1)))

    def set_contract(self, contract):
        """Arithmetic contract of the predictor: 1 = TZ-PA1 (direct fmaf chains), 2 = TZ-PA2 (Winograd chains on the
        same-resolution sources of levels >= 1).  Encoder and decoder must agree."""
        self._ck(self.lib.tz_set_contract(self.h, int(contract)))

    def get_contract(self):
        return int(self.lib.tz_get_contract(self.h))

    def rollout_contract(self):
        """The contract the resident prediction stack was made under (the stamp tz_rollout / tz_rollout_decode left)."""
        rc = int(self.lib.tz_rollout_contract(self.h))
        if rc < 0:
            self._ck(rc)
        return rc

    def scan_fault_inject(self, epoch_skew=0, poll_limit=0):
        """Diagnostic: make the inverse scan's bounded wait expire (see tz_scan_fault_inject); (0, 0) = normal."""
        self._ck(self.lib.tz_scan_fault_inject(self.h, int(epoch_skew), int(poll_limit)))

    # ---- rollout + encode / decode
    @staticmethod
    def _check_stack(x, what):
        if len(x.shape) != 4 or x.shape[3] != 3:  # compress.py:114: grayscale is expanded to 3 channels first
            raise ValueError("%s must be a (nt, H, W, 3) uint8 stack, got shape %r" % (what, tuple(x.shape)))

    # streaming ingestion / delivery (tz_frames_* / tz_payload_get)
    def frames_begin(self, nt, h, w):
        self._ck(self.lib.tz_frames_begin(self.h, nt, h, w))
        self._staged = (nt, h, w)

    def frames_put(self, first, frames):
        self._check_stack(frames, "frames")
        self._ck(self.lib.tz_frames_put(self.h, int(first), int(frames.shape[0]), _ptr(frames, np.uint8)))

    def frames_fence(self):
        self._ck(self.lib.tz_frames_fence(self.h))

    def frames_get(self, first, count, out=None):
        nt, h, w = self._shape
        if out is None:
            out = np.empty((count, h, w, 3), np.uint8)
        self._ck(self.lib.tz_frames_get(self.h, int(first), int(count), _ptr(out)))
        return out

    def payload_begin(self, count):
        self._ck(self.lib.tz_payload_begin(self.h, int(count)))

    def payload_put(self, offset, piece):
        self._ck(self.lib.tz_payload_put(self.h, int(offset), _numel(piece), _ptr(piece, np.int16)))

    def decoded_get(self, first, count, out=None):
        nt, h, w = self._shape
        if out is None:
            out = np.empty((count, h, w, 3), np.uint8)
        self._ck(self.lib.tz_decoded_get(self.h, int(first), int(count), _ptr(out)))
        return out

    def payload_get(self, offset, count, out=None):
        if out is None:
            out = np.empty(count, np.int16)
        self._ck(self.lib.tz_payload_get(self.h, int(offset), int(count), _ptr(out)))
        return out

    def rollout(self, frames, warm_up, window, threshold=0.0, want_mse=False):
        """frames: (nt,H,W,3) uint8 stack (host or device), or None after frames_begin / frames_put."""
        if frames is None:
            nt, h, w = self._staged
        else:
            self._check_stack(frames, "frames")
            nt, h, w = frames.shape[:3]
        key = np.zeros(nt, np.uint8)
        mse = np.zeros(nt, np.float64) if want_mse else None
        self._ck(self.lib.tz_rollout(self.h, None if frames is None else _ptr(frames, np.uint8), nt, h, w, warm_up, int(window or 0),
                                     float(threshold or 0.0), key.ctypes.data, _ptr(mse)))
        self._shape = (nt, h, w)
        return key.astype(bool), mse

    def rollout_decode(self, key_frames, warm_up):
        """key_frames: the (nt,H,W,3) key stack, or None after frames_begin / frames_put."""
        if key_frames is None:
            nt, h, w = self._staged
        else:
            self._check_stack(key_frames, "key_frames")
            nt, h, w = key_frames.shape[:3]
        key = np.zeros(nt, np.uint8)
        self._ck(self.lib.tz_rollout_decode(self.h, None if key_frames is None else _ptr(key_frames, np.uint8), nt, h, w,
                                            warm_up, key.ctypes.data))
        self._shape = (nt, h, w)
        return key.astype(bool)

    def get_predictions(self, out=None):
        """out: a host or device buffer of nt*Hp*Wp*3 float32 (default: a new numpy array)."""
        nt, h, w = self._shape
        if out is None:
            out = np.empty((nt, pad8(h), pad8(w), 3), np.float32)
        elif _numel(out) != nt * pad8(h) * pad8(w) * 3:
            raise ValueError("prediction buffer holds %d elements, expected %d" % (_numel(out), nt * pad8(h) * pad8(w) * 3))
        self._ck(self.lib.tz_get_predictions(self.h, _ptr(out)))
        return out

    def byte_shuffle(self, x, out=None):
        n = _numel(x)
        if out is None:
            out = np.empty(2 * n, np.uint8)
        self._ck(self.lib.tz_byte_shuffle(self.h, _ptr(x), n, _ptr(out)))
        return out

    def byte_unshuffle(self, planes, out=None):
        n = _numel(planes) // 2
        if out is None:
            out = np.empty(n, np.int16)
        self._ck(self.lib.tz_byte_unshuffle(self.h, _ptr(planes), n, _ptr(out)))
        return out

    def encode(self, mode, bound, entropy=True, payload=None, want_delta=False, shuffle=False, delta_out=None):
        """shuffle=True (not a reference format): `payload` then holds the two byte planes of the
        int16 payload (same buffer size), see tz_byte_shuffle.  want_delta / delta_out (a host or
        device buffer of nt*H*W*3 int16): also return the quantised delta stack."""
        nt, h, w = self._shape
        b0 = float(bound[0])
        b1 = float(bound[1]) if len(bound) > 1 else 0.0
        resident = isinstance(payload, str) and payload == "resident"  # stays in the context: payload_get
        if resident:
            payload = None
        elif payload is None:
            payload = _RESULTS.empty(nt * h * w * 3, np.int16)
        table = np.zeros(TZ_MAX_TABLE, np.int16)
        tlen = C.c_int(0)
        delta = delta_out if delta_out is not None else (np.empty((nt, h, w, 3), np.int16) if want_delta else None)
        if delta is not None and _numel(delta) != nt * h * w * 3:
            raise ValueError("delta buffer holds %d elements, expected %d" % (_numel(delta), nt * h * w * 3))
        self._ck(self.lib.tz_encode(self.h, MODES[mode], b0, b1, int(bool(entropy)) | (2 if shuffle else 0), _ptr(payload), table.ctypes.data,
                                    C.byref(tlen), _ptr(delta)))
        t = table[: tlen.value].copy() if tlen.value >= 0 else None
        return payload, t, delta

    def set_payload_deferred(self, on=True):
        """tz_set_payload_deferred: encode(payload=<pinned host buffer>) returns with the device -> host transfer of
        the payload still running; payload_wait() completes it (it overlaps the next sequence's rollout)."""
        self._ck(self.lib.tz_set_payload_deferred(self.h, int(bool(on))))

    def payload_wait(self):
        self._ck(self.lib.tz_payload_wait(self.h))

    def encode_begin(self, mode, bound, entropy=True):
        """First phase of a window-sharded encode (tz_encode_begin): -> (hist uint64[2111] | None,
        first, last) where hist counts this shard's symbols taken without a carry and first / last
        are the edge elements of its quantised delta stack.  The symbols stay in the context."""
        b1 = float(bound[1]) if len(bound) > 1 else 0.0
        hist = np.zeros(TZ_NBINS, np.uint64) if entropy else None
        edge = np.zeros(2, np.int16)
        self._ck(self.lib.tz_encode_begin(self.h, MODES[mode], float(bound[0]), b1, int(bool(entropy)),
                                          None if hist is None else hist.ctypes.data, edge.ctypes.data))
        return hist, int(edge[0]), int(edge[1])

    def encode_finish(self, carry, table, out=None):
        """Second phase (tz_encode_finish): carry = last delta element of the previous shard (None for
        the first shard), table = the table of the summed histogram (None: no remap).  out: host or
        device buffer, "resident" keeps the payload in the context (payload_get)."""
        nt, h, w = self._shape
        resident = isinstance(out, str) and out == "resident"
        if resident:
            out = None
        elif out is None:
            out = _RESULTS.empty(nt * h * w * 3, np.int16)
        tb = None if table is None else np.ascontiguousarray(table, np.int16)
        self._ck(self.lib.tz_encode_finish(self.h, int(carry is not None), int(carry or 0), _ptr(tb),
                                           -1 if tb is None else len(tb), _ptr(out)))
        return out

    def stream_ptr(self):
        return self.lib.tz_ctx_stream(self.h)

    def encode_delta(self, mode, bound, out=None):
        nt, h, w = self._shape
        if out is None:
            out = np.empty((nt, h, w, 3), np.int16)
        b1 = float(bound[1]) if len(bound) > 1 else 0.0
        self._ck(self.lib.tz_encode_delta(self.h, MODES[mode], float(bound[0]), b1, _ptr(out)))
        return out

    def decode_delta(self, delta, out=None):
        nt, h, w = self._shape
        if _numel(delta) != nt * h * w * 3:  # decompress.py:240: the reference's reshape raises
            raise ValueError("delta stack holds %d elements, expected %d" % (_numel(delta), nt * h * w * 3))
        if out is None:
            out = np.empty((nt, h, w, 3), np.uint8)
        self._ck(self.lib.tz_decode_delta(self.h, _ptr(delta), _ptr(out)))
        return out

    def decode(self, payload, table, out=None):
        """payload None: the pieces staged with payload_begin / payload_put.  out="resident": the
        frames stay in the context (decoded_get)."""
        nt, h, w = self._shape
        n = nt * h * w * 3
        if payload is not None and _numel(payload) != n:  # decompress.py:240: the reference's reshape raises
            raise ValueError("payload holds %d elements, expected %d" % (_numel(payload), n))
        resident = isinstance(out, str) and out == "resident"
        if resident:
            out = None
        elif out is None:
            out = _RESULTS.empty((nt, h, w, 3), np.uint8)
        tl = -1 if table is None else len(table)
        tb = None if table is None else np.ascontiguousarray(table, np.int16)
        self._ck(self.lib.tz_decode(self.h, None if payload is None else _ptr(payload), n, _ptr(tb), tl, _ptr(out)))
        return out

    # ---- operator seams
    def delta_encode(self, pred, orig, zero_mask=None, out=None):
        n, h, w = orig.shape[:3]
        if out is None:
            out = np.empty((n, h, w, 3), np.int16)
        zm = None if zero_mask is None else np.ascontiguousarray(zero_mask, np.uint8)
        self._ck(self.lib.tz_delta_encode(self.h, _ptr(pred), _ptr(orig), _ptr(zm), n, h, w, _ptr(out)))
        return out

    def error_bound(self, orig, diff, mode, bound, skip_mask=None):
        n, h, w = orig.shape[:3]
        b1 = float(bound[1]) if len(bound) > 1 else 0.0
        sk = None if skip_mask is None else np.ascontiguousarray(skip_mask, np.uint8)
        self._ck(self.lib.tz_error_bound(self.h, _ptr(orig), _ptr(diff), _ptr(sk), n, h, w, MODES[mode], float(bound[0]), b1))
        return diff

    def spatial_delta(self, x, offset, carry=None, hist=None, out=None):
        n = int(np.prod(x.shape))
        if out is None:
            out = np.empty(n, np.int16)
        self._ck(self.lib.tz_spatial_delta(self.h, _ptr(x), n, int(carry is not None), int(carry or 0), int(offset),
                                           _ptr(out), _ptr(hist)))
        return out

    def build_table(self, hist):
        hist = np.ascontiguousarray(hist, np.uint64)
        table = np.zeros(TZ_MAX_TABLE, np.int16)
        tlen = C.c_int(0)
        self._ck(self.lib.tz_build_table(hist.ctypes.data, len(hist), table.ctypes.data, C.byref(tlen)))
        return table[: tlen.value].copy()

    def remap(self, x, table, out=None):
        n = int(np.prod(x.shape))
        if out is None:
            out = np.empty(n, np.int16)
        tb = np.ascontiguousarray(table, np.int16)
        self._ck(self.lib.tz_remap(self.h, _ptr(x), n, tb.ctypes.data, len(tb), _ptr(out)))
        return out

    def unmap(self, x, table, offset=True, out=None):
        n = int(np.prod(x.shape))
        if out is None:
            out = np.empty(n, np.int16)
        tb = np.ascontiguousarray(table, np.int16)
        self._ck(self.lib.tz_unmap(self.h, _ptr(x), n, tb.ctypes.data, len(tb), int(offset), _ptr(out)))
        return out

    def spatial_undelta(self, x, carry=None, out=None):
        n = int(np.prod(x.shape))
        if out is None:
            out = np.empty(n, np.int16)
        self._ck(self.lib.tz_spatial_undelta(self.h, _ptr(x), n, int(carry is not None), int(carry or 0), _ptr(out)))
        return out

    def reconstruct(self, pred, key_frames, key_mask, diff, out=None):
        n, h, w = diff.shape[:3]
        if out is None:
            out = np.empty((n, h, w, 3), np.uint8)
        km = None if key_mask is None else np.ascontiguousarray(key_mask, np.uint8)
        self._ck(self.lib.tz_reconstruct(self.h, _ptr(pred), _ptr(key_frames), _ptr(km), _ptr(diff), n, h, w, _ptr(out)))
        return out

    def window_sse(self, orig, pred):
        n, h, w = orig.shape[:3]
        sse = np.zeros(n, np.float64)
        self._ck(self.lib.tz_window_sse(self.h, _ptr(orig), _ptr(pred), n, h, w, sse.ctypes.data))
        return sse

    # ---- timing
    def timer_start(self):
        self._ck(self.lib.tz_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float(0)
        self._ck(self.lib.tz_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def prof_enable(self, on=True):
        self._ck(self.lib.tz_prof_enable(self.h, int(on)))

    def prof_reset(self):
        self._ck(self.lib.tz_prof_reset(self.h))

    def prof_get(self):
        out = {}
        for i in range(self.lib.tz_prof_count()):
            ms, n = C.c_double(0), C.c_longlong(0)
            self._ck(self.lib.tz_prof_get(self.h, i, C.byref(ms), C.byref(n)))
            out[self.lib.tz_prof_name(i).decode()] = (ms.value, n.value)
        return out
