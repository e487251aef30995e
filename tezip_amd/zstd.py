"""zstd one-shot compress/decompress through ctypes on libzstd.

The reference uses the PyPI `zstd` module: `zstd.compress(data, 9)` / `zstd.decompress(data)`
(/root/reference/src/compress.py:276,398; decompress.py:89,98), i.e. one standard zstd frame
with the content size in the header.  ZSTD_compress(level 9) produces the same kind of frame;
byte sizes can differ slightly between libzstd versions (the reference pins 1.4.5), so
ratios are only compared between runs that use the same library.

`compress_array(..., threads=n)` uses libzstd's own job-parallel compressor
(ZSTD_c_nbWorkers): still ONE standard frame with the content size in its header -- exactly what
the reference's `zstd.decompress` reads -- but level 9 of a 120 MB entropy stream takes a fraction
of the 1.8 s the single thread needs.  It needs a libzstd built with
multithreading: the loader prefers one that has it (this image: /opt/conda/lib, 1.4.9; the system
1.4.8 is single-threaded) and silently compresses on one thread otherwise."""
import ctypes as C
import os

_L = None
_MT = False
ZSTD_c_compressionLevel, ZSTD_c_nbWorkers, ZSTD_c_jobSize = 100, 400, 401


def _job_size(total_bytes, threads):
    """Bytes per job of libzstd's multi-threaded compressor, 0 = its default (4 windows: 8 MB at level 9 for
    this payload size, i.e. 16 jobs for cfg3's 126 MB).  Smaller jobs buy nothing on 16 cores and cost size
    (2 MB: +0.75 %, 1 MB: +1.5 % against the single-threaded frame; default: +0.19 %;
    `scripts/zstd_jobs.py`).  TEZIP_ZSTD_JOB_MB overrides (diagnostic)."""
    mb = os.environ.get("TEZIP_ZSTD_JOB_MB")
    return (int(mb) << 20) if mb else 0


class _InBuffer(C.Structure):
    _fields_ = [("src", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]


class _OutBuffer(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]


def _bind(L):
    L.ZSTD_compressBound.restype = C.c_size_t
    L.ZSTD_compressBound.argtypes = [C.c_size_t]
    L.ZSTD_compress.restype = C.c_size_t
    L.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    L.ZSTD_decompress.restype = C.c_size_t
    L.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.ZSTD_getFrameContentSize.restype = C.c_ulonglong
    L.ZSTD_getFrameContentSize.argtypes = [C.c_void_p, C.c_size_t]
    L.ZSTD_isError.restype = C.c_uint
    L.ZSTD_isError.argtypes = [C.c_size_t]
    L.ZSTD_getErrorName.restype = C.c_char_p
    L.ZSTD_getErrorName.argtypes = [C.c_size_t]
    L.ZSTD_versionNumber.restype = C.c_uint
    L.ZSTD_createCCtx.restype = C.c_void_p
    L.ZSTD_freeCCtx.argtypes = [C.c_void_p]
    L.ZSTD_CCtx_setParameter.restype = C.c_size_t
    L.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.ZSTD_compress2.restype = C.c_size_t
    L.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.ZSTD_CCtx_setPledgedSrcSize.restype = C.c_size_t
    L.ZSTD_CCtx_setPledgedSrcSize.argtypes = [C.c_void_p, C.c_ulonglong]
    L.ZSTD_compressStream2.restype = C.c_size_t
    L.ZSTD_compressStream2.argtypes = [C.c_void_p, C.POINTER(_OutBuffer), C.POINTER(_InBuffer), C.c_int]
    L.ZSTD_CStreamOutSize.restype = C.c_size_t
    L.ZSTD_createDCtx.restype = C.c_void_p
    L.ZSTD_freeDCtx.argtypes = [C.c_void_p]
    L.ZSTD_decompressStream.restype = C.c_size_t
    L.ZSTD_decompressStream.argtypes = [C.c_void_p, C.POINTER(_OutBuffer), C.POINTER(_InBuffer)]


def _has_workers(L):
    cctx = L.ZSTD_createCCtx()
    try:
        return not L.ZSTD_isError(L.ZSTD_CCtx_setParameter(cctx, ZSTD_c_nbWorkers, 2))
    finally:
        L.ZSTD_freeCCtx(cctx)


def _lib():
    global _L, _MT
    if _L is None:
        last, first = None, None
        names = [os.environ["TEZIP_LIBZSTD"]] if os.environ.get("TEZIP_LIBZSTD") else \
            ["libzstd.so.1", "libzstd.so", "/opt/conda/lib/libzstd.so.1", "/opt/conda/lib/libzstd.so"]
        for name in names:
            try:
                # RTLD_DEEPBIND: a second libzstd may already be in the process (rocprofv3, other
                # extensions); without it this library's internal calls bind to THAT one's symbols
                L = C.CDLL(name, mode=os.RTLD_LOCAL | getattr(os, "RTLD_DEEPBIND", 0))
                _bind(L)
            except (OSError, AttributeError) as e:
                last = e
                continue
            if first is None:
                first = L
            if _has_workers(L):
                _L, _MT = L, True
                break
        if _L is None:
            _L = first
        if _L is None:
            raise ImportError("libzstd not found: %s" % last)
    return _L


def multithreaded():
    """True when the loaded libzstd can compress one frame on several threads."""
    _lib()
    return _MT


def version():
    return _lib().ZSTD_versionNumber()


def compress(data, level=3):
    L = _lib()
    data = bytes(data) if not isinstance(data, (bytes, bytearray, memoryview)) else data
    src = (C.c_char * len(data)).from_buffer_copy(data) if len(data) else None
    cap = L.ZSTD_compressBound(len(data))
    dst = C.create_string_buffer(cap)
    n = L.ZSTD_compress(dst, cap, src, len(data), int(level))
    if L.ZSTD_isError(n):
        raise RuntimeError("zstd: " + L.ZSTD_getErrorName(n).decode())
    return dst.raw[:n]


def default_threads():
    """Worker threads compress.run uses: TEZIP_ZSTD_THREADS, else the CPUs of this process (<= 16)."""
    if os.environ.get("TEZIP_ZSTD_THREADS"):
        return max(0, int(os.environ["TEZIP_ZSTD_THREADS"]))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return min(n, 16)


def compress_array(arr, level=9, threads=0):
    """Compress a C-contiguous numpy array into one zstd frame without an intermediate bytes copy.
    threads > 1: libzstd job-parallel compression (same frame format; bytes differ from the
    single-threaded output, sizes within a fraction of a percent)."""
    L = _lib()
    n_in = arr.nbytes
    cap = L.ZSTD_compressBound(n_in)
    dst = C.create_string_buffer(cap)
    if threads > 1 and _MT and n_in > (1 << 22):
        cctx = L.ZSTD_createCCtx()
        try:
            L.ZSTD_CCtx_setParameter(cctx, ZSTD_c_compressionLevel, int(level))
            L.ZSTD_CCtx_setParameter(cctx, ZSTD_c_nbWorkers, int(threads))
            if _job_size(n_in, threads):
                L.ZSTD_CCtx_setParameter(cctx, ZSTD_c_jobSize, _job_size(n_in, threads))
            n = L.ZSTD_compress2(cctx, dst, cap, arr.ctypes.data, n_in)
        finally:
            L.ZSTD_freeCCtx(cctx)
    else:
        n = L.ZSTD_compress(dst, cap, arr.ctypes.data, n_in, int(level))
    if L.ZSTD_isError(n):
        raise RuntimeError("zstd: " + L.ZSTD_getErrorName(n).decode())
    return dst.raw[:n]


def decompress(data):
    L = _lib()
    data = bytes(data)
    size = L.ZSTD_getFrameContentSize(data, len(data))
    if size in (2 ** 64 - 1, 2 ** 64 - 2):
        raise RuntimeError("zstd: frame without a content size (not written by zstd.compress)")
    dst = C.create_string_buffer(max(int(size), 1))
    n = L.ZSTD_decompress(dst, int(size), data, len(data))
    if L.ZSTD_isError(n):
        raise RuntimeError("zstd: " + L.ZSTD_getErrorName(n).decode())
    return dst.raw[:n]


class StreamCompressor:
    """One standard zstd frame written piece by piece (ZSTD_compressStream2): the input never has
    to exist as a whole in memory, and with `threads` > 1 libzstd compresses the pieces on its own
    worker threads while the caller fetches the next one.  The total size is pledged up front so the
    frame header carries the content size -- the reference's `zstd.decompress` needs it
    (decompress.py:89,98)."""

    def __init__(self, fileobj, total_bytes, level=9, threads=0):
        self.L, self.f = _lib(), fileobj
        self.cctx = self.L.ZSTD_createCCtx()
        self.L.ZSTD_CCtx_setParameter(self.cctx, ZSTD_c_compressionLevel, int(level))
        if threads > 1 and _MT:
            self.L.ZSTD_CCtx_setParameter(self.cctx, ZSTD_c_nbWorkers, int(threads))
            if _job_size(total_bytes, threads):
                self.L.ZSTD_CCtx_setParameter(self.cctx, ZSTD_c_jobSize, _job_size(total_bytes, threads))
        self._ck(self.L.ZSTD_CCtx_setPledgedSrcSize(self.cctx, int(total_bytes)))
        self.cap = max(int(self.L.ZSTD_CStreamOutSize()), 1 << 20)
        self.out = C.create_string_buffer(self.cap)
        self.written = 0
        self.fed = 0

    def _ck(self, n):
        if self.L.ZSTD_isError(n):
            raise RuntimeError("zstd: " + self.L.ZSTD_getErrorName(n).decode())
        return n

    def _pump(self, ptr, size, directive):
        ib = _InBuffer(ptr, size, 0)
        while True:
            ob = _OutBuffer(C.addressof(self.out), self.cap, 0)
            left = self._ck(self.L.ZSTD_compressStream2(self.cctx, C.byref(ob), C.byref(ib), directive))
            if ob.pos:
                self.f.write(self.out.raw[:ob.pos] if ob.pos < self.cap else self.out.raw)
                self.written += ob.pos
            if directive == 0 and ib.pos == ib.size:
                return
            if directive != 0 and left == 0:
                return

    def write(self, arr):
        """arr: C-contiguous numpy array (its memory is read in place)."""
        self.fed += arr.nbytes
        if arr.nbytes:
            self._pump(arr.ctypes.data, arr.nbytes, 0)  # ZSTD_e_continue

    def close(self):
        if self.cctx:
            try:
                self._pump(None, 0, 2)                     # ZSTD_e_end
            finally:
                self.L.ZSTD_freeCCtx(self.cctx)
                self.cctx = None
        return self.written


def content_size(head):
    """Decompressed size announced by the frame header (`head`: at least the first 18 bytes)."""
    L = _lib()
    size = L.ZSTD_getFrameContentSize(bytes(head), len(head))
    if size in (2 ** 64 - 1, 2 ** 64 - 2):
        raise RuntimeError("zstd: frame without a content size (not written by zstd.compress)")
    return int(size)


def stream_decompress(fileobj, piece_bytes=16 << 20, read_bytes=4 << 20):
    """Generator over the decompressed content of ONE zstd frame read from `fileobj`: yields
    (content_size, uint8 numpy piece) with pieces of `piece_bytes` (the last one shorter); the
    yielded array is reused by the next iteration.  Raises like decompress() when the frame has
    no content size (the reference's frames always carry it)."""
    import numpy as np
    L = _lib()
    head = fileobj.read(read_bytes)
    size = L.ZSTD_getFrameContentSize(head, len(head))
    if size in (2 ** 64 - 1, 2 ** 64 - 2):
        raise RuntimeError("zstd: frame without a content size (not written by zstd.compress)")
    dctx = L.ZSTD_createDCtx()
    out = np.empty(max(1, min(piece_bytes, int(size))), np.uint8)
    try:
        src = head
        produced = 0
        ob = _OutBuffer(out.ctypes.data, out.size, 0)
        eof = False
        while produced < size:
            buf = C.create_string_buffer(src, len(src)) if src else C.create_string_buffer(1)
            ib = _InBuffer(C.addressof(buf), len(src), 0)
            progressed = False
            while produced + ob.pos < size:
                before = (ib.pos, ob.pos)
                n = L.ZSTD_decompressStream(dctx, C.byref(ob), C.byref(ib))
                if L.ZSTD_isError(n):
                    raise RuntimeError("zstd: " + L.ZSTD_getErrorName(n).decode())
                progressed = progressed or (ib.pos, ob.pos) != before
                if ob.pos == ob.size:
                    produced += ob.pos
                    yield int(size), out[:ob.pos]
                    ob = _OutBuffer(out.ctypes.data, out.size, 0)
                elif ib.pos == ib.size:
                    break  # needs more input (or only flushes what it still holds)
            if produced + ob.pos >= size:
                break
            if eof and not progressed:
                raise RuntimeError("zstd: truncated frame")
            src = fileobj.read(read_bytes)
            eof = not src
        if ob.pos:
            produced += ob.pos
            yield int(size), out[:ob.pos]
    finally:
        L.ZSTD_freeDCtx(dctx)
