"""CPU ORACLE (test infrastructure, NOT product code).

A numpy restatement of the TEZip hot path -- padding, rollout state machine, delta,
error-bound quantiser, spatial delta, rank remap, trailer, and every inverse -- written
from the behaviour of the reference at /root/reference/src (file:line cited per function).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; tezip_amd/ never does (the product path fails loudly without its HIP library).

Pinning: every function here is checked in tests/test_oracle_golden.py against
tests/golden/ref_helpers.npz (outputs of the reference's own helpers) and
tests/golden/ref_runs.npz (outputs of the reference's compress.run / decompress.run driven
by tests/golden/fake_predictor.py), plus the doc KATs of SURVEY.md §4.2.
The PredNet arithmetic itself (Keras 2.2.4 / TF 1.15, not under /root/reference) is
"parity unpinned": see oracle/prednet_np.py and DESIGN.md.

Pure-Python loops are used where the reference is sequential (error_bound, decoder
scan): keep inputs small, or use the C restatement (oracle/tz_oracle.c via
oracle/coracle.py) for big ones.
"""
import numpy as np

OFFSET = 1600  # compress.py:348


# --------------------------------------------------------------------------- a1 padding
def padding_size(num):
    """data_utils.py:103-107: round up to a multiple of 8."""
    return num if num % 8 == 0 else (int(num / 8) + 1) * 8


def padding_shape(height, width):
    """data_utils.py:94-100."""
    return (padding_size(height), padding_size(width))


def data_padding(x):
    """data_utils.py:77-91: zero-pad bottom/right, result is float64."""
    hp, wp = padding_shape(x.shape[2], x.shape[3])
    out = np.zeros((x.shape[0], x.shape[1], hp, wp, x.shape[4]))
    out[:, :, : x.shape[2], : x.shape[3]] = x
    return out


# ----------------------------------------------------------------------- a8 delta (A.2)
def delta_group(pred_pad_f32, orig_u8, zero_first=True):
    """compress.py:294-314 for one group.

    pred_pad_f32: (k, Hp, Wp, 3) float32, orig_u8: (k, H, W, 3) uint8.
    int(pred * 255.0) uses a FLOAT32 multiply (numpy keeps float32 for array*pyfloat),
    then truncation toward zero; orig/255*255 -> int is the identity on uint8.
    """
    k, h, w, _ = orig_u8.shape
    crop = np.asarray(pred_pad_f32, dtype=np.float32)[:, :h, :w]
    x_hat = (crop * np.float32(255.0)).astype(np.int64)
    d = x_hat - orig_u8.astype(np.int64)
    if zero_first:
        d[0] = 0
    return d


# ------------------------------------------------------------------ a9 quantiser (A.3)
def error_bound(orig, diff, mode, value):
    """compress.py:23-70 restated; returns an int64 array like `diff`.

    orig/diff: integer arrays of one (frame, channel) slab, any shape (row-major chain).
    The reference assigns float medians into an int64 array (compress.py:61,67 via 319):
    truncation toward zero is part of the contract.
    """
    diff = np.asarray(diff)
    if value[0] == 0:
        return diff.astype(np.int64)
    bf = np.asarray(orig).reshape(-1).astype(np.int64)
    df = diff.reshape(-1).astype(np.int64)
    if (mode == "rel" and value[0] < 0) or (mode == "absrel" and value[1] < 0):
        # a negative tolerance: the reference assigns (inf + -inf)/2 = NaN into its int array at the first
        # element (compress.py:60-61) and raises; rejected up front like pwrel below
        raise ValueError("%s bound must be >= 0" % mode)
    if mode == "abs":
        e = np.full(df.shape, float(abs(value[0])))
    elif mode == "rel":
        e = np.full(df.shape, float(bf.max() - bf.min()) * float(value[0]))
    elif mode == "absrel":
        if value[1] == 0:
            return diff.astype(np.int64)
        a = float(abs(value[0]))
        r = float(bf.max() - bf.min()) * float(value[1])
        e = np.full(df.shape, a if a < r else r)
    elif mode == "pwrel":
        e = bf.astype(np.float64) * float(value[0])
        if (e < 0).any():
            # the reference raises here (NaN assigned into an int slice); we reject up front
            raise ValueError("pwrel bound must be >= 0")
    else:
        raise ValueError("unknown mode %r" % (mode,))
    du = df.astype(np.float64) + e
    dl = df.astype(np.float64) - e
    out = df.copy()
    u, l, head = np.inf, -np.inf, 0
    n = len(df)
    for i in range(n):
        if min(u, du[i]) - max(l, dl[i]) < 0.0:
            out[head:i] = int((u + l) / 2)  # int(): truncation toward zero, as the int64 store
            u, l, head = np.inf, -np.inf, i
        if du[i] < u:
            u = du[i]
        if l < dl[i]:
            l = dl[i]
    if n:
        out[head:n] = int((u + l) / 2)
    return out.reshape(diff.shape)


# ------------------------------------------------------------ a11 / a19 spatial delta
def finding_difference_enc(arr):
    """compress.py:73-77: out[0]=in[0], out[i]=in[i-1]-in[i] over the flattened array
    (int16 wrap-around for int16 input)."""
    a = np.asarray(arr)
    f = a.reshape(-1).copy()
    with np.errstate(over="ignore"):
        f[1:] = f[:-1].copy() - a.reshape(-1)[1:]
    return f.reshape(a.shape)


def finding_difference_dec(arr):
    """decompress.py:22-29: in[i] = in[i-1] - out[i], sequential; as a wrap-around scan:
    in[i] = out[0] - sum(out[1..i]) (mod 2^16 for int16)."""
    a = np.asarray(arr)
    f = a.reshape(-1)
    if f.size == 0:
        return a.copy()
    if a.dtype == np.int16:
        acc = np.cumsum(f[1:].astype(np.int64))
        res = np.empty(f.shape, dtype=np.int64)
        res[0] = f[0]
        res[1:] = int(f[0]) - acc
        res = ((res + 32768) % 65536 - 32768).astype(np.int16)
    else:
        res = np.empty_like(f)
        res[0] = f[0]
        res[1:] = f[0] - np.cumsum(f[1:])
    return res.reshape(a.shape)


# --------------------------------------------------------------- a12 / a13 / a18 remap
def build_table(y):
    """compress.py:352-361: symbols with count>0, by count descending; the stable sort
    with reverse=True keeps ascending symbol order among equal counts."""
    counts = np.bincount(np.asarray(y).reshape(-1).astype(np.int64))
    syms = np.nonzero(counts)[0]
    order = sorted(zip(syms.tolist(), counts[syms].tolist()), key=lambda e: e[1], reverse=True)
    return np.array([s for s, _ in order], dtype=np.int16)


def remap_enc(y, table):
    """compress.py:84-90 as one LUT pass.  Equivalent to the T sequential `where` passes
    because ranks (<1021) never collide with the remaining symbols (>=1090)."""
    y = np.asarray(y)
    lut = np.arange(65536, dtype=np.int64) - 32768  # identity on int16 values
    for idx, sym in enumerate(np.asarray(table).tolist()):
        lut[sym + 32768] = idx
    return lut[y.astype(np.int64) + 32768].astype(np.int16)


def unmap_lut(table):
    """decompress.py:31-36 as a LUT that reproduces the sequential-pass semantics exactly,
    including chained substitutions for (non-reference) tables whose symbols are < T."""
    t = [int(v) for v in np.asarray(table).tolist()]
    n = len(t)
    lut = np.arange(65536, dtype=np.int64) - 32768
    for v in range(n):
        cur, last = v, -1
        while 0 <= cur < n and cur > last:
            last, cur = cur, t[cur]
        lut[v + 32768] = cur
    return lut


def remap_dec(ranks, table):
    lut = unmap_lut(table)
    return lut[np.asarray(ranks).astype(np.int64) + 32768].astype(np.int16)


# ------------------------------------------------------------- a6 / a7 rollout (A.1)
class FnPredictor:
    """Adapter: c0(hp, wp) -> (Hp,Wp,3) f32 ; next(frame (Hp,Wp,3)) -> (Hp,Wp,3) f32."""

    def __init__(self, c0_fn, next_fn):
        self._c0, self._next = c0_fn, next_fn

    def c0(self, hp, wp):
        return self._c0(hp, wp)

    def next(self, frame):
        return self._next(frame)


def check_lengths(nt, p):
    """The reference misbehaves for nt <= p+1; we reject."""
    if p < 0 or nt < p + 2:
        raise ValueError("need at least warm_up+2 frames (nt=%d, warm_up=%d)" % (nt, p))


def rollout(frames_u8, p, window, threshold, predictor):
    """compress.py:183-268 restated.

    frames_u8: (nt, H, W, 3).  Returns dict(groups=[(start, [pred,...])...], key=bool[nt],
    mse=[...], x_pad=float64 (nt,Hp,Wp,3)).  A group covers frames start..start+len-1 and
    slot 0 of every group is a placeholder prediction (C0, or the last prediction in the
    nt-1 special case of compress.py:260-262).
    """
    nt, h, w, _ = frames_u8.shape
    check_lengths(nt, p)
    x = frames_u8.astype(np.float32) / np.float32(255)  # compress.py:138 (float32 divide)
    x_pad = data_padding(x[None])[0]  # float64 holding float32 values
    hp, wp = x_pad.shape[1:3]
    c0 = np.asarray(predictor.c0(hp, wp), dtype=np.float32)
    key = np.zeros(nt, dtype=bool)
    groups, mses = [], []
    cur = None
    if p > 0:
        key[:p] = True
        groups.append((0, [c0] * p))
        cur = (p, [c0])
    key_idx = idx = p + 1
    while idx < nt:
        if idx == key_idx:
            inp = x_pad[idx - 1]
            key[idx - 1] = True
        else:
            inp = cur[1][-1]
        pred = np.asarray(predictor.next(inp), dtype=np.float32)
        if idx == 1:
            cur = (0, [c0, pred])
        else:
            cur[1].append(pred)
        stack = np.stack(cur[1][1:]).astype(np.float64)
        stop = float(np.mean((x_pad[key_idx: idx + 1] - stack) ** 2))
        mses.append(stop)
        if (threshold is not None and stop > threshold) or (window is not None and (idx - p) % window == 0):
            groups.append((cur[0], cur[1][:-1]))
            cur = (idx, [c0])
            if idx == nt - 1:
                key[idx] = True
                cur = (idx, [pred])
            key_idx = idx + 1
        idx += 1
    groups.append(cur)
    return dict(groups=groups, key=key, mse=mses, x_pad=x_pad, c0=c0)


# -------------------------------------------------- a8-a14 encoder back half (3.1 G-J)
def encode_stream(frames_u8, ro, p, mode, bound, entropy=True):
    """compress.py:289-395 restated: groups -> pre-zstd int16 stream (with trailer)."""
    nt, h, w, _ = frames_u8.shape
    parts = []
    for g, (start, preds) in enumerate(ro["groups"]):
        k = len(preds)
        orig = frames_u8[start: start + k]
        d = delta_group(np.stack(preds), orig)
        if not (p != 0 and g == 0):
            oi = orig.astype(np.int64)
            for j in range(1, k):
                for c in range(3):
                    d[j, :, :, c] = error_bound(oi[j, :, :, c], d[j, :, :, c], mode, bound)
        parts.append(d)
    dm = np.concatenate(parts, axis=0).astype(np.int16)
    assert dm.shape[0] == nt
    sd = finding_difference_enc(dm).reshape(-1)
    if entropy:
        y = (OFFSET - sd.astype(np.int64)).astype(np.int16)
        table = build_table(y)
        payload = remap_enc(y, table)
    else:
        table = None
        payload = sd
    stream = build_stream(payload, table, nt, h, w, p)
    return dict(stream=stream, delta=dm, sd=sd, table=table)


def build_stream(payload, table, nt, h, w, p):
    """compress.py:375-395: payload | table | len(table)  (or payload | -1), then the shape
    (1, nt, H, W, 3) and PREPROCESS, everything as int16."""
    if table is not None:
        tail = np.concatenate([np.asarray(table).astype(np.int64), [len(table)]])
    else:
        tail = np.array([-1], dtype=np.int64)
    trailer = np.concatenate([tail, [1, nt, h, w, 3], [p]]).astype(np.int16)
    return np.concatenate([np.asarray(payload, dtype=np.int16).reshape(-1), trailer])


def key_frame_stream(frames_u8, key):
    """compress.py:183,190,220,261,271-273: uint8 stack, zero except at key frames."""
    kf = np.zeros_like(frames_u8)
    kf[key] = frames_u8[key]
    return kf.reshape(-1)


def filename_txt(names, is_rgb):
    """compress.py:133-136."""
    return "%d\n" % int(is_rgb) + "".join("%s\n" % n for n in names)


def compress_oracle(frames_u8, p, window, threshold, mode, bound, predictor, entropy=True):
    ro = rollout(frames_u8, p, window, threshold, predictor)
    enc = encode_stream(frames_u8, ro, p, mode, bound, entropy)
    enc["key_frame"] = key_frame_stream(frames_u8, ro["key"])
    enc["key"] = ro["key"]
    enc["mse"] = ro["mse"]
    enc["rollout"] = ro
    return enc


# ------------------------------------------------------------------- a16-a20 decoder
def parse_stream(stream):
    """decompress.py:105-113, 203-221: trailer -> (payload, table|None, shape, warm_up)."""
    s = np.asarray(stream, dtype=np.int16)
    warm_up = int(s[-1])
    shape = tuple(int(v) for v in s[-6:-1])
    tlen = int(s[-7])
    if tlen == -1:
        return s[:-7], None, shape, warm_up
    return s[: -7 - tlen], s[-7 - tlen: -7], shape, warm_up


def key_frame_check(x_pad):
    """decompress.py:123-129: indices of frames with any non-zero sample, plus nt."""
    nt = x_pad.shape[0]
    return [i for i in range(nt) if not np.all(x_pad[i] == 0)] + [nt]


def decoder_rollout(key_u8, warm_up, predictor):
    """decompress.py:115-186: -> float64 (nt,Hp,Wp,3) prediction stack with key frames in
    their own slots (key/255 in float64) and frame 0 forced to the key frame."""
    nt = key_u8.shape[0]
    x = key_u8 / 255  # float64 divide (decompress.py:117)
    x_pad = data_padding(x[None])[0]
    hp, wp = x_pad.shape[1:3]
    kfc = key_frame_check(x_pad)
    c0 = np.asarray(predictor.c0(hp, wp), dtype=np.float32)
    res = [c0.astype(np.float64) for _ in range(warm_up)]
    for idx in range(warm_up, len(kfc[warm_up:]) + warm_up - 1):
        for pi in range(kfc[idx], kfc[idx + 1]):
            if pi == kfc[idx]:
                res.append(x_pad[pi])
            elif pi == kfc[idx] + 1:
                last = np.asarray(predictor.next(x_pad[pi - 1].astype(np.float32)), dtype=np.float32)
                res.append(last.astype(np.float64))
            else:
                last = np.asarray(predictor.next(last), dtype=np.float32)
                res.append(last.astype(np.float64))
    out = np.stack(res)
    out[0] = x_pad[0]
    return out, kfc


def decode_stream(stream, key_bytes, predictor):
    """decompress.py:87-256 restated: -> uint8 (nt,H,W,3)."""
    payload, table, shape, warm_up = parse_stream(stream)
    _, nt, h, w, c = shape
    key_u8 = np.asarray(key_bytes, dtype=np.uint8).reshape(nt, h, w, c)
    x_hat, _ = decoder_rollout(key_u8, warm_up, predictor)
    if x_hat.shape[0] != nt:
        raise ValueError("key frames do not cover the sequence")
    x_hat = x_hat[:, :h, :w]
    if table is not None:
        sym = remap_dec(payload, table)
        sd = (OFFSET - sym.astype(np.int64)).astype(np.int16)
    else:
        sd = payload
    diff = finding_difference_dec(sd.reshape(nt, h, w, c))
    return reconstruct(x_hat, diff)


def reconstruct(x_hat_f64, diff_i16):
    """decompress.py:252-256,269: pred*255 (FLOAT64) - diff, clip to [0,255], truncate.

    The encoder truncated a FLOAT32 product (compress.py:307,311).  The two agree:
    255 = 2^8-1 and 2^8 = 1 (mod 255), so a float32 product p*255 can never round up onto
    an integer, i.e. trunc(f32(p*255)) == floor(p*255) for every float32 p >= 0
    (tests/test_oracle_golden.py::test_f32_product_truncation_is_exact).  Hence this
    float64 formula equals the integer form `reconstruct_int` below."""
    rec = np.asarray(x_hat_f64, dtype=np.float64) * 255 - diff_i16
    rec = np.where(rec > 255, 255, rec)
    rec = np.where(rec < 0, 0, rec)
    return rec.astype(np.uint8)


def reconstruct_int(base_int, diff_i16):
    """Integer form of `reconstruct`: base = trunc(f32(pred*255)) at predicted slots and
    the key byte at key slots; result = clamp(base - diff, 0, 255).  This is what the HIP
    kernel computes."""
    return np.clip(np.asarray(base_int, dtype=np.int64) - np.asarray(diff_i16, dtype=np.int64), 0, 255).astype(np.uint8)
