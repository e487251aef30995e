"""Minimal read-only HDF5 reader: just enough to pull float32 weight arrays out of the
Keras-2.2.4 / h5py-2.10 checkpoints the reference writes (train.py:109 `ModelCheckpoint` ->
full-model file with /model_weights/<layer>/<weight name>; `save_weights` files have the layer
groups at the root).  h5py is not part of this image, and the format subset these files use is
small: superblock v0/v1, old-style groups (B-tree v1 + local heap + symbol-table nodes), object
headers v1 (with continuation blocks), contiguous or compact datasets of little-endian IEEE
floats.  Anything else (chunking, compression, new-style groups) raises NotImplementedError.
"""
import struct

import numpy as np

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5File:
    def __init__(self, path):
        import mmap
        with open(path, "rb") as f:
            try:  # datasets of training sets are GBs: map the file instead of reading it
                self.b = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
            except (ValueError, OSError):
                self.b = f.read()
        off = 0
        while self.b[off:off + 8] != SIG:  # the superblock may sit at 0, 512, 1024, ...
            off = 512 if off == 0 else off * 2
            if off + 8 > len(self.b):
                raise ValueError("not an HDF5 file")
        ver = self.b[off + 8]
        if ver not in (0, 1):
            raise NotImplementedError("HDF5 superblock version %d (only 0/1, as written by h5py's default)" % ver)
        self.O, self.L = self.b[off + 13], self.b[off + 14]
        if (self.O, self.L) != (8, 8):
            raise NotImplementedError("offset/length sizes %d/%d" % (self.O, self.L))
        p = off + 24 + (4 if ver == 1 else 0)
        self.base = self._u(p, 8)
        root_entry = p + 32  # base, free-space, eof, driver addresses
        self.root = self._sym_entry(root_entry)

    # ---- primitives
    def _u(self, p, n):
        return int.from_bytes(self.b[p:p + n], "little")

    def _sym_entry(self, p):
        name_off, hdr, cache = self._u(p, 8), self._u(p + 8, 8), self._u(p + 16, 4)
        btree = heap = None
        if cache == 1:
            btree, heap = self._u(p + 24, 8), self._u(p + 32, 8)
        return dict(name_off=name_off, hdr=hdr, btree=btree, heap=heap)

    def _messages(self, hdr):
        """Yield (type, payload offset, size) of an object header v1, following continuations."""
        p = hdr + self.base
        if self.b[p] != 1:
            raise NotImplementedError("object header version %d (new-style file; re-save with libver='earliest')" % self.b[p])
        nmsg, size = self._u(p + 2, 2), self._u(p + 8, 4)
        blocks = [(p + 16, size)]
        seen = 0
        while blocks and seen < nmsg:
            q, left = blocks.pop(0)
            end = q + left
            while q + 8 <= end and seen < nmsg:
                mtype, msize = self._u(q, 2), self._u(q + 2, 2)
                data = q + 8
                if mtype == 0x10:  # continuation
                    blocks.append((self._u(data, 8) + self.base, self._u(data + 8, 8)))
                yield mtype, data, msize
                seen += 1
                q = data + ((msize + 7) & ~7)

    # ---- groups
    def _heap_name(self, heap, off):
        p = heap + self.base
        if self.b[p:p + 4] != b"HEAP":
            raise ValueError("bad local heap")
        data = self._u(p + 24, 8) + self.base
        end = self.b.find(b"\x00", data + off)
        return self.b[data + off:end].decode()

    def _btree_entries(self, addr, heap):
        p = addr + self.base
        if self.b[p:p + 4] != b"TREE":
            raise ValueError("bad B-tree node")
        ntype, level, used = self.b[p + 4], self.b[p + 5], self._u(p + 6, 2)
        if ntype != 0:
            raise NotImplementedError("chunked dataset B-tree")
        q = p + 24
        for i in range(used):
            child = self._u(q + 8 + i * 16, 8)
            if level > 0:
                yield from self._btree_entries(child, heap)
            else:
                s = child + self.base
                if self.b[s:s + 4] != b"SNOD":
                    raise ValueError("bad symbol table node")
                for k in range(self._u(s + 6, 2)):
                    e = self._sym_entry(s + 8 + k * 40)
                    yield self._heap_name(heap, e["name_off"]), e

    def _group_tables(self, entry):
        if entry["btree"] is not None:
            return entry["btree"], entry["heap"]
        for mtype, data, _ in self._messages(entry["hdr"]):
            if mtype == 0x11:
                return self._u(data, 8), self._u(data + 8, 8)
            if mtype in (0x02, 0x06):
                raise NotImplementedError("new-style (link message) groups")
        return None

    def _dataset(self, hdr):
        shape = dtype = None
        layout = None
        for mtype, data, size in self._messages(hdr):
            if mtype == 0x01:  # dataspace
                ver, rank, flags = self.b[data], self.b[data + 1], self.b[data + 2]
                q = data + (8 if ver == 1 else 4)
                shape = tuple(self._u(q + 8 * i, 8) for i in range(rank))
            elif mtype == 0x03:  # datatype
                cls, bits0, sz = self.b[data] & 0x0F, self.b[data + 1], self._u(data + 4, 4)
                if cls == 1 and not (bits0 & 1) and sz in (4, 8):
                    dtype = np.dtype("<f%d" % sz)
                elif cls == 0 and not (bits0 & 1) and sz in (1, 2, 4, 8):  # fixed point: bit 3 = signed
                    dtype = np.dtype("<%s%d" % ("i" if bits0 & 8 else "u", sz))
                elif cls == 3:                                             # fixed-length string
                    dtype = np.dtype("S%d" % sz)
                else:
                    raise NotImplementedError("only little-endian float / integer and fixed-length string datasets")
            elif mtype == 0x08:  # layout
                ver, lclass = self.b[data], self.b[data + 1]
                if ver != 3:
                    raise NotImplementedError("data layout version %d" % ver)
                if lclass == 1:
                    layout = ("contiguous", self._u(data + 2, 8), self._u(data + 10, 8))
                elif lclass == 0:
                    layout = ("compact", data + 4, self._u(data + 2, 2))
                else:
                    raise NotImplementedError("chunked/compressed datasets (Keras writes contiguous ones)")
        if shape is None or dtype is None or layout is None:
            return None
        n = int(np.prod(shape)) if shape else 1
        if layout[0] == "contiguous":
            addr = layout[1]
            if addr == UNDEF:
                return np.zeros(shape, dtype)
            start = addr + self.base
        else:
            start = layout[1]
        return np.frombuffer(self.b, dtype=dtype, count=n, offset=start).reshape(shape).copy()

    def walk(self):
        """-> {full path: ndarray} for every dataset in the file."""
        out = {}

        def rec(entry, prefix):
            tabs = self._group_tables(entry)
            if tabs is None:
                arr = self._dataset(entry["hdr"])
                if arr is not None:
                    out[prefix] = arr
                return
            for name, e in self._btree_entries(*tabs):
                rec(e, prefix + "/" + name)

        rec(self.root, "")
        return out


def load_prednet_weights(path, names):
    """names: ['a0/kernel', 'a0/bias', ...] (PredNetConfig.weight_shapes order). Looks the arrays
    up by Keras' variable names `.../layer_<key>_<level>/{kernel,bias}:0` (prednet.py:225)."""
    data = H5File(path).walk()
    out = []
    for n in names:
        key, kind = n.split("/")
        stem = key.rstrip("0123456789")
        want = "/layer_%s_%s/" % (stem, key[len(stem):])
        hit = [v for k, v in data.items() if want in k + "/" and k.rsplit("/", 1)[-1].startswith(kind)]
        if len(hit) != 1:
            raise ValueError("cannot locate %s in %s (%d matches)" % (n, path, len(hit)))
        out.append(hit[0].astype(np.float32))
    return out


# ------------------------------------------------------------------------------------- writer
# The mirror image of the reader: the same HDF5 subset (superblock v0, old-style groups, object
# headers v1, contiguous little-endian float datasets) plus fixed-length string attributes, which
# is all Keras 2.2.4's `load_weights` / `load_weights_from_hdf5_group` looks at
# (`layer_names`, `weight_names`, `keras_version`, `backend`).  Written files are read back by real
# h5py / libhdf5 (tests/test_host.py runs that check with the conda interpreter when it exists).
LEAF_K, INTERNAL_K = 16, 16   # symbol-table node holds 2*LEAF_K entries: every group fits one node


class Group:
    def __init__(self, attrs=None):
        self.attrs = dict(attrs or {})   # name -> bytes (scalar string) | list of bytes (1-D string array)
        self.children = {}               # name -> Group | numpy array

    def group(self, name, attrs=None):
        g = self.children[name] = Group(attrs)
        return g

    def dataset(self, name, arr, attrs=None):
        arr = np.ascontiguousarray(arr)
        self.children[name] = Dataset(arr, attrs) if attrs else arr


class Dataset:
    def __init__(self, arr, attrs=None):
        self.arr, self.attrs = np.ascontiguousarray(arr), dict(attrs or {})


def _pad8(b):
    return b + b"\x00" * (-len(b) % 8)


def _string_dtype(size):
    return bytes([0x13, 0x01, 0, 0]) + struct.pack("<I", max(size, 1))  # fixed string, null padded, ASCII


def _float_dtype(itemsize):
    if itemsize == 4:
        prop = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
        return bytes([0x11, 0x20, 31, 0]) + struct.pack("<I", 4) + prop
    prop = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
    return bytes([0x11, 0x20, 63, 0]) + struct.pack("<I", 8) + prop


def _int_dtype(dt):
    return bytes([0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0]) + struct.pack("<IHH", dt.itemsize, 0, 8 * dt.itemsize)


def _dataspace(shape):
    return bytes([1, len(shape), 0, 0, 0, 0, 0, 0]) + b"".join(struct.pack("<Q", d) for d in shape)


def _attr_message(name, value):
    if isinstance(value, (bytes, str)):
        v = value.encode() if isinstance(value, str) else value
        size, shape, data = max(len(v), 1), (), v.ljust(max(len(v), 1), b"\x00")
    else:
        items = [x.encode() if isinstance(x, str) else bytes(x) for x in value]
        size = max([len(x) for x in items] + [1])
        shape, data = (len(items),), b"".join(x.ljust(size, b"\x00") for x in items)
    nm = name.encode() + b"\x00"
    dt, ds = _string_dtype(size), _dataspace(shape)
    head = struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(ds))
    return 0x0C, head + _pad8(nm) + _pad8(dt) + _pad8(ds) + data


def _object_header(messages):
    body = b""
    for mtype, payload in messages:
        p = _pad8(payload)
        if len(p) > 0xFFF8:
            raise ValueError("HDF5 header message too large (%d bytes)" % len(p))
        body += struct.pack("<HHB3x", mtype, len(p), 0) + p
    return struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(body)) + body


class _Writer:
    def __init__(self):
        self.buf = bytearray(96)  # superblock goes here at the end

    def alloc(self, data):
        off = len(self.buf)
        self.buf += _pad8(bytes(data))
        return off

    def reserve(self, n):
        off = len(self.buf)
        self.buf += b"\x00" * ((n + 7) & ~7)
        return off

    def put(self, off, data):
        self.buf[off:off + len(data)] = data

    def write_dataset(self, arr, attrs=None):
        if arr.dtype in (np.float32, np.float64):
            dt = _float_dtype(arr.dtype.itemsize)
        elif arr.dtype.kind in "iu" and arr.dtype.itemsize in (1, 2, 4, 8):
            dt = _int_dtype(arr.dtype)
        elif arr.dtype.kind == "S":
            dt = _string_dtype(arr.dtype.itemsize)
        else:
            raise NotImplementedError("dataset dtype %s" % arr.dtype)
        if arr.dtype.kind != "S" and arr.dtype.byteorder == ">":
            arr = arr.astype(arr.dtype.newbyteorder("<"))
        raw = UNDEF
        if arr.size:
            raw = len(self.buf)
            self.buf += memoryview(arr).cast("B")   # one copy, no intermediate bytes object
            self.buf += b"\x00" * (-len(self.buf) % 8)
        msgs = [(0x01, _dataspace(arr.shape)), (0x03, dt),
                (0x05, bytes([2, 2, 2, 0])),  # fill value v2: late allocation, written if set, none defined
                (0x08, bytes([3, 1]) + struct.pack("<QQ", raw, arr.nbytes))]
        msgs += [_attr_message(k, v) for k, v in (attrs or {}).items()]
        return self.alloc(_object_header(msgs))

    def write_group(self, g):
        """-> (object header address, b-tree address, heap address)"""
        names = sorted(g.children, key=lambda s: s.encode())
        if len(names) > 2 * LEAF_K:
            raise NotImplementedError("more than %d entries in one group" % (2 * LEAF_K))
        kids = {}
        for n in names:  # children first: their addresses go into the symbol table node
            c = g.children[n]
            if isinstance(c, Group):
                kids[n] = self.write_group(c)
            elif isinstance(c, Dataset):
                kids[n] = (self.write_dataset(c.arr, c.attrs), None, None)
            else:
                kids[n] = (self.write_dataset(c), None, None)
        heap_data, offs = bytearray(8), {}
        for n in names:
            offs[n] = len(heap_data)
            heap_data += _pad8(n.encode() + b"\x00")
        free_off = len(heap_data)
        free_len = max(16, 88 - len(heap_data) if len(heap_data) < 72 else 16)
        heap_data += struct.pack("<QQ", 1, free_len) + b"\x00" * (free_len - 16)
        data_addr = self.alloc(heap_data)
        heap = self.alloc(b"HEAP" + bytes(4) + struct.pack("<QQQ", len(heap_data), free_off, data_addr))
        snod = b"SNOD" + struct.pack("<BBH", 1, 0, len(names))
        for n in names:
            hdr, bt, hp = kids[n]
            if bt is None:
                snod += struct.pack("<QQII16x", offs[n], hdr, 0, 0)
            else:
                snod += struct.pack("<QQIIQQ", offs[n], hdr, 1, 0, bt, hp)
        snod = snod.ljust(8 + 2 * LEAF_K * 40, b"\x00")
        tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if names else 0, UNDEF, UNDEF)
        if names:
            snod_addr = self.alloc(snod)
            tree += struct.pack("<QQQ", 0, snod_addr, offs[names[-1]])
        tree = tree.ljust(24 + (2 * INTERNAL_K + 1) * 8 + 2 * INTERNAL_K * 8, b"\x00")
        btree = self.alloc(tree)
        msgs = [(0x11, struct.pack("<QQ", btree, heap))] + [_attr_message(k, v) for k, v in g.attrs.items()]
        return self.alloc(_object_header(msgs)), btree, heap

    def finish(self, root):
        hdr, bt, hp = self.write_group(root)
        sb = SIG + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", LEAF_K, INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(self.buf), UNDEF)
        sb += struct.pack("<QQIIQQ", 0, hdr, 1, 0, bt, hp)
        assert len(sb) == 96
        self.put(0, sb)
        return bytes(self.buf)


def write_file(path, root):
    data = _Writer().finish(root)
    with open(path, "wb") as f:
        f.write(data)


def save_prednet_checkpoint(path, cfg, weights, nt=2, model_config=None):
    """Writes `prednet_weights.hdf5` in the layout of the reference's Keras 2.2.4
    `ModelCheckpoint` (train.py:109: a full-model file, weights under /model_weights): layers
    input_1, pred_net_1 (the 46 arrays of prednet.py:210-227 as
    pred_net_1/layer_<key>_<level>/{kernel,bias}:0), time_distributed_1 and dense_2 with the fixed
    loss weights of train.py:56-59,67-69 (the reference's `load_weights` insists on every weighted
    layer of the model json being present), flatten_1."""
    L = cfg.nb_layers
    kv = b"2.2.4"
    root = Group({"keras_version": kv, "backend": b"tensorflow"})
    if model_config is not None:
        root.attrs["model_config"] = model_config if isinstance(model_config, bytes) else model_config.encode()
    mw = root.group("model_weights", {"layer_names": [b"input_1", b"pred_net_1", b"time_distributed_1", b"flatten_1", b"dense_2"],
                                      "backend": b"tensorflow", "keras_version": kv})
    mw.group("input_1", {"weight_names": []})
    mw.group("flatten_1", {"weight_names": []})
    names = []
    pn = mw.group("pred_net_1")
    inner = pn.group("pred_net_1")
    for (n, shape), w in zip(cfg.weight_shapes(), weights):
        key, kind = n.split("/")
        stem = key.rstrip("0123456789")
        lname = "layer_%s_%s" % (stem, key[len(stem):])
        if lname not in inner.children:
            inner.group(lname)
        inner.children[lname].dataset(kind + ":0", np.asarray(w, np.float32).reshape(shape))
        names.append(("pred_net_1/%s/%s:0" % (lname, kind)).encode())
    pn.attrs["weight_names"] = names
    layer_loss = np.zeros((L, 1), np.float32)
    layer_loss[0] = 1.0                                  # train.py:56-57: the "L_0" model
    time_loss = np.full((nt, 1), 1.0 / (nt - 1), np.float32)
    time_loss[0] = 0.0                                   # train.py:58-59
    td = mw.group("time_distributed_1", {"weight_names": [b"time_distributed_1/kernel:0", b"time_distributed_1/bias:0"]})
    tdi = td.group("time_distributed_1")
    tdi.dataset("kernel:0", layer_loss)
    tdi.dataset("bias:0", np.zeros(1, np.float32))
    de = mw.group("dense_2", {"weight_names": [b"dense_2/kernel:0", b"dense_2/bias:0"]})
    dei = de.group("dense_2")
    dei.dataset("kernel:0", time_loss)
    dei.dataset("bias:0", np.zeros(1, np.float32))
    write_file(path, root)
