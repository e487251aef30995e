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
        with open(path, "rb") as f:
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
        end = self.b.index(b"\x00", data + off)
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
                if cls != 1 or (bits0 & 1) or sz not in (4, 8):
                    raise NotImplementedError("only little-endian IEEE float datasets")
                dtype = np.dtype("<f%d" % sz)
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
