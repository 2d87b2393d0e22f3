"""Read-only HDF5 reader for the subset of the format that `keras.Model.save_weights` produces
through h5py (what the reference's CheckpointManager writes: custom_train_objects/
checkpoint_manager.py:148-167), in numpy only - h5py is not part of this image's interpreter.

Covered (HDF5 file format specification, version 3): superblock versions 0-3; groups stored as
symbol tables (version-1 B-tree + local heap + symbol nodes, what h5py writes by default) and as
link messages inside version-2 object headers (`libver = 'latest'` with few links per group);
version-1 and version-2 object headers with continuation blocks; datasets with contiguous, compact
or chunked layout (version-1 chunk B-tree; deflate and shuffle filters; not the chunk indexes of
`libver = 'latest'`); fixed-point and
floating-point datatypes of either byte order.  Anything else (dense link storage in fractal
heaps, variable-length or compound types, external files, other filters) raises `Hdf5Error` with
the name of the feature, never a wrong array.

    with Hdf5File(path) as f:
        for name, shape in f.datasets():      # 'encoder/norm/vars/0', (128,)
            array = f.read(name)
"""
import zlib

import numpy as np

UNDEFINED = 0xFFFFFFFFFFFFFFFF
_SIGNATURE = b"\x89HDF\r\n\x1a\n"


class Hdf5Error(ValueError):
    pass


class _Dataset(object):
    __slots__ = ("shape", "dtype", "layout", "filters", "fill")

    def __init__(self):
        self.shape, self.dtype, self.layout, self.filters, self.fill = None, None, None, [], None


class Hdf5File(object):
    def __init__(self, path):
        self._f = open(path, "rb")
        self._path = path
        try:
            self._read_superblock()
            self._index = {}                  # 'a/b/c' -> object header address of every dataset
            self._walk(self._root, "", set())
        except Exception:
            self._f.close()
            raise

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        self._f.close()

    # -- low level ----------------------------------------------------------------------------
    def _at(self, address, size):
        self._f.seek(self._base + address)
        data = self._f.read(size)
        if len(data) != size:
            raise Hdf5Error("%s: truncated file (wanted %d bytes at %#x)" % (self._path, size, address))
        return data

    def _uint(self, data, offset, size):
        return int.from_bytes(data[offset: offset + size], "little")

    def _read_superblock(self):
        self._base = 0
        start = 0
        while True:                            # the superblock sits at 0, 512, 1024, 2048, ...
            self._f.seek(start)
            if self._f.read(8) == _SIGNATURE:
                break
            start = 512 if start == 0 else start * 2
            if start > (1 << 26):
                raise Hdf5Error("%s is not an HDF5 file" % self._path)
        head = self._at(start, 64 + 64)
        version = head[8]
        if version in (0, 1):
            self._so, self._sl = head[13], head[14]
            p = 24 + (4 if version == 1 else 0)
            self._base = self._uint(head, p, self._so)
            p += 4 * self._so                  # base, free-space info, end of file, driver info
            entry = head[p: p + 2 * self._so + 24]
            self._root = self._uint(entry, self._so, self._so)       # object header of the root group
        elif version in (2, 3):
            self._so, self._sl = head[9], head[10]
            p = 12
            self._base = self._uint(head, p, self._so)
            self._root = self._uint(head, p + 3 * self._so, self._so)
        else:
            raise Hdf5Error("superblock version %d" % version)
        if self._base == UNDEFINED & ((1 << (8 * self._so)) - 1):
            self._base = 0
        self._undef = (1 << (8 * self._so)) - 1

    # -- object headers -------------------------------------------------------------------------
    def _messages(self, address):
        """[(type, flags, payload bytes)] of the object header at `address` (both header versions)."""
        out = []
        head = self._at(address, 16)
        if head[:4] == b"OHDR":
            flags = head[5]
            p = 6
            if flags & 0x20:
                p += 16                          # access, modification, change, birth times
            if flags & 0x10:
                p += 4                           # attribute storage phase change values
            width = 1 << (flags & 3)
            head = self._at(address, p + width)
            chunk0 = self._uint(head, p, width)
            blocks = [(address + p + width, chunk0)]
            tracked = bool(flags & 0x04)
            i = 0
            while i < len(blocks):
                start, size = blocks[i]
                i += 1
                data = self._at(start, size)
                q, end = 0, size - 4             # a checksum closes every chunk
                if i > 1:
                    if data[:4] != b"OCHK":
                        raise Hdf5Error("bad object header continuation")
                    q = 4
                while q + 4 <= end:
                    mtype, msize, mflags = data[q], self._uint(data, q + 1, 2), data[q + 3]
                    q += 4 + (2 if tracked else 0)
                    body = data[q: q + msize]
                    q += msize
                    if mtype == 0x10:
                        blocks.append((self._uint(body, 0, self._so), self._uint(body, self._so, self._sl)))
                    elif mtype:
                        out.append((mtype, mflags, body))
            return out
        if head[0] != 1:
            raise Hdf5Error("object header version %d at %#x" % (head[0], address))
        blocks = [(address + 16, self._uint(head, 8, 4))]
        i = 0
        while i < len(blocks):
            start, size = blocks[i]
            i += 1
            data = self._at(start, size)
            q = 0
            while q + 8 <= size:
                mtype, msize, mflags = self._uint(data, q, 2), self._uint(data, q + 2, 2), data[q + 4]
                body = data[q + 8: q + 8 + msize]
                q += 8 + msize
                if mtype == 0x10:
                    blocks.append((self._uint(body, 0, self._so), self._uint(body, self._so, self._sl)))
                elif mtype:
                    out.append((mtype, mflags, body))
        return out

    # -- groups ---------------------------------------------------------------------------------
    def _walk(self, address, prefix, seen):
        if address in seen:
            return
        seen.add(address)
        messages = self._messages(address)
        kinds = {m[0] for m in messages}
        if 0x08 in kinds and 0x01 in kinds:                          # layout + dataspace: a dataset
            self._index[prefix.strip("/")] = address
            return
        for mtype, mflags, body in messages:
            if mtype == 0x11:                                         # symbol table: B-tree + local heap
                btree, heap = self._uint(body, 0, self._so), self._uint(body, self._so, self._so)
                heap_head = self._at(heap, 8 + 2 * self._sl + self._so)
                if heap_head[:4] != b"HEAP":
                    raise Hdf5Error("bad local heap at %#x" % heap)
                names = self._at(self._uint(heap_head, 8 + 2 * self._sl, self._so), self._uint(heap_head, 8, self._sl))
                for name_offset, child in self._group_entries(btree):
                    end = names.index(b"\0", name_offset)
                    self._walk(child, prefix + "/" + names[name_offset:end].decode("utf-8"), seen)
            elif mtype == 0x06:                                       # link message
                name, child = self._link(body)
                if child is not None:
                    self._walk(child, prefix + "/" + name, seen)
            elif mtype == 0x02:                                       # link info: dense storage?
                p = 2 + (8 if body[1] & 1 else 0)
                if self._uint(body, p, self._so) != self._undef:
                    raise Hdf5Error("group %r stores its links in a fractal heap (dense link storage): not "
                                    "supported, rewrite the file with h5py's default settings" % (prefix or "/"))

    def _group_entries(self, address):
        node = self._at(address, 8 + 2 * self._so)
        if node[:4] == b"SNOD":
            count = self._uint(node, 6, 2)
            size = 2 * self._so + 24
            data = self._at(address + 8, count * size)
            return [(self._uint(data, i * size, self._so), self._uint(data, i * size + self._so, self._so))
                    for i in range(count)]
        if node[:4] != b"TREE" or node[4] != 0:
            raise Hdf5Error("bad group B-tree node at %#x" % address)
        used = self._uint(node, 6, 2)
        data = self._at(address + 8 + 2 * self._so, (2 * used + 1) * max(self._so, self._sl) + 16)
        out, p = [], self._sl                                       # key 0, then (child, key) pairs
        for _ in range(used):
            out += self._group_entries(self._uint(data, p, self._so))
            p += self._so + self._sl
        return out

    def _link(self, body):
        flags = body[1]
        p = 2
        kind = 0
        if flags & 0x08:
            kind = body[p]
            p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        width = 1 << (flags & 3)
        n = self._uint(body, p, width)
        p += width
        name = body[p: p + n].decode("utf-8")
        p += n
        if kind != 0:                                                # soft / external links are not followed
            return name, None
        return name, self._uint(body, p, self._so)

    # -- datasets -------------------------------------------------------------------------------
    def datasets(self):
        """[(path, shape)] of every dataset, sorted by path."""
        return [(name, self._describe(addr).shape) for name, addr in sorted(self._index.items())]

    def __contains__(self, name):
        return name.strip("/") in self._index

    def names(self):
        return sorted(self._index)

    def _describe(self, address):
        d = _Dataset()
        for mtype, mflags, body in self._messages(address):
            if mtype == 0x01:
                version, rank, flags = body[0], body[1], body[2]
                p = 8 if version == 1 else 4
                d.shape = tuple(self._uint(body, p + i * self._sl, self._sl) for i in range(rank))
                if version >= 2 and body[3] == 2:
                    raise Hdf5Error("null dataspace")
            elif mtype == 0x03:
                d.dtype = self._dtype(body)
                if isinstance(d.dtype, tuple):
                    d.dtype, d.fill = d.dtype
            elif mtype == 0x08:
                d.layout = self._layout(body)
            elif mtype == 0x0B:
                d.filters = self._filters(body)
        if d.shape is None or d.dtype is None or d.layout is None:
            raise Hdf5Error("dataset header without dataspace, datatype or layout")
        return d

    def _dtype(self, body):
        cls, version = body[0] & 15, body[0] >> 4
        bits0 = body[1]
        size = self._uint(body, 4, 4)
        order = ">" if bits0 & 1 else "<"
        if cls == 0:
            return np.dtype("%s%s%d" % (order, "i" if bits0 & 8 else "u", size))
        if cls == 1:
            if size not in (2, 4, 8):
                raise Hdf5Error("floating-point type of %d bytes" % size)
            exp_size, man_size = body[8 + 5], body[8 + 7]
            if (size, exp_size, man_size) not in ((2, 5, 10), (4, 8, 23), (8, 11, 52)):
                if (size, exp_size, man_size) == (2, 8, 7):
                    return np.dtype(order + "u2"), "bfloat16"     # widened to float32 by read()
                raise Hdf5Error("non-IEEE float (%d bytes, %d exponent bits)" % (size, exp_size))
            return np.dtype("%sf%d" % (order, size))
        raise Hdf5Error("datatype class %d (only integers and floats are supported)" % cls)

    def _layout(self, body):
        version = body[0]
        if version in (3, 4):
            cls = body[1]
            if version == 4 and cls == 2:
                raise Hdf5Error("chunked dataset written with libver = 'latest' (version-4 chunk index)")
            if cls == 0:
                size = self._uint(body, 2, 2)
                return ("compact", body[4: 4 + size])
            if cls == 1:
                return ("contiguous", self._uint(body, 2, self._so), self._uint(body, 2 + self._so, self._sl))
            if cls == 2:
                rank = body[2]
                btree = self._uint(body, 3, self._so)
                dims = [self._uint(body, 3 + self._so + 4 * i, 4) for i in range(rank)]
                return ("chunked", btree, dims[:-1], dims[-1])
            raise Hdf5Error("data layout class %d" % cls)
        if version in (1, 2):
            rank, cls = body[1], body[2]
            p = 8
            address = None
            if cls != 0:
                address = self._uint(body, p, self._so)
                p += self._so
            dims = [self._uint(body, p + 4 * i, 4) for i in range(rank)]
            p += 4 * rank
            if cls == 0:
                size = self._uint(body, p, 4)
                return ("compact", body[p + 4: p + 4 + size])
            if cls == 1:
                return ("contiguous", address, None)
            return ("chunked", address, dims[:-1], dims[-1])
        raise Hdf5Error("data layout message version %d (written with libver = 'latest'?)" % version)

    def _filters(self, body):
        version, count = body[0], body[1]
        p = 8 if version == 1 else 2
        out = []
        for _ in range(count):
            fid = self._uint(body, p, 2)
            if version == 1 or fid >= 256:
                name_len = self._uint(body, p + 2, 2)
                p += 4
            else:
                name_len = 0
                p += 2
            n_values = self._uint(body, p + 2, 2)
            p += 4
            p += (name_len + 7) // 8 * 8 if version == 1 else name_len
            values = [self._uint(body, p + 4 * i, 4) for i in range(n_values)]
            p += 4 * n_values
            if version == 1 and n_values % 2:
                p += 4
            if fid not in (1, 2):
                raise Hdf5Error("filter %d (only deflate and shuffle are supported)" % fid)
            out.append((fid, values))
        return out

    def shape(self, name):
        return self._describe(self._index[name.strip("/")]).shape

    def read(self, name):
        """The dataset as a numpy array in native byte order."""
        name = name.strip("/")
        if name not in self._index:
            raise KeyError(name)
        d = self._describe(self._index[name])
        dtype = d.dtype
        count = int(np.prod(d.shape, dtype=np.int64)) if d.shape else 1
        kind = d.layout[0]
        if kind == "compact":
            raw = bytes(d.layout[1])[: count * dtype.itemsize]
            arr = np.frombuffer(raw, dtype=dtype, count=count)
        elif kind == "contiguous":
            if d.layout[1] == self._undef:                           # never written: all fill value (zeros)
                arr = np.zeros(count, dtype=dtype)
            else:
                arr = np.frombuffer(self._at(d.layout[1], count * dtype.itemsize), dtype=dtype, count=count)
        else:
            arr = self._read_chunked(d, count)
        arr = arr.reshape(d.shape).astype(dtype.newbyteorder("="), copy=True)
        if d.fill == "bfloat16":
            arr = (arr.astype(np.uint32) << 16).view(np.float32)
        return arr

    def _read_chunked(self, d, count):
        _, btree, chunk, item = d.layout
        rank = len(d.shape)
        if item != d.dtype.itemsize or len(chunk) != rank:
            raise Hdf5Error("chunk description does not match the dataset")
        out = np.zeros(d.shape, dtype=d.dtype)
        if btree == self._undef:
            return out.reshape(-1)
        for offsets, address, size, mask in self._chunks(btree, rank):
            raw = self._at(address, size)
            for index in range(len(d.filters) - 1, -1, -1):           # undo the pipeline back to front
                if mask & (1 << index):
                    continue
                fid, values = d.filters[index]
                if fid == 1:
                    raw = zlib.decompress(raw)
                else:
                    width = values[0] if values else item
                    n = len(raw) // width
                    raw = np.frombuffer(raw, np.uint8)[: n * width].reshape(width, n).T.tobytes()
            block = np.frombuffer(raw, dtype=d.dtype, count=int(np.prod(chunk))).reshape(chunk)
            sel_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offsets, chunk, d.shape))
            sel_in = tuple(slice(0, s.stop - s.start) for s in sel_out)
            out[sel_out] = block[sel_in]
        return out.reshape(-1)

    def _chunks(self, address, rank):
        node = self._at(address, 8 + 2 * self._so)
        if node[:4] != b"TREE" or node[4] != 1:
            raise Hdf5Error("bad chunk B-tree node at %#x" % address)
        level, used = node[5], self._uint(node, 6, 2)
        key = 8 + 8 * (rank + 1)
        data = self._at(address + 8 + 2 * self._so, used * (key + self._so) + key)
        p = 0
        for _ in range(used):
            size, mask = self._uint(data, p, 4), self._uint(data, p + 4, 4)
            offsets = [self._uint(data, p + 8 + 8 * i, 8) for i in range(rank)]
            child = self._uint(data, p + key, self._so)
            p += key + self._so
            if level:
                for c in self._chunks(child, rank):
                    yield c
            else:
                yield offsets, child, size, mask
