"""Minimal HDF5 reader / writer for the chain files either side of the sampling step.

The reference keeps its chains in HDF5 through h5py: ``chemcee_256.h5`` written by emcee's
``HDFBackend`` subclass ``Transformbackend`` (reference sampler.py:322-368: group ``mcmc`` with
attributes ``nwalkers``, ``ndim``, ``iteration``, ``has_blobs`` and datasets ``chain``,
``chain_transformed``, ``log_prob``, ``accepted``) and ``zeus_256.h5`` written by
``ZeusTransformCallback`` (sampler.py:556-577: root datasets ``samples``, ``chain_transformed``,
``logprob``, gzip-compressed chunks).  ``read_chain_and_cut`` (util.py:68-94) reads them back for
the next iteration's training points.  h5py / libhdf5 are not part of this image, so the subset of
the file format those files use is implemented here from the HDF5 File Format Specification
(version 0 superblock, version 1 object headers, symbol-table groups, B-tree v1 chunk indices,
deflate / shuffle / fletcher32 filters, global heaps for variable-length strings).

Reader: checked against the reference's own fixture ``chemcee_256.h5`` (written by libhdf5) and
the numbers the reference's ``tests/test_main.py:50-51`` asserts on it.  Writer: contiguous
datasets, or gzip chunks along axis 0 under a one-node chunk B-tree, in "earliest" format structures
laid out like libhdf5's own (same superblock, node sizes and message versions as the fixture); it is
checked through the reader only -- libhdf5 is not available here to cross-check it.
"""
import mmap
import os
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIGNATURE = b"\x89HDF\r\n\x1a\n"


class H5Error(IOError):
    pass


# ---------------------------------------------------------------------------------------- reading
class _Datatype(object):
    """Decoded datatype message: ``kind`` in {"num", "str", "vlen_str", "vlen", "enum", "other"}."""

    def __init__(self, kind, size, dtype=None, base=None, charset=0):
        self.kind, self.size, self.dtype, self.base, self.charset = kind, size, dtype, base, charset


def _parse_datatype(buf, off=0):
    cv, b0, b1, b2, size = struct.unpack_from("<BBBBI", buf, off)
    cls, p = cv & 0x0F, off + 8
    order = ">" if (b0 & 1) else "<"
    if cls == 0:                                                    # fixed point
        signed = (b0 >> 3) & 1
        return _Datatype("num", size, np.dtype("%s%s%d" % (order, "i" if signed else "u", size))), p + 4
    if cls == 1:                                                    # floating point
        return _Datatype("num", size, np.dtype("%sf%d" % (order, size))), p + 12
    if cls == 3:                                                    # fixed-length string
        return _Datatype("str", size, np.dtype("S%d" % size), charset=(b0 >> 4) & 0xF), p
    if cls == 9:                                                    # variable length
        base, q = _parse_datatype(buf, p)
        kind = "vlen_str" if (b0 & 0xF) == 1 else "vlen"
        return _Datatype(kind, size, None, base=base, charset=b1 & 0xF), q
    if cls == 8:                                                    # enumeration (h5py's bool)
        base, q = _parse_datatype(buf, p)
        nmemb = b0 | (b1 << 8)
        for _ in range(nmemb):                                      # names: null-terminated, padded to 8
            end = buf.index(b"\x00", q)
            q += ((end - q) // 8 + 1) * 8
        q += nmemb * base.size
        return _Datatype("enum", size, base.dtype, base=base), q
    return _Datatype("other", size), p


def _parse_dataspace(buf):
    ver, rank, flags = struct.unpack_from("<BBB", buf, 0)
    if ver == 1:
        p = 8
    elif ver == 2:
        if buf[3] == 2:                                             # null dataspace
            return None, None
        p = 4
    else:
        raise H5Error("dataspace message version %d" % ver)
    dims = struct.unpack_from("<%dQ" % rank, buf, p)
    maxdims = struct.unpack_from("<%dQ" % rank, buf, p + 8 * rank) if flags & 1 else dims
    return tuple(dims), tuple(maxdims)


class _Object(object):
    def __init__(self, f, addr):
        self.f, self.addr = f, addr
        self.msgs = f._read_header(addr)

    def _first(self, mtype):
        for t, body in self.msgs:
            if t == mtype:
                return body
        return None

    @property
    def attrs(self):
        out = {}
        for t, body in self.msgs:
            if t == 0x000C:
                name, val = self.f._parse_attribute(body)
                out[name] = val
        return out


class Dataset(_Object):
    def __init__(self, f, addr):
        super(Dataset, self).__init__(f, addr)
        self.shape, self.maxshape = _parse_dataspace(self._first(0x0001))
        self.dt, _ = _parse_datatype(self._first(0x0003))
        if self.dt.kind not in ("num", "str", "enum"):
            raise H5Error("dataset datatype class not supported")
        self.dtype = self.dt.dtype

    @property
    def chunks(self):
        """Chunk shape of a chunked dataset (h5py's ``Dataset.chunks``), None for compact / contiguous layouts."""
        body = self._first(0x0008)
        if body is None or body[0] != 3 or body[1] != 2:
            return None
        ndim = body[2]
        return tuple(int(c) for c in struct.unpack_from("<%dI" % ndim, body, 11)[:-1])

    def _filters(self):
        body = self._first(0x000B)
        if body is None:
            return []
        ver, nf = body[0], body[1]
        p, out = (8 if ver == 1 else 2), []
        for _ in range(nf):
            fid, = struct.unpack_from("<H", body, p)
            p += 2
            if ver == 1 or fid >= 256:
                nlen, = struct.unpack_from("<H", body, p)
                p += 2
            else:
                nlen = 0
            flags, ncd = struct.unpack_from("<HH", body, p)
            p += 4
            p += ((nlen + 7) // 8) * 8 if ver == 1 else nlen
            cd = struct.unpack_from("<%dI" % ncd, body, p)
            p += 4 * ncd
            if ver == 1 and ncd % 2:
                p += 4
            out.append((fid, cd))
        return out

    def _unfilter(self, raw, mask, filters):
        for k in range(len(filters) - 1, -1, -1):
            if mask & (1 << k):
                continue
            fid, cd = filters[k]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                es = cd[0] if cd else self.dtype.itemsize
                n = len(raw) // es
                a = np.frombuffer(raw, np.uint8, n * es).reshape(es, n)
                raw = a.T.tobytes() + raw[n * es:]
            elif fid == 3:
                raw = raw[:-4]
            else:
                raise H5Error("filter %d not supported" % fid)
        return raw

    def read(self, nrows=None):
        """The whole dataset, or its first ``nrows`` entries along axis 0 (emcee allocates the chain for
        the requested number of steps and keeps the filled length in the ``iteration`` attribute)."""
        if self.shape is None:
            return None
        full = self.shape
        if nrows is not None and self.shape and nrows < self.shape[0]:
            self.shape = (int(nrows),) + tuple(self.shape[1:])
        try:
            return self._read()
        finally:
            self.shape = full

    def _read(self):
        body = self._first(0x0008)
        ver, cls = body[0], body[1]
        if ver != 3:
            raise H5Error("data layout message version %d" % ver)
        count = int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1
        nbytes = count * self.dtype.itemsize
        if cls == 0:
            size, = struct.unpack_from("<H", body, 2)
            raw = bytes(body[4:4 + size])
            return np.frombuffer(raw, self.dtype, count).reshape(self.shape).copy()
        if cls == 1:
            addr, size = struct.unpack_from("<QQ", body, 2)
            if addr == UNDEF or count == 0:
                return np.zeros(self.shape, self.dtype)
            if addr + self.f.base + nbytes > len(self.f.buf):
                raise H5Error("address beyond the end of the file")
            return np.frombuffer(self.f.buf, self.dtype, count, offset=addr + self.f.base).reshape(self.shape).copy()
        if cls != 2:
            raise H5Error("data layout class %d" % cls)
        ndim = body[2]
        btree, = struct.unpack_from("<Q", body, 3)
        cdims = struct.unpack_from("<%dI" % ndim, body, 11)[:-1]
        out = np.zeros(self.shape, self.dtype)
        if btree == UNDEF or count == 0:
            return out
        filters = self._filters()
        ccount = int(np.prod(cdims, dtype=np.int64))
        for size, mask, offs, addr in self.f._chunks(btree, ndim):
            if any(o >= n for o, n in zip(offs, self.shape)):
                continue
            raw = self.f._at(addr, size)
            if filters:
                raw = self._unfilter(raw, mask, filters)
            chunk = np.frombuffer(raw, self.dtype, ccount).reshape(cdims)
            sel_out = tuple(slice(o, min(o + c, n)) for o, c, n in zip(offs, cdims, self.shape))
            sel_in = tuple(slice(0, s.stop - s.start) for s in sel_out)
            out[sel_out] = chunk[sel_in]
        return out

    def __getitem__(self, key):
        return self.read()[key]


class Group(_Object):
    def __init__(self, f, addr, btree=None, heap=None):
        super(Group, self).__init__(f, addr)
        if btree is None:
            st = self._first(0x0011)
            if st is None:
                raise H5Error("group without symbol table (new-style groups are not supported)")
            btree, heap = struct.unpack_from("<QQ", st, 0)
        self._links = f._group_entries(btree, heap)

    def keys(self):
        return sorted(self._links)

    def __contains__(self, name):
        name = name.strip("/")
        head, _, rest = name.partition("/")
        if head not in self._links:
            return False
        return True if not rest else (rest in self[head])

    def __getitem__(self, name):
        name = name.strip("/")
        head, _, rest = name.partition("/")
        if head not in self._links:
            raise KeyError(name)
        obj = self.f._open(self._links[head])
        return obj[rest] if rest else obj


class File(Group):
    """Read-only view of an HDF5 file held in memory (``File(path)['mcmc/chain'].read()``)."""

    def __init__(self, path):
        with open(path, "rb") as fh:                                # mapped, not read: a chain file can be gigabytes
            size = os.fstat(fh.fileno()).st_size
            self.buf = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ) if size else b""
        b = self.buf
        if b[:8] != SIGNATURE:
            raise H5Error("%s: not an HDF5 file" % path)
        ver = b[8]
        if ver not in (0, 1):
            raise H5Error("superblock version %d not supported" % ver)
        if b[13] != 8 or b[14] != 8:
            raise H5Error("only 8-byte offsets and lengths are supported")
        p = 24 + (4 if ver == 1 else 0)
        self.base, _, self.eof, _ = struct.unpack_from("<QQQQ", b, p)
        p += 32
        _, ohdr, cache, _, bt, hp = struct.unpack_from("<QQIIQQ", b, p)
        self.f = self
        if cache == 1:
            Group.__init__(self, self, ohdr, bt, hp)
        else:
            Group.__init__(self, self, ohdr)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def _at(self, addr, n):
        addr += self.base
        if addr + n > len(self.buf):
            raise H5Error("address beyond the end of the file")
        return self.buf[addr:addr + n]

    def _read_header(self, addr):
        b = self.buf
        ver, _, nmsg, _, hsize = struct.unpack_from("<BBHII", b, addr + self.base)
        if ver != 1:
            raise H5Error("object header version %d not supported" % ver)
        blocks, msgs = [(addr + 16, hsize)], []
        while blocks and len(msgs) < nmsg:
            p, n = blocks.pop(0)
            p += self.base
            end = p + n
            while p + 8 <= end and len(msgs) < nmsg:
                t, size, flags = struct.unpack_from("<HHB", b, p)
                body = b[p + 8:p + 8 + size]
                p += 8 + size
                if t == 0x0010:
                    blocks.append(struct.unpack_from("<QQ", body, 0))
                msgs.append((t, body))
        return msgs

    def _open(self, addr):
        types = [t for t, _ in self._read_header(addr)]
        if 0x0011 in types:
            return Group(self, addr)
        if 0x0008 in types:
            return Dataset(self, addr)
        raise H5Error("object at %d is neither an old-style group nor a dataset" % addr)

    def _heap_string(self, heap_data, off):
        end = self.buf.find(b"\x00", heap_data + off)
        return self.buf[heap_data + off:end].decode("utf-8")

    def _group_entries(self, btree, heap):
        hb = self._at(heap, 32)
        if hb[:4] != b"HEAP":
            raise H5Error("local heap signature")
        heap_data = struct.unpack_from("<Q", hb, 24)[0] + self.base
        out = {}

        def walk(addr):
            nb = self._at(addr, 24)
            if nb[:4] != b"TREE" or nb[4] != 0:
                raise H5Error("group B-tree node")
            level, used = nb[5], struct.unpack_from("<H", nb, 6)[0]
            body = self._at(addr + 24, (2 * used + 1) * 8)
            for k in range(used):
                child, = struct.unpack_from("<Q", body, 8 + 16 * k)
                if level > 0:
                    walk(child)
                    continue
                sn = self._at(child, 8)
                if sn[:4] != b"SNOD":
                    raise H5Error("symbol table node signature")
                nsym, = struct.unpack_from("<H", sn, 6)
                ent = self._at(child + 8, 40 * nsym)
                for e in range(nsym):
                    noff, ohdr = struct.unpack_from("<QQ", ent, 40 * e)
                    out[self._heap_string(heap_data, noff)] = ohdr
        walk(btree)
        return out

    def _chunks(self, addr, ndim):
        nb = self._at(addr, 24)
        if nb[:4] != b"TREE" or nb[4] != 1:
            raise H5Error("chunk B-tree node")
        level, used = nb[5], struct.unpack_from("<H", nb, 6)[0]
        ksz = 8 + 8 * ndim
        body = self._at(addr + 24, used * (ksz + 8) + ksz)
        for k in range(used):
            p = k * (ksz + 8)
            size, mask = struct.unpack_from("<II", body, p)
            offs = struct.unpack_from("<%dQ" % ndim, body, p + 8)[:-1]
            child, = struct.unpack_from("<Q", body, p + ksz)
            if level > 0:
                for c in self._chunks(child, ndim):
                    yield c
            else:
                yield size, mask, offs, child

    def _global_heap_object(self, addr, index):
        hb = self._at(addr, 16)
        if hb[:4] != b"GCOL":
            raise H5Error("global heap signature")
        total, = struct.unpack_from("<Q", hb, 8)
        blk = self._at(addr, total)
        p = 16
        while p + 16 <= total:
            idx, _, _, size = struct.unpack_from("<HHIQ", blk, p)
            if idx == index:
                return blk[p + 16:p + 16 + size]
            if idx == 0:
                break
            p += 16 + ((size + 7) // 8) * 8
        raise H5Error("global heap object %d not found" % index)

    def _parse_attribute(self, body):
        ver = body[0]
        nsz, dsz, ssz = struct.unpack_from("<HHH", body, 2)
        if ver == 1:
            p, pad = 8, lambda n: ((n + 7) // 8) * 8
        elif ver in (2, 3):
            p, pad = (8 if ver == 2 else 9), lambda n: n
        else:
            raise H5Error("attribute message version %d" % ver)
        name = bytes(body[p:p + nsz]).split(b"\x00")[0].decode("utf-8")
        p += pad(nsz)
        dt, _ = _parse_datatype(body, p)
        p += pad(dsz)
        shape, _ = _parse_dataspace(body[p:p + ssz])
        p += pad(ssz)
        if shape is None:
            return name, None
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        data = body[p:]
        if dt.kind in ("num", "enum"):
            v = np.frombuffer(data, dt.dtype, count).reshape(shape).copy()
            if dt.kind == "enum" and dt.size == 1:
                v = v.astype(bool)
        elif dt.kind == "str":
            v = np.array([bytes(data[i * dt.size:(i + 1) * dt.size]).split(b"\x00")[0].decode("utf-8", "replace")
                          for i in range(count)], dtype=object).reshape(shape)
        elif dt.kind == "vlen_str":
            vals = []
            for i in range(count):
                _, gaddr, gidx = struct.unpack_from("<IQI", data, 16 * i)
                vals.append(self._global_heap_object(gaddr, gidx).split(b"\x00")[0].decode("utf-8", "replace")
                            if gaddr not in (0, UNDEF) else "")
            v = np.array(vals, dtype=object).reshape(shape)
        else:
            return name, None
        return name, (v[()] if shape == () else v)


# ---------------------------------------------------------------------------------------- writing
def _pad8(b):
    return b + b"\x00" * (-len(b) % 8)


def _dtype_message(dt):
    dt = np.dtype(dt)
    if dt.kind == "f" and dt.itemsize in (4, 8):
        if dt.itemsize == 8:
            return struct.pack("<BBBBIHHBBBBI", 0x11, 0x20, 63, 0, 8, 0, 64, 52, 11, 0, 52, 1023)
        return struct.pack("<BBBBIHHBBBBI", 0x11, 0x20, 31, 0, 4, 0, 32, 23, 8, 0, 23, 127)
    if dt.kind in "iu":
        return struct.pack("<BBBBIHH", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, dt.itemsize, 0, 8 * dt.itemsize)
    if dt.kind == "S":
        return struct.pack("<BBBBI", 0x13, 0x00, 0, 0, dt.itemsize)          # null-terminated ASCII
    raise H5Error("dtype %s cannot be written" % dt)


def _dataspace_message(shape):
    rank = len(shape)
    return struct.pack("<BBBB4x", 1, rank, 0, 0) + struct.pack("<%dQ" % rank, *shape)


def _message(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _attribute_message(name, value):
    if isinstance(value, (str, bytes)):
        raw = value.encode("utf-8") if isinstance(value, str) else value
        arr = np.array(raw + b"\x00", dtype="S%d" % (len(raw) + 1))
    else:
        arr = np.asarray(value)
        if arr.dtype == np.bool_:
            arr = arr.astype(np.int8)
        if arr.dtype.byteorder == ">":
            arr = arr.astype(arr.dtype.newbyteorder("<"))
    nm = name.encode("utf-8") + b"\x00"
    dtm, dsm = _dtype_message(arr.dtype), _dataspace_message(arr.shape)
    body = struct.pack("<BBHHH", 1, 0, len(nm), len(dtm), len(dsm)) + _pad8(nm) + _pad8(dtm) + _pad8(dsm) \
        + np.ascontiguousarray(arr).tobytes()
    return _message(0x000C, body)


def _object_header(messages):
    blob = b"".join(messages)
    return struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(blob)) + blob


class _Blocks(object):
    """Concatenation along axis 0 of arrays with equal trailing dimensions, never materialised."""

    def __init__(self, blocks):
        self.blocks = blocks
        self.dtype = blocks[0].dtype
        self.shape = (sum(len(b) for b in blocks),) + tuple(blocks[0].shape[1:])
        self.ndim = len(self.shape)
        self.nbytes = sum(b.nbytes for b in blocks)


class Writer(object):
    """Builds a file of old-style groups and contiguous datasets, written in one pass::

        w = Writer(); g = w.group("mcmc", attrs={...}); w.dataset(g, "chain", array); w.save(path)

    ``w.root`` is the root group.  At most 8 links per group (one symbol-table node, the default
    leaf K = 4 of libhdf5)."""

    LEAF_K, INTERNAL_K = 4, 16
    CHUNK_ENTRIES = 64                                              # 2K of the chunk B-tree, default K = 32

    class _G(object):
        def __init__(self, attrs):
            self.attrs, self.children = dict(attrs or {}), []

    def __init__(self, attrs=None):
        self.root = Writer._G(attrs)

    def group(self, name, parent=None, attrs=None):
        g = Writer._G(attrs)
        (parent or self.root).children.append((name, g))
        return g

    def dataset(self, parent, name, array, attrs=None, compression=None, shuffle=False):
        """``compression="gzip"``: chunked along axis 0 with deflate (level 4) chunks and an unlimited first
        dimension, the layout ``ZeusTransformCallback`` asks h5py for (sampler.py:567-569).
        ``array`` may be a list of blocks that share dtype and trailing dimensions: stored as their concatenation
        along axis 0 without building it in memory (contiguous layout only)."""
        if isinstance(array, (list, tuple)):
            blocks = [np.ascontiguousarray(b) for b in array]
            dt = np.result_type(*[b.dtype for b in blocks])
            blocks = [b.astype(dt.newbyteorder("<") if dt.byteorder == ">" else dt, copy=False) for b in blocks]
            if compression is not None or len(set(b.shape[1:] for b in blocks)) != 1:
                array = np.concatenate(blocks)
            else:
                (parent or self.root).children.append((name, (_Blocks(blocks), dict(attrs or {}), None, False)))
                return
        array = np.ascontiguousarray(array)
        if array.dtype.byteorder == ">":
            array = array.astype(array.dtype.newbyteorder("<"))
        if compression not in (None, "gzip"):
            raise H5Error("compression %r" % (compression,))
        (parent or self.root).children.append((name, (array, dict(attrs or {}), compression, shuffle)))

    def save(self, path):
        # pieces in file order; large arrays go in as memoryviews (no copy), the superblock piece is patched last
        pieces, pos = [bytearray(96)], [96]

        def alloc(data, align=8):
            pad = -pos[0] % align
            if pad:
                pieces.append(b"\x00" * pad)
                pos[0] += pad
            addr = pos[0]
            pieces.append(data)
            pos[0] += data.nbytes if isinstance(data, memoryview) else len(data)
            return addr

        def write_chunked(array, attrs, shuffle):
            rank, es = array.ndim, array.dtype.itemsize
            n = array.shape[0]
            rows = max(1, -(-n // self.CHUNK_ENTRIES))              # one B-tree node: at most 64 chunks
            cdims = (rows,) + tuple(array.shape[1:])
            ksz = 8 + 8 * (rank + 1)
            node = bytearray()
            nchunks = 0
            for r0 in range(0, n, rows):
                blk = np.zeros(cdims, array.dtype)
                blk[:min(rows, n - r0)] = array[r0:r0 + rows]
                raw = blk.tobytes()
                if shuffle:
                    raw = np.frombuffer(raw, np.uint8).reshape(-1, es).T.tobytes()
                raw = zlib.compress(raw, 4)
                addr = alloc(raw)
                node += struct.pack("<II", len(raw), 0) + struct.pack("<%dQ" % (rank + 1), r0, *([0] * rank))
                node += struct.pack("<Q", addr)
                nchunks += 1
            node += struct.pack("<II", 0, 0) + struct.pack("<%dQ" % (rank + 1), nchunks * rows, *([0] * rank))
            head = struct.pack("<4sBBHQQ", b"TREE", 1, 0, nchunks, UNDEF, UNDEF)
            full = 24 + self.CHUNK_ENTRIES * (ksz + 8) + ksz
            bt = alloc(head + bytes(node) + b"\x00" * (full - 24 - len(node))) if nchunks else UNDEF
            filt = b""
            nf = 0
            if shuffle:
                filt += struct.pack("<HHHH", 2, 0, 1, 1) + struct.pack("<II", es, 0)
                nf += 1
            filt += struct.pack("<HHHH", 1, 0, 1, 1) + struct.pack("<II", 4, 0)
            nf += 1
            rank_dims = struct.pack("<%dQ" % rank, *array.shape) + struct.pack("<%dQ" % rank, UNDEF, *array.shape[1:])
            msgs = [_message(0x0001, struct.pack("<BBBB4x", 1, rank, 1, 0) + rank_dims),
                    _message(0x0003, _dtype_message(array.dtype), flags=1),
                    _message(0x0005, struct.pack("<BBBB", 2, 3, 2, 0)),    # incremental allocation
                    _message(0x000B, struct.pack("<BB6x", 1, nf) + filt, flags=1),
                    _message(0x0008, struct.pack("<BBBQ", 3, 2, rank + 1, bt) + struct.pack("<%dI" % (rank + 1), *(cdims + (es,))))]
            msgs += [_attribute_message(k, v) for k, v in attrs.items()]
            return alloc(_object_header(msgs))

        def write_extensible(ext, attrs):
            rank, es = ext.ndim, ext.dtype.itemsize
            dims = struct.pack("<%dQ" % rank, *ext.shape) + struct.pack("<%dQ" % rank, UNDEF, *ext.tail)
            msgs = [_message(0x0001, struct.pack("<BBBB4x", 1, rank, 1, 0) + dims),
                    _message(0x0003, _dtype_message(ext.dtype), flags=1),
                    _message(0x0005, struct.pack("<BBBB", 2, 3, 2, 0)),    # incremental allocation
                    _message(0x0008, struct.pack("<BBBQ", 3, 2, rank + 1, UNDEF) +
                             struct.pack("<%dI" % (rank + 1), *((ext.chunk_rows,) + ext.tail + (es,))))]
            msgs += [_attribute_message(k, v) for k, v in attrs.items()]
            return alloc(_object_header(msgs))

        def write_dataset(array, attrs, compression=None, shuffle=False):
            if isinstance(array, _Extensible):
                return write_extensible(array, attrs)
            if compression and array.ndim >= 1:
                return write_chunked(array, attrs, shuffle)
            nbytes = array.nbytes
            if isinstance(array, _Blocks):
                daddr = UNDEF
                for b in array.blocks:                              # back to back: the first one aligned, the rest unpadded
                    if b.nbytes:
                        a = alloc(memoryview(b.reshape(-1)).cast("B"), 8 if daddr == UNDEF else 1)
                        daddr = a if daddr == UNDEF else daddr
            else:
                daddr = alloc(memoryview(array.reshape(-1)).cast("B")) if nbytes else UNDEF
            msgs = [_message(0x0001, _dataspace_message(array.shape)),
                    _message(0x0003, _dtype_message(array.dtype), flags=1),
                    _message(0x0005, struct.pack("<BBBB", 2, 2, 2, 0)),    # fill value: late allocation, if-set, undefined
                    _message(0x0008, struct.pack("<BBQQ", 3, 1, daddr, nbytes))]
            msgs += [_attribute_message(k, v) for k, v in attrs.items()]
            return alloc(_object_header(msgs))

        def write_group(g):
            """Returns (object header address, B-tree address, local heap address)."""
            entries = []
            for name, child in g.children:
                if isinstance(child, Writer._G):
                    entries.append((name, write_group(child)))
                else:
                    entries.append((name, (write_dataset(*child), None, None)))
            if len(entries) > 2 * self.LEAF_K:
                raise H5Error("more than %d links in one group" % (2 * self.LEAF_K))
            entries.sort(key=lambda e: e[0].encode("utf-8"))
            heap, offs = bytearray(8), []                           # offset 0: the empty name
            for name, _ in entries:
                offs.append(len(heap))
                heap += _pad8(name.encode("utf-8") + b"\x00")
            free = len(heap)
            heap_size = max(88, free + 16)                          # room for one free-list block
            heap += struct.pack("<QQ", 1, heap_size - free)         # free block: next = 1 (last), size
            heap += b"\x00" * (heap_size - len(heap))
            snod = bytearray(struct.pack("<4sBBH", b"SNOD", 1, 0, len(entries)))
            for (name, (ohdr, bt, hp)), noff in zip(entries, offs):
                if bt is None:
                    snod += struct.pack("<QQII16x", noff, ohdr, 0, 0)
                else:
                    snod += struct.pack("<QQIIQQ", noff, ohdr, 1, 0, bt, hp)
            snod += b"\x00" * (8 + 2 * self.LEAF_K * 40 - len(snod))
            snod_addr = alloc(bytes(snod))
            tree = bytearray(struct.pack("<4sBBHQQ", b"TREE", 0, 0, 1 if entries else 0, UNDEF, UNDEF))
            if entries:
                tree += struct.pack("<QQQ", 0, snod_addr, offs[-1])
            tree += b"\x00" * (24 + (4 * self.INTERNAL_K + 1) * 8 - len(tree))
            bt_addr = alloc(bytes(tree))
            hdata_addr = alloc(bytes(heap))
            hp_addr = alloc(struct.pack("<4sB3xQQQ", b"HEAP", 0, heap_size, free, hdata_addr))
            msgs = [_message(0x0011, struct.pack("<QQ", bt_addr, hp_addr))]
            msgs += [_attribute_message(k, v) for k, v in g.attrs.items()]
            return alloc(_object_header(msgs)), bt_addr, hp_addr

        ohdr, bt, hp = write_group(self.root)
        alloc(b"", 8)
        sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, self.LEAF_K, self.INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, pos[0], UNDEF)
        sb += struct.pack("<QQIIQQ", 0, ohdr, 1, 0, bt, hp)
        pieces[0][:len(sb)] = sb
        with open(path, "wb") as fh:
            for piece in pieces:
                fh.write(piece)


# ------------------------------------------------------------------------------------ appending
class Appender(object):
    """An HDF5 file whose datasets grow along axis 0, one uncompressed chunk per append -- what emcee's
    ``HDFBackend`` does through h5py (``grow`` + ``save_step``, reference sampler.py:340-368) -- so that the chain
    file is current after every flush and nothing has to be consolidated at the end.

    ``Appender.create(path, layout)`` writes the skeleton (groups, empty extensible datasets, attributes);
    ``append({name: block, ...})`` adds ``len(block)`` rows to each named dataset (a chunk holds ``chunk_rows`` rows;
    a short append leaves the chunk partly filled for the next one, a long one is split); ``set_attr`` / ``set_data`` overwrite fixed-size
    attribute / dataset values in place; ``Appender.open(path)`` continues a file written this way.
    Write order of an append: chunk data, B-tree nodes, dataspace, end-of-file address -- a reader that comes
    in between sees the state before the append."""

    K2 = 64                                                         # entries per chunk B-tree node (2K, K = 32)

    def __init__(self, path):
        self.path = path
        self.fh = open(path, "r+b")
        self.ds = {}                                                # name -> state
        self.attr_off = {}                                          # (object name, attribute) -> (file offset, dtype)
        self.data_off = {}                                          # fixed dataset name -> (offset, dtype, shape)
        self.fh.seek(0, 2)
        self.eof = self.fh.tell()
        self._bulk, self._pool = [], None

    # Bulk chunk data goes out by positional writes (os.pwrite releases the GIL) off the caller's thread.  Buffered writes
    # to ONE file are serialised by the kernel (the inode lock): measured on the MI355X boxes (tools/io_write_probe.py, R5) one
    # writer sustains 5.8 GB/s into tmpfs and 9.8 GB/s into the overlay file system's page cache, four writers 4.6 / 9.8,
    # eight 3.0 / 9.3, copies into a shared mapping 3-6 -- so one writer it is (a 4096-walker chain produces 13.8 GB/s of
    # samples at the full sampling rate: at that size the chain file, not the GPU, bounds a run).
    BULK_MIN = 4 << 20
    BULK_PIECE = int(os.environ.get("LINNA_H5_PIECE_MB", "32")) << 20
    BULK_THREADS = int(os.environ.get("LINNA_H5_WRITERS", "1"))

    def _write_at(self, addr, view):
        if view.nbytes < self.BULK_MIN:
            self.fh.seek(addr); self.fh.write(view)
        else:
            for off in range(0, view.nbytes, self.BULK_PIECE):
                self._bulk.append((addr + off, view[off:off + self.BULK_PIECE]))

    @staticmethod
    def _pwrite_all(fd, addr, view):
        done = 0
        while done < view.nbytes:
            done += os.pwrite(fd, view[done:], addr + done)

    def _run_bulk(self):
        if not self._bulk:
            return
        self.fh.flush()
        fd = self.fh.fileno()
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=self.BULK_THREADS, thread_name_prefix="linna-h5-write")
        jobs = [self._pool.submit(self._pwrite_all, fd, addr, view) for addr, view in self._bulk]
        self._bulk = []
        for j in jobs:
            j.result()                                              # raises what a write raised

    # -- skeleton ------------------------------------------------------------------------------
    @staticmethod
    def create(path, spec, attrs=None, group=None, group_attrs=None, fixed=None, chunk_rows=100):
        """``spec``: {dataset name: (trailing shape, dtype)} extensible datasets inside ``group`` (or the root);
        ``fixed``: {name: array} small contiguous datasets next to them; attributes of the root / the group."""
        w = Writer(attrs=attrs)
        g = w.group(group, attrs=group_attrs) if group else None
        for name, a in (fixed or {}).items():
            w.dataset(g, name, np.asarray(a))
        for name, (tail, dt) in spec.items():
            (g or w.root).children.append((name, (_Extensible(tuple(tail), np.dtype(dt), int(chunk_rows)), {}, None, False)))
        w.save(path)
        return Appender.open(path)

    @staticmethod
    def open(path):
        ap = Appender(path)
        f = File(path)
        def visit(grp, prefix):
            for name in grp.keys():
                obj = grp[name]
                full = prefix + name
                if isinstance(obj, Group):
                    ap._index_attrs(f, obj, full)
                    visit(obj, full + "/")
                else:
                    ap._index_dataset(f, obj, full)
        ap._index_attrs(f, f, "")
        visit(f, "")
        return ap

    def _messages_with_offsets(self, f, addr):
        """(type, file offset of the body, size) of every header message of the object at ``addr``."""
        b = f.buf
        ver, _, nmsg, _, hsize = struct.unpack_from("<BBHII", b, addr + f.base)
        blocks, out = [(addr + 16, hsize)], []
        while blocks and len(out) < nmsg:
            p, n = blocks.pop(0)
            p += f.base
            end = p + n
            while p + 8 <= end and len(out) < nmsg:
                t, size, _ = struct.unpack_from("<HHB", b, p)
                if t == 0x0010:
                    blocks.append(struct.unpack_from("<QQ", b, p + 8))
                out.append((t, p + 8, size))
                p += 8 + size
        return out

    def _index_attrs(self, f, obj, name):
        for t, off, size in self._messages_with_offsets(f, obj.addr):
            if t != 0x000C:
                continue
            body = f.buf[off:off + size]
            if body[0] != 1:
                continue
            nsz, dsz, ssz = struct.unpack_from("<HHH", body, 2)
            pad = lambda n: ((n + 7) // 8) * 8
            aname = bytes(body[8:8 + nsz]).split(b"\x00")[0].decode("utf-8")
            dt, _ = _parse_datatype(body, 8 + pad(nsz))
            if dt.kind == "num":
                self.attr_off[(name, aname)] = (off + 8 + pad(nsz) + pad(dsz) + pad(ssz), dt.dtype)

    def _index_dataset(self, f, ds, name):
        msgs = self._messages_with_offsets(f, ds.addr)
        lay = [(off, size) for t, off, size in msgs if t == 0x0008][0]
        spc = [(off, size) for t, off, size in msgs if t == 0x0001][0]
        body = f.buf[lay[0]:lay[0] + lay[1]]
        if body[1] == 1:                                            # contiguous: a fixed dataset, writable in place
            addr, = struct.unpack_from("<Q", body, 2)
            self.data_off[name] = (addr + f.base, ds.dtype, ds.shape)
            return
        if body[1] != 2 or ds._filters():
            return
        rank = body[2] - 1
        btree, = struct.unpack_from("<Q", body, 3)
        cdims = struct.unpack_from("<%dI" % (rank + 1), body, 11)
        st = dict(rank=rank, dtype=ds.dtype, tail=tuple(ds.shape[1:]), chunk_rows=cdims[0], nrows=ds.shape[0],
                  dims_off=spc[0] + 8, btree_off=lay[0] + 3, path=[])
        st["chunk_bytes"] = int(np.prod(cdims[:-1], dtype=np.int64)) * cdims[-1]
        # rightmost path of the chunk B-tree (root first)
        addr = btree
        while addr != UNDEF:
            node = self._read_node(f, addr, rank)
            st["path"].append(node)
            addr = node["children"][-1] if node["level"] > 0 and node["children"] else UNDEF
        st["path"].reverse()                                        # path[level]
        self.ds[name] = st

    def _read_node(self, f, addr, rank):
        nb = f._at(addr, 24)
        level, used = nb[5], struct.unpack_from("<H", nb, 6)[0]
        left, right = struct.unpack_from("<QQ", nb, 8)
        ksz = 8 + 8 * (rank + 1)
        body = f._at(addr + 24, used * (ksz + 8) + ksz)
        keys, children = [], []
        for k in range(used + 1):
            p = k * (ksz + 8)
            size, mask = struct.unpack_from("<II", body, p)
            offs = struct.unpack_from("<%dQ" % (rank + 1), body, p + 8)
            keys.append((size, offs[0]))
            if k < used:
                children.append(struct.unpack_from("<Q", body, p + ksz)[0])
        return dict(addr=addr, level=level, keys=keys[:-1], final=keys[-1][1], children=children, left=left, right=right)

    # -- writing -------------------------------------------------------------------------------
    def _alloc(self, nbytes):
        self.eof += -self.eof % 8
        addr = self.eof
        self.eof += nbytes
        return addr

    def _node_bytes(self, st):
        return 24 + self.K2 * (8 + 8 * (st["rank"] + 1) + 8) + 8 + 8 * (st["rank"] + 1)

    def _write_node(self, st, node):
        rank = st["rank"]
        out = bytearray(struct.pack("<4sBBHQQ", b"TREE", 1, node["level"], len(node["children"]), node["left"], node["right"]))
        for (size, row), child in zip(node["keys"], node["children"]):
            out += struct.pack("<II", size, 0) + struct.pack("<%dQ" % (rank + 1), row, *([0] * rank)) + struct.pack("<Q", child)
        out += struct.pack("<II", 0, 0) + struct.pack("<%dQ" % (rank + 1), node["final"], *([0] * rank))
        out += b"\x00" * (self._node_bytes(st) - len(out))
        self.fh.seek(node["addr"]); self.fh.write(out)

    def _insert(self, st, level, key, child, final):
        """Append (key, child) at ``level`` of the rightmost path; ``final``: row just past everything stored."""
        path = st["path"]
        node = path[level]
        if len(node["children"]) < self.K2:
            node["keys"].append(key); node["children"].append(child); node["final"] = final
            self._write_node(st, node)
            for up in path[level + 1:]:                             # the boundary key moves up the rightmost path
                up["final"] = final
                self._write_node(st, up)
            return
        # full: a new rightmost node at this level, linked as sibling, announced one level up
        new = dict(addr=self._alloc(self._node_bytes(st)), level=level, keys=[key], children=[child], final=final, left=node["addr"], right=UNDEF)
        node["right"] = new["addr"]
        self._write_node(st, node)
        self._write_node(st, new)
        if level + 1 == len(path):                                  # the full node was the root: new root over both
            root = dict(addr=self._alloc(self._node_bytes(st)), level=level + 1, keys=[node["keys"][0]], children=[node["addr"]],
                        final=final, left=UNDEF, right=UNDEF)
            path.append(root)
            self.fh.seek(st["btree_off"]); self.fh.write(struct.pack("<Q", root["addr"]))
        path[level] = new
        self._insert(st, level + 1, key, new["addr"], final)

    def append(self, blocks):
        """Order on disk: chunk DATA first (buffered small writes flushed, bulk pieces by parallel ``pwrite``), then the
        B-tree entries that point at it, then the datasets' new lengths -- a reader (or a run killed) in between sees the
        state before the append and never an index entry over partly written data; space allocated for data that was
        never indexed lies past the recorded end of file (offset 40 is rewritten last) and is reused on resume."""
        n = None
        pending = []                                                # (dataset state, first row of the chunk, its address, key boundary)
        for name, a in blocks.items():
            st = self.ds[name]
            a = np.ascontiguousarray(a, dtype=st["dtype"])
            if tuple(a.shape[1:]) != st["tail"]:
                raise H5Error("append: %s has trailing shape %s, the dataset %s" % (name, a.shape[1:], st["tail"]))
            n = len(a) if n is None else n
            if len(a) != n:
                raise H5Error("append: blocks of different length")
            fill = st["nrows"] % st["chunk_rows"]
            if fill:                                                # the last chunk is partial: continue inside it
                take = min(len(a), st["chunk_rows"] - fill)
                row_bytes = st["chunk_bytes"] // st["chunk_rows"]
                self._write_at(st["path"][0]["children"][-1] + fill * row_bytes, memoryview(np.ascontiguousarray(a[:take]).reshape(-1)).cast("B"))
                st["nrows"] += take
                a = a[take:]
            for r0 in range(0, len(a), st["chunk_rows"]):
                part = a[r0:r0 + st["chunk_rows"]]
                addr = self._alloc(st["chunk_bytes"])
                self._write_at(addr, memoryview(part.reshape(-1)).cast("B"))
                if part.nbytes < st["chunk_bytes"]:
                    self.fh.seek(addr + part.nbytes); self.fh.write(b"\x00" * (st["chunk_bytes"] - part.nbytes))
                row = st["nrows"]
                st["nrows"] += len(part)
                pending.append((st, row, addr, (row // st["chunk_rows"] + 1) * st["chunk_rows"]))
        self.fh.flush()
        self._run_bulk()                                            # chunk data is in the file before anything points at it
        for st, row, addr, final in pending:
            if not st["path"]:                                      # first chunk: the root leaf
                node = dict(addr=self._alloc(self._node_bytes(st)), level=0, keys=[], children=[], final=final, left=UNDEF, right=UNDEF)
                st["path"].append(node)
                self.fh.seek(st["btree_off"]); self.fh.write(struct.pack("<Q", node["addr"]))
            self._insert(st, 0, (st["chunk_bytes"], row), addr, final)
        self.fh.flush()
        for name in blocks:                                         # the new length becomes visible last
            st = self.ds[name]
            self.fh.seek(st["dims_off"]); self.fh.write(struct.pack("<Q", st["nrows"]))
        self._finish()

    def _finish(self):
        self.fh.seek(40); self.fh.write(struct.pack("<Q", self.eof))
        self.fh.seek(0, 2)
        if self.fh.tell() < self.eof:
            self.fh.write(b"\x00" * (self.eof - self.fh.tell()))
        self.fh.flush()

    def set_attr(self, obj, name, value):
        off, dt = self.attr_off[(obj, name)]
        self.fh.seek(off); self.fh.write(np.asarray(value, dt).tobytes()); self.fh.flush()

    def set_data(self, name, array):
        off, dt, shape = self.data_off[name]
        a = np.ascontiguousarray(array, dt)
        if a.shape != tuple(shape):
            raise H5Error("set_data: %s is %s, got %s" % (name, shape, a.shape))
        self.fh.seek(off); self.fh.write(a.tobytes()); self.fh.flush()

    def nrows(self, name):
        return self.ds[name]["nrows"]

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True); self._pool = None
        if self.fh is not None:
            self.fh.close(); self.fh = None


class _Extensible(object):
    """Placeholder for Writer: an empty chunked dataset that an Appender will grow along axis 0."""

    def __init__(self, tail, dtype, chunk_rows):
        self.tail, self.dtype, self.chunk_rows = tail, dtype, chunk_rows
        self.shape = (0,) + tail
        self.ndim = len(self.shape)
        self.nbytes = 0
