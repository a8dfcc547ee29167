"""A small pure-Python reader/writer for the subset of HDF5 the reference's files use.

The reference keeps its checkpoints as Keras HDF5 weight files (`model.save_weights(...hdf5)` / `load_weights(by_name=True,
skip_mismatch=True)`, nets/mj_uwyhNets_ba.py:536-551,610-630; mains/mj_trainUWYHGaitNet_DataGen_CasiaB.py:475-489,524-527)
and its samples as deepdish (PyTables) files (data/generateOFData.py:137-149).  Neither h5py nor PyTables exists on the
target image, so this module speaks the file format itself (HDF5 File Format Specification, version 0/2 superblocks):

  read : old-style groups (symbol table: v1 B-tree + local heap), new-style groups with compact link messages, v1 and v2
         object headers with continuation blocks, contiguous / compact / chunked (v1 B-tree) layouts, the deflate, shuffle
         and fletcher32 filters, fixed-point / floating-point / fixed- and variable-length string datatypes (global heap),
         attributes (message versions 1-3).
  write: version-0 superblock, old-style groups, contiguous datasets, attributes with numeric / fixed-length string values --
         what h5py writes by default, so that Keras/h5py on the reference side reads the result.

Anything outside the subset raises H5Error naming the feature (never a silent wrong answer).
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIGNATURE = b"\x89HDF\r\n\x1a\n"


class H5Error(Exception):
    pass


# --------------------------------------------------------------------------------------------------------------------
# reader
# --------------------------------------------------------------------------------------------------------------------
class _Type:
    """Decoded datatype message: numpy dtype for numeric / fixed strings, or kind 'vlen_str'."""

    def __init__(self, kind, dtype=None, size=0, utf8=False):
        self.kind, self.dtype, self.size, self.utf8 = kind, dtype, size, utf8


def _parse_datatype(buf, off=0):
    cv, b0, b1, b2, size = struct.unpack_from("<BBBBI", buf, off)
    cls, ver = cv & 15, cv >> 4
    if cls == 0:    # fixed-point
        order = ">" if b0 & 1 else "<"
        return _Type("num", np.dtype("%s%s%d" % (order, "i" if b0 & 8 else "u", size)), size)
    if cls == 1:    # floating-point
        order = ">" if b0 & 1 else "<"
        if size not in (2, 4, 8):
            raise H5Error("floating-point type of %d bytes" % size)
        return _Type("num", np.dtype("%sf%d" % (order, size)), size)
    if cls == 3:    # fixed-length string
        return _Type("str", np.dtype("S%d" % size), size, utf8=bool((b0 >> 4) & 15))
    if cls == 9:    # variable-length
        if (b0 & 15) != 1:
            raise H5Error("variable-length sequences (only variable-length strings are supported)")
        return _Type("vlen_str", None, 16, utf8=bool(b1 & 15))
    if cls == 4:    # bit field: read as the unsigned integer of its size
        return _Type("num", np.dtype("%su%d" % (">" if b0 & 1 else "<", size)), size)
    if cls == 8:    # enumeration: the base integer type; the two-member one-byte one (h5py / PyTables booleans) as bool
        base = _parse_datatype(buf, off + 8)
        if size == 1 and (b0 | (b1 << 8)) == 2:
            return _Type("num", np.dtype(np.bool_), 1)
        return base
    raise H5Error("datatype class %d (version %d)" % (cls, ver))


def _parse_dataspace(buf, off=0):
    ver = buf[off]
    if ver == 1:
        rank, flags = buf[off + 1], buf[off + 2]
        p = off + 8
    elif ver == 2:
        rank, flags, typ = buf[off + 1], buf[off + 2], buf[off + 3]
        if typ == 2:
            return None        # null dataspace
        p = off + 4
    else:
        raise H5Error("dataspace message version %d" % ver)
    return tuple(struct.unpack_from("<%dQ" % rank, buf, p)) if rank else ()


class _Node:
    """An object header, decoded: messages by type."""

    def __init__(self, f, addr):
        self.f, self.addr = f, addr
        self.msgs = []       # (type, flags, bytes)
        f._read_object_header(addr, self.msgs)

    def first(self, mtype):
        for t, _, d in self.msgs:
            if t == mtype:
                return d
        return None


class Attributes(dict):
    pass


class Object:
    def __init__(self, f, node, name):
        self._f, self._node, self.name = f, node, name
        self._attrs = None

    @property
    def attrs(self):
        if self._attrs is None:
            self._attrs = Attributes()
            for t, _, d in self._node.msgs:
                if t == 0x000C:
                    k, v = self._f._parse_attribute(d)
                    self._attrs[k] = v
                elif t == 0x0015:
                    ver, flags = d[0], d[1]
                    p = 2 + (2 if flags & 1 else 0)
                    if struct.unpack_from("<Q", d, p)[0] != UNDEF:
                        raise H5Error("densely stored attributes (fractal heap) on %s" % self.name)
        return self._attrs


class Group(Object):
    def __init__(self, f, node, name):
        super().__init__(f, node, name)
        self._links = None

    def _load(self):
        if self._links is not None:
            return self._links
        links = {}
        st = self._node.first(0x0011)
        if st is not None:
            btree, heap = struct.unpack_from("<QQ", st, 0)
            self._f._walk_group_btree(btree, self._f._local_heap(heap), links)
        for t, _, d in self._node.msgs:
            if t == 0x0006:
                name, addr = self._f._parse_link(d)
                if addr is not None:
                    links[name] = addr
            elif t == 0x0002:
                flags = d[1]
                p = 2 + (8 if flags & 1 else 0)
                if struct.unpack_from("<Q", d, p)[0] != UNDEF:
                    raise H5Error("densely stored links (fractal heap) in group %s" % self.name)
        self._links = links
        return links

    def keys(self):
        return sorted(self._load())

    def __contains__(self, key):
        try:
            self[key]
            return True
        except KeyError:
            return False

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self._load())

    def __getitem__(self, path):
        obj = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(obj, Group):
                raise KeyError(path)
            links = obj._load()
            if part not in links:
                raise KeyError("%s (no %r in %s)" % (path, part, obj.name))
            obj = obj._f._object(links[part], (obj.name.rstrip("/") + "/" + part))
        return obj

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def visit_datasets(self, prefix=""):
        """[(relative path, Dataset)] of every dataset below this group, depth first in name order."""
        out = []
        for k in self.keys():
            o = self[k]
            if isinstance(o, Group):
                out += o.visit_datasets(prefix + k + "/")
            else:
                out.append((prefix + k, o))
        return out


class Dataset(Object):
    def __init__(self, f, node, name):
        super().__init__(f, node, name)
        self._type = _parse_datatype(node.first(0x0003))
        self.shape = _parse_dataspace(node.first(0x0001))
        self.dtype = self._type.dtype if self._type.kind != "vlen_str" else np.dtype(object)

    def read(self):
        f, t = self._f, self._type
        if self.shape is None:
            return None
        count = int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1
        lay = self._node.first(0x0008)
        ver = lay[0]
        if ver == 4 and lay[1] == 2:
            raise H5Error("version-4 chunked layout (chunk index types of libver='latest') on %s" % self.name)
        if ver in (3, 4):   # (version 4 keeps the compact and contiguous forms of version 3)
            cls = lay[1]
            if cls == 0:
                size = struct.unpack_from("<H", lay, 2)[0]
                raw = bytes(lay[4:4 + size])
            elif cls == 1:
                addr, size = struct.unpack_from("<QQ", lay, 2)
                raw = b"\0" * (count * t.size) if addr == UNDEF else f._at(addr, size)
            elif cls == 2:
                nd = lay[2]
                btree = struct.unpack_from("<Q", lay, 3)[0]
                cdims = struct.unpack_from("<%dI" % nd, lay, 11)
                raw = self._read_chunked(btree, cdims[:-1], cdims[-1])
            else:
                raise H5Error("data layout class %d" % cls)
        elif ver in (1, 2):
            nd, cls = lay[1], lay[2]
            p = 8
            addr = None
            if cls != 0:
                addr = struct.unpack_from("<Q", lay, p)[0]
                p += 8
            dims = struct.unpack_from("<%dI" % nd, lay, p)
            p += 4 * nd
            if cls == 0:
                size = struct.unpack_from("<I", lay, p)[0]
                raw = bytes(lay[p + 4:p + 4 + size])
            elif cls == 1:
                raw = b"\0" * (count * t.size) if addr == UNDEF else f._at(addr, count * t.size)
            else:
                raw = self._read_chunked(addr, dims[:-1], dims[-1])
        else:
            raise H5Error("data layout message version %d (written with libver='latest'?)" % ver)
        return f._decode(raw, t, self.shape)

    def __getitem__(self, key):
        a = self.read()
        return a[key] if isinstance(a, np.ndarray) else a

    def _read_chunked(self, btree, cdims, esize):
        shape = self.shape
        rank = len(shape)
        if len(cdims) != rank:
            raise H5Error("chunk rank %d for a rank-%d dataset" % (len(cdims), rank))
        filters = self._filters()
        out = np.zeros(shape, dtype=np.dtype("V%d" % esize))
        if btree == UNDEF:
            return out.tobytes()
        chunks = []
        self._f._walk_chunk_btree(btree, rank, chunks)
        cbytes = int(np.prod(cdims, dtype=np.int64)) * esize
        for size, mask, offs, addr in chunks:
            raw = self._f._at(addr, size)
            for i, (fid, cd) in reversed(list(enumerate(filters))):
                if mask & (1 << i):
                    continue
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:
                    w = cd[0] if cd else esize
                    n = len(raw) // w
                    raw = np.frombuffer(raw, np.uint8, n * w).reshape(w, n).T.tobytes() + raw[n * w:]
                elif fid == 3:
                    raw = raw[:-4]
                else:
                    raise H5Error("filter id %d (%s) on %s" % (fid, {32001: "blosc", 4: "szip", 307: "bzip2", 305: "lzo"}.get(fid, "?"), self.name))
            if len(raw) != cbytes:
                raise H5Error("chunk of %d bytes, expected %d" % (len(raw), cbytes))
            chunk = np.frombuffer(raw, dtype=out.dtype).reshape(cdims)
            sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
            out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]
        return out.tobytes()

    def _filters(self):
        d = self._node.first(0x000B)
        if d is None:
            return []
        ver, n = d[0], d[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid = struct.unpack_from("<H", d, p)[0]
            p += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = struct.unpack_from("<H", d, p)[0]
                p += 2
            _, ncd = struct.unpack_from("<HH", d, p)
            p += 4
            if ver == 1:
                nlen = (nlen + 7) & ~7
            p += nlen
            cd = struct.unpack_from("<%dI" % ncd, d, p)
            p += 4 * ncd
            if ver == 1 and ncd & 1:
                p += 4
            out.append((fid, cd))
        return out


class File(Group):
    """Read-only view of an HDF5 file (the whole file is held in memory: checkpoints are a few 10 MB, samples 100s of KB)."""

    def __init__(self, path):
        with open(path, "rb") as fh:
            self._buf = fh.read()
        self.filename = path
        self._cache = {}
        self._gheaps = {}
        base = self._buf.find(SIGNATURE)
        if base != 0:
            raise H5Error("%s: not an HDF5 file (or a user block precedes the superblock)" % path)
        ver = self._buf[8]
        if ver in (0, 1):
            so, sl = self._buf[13], self._buf[14]
            if (so, sl) != (8, 8):
                raise H5Error("offset/length sizes %d/%d (only 8/8)" % (so, sl))
            p = 24 + (4 if ver == 1 else 0)
            p += 32            # base, free-space, eof, driver addresses
            root_hdr = struct.unpack_from("<Q", self._buf, p + 8)[0]
        elif ver in (2, 3):
            if (self._buf[9], self._buf[10]) != (8, 8):
                raise H5Error("offset/length sizes other than 8/8")
            root_hdr = struct.unpack_from("<Q", self._buf, 12 + 24)[0]
        else:
            raise H5Error("superblock version %d" % ver)
        super().__init__(self, _Node(self, root_hdr), "/")

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    # ---- low level ----------------------------------------------------------------------------------------------
    def _at(self, addr, size):
        if addr + size > len(self._buf):
            raise H5Error("truncated file: %d bytes at %d" % (size, addr))
        return self._buf[addr:addr + size]

    def _object(self, addr, name):
        hit = self._cache.get(addr)
        if hit is None:
            node = _Node(self, addr)
            hit = Dataset(self, node, name) if node.first(0x0008) is not None else Group(self, node, name)
            self._cache[addr] = hit
        return hit

    def _read_object_header(self, addr, msgs):
        b = self._buf
        if b[addr:addr + 4] == b"OHDR":
            if b[addr + 4] != 2:
                raise H5Error("object header version %d" % b[addr + 4])
            flags = b[addr + 5]
            p = addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            w = 1 << (flags & 3)
            size = int.from_bytes(b[p:p + w], "little")
            p += w
            blocks = [(p, size)]
            tracked = bool(flags & 4)
            while blocks:
                p, size = blocks.pop(0)
                end = p + size
                while p + 4 <= end:
                    mtype, msize, mflags = b[p], struct.unpack_from("<H", b, p + 1)[0], b[p + 3]
                    p += 4 + (2 if tracked else 0)
                    data = b[p:p + msize]
                    p += msize
                    if mtype == 0x10:
                        caddr, clen = struct.unpack_from("<QQ", data, 0)
                        if b[caddr:caddr + 4] != b"OCHK":
                            raise H5Error("bad object header continuation")
                        blocks.append((caddr + 4, clen - 8))
                    elif mtype != 0:
                        msgs.append((mtype, mflags, data))
            return
        ver, _, nmsg, _, hsize = struct.unpack_from("<BBHII", b, addr)
        if ver != 1:
            raise H5Error("object header version %d at %d" % (ver, addr))
        blocks = [(addr + 16, hsize)]
        while blocks and nmsg > 0:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end and nmsg > 0:
                mtype, msize, mflags = struct.unpack_from("<HHB", b, p)
                data = b[p + 8:p + 8 + msize]
                p += 8 + msize
                nmsg -= 1
                if mtype == 0x10:
                    blocks.append(struct.unpack_from("<QQ", data, 0))
                elif mtype != 0:
                    if mflags & 2:
                        raise H5Error("shared object header messages")
                    msgs.append((mtype, mflags, data))

    def _local_heap(self, addr):
        b = self._buf
        if b[addr:addr + 4] != b"HEAP":
            raise H5Error("bad local heap")
        return struct.unpack_from("<Q", b, addr + 24)[0]

    def _cstr(self, addr):
        end = self._buf.index(b"\0", addr)
        return self._buf[addr:end].decode("utf-8")

    def _walk_group_btree(self, addr, heap_data, links):
        b = self._buf
        if b[addr:addr + 4] == b"SNOD":
            n = struct.unpack_from("<H", b, addr + 6)[0]
            for i in range(n):
                noff, hdr = struct.unpack_from("<QQ", b, addr + 8 + 40 * i)
                links[self._cstr(heap_data + noff)] = hdr
            return
        if b[addr:addr + 4] != b"TREE" or b[addr + 4] != 0:
            raise H5Error("bad group B-tree node")
        n = struct.unpack_from("<H", b, addr + 6)[0]
        p = addr + 24
        for i in range(n):
            child = struct.unpack_from("<Q", b, p + 8)[0]
            self._walk_group_btree(child, heap_data, links)
            p += 16

    def _walk_chunk_btree(self, addr, rank, out):
        b = self._buf
        if b[addr:addr + 4] != b"TREE" or b[addr + 4] != 1:
            raise H5Error("bad chunk B-tree node")
        level, n = b[addr + 5], struct.unpack_from("<H", b, addr + 6)[0]
        ksize = 8 + 8 * (rank + 1)
        p = addr + 24
        for i in range(n):
            size, mask = struct.unpack_from("<II", b, p)
            offs = struct.unpack_from("<%dQ" % rank, b, p + 8)
            child = struct.unpack_from("<Q", b, p + ksize)[0]
            if level == 0:
                out.append((size, mask, offs, child))
            else:
                self._walk_chunk_btree(child, rank, out)
            p += ksize + 8

    def _parse_link(self, d):
        ver, flags = d[0], d[1]
        p = 2
        ltype = 0
        if flags & 8:
            ltype = d[p]
            p += 1
        if flags & 4:
            p += 8
        if flags & 0x10:
            p += 1
        w = 1 << (flags & 3)
        n = int.from_bytes(d[p:p + w], "little")
        p += w
        name = bytes(d[p:p + n]).decode("utf-8")
        p += n
        if ltype != 0:
            return name, None     # soft / external links are not followed
        return name, struct.unpack_from("<Q", d, p)[0]

    def _parse_attribute(self, d):
        ver = d[0]
        nsize, tsize, ssize = struct.unpack_from("<HHH", d, 2)
        if ver == 1:
            p = 8
            pad = lambda n: (n + 7) & ~7
        elif ver in (2, 3):
            if d[1] & 3:
                raise H5Error("attribute with a shared datatype/dataspace")
            p = 8 + (1 if ver == 3 else 0)
            pad = lambda n: n
        else:
            raise H5Error("attribute message version %d" % ver)
        name = bytes(d[p:p + nsize]).split(b"\0")[0].decode("utf-8")
        p += pad(nsize)
        t = _parse_datatype(d, p)
        p += pad(tsize)
        shape = _parse_dataspace(d, p)
        p += pad(ssize)
        if shape is None:
            return name, None
        return name, self._decode(d[p:], t, shape)

    def _global_heap_object(self, addr, index):
        heap = self._gheaps.get(addr)
        if heap is None:
            b = self._buf
            if b[addr:addr + 4] != b"GCOL":
                raise H5Error("bad global heap collection")
            total = struct.unpack_from("<Q", b, addr + 8)[0]
            heap = {}
            p = addr + 16
            while p + 16 <= addr + total:
                idx, _, _, size = struct.unpack_from("<HHIQ", b, p)
                if idx == 0:
                    break
                heap[idx] = b[p + 16:p + 16 + size]
                p += 16 + ((size + 7) & ~7)
            self._gheaps[addr] = heap
        return heap[index]

    def _decode(self, raw, t, shape):
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if t.kind == "vlen_str":
            vals = []
            for i in range(count):
                n, addr, idx = struct.unpack_from("<IQI", raw, 16 * i)
                s = bytes(self._global_heap_object(addr, idx)[:n]) if addr not in (0, UNDEF) and n else b""
                vals.append(s.decode("utf-8") if t.utf8 else s)
            if not shape:
                return vals[0]
            a = np.empty(count, dtype=object)
            a[:] = vals
            return a.reshape(shape)
        a = np.frombuffer(raw, dtype=t.dtype, count=count)
        if t.kind == "num" and t.dtype.byteorder == ">":
            a = a.astype(t.dtype.newbyteorder("<"))
        a = a.reshape(shape)
        return a[()] if not shape else a.copy()


# --------------------------------------------------------------------------------------------------------------------
# writer
# --------------------------------------------------------------------------------------------------------------------
def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _dtype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind in "iu":
        bits = 0x08 if dt.kind == "i" else 0
        return struct.pack("<BBBBIHH", 0x10, bits, 0, 0, dt.itemsize, 0, 8 * dt.itemsize)
    if dt.kind == "f":
        # (sign location, exponent location/size, mantissa location/size, exponent bias) of IEEE binary32 / binary64 / binary16
        spec = {4: (31, 23, 8, 0, 23, 127), 8: (63, 52, 11, 0, 52, 1023), 2: (15, 10, 5, 0, 10, 15)}[dt.itemsize]
        sign, eloc, esz, mloc, msz, bias = spec
        return struct.pack("<BBBBIHHBBBBI", 0x11, 0x20, sign, 0, dt.itemsize, 0, 8 * dt.itemsize, eloc, esz, mloc, msz, bias)
    if dt.kind == "S":
        return struct.pack("<BBBBI", 0x13, 0x01, 0, 0, max(dt.itemsize, 1))     # null-padded ASCII
    if dt.kind == "b":   # the int8 enumeration FALSE = 0 / TRUE = 1 that h5py and PyTables store booleans as
        base = struct.pack("<BBBBIHH", 0x10, 0x08, 0, 0, 1, 0, 8)
        return struct.pack("<BBBBI", 0x18, 2, 0, 0, 1) + base + _pad8(b"FALSE\0") + _pad8(b"TRUE\0") + bytes([0, 1])
    raise H5Error("cannot write dtype %r" % (dt,))


def _space_msg(shape):
    if shape == ():
        return struct.pack("<BBBBI", 1, 0, 0, 0, 0)
    return struct.pack("<BBBBI", 1, len(shape), 0, 0, 0) + struct.pack("<%dQ" % len(shape), *shape)


class Bool:
    """Marks a scalar to be written as an HDF5 boolean (enumeration) instead of the uint8 other booleans become."""

    def __init__(self, v):
        self.v = bool(v)


def _as_array(v):
    """Attribute / dataset value -> numpy array of a writable dtype (str -> fixed-length bytes)."""
    if isinstance(v, Bool):
        return np.array(v.v, dtype=np.bool_)
    if isinstance(v, str):
        v = v.encode("utf-8")
    if isinstance(v, bytes):
        return np.array(v, dtype="S%d" % max(len(v), 1))
    if isinstance(v, (list, tuple)) and v and isinstance(v[0], (str, bytes)):
        v = [s.encode("utf-8") if isinstance(s, str) else s for s in v]
        return np.array(v, dtype="S%d" % max(max(len(s) for s in v), 1))
    a = np.asarray(v)
    if a.dtype.kind == "U":
        a = np.char.encode(a, "utf-8")
    if a.dtype.kind == "b":
        a = a.astype(np.uint8)
    if a.dtype.kind == "S" and a.dtype.itemsize == 0:
        a = a.astype("S1")
    if a.dtype.byteorder == ">":
        a = a.astype(a.dtype.newbyteorder("<"))
    return a


class _WGroup:
    def __init__(self):
        self.children = {}     # name -> _WGroup | np.ndarray
        self.attrs = {}
        self.child_attrs = {}  # dataset name -> attrs


class Writer:
    """Collects a tree of groups / datasets / attributes and writes it in one pass.

        w = Writer(); w.create_dataset("a/b/kernel:0", array); w.set_attr("a", "weight_names", [b"b/kernel:0"]); w.save(path)
    """

    LEAF_K = 4      # the library's defaults (symbol-table nodes of 2K entries, internal nodes of 2K children)
    INTERNAL_K = 16

    def __init__(self):
        self.root = _WGroup()

    def _group(self, path, create=True):
        g = self.root
        for part in [p for p in path.split("/") if p]:
            nxt = g.children.get(part)
            if nxt is None:
                if not create:
                    raise KeyError(path)
                nxt = g.children[part] = _WGroup()
            if not isinstance(nxt, _WGroup):
                raise H5Error("%s: %s is a dataset" % (path, part))
            g = nxt
        return g

    def create_group(self, path):
        self._group(path)

    def create_dataset(self, path, data):
        parts = [p for p in path.split("/") if p]
        g = self._group("/".join(parts[:-1]))
        a = _as_array(data)
        g.children[parts[-1]] = a if a.ndim == 0 else np.ascontiguousarray(a)   # (ascontiguousarray would make a scalar 1-d)

    def set_attr(self, path, name, value):
        parts = [p for p in path.split("/") if p]
        if parts:
            parent = self._group("/".join(parts[:-1]))
            node = parent.children.get(parts[-1])
            if node is not None and not isinstance(node, _WGroup):
                parent.child_attrs.setdefault(parts[-1], {})[name] = value
                return
        self._group(path).attrs[name] = value

    # ---- serialisation ------------------------------------------------------------------------------------------
    def save(self, path):
        self.buf = bytearray(96)        # superblock (version 0: 24 + 32 + 40 bytes) written last
        root_hdr, btree, heap = self._write_group(self.root)
        sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, self.LEAF_K, self.INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(self.buf), UNDEF)
        sb += struct.pack("<QQII", 0, root_hdr, 1, 0) + struct.pack("<QQ", btree, heap)
        self.buf[0:96] = sb
        with open(path, "wb") as fh:
            fh.write(self.buf)

    def _alloc(self, data):
        self.buf += b"\0" * (-len(self.buf) % 8)
        addr = len(self.buf)
        self.buf += data
        return addr

    def _attr_msgs(self, attrs):
        out = []
        for name, value in attrs.items():
            a = _as_array(value)
            nm = name.encode("utf-8") + b"\0"
            t, s = _dtype_msg(a.dtype), _space_msg(a.shape)
            body = struct.pack("<BBHHH", 1, 0, len(nm), len(t), len(s)) + _pad8(nm) + _pad8(t) + _pad8(s) + a.tobytes()
            out.append((0x000C, body))
        return out

    def _object_header(self, msgs):
        body = b""
        for mtype, data in msgs:
            data = _pad8(data)
            if len(data) > 0xFFFF:
                raise H5Error("attribute of %d bytes does not fit an object-header message (64 KB limit)" % len(data))
            body += struct.pack("<HHBBBB", mtype, len(data), 0, 0, 0, 0) + data
        return self._alloc(struct.pack("<BBHII", 1, 0, len(msgs), 1, len(body)) + b"\0" * 4 + body)

    def _write_dataset(self, a, attrs):
        addr = self._alloc(a.tobytes()) if a.size else UNDEF
        msgs = [(0x0001, _space_msg(a.shape)), (0x0003, _dtype_msg(a.dtype)),
                (0x0005, struct.pack("<BBBB", 2, 2, 2, 0)),                       # fill value: late allocation, undefined value
                (0x0008, struct.pack("<BBQQ", 3, 1, addr, a.nbytes))]
        return self._object_header(msgs + self._attr_msgs(attrs))

    def _write_group(self, g):
        entries = []    # (name, header address, cache type, scratch)
        for name in sorted(g.children, key=lambda s: s.encode("utf-8")):
            child = g.children[name]
            if isinstance(child, _WGroup):
                hdr, bt, hp = self._write_group(child)
                entries.append((name, hdr, 1, struct.pack("<QQ", bt, hp)))
            else:
                entries.append((name, self._write_dataset(child, g.child_attrs.get(name, {})), 0, b"\0" * 16))
        # local heap: offset 0 holds the empty string, then the names
        heap_data = bytearray(b"\0" * 8)
        offs = []
        for name, *_ in entries:
            offs.append(len(heap_data))
            heap_data += _pad8(name.encode("utf-8") + b"\0")
        free = len(heap_data)
        heap_data += struct.pack("<QQ", 1, 16)           # one free block at the end (next = 1: none, size 16)
        data_addr = self._alloc(bytes(heap_data))
        heap = self._alloc(b"HEAP" + struct.pack("<BBBBQQQ", 0, 0, 0, 0, len(heap_data), free, data_addr))
        # symbol-table nodes of at most 2K entries under one B-tree node (at most 2K' children)
        per = 2 * self.LEAF_K
        nodes = [list(range(i, min(i + per, len(entries)))) for i in range(0, len(entries), per)]
        if len(nodes) > 2 * self.INTERNAL_K:
            raise H5Error("group with %d entries (limit %d)" % (len(entries), 2 * self.INTERNAL_K * per))
        keys, kids = [0], []
        for idxs in nodes:
            body = b"SNOD" + struct.pack("<BBH", 1, 0, len(idxs))
            for i in idxs:
                name, hdr, ctype, scratch = entries[i]
                body += struct.pack("<QQII", offs[i], hdr, ctype, 0) + scratch
            body += b"\0" * (40 * (per - len(idxs)))
            kids.append(self._alloc(body))
            keys.append(offs[idxs[-1]])
        tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(kids), UNDEF, UNDEF)
        for i, kid in enumerate(kids):
            tree += struct.pack("<QQ", keys[i], kid)
        tree += struct.pack("<Q", keys[len(kids)])
        tree += b"\0" * (16 * (2 * self.INTERNAL_K - len(kids)))
        btree = self._alloc(tree)
        hdr = self._object_header([(0x0011, struct.pack("<QQ", btree, heap))] + self._attr_msgs(g.attrs))
        return hdr, btree, heap
