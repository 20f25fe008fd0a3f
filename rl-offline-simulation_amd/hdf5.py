"""A reader for the HDF5 files the reference's datasets are stored in -- NumPy + zlib only.

The reference writes a dataset with h5py (offsim4rl/data.py:85-98: `create_dataset(key, data=..., compression='gzip')` per experience
key, three opaque attributes holding pickles) and reads it back with h5py (data.py:120-146, utils/dataset_utils.py:13-34).  h5py is not
part of this stack (the interpreter that runs the engine has none), so `OfflineDataset.load_hdf5` falls back on this module: the subset
of the HDF5 file format that h5py / libhdf5 1.8 - 1.14 produce for such a file, read straight from the bytes:

  superblock          versions 0 - 3
  object headers      version 1, and version 2 ("OHDR" / "OCHK", what libver='latest' writes), continuation blocks
  groups              symbol tables (B-tree v1 + local heap + SNOD), and compact link messages of v2 headers
  dataspaces          scalar / simple / null, versions 1 and 2
  datatypes           integers, IEEE floats (2 / 4 / 8 bytes, either byte order), fixed-length strings, opaque, enums over integers
                      (h5py's bool is an enum {FALSE, TRUE} over int8 and comes back as numpy bool)
  layouts             compact, contiguous, chunked with a version-1 B-tree (data layout message version 3; versions 1 / 2 as well)
  filters             deflate, shuffle, fletcher32
  attributes          message versions 1 - 3
  fill values         chunks that were never written come back as the fill value (default 0)

Refused with NotImplementedError, never guessed at: variable-length, compound, reference and array datatypes, layout-version-4 chunk indexes
(single chunk / fixed array / extensible array / B-tree v2 -- written only under libver='latest'), virtual datasets (data.py:148-175's
`concatenate` output), dense link / attribute storage (fractal heaps), external links, other filters (lzf, szip, ...).

The objects mimic the slice of h5py's API the reference touches: `File(path)` is a `Group`; groups have `.attrs`, `keys()`, iteration,
`in`, `[name]` (paths with "/" allowed), `.get(name, default)`, `visititems(fn)`; datasets have `.shape`, `.dtype`, `len()`, `[...]` / `[()]`
and convert with `np.asarray`.  A dataset is read whole on every access and comes back as a fresh, writable array, as with h5py (the
device table copies the columns anyway).
Format source: the HDF5 File Format Specification, version 3.0 (public; restated here, no code taken from libhdf5 or h5py).
"""
import mmap
import struct
import zlib

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class HDF5FormatError(ValueError):
    """The bytes are not what the format specification says they should be (truncated or corrupt file, or not HDF5 at all)."""


def _unsupported(what):
    return NotImplementedError(f"HDF5 feature outside this reader's subset: {what} (install h5py to read this file)")


class _Buf:
    """The file's bytes with bounds-checked little-endian reads."""

    def __init__(self, data):
        self.d = data
        self.n = len(data)

    def bytes(self, off, n):
        if off < 0 or n < 0 or off + n > self.n:
            raise HDF5FormatError(f"read of {n} bytes at offset {off} runs past the end of the file ({self.n} bytes)")
        return self.d[off:off + n]

    def u(self, off, n):
        return int.from_bytes(self.bytes(off, n), "little")


def _pad8(n):
    return (n + 7) & ~7


# ---------------------------------------------------------------------------------------------------------------------------------
# messages

def _parse_dataspace(b, lsz=8):
    ver = b[0]
    rank = b[1]
    flags = b[2]
    if ver == 1:
        off = 8
    elif ver == 2:
        if b[3] == 2:
            return None  # null dataspace
        off = 4
    else:
        raise _unsupported(f"dataspace message version {ver}")
    dims = tuple(int.from_bytes(b[off + lsz * i: off + lsz * (i + 1)], "little") for i in range(rank))
    return dims


class _Type:
    __slots__ = ("dtype", "size", "as_bool", "consumed")


def _parse_datatype(b, lsize=8):
    """-> _Type: numpy dtype of an element as stored, its size, whether it is h5py's bool enum, and the bytes the message took."""
    cv = b[0]
    cls, ver = cv & 0x0F, cv >> 4
    bits = b[1] | (b[2] << 8) | (b[3] << 16)
    size = struct.unpack_from("<I", b, 4)[0]
    t = _Type()
    t.size = size
    t.as_bool = False
    p = 8
    if cls == 0:  # fixed point
        order = ">" if bits & 1 else "<"
        signed = bool(bits & 8)
        if size not in (1, 2, 4, 8):
            raise _unsupported(f"{size}-byte integers")
        t.dtype = np.dtype(f"{order}{'i' if signed else 'u'}{size}")
        p += 4
    elif cls == 1:  # floating point
        order = ">" if bits & 1 else "<"
        if bits & 0x40:
            raise _unsupported("VAX byte order")
        if size not in (2, 4, 8):
            raise _unsupported(f"{size}-byte floats")
        bit_off, prec, e_loc, e_size, m_loc, m_size, bias = struct.unpack_from("<HHBBBBI", b, 8)
        want = {2: (10, 5, 0, 10, 15), 4: (23, 8, 0, 23, 127), 8: (52, 11, 0, 52, 1023)}[size]
        if (e_loc, e_size, m_loc, m_size, bias) != want or bit_off != 0 or prec != 8 * size:
            raise _unsupported("a floating-point layout that is not IEEE binary16/32/64")
        t.dtype = np.dtype(f"{order}f{size}")
        p += 12
    elif cls == 3:  # fixed-length string
        t.dtype = np.dtype(f"S{size}")
    elif cls == 5:  # opaque
        t.dtype = np.dtype(f"V{size}")
        p += _pad8(bits & 0xFF)
    elif cls == 8:  # enumeration
        n = bits & 0xFFFF
        base = _parse_datatype(b[8:], lsize)
        if base.dtype.kind not in "iu":
            raise _unsupported("an enumeration over a non-integer type")
        p = 8 + base.consumed
        names = []
        for _ in range(n):
            e = b.index(b"\0", p)
            names.append(bytes(b[p:e]))
            p = e + 1 if ver >= 3 else p + _pad8(e + 1 - p)
        vals = np.frombuffer(bytes(b[p:p + n * base.size]), dtype=base.dtype)
        p += n * base.size
        t.dtype = base.dtype
        t.as_bool = base.size == 1 and sorted(zip(vals.tolist(), names)) == [(0, b"FALSE"), (1, b"TRUE")]
    else:
        name = {2: "time", 4: "bitfield", 6: "compound", 7: "reference", 9: "variable-length", 10: "array"}.get(cls, f"class {cls}")
        raise _unsupported(f"{name} datatype")
    t.consumed = p
    return t


def _parse_fill(b, msg_type):
    """-> the fill value's bytes, or None (zeros)."""
    if msg_type == 0x0004:  # old fill value message
        n = struct.unpack_from("<I", b, 0)[0]
        return bytes(b[4:4 + n]) if n else None
    ver = b[0]
    if ver in (1, 2):
        defined = b[3]
        if ver == 1 or defined:
            n = struct.unpack_from("<I", b, 4)[0]
            return bytes(b[8:8 + n]) if n else None
        return None
    if ver == 3:
        if b[1] & 0x20:
            n = struct.unpack_from("<I", b, 2)[0]
            return bytes(b[6:6 + n]) if n else None
        return None
    raise _unsupported(f"fill value message version {ver}")


def _parse_filters(b):
    ver, n = b[0], b[1]
    p = 8 if ver == 1 else 2
    if ver not in (1, 2):
        raise _unsupported(f"filter pipeline message version {ver}")
    out = []
    for _ in range(n):
        fid = struct.unpack_from("<H", b, p)[0]
        p += 2
        if ver == 1 or fid >= 256:
            nlen = struct.unpack_from("<H", b, p)[0]
            p += 2
        else:
            nlen = 0
        flags, ncd = struct.unpack_from("<HH", b, p)
        p += 4
        p += _pad8(nlen) if ver == 1 else nlen
        cd = struct.unpack_from(f"<{ncd}I", b, p)
        p += 4 * ncd
        if ver == 1 and ncd % 2:
            p += 4
        out.append((fid, flags, cd))
    return out


class _Layout:
    __slots__ = ("kind", "addr", "size", "chunk", "data")


def _parse_layout(b, osz, lsz):
    ver = b[0]
    lay = _Layout()
    lay.chunk = lay.data = None
    lay.addr = _UNDEF
    lay.size = 0
    if ver in (1, 2):
        ndim, cls = b[1], b[2]
        p = 8
        if cls != 0:
            lay.addr = int.from_bytes(b[p:p + osz], "little")
            p += osz
        dims = struct.unpack_from(f"<{ndim}I", b, p)
        p += 4 * ndim
        if cls == 2:
            p += 4  # element size
            lay.kind, lay.chunk = "chunked", tuple(dims)
            esz = struct.unpack_from("<I", b, p - 4)[0]
            lay.chunk = tuple(dims) + (esz,)
        elif cls == 1:
            lay.kind = "contiguous"
        else:
            n = struct.unpack_from("<I", b, p)[0]
            lay.kind, lay.data = "compact", bytes(b[p + 4:p + 4 + n])
        return lay
    if ver == 3:
        cls = b[1]
        if cls == 0:
            n = struct.unpack_from("<H", b, 2)[0]
            lay.kind, lay.data = "compact", bytes(b[4:4 + n])
        elif cls == 1:
            lay.kind = "contiguous"
            lay.addr = int.from_bytes(b[2:2 + osz], "little")
            lay.size = int.from_bytes(b[2 + osz:2 + osz + lsz], "little")
        elif cls == 2:
            ndim = b[2]
            lay.kind = "chunked"
            lay.addr = int.from_bytes(b[3:3 + osz], "little")
            lay.chunk = struct.unpack_from(f"<{ndim}I", b, 3 + osz)  # rank + 1 entries: the last one is the element size
        else:
            raise _unsupported(f"data layout class {cls}" + (" (a virtual dataset)" if cls == 3 else ""))
        return lay
    if ver == 4:
        cls = b[1]
        if cls == 0:
            n = struct.unpack_from("<H", b, 2)[0]
            lay.kind, lay.data = "compact", bytes(b[4:4 + n])
            return lay
        if cls == 1:
            lay.kind = "contiguous"
            lay.addr = int.from_bytes(b[2:2 + osz], "little")
            lay.size = int.from_bytes(b[2 + osz:2 + osz + lsz], "little")
            return lay
        raise _unsupported("a version-4 chunk index (file written with libver='latest')" if cls == 2 else "a virtual dataset")
    raise _unsupported(f"data layout message version {ver}")


# ---------------------------------------------------------------------------------------------------------------------------------
# objects

class _Header:
    """The messages of one object header: a list of (type, bytes)."""

    def __init__(self, f, addr):
        self.f = f
        self.msgs = []
        buf = f.buf
        if buf.bytes(addr, 4) == b"OHDR":
            self._v2(addr)
        else:
            self._v1(addr)

    def _v1(self, addr):
        buf, f = self.f.buf, self.f
        if buf.u(addr, 1) != 1:
            raise HDF5FormatError(f"no object header at offset {addr}")
        nmsg = buf.u(addr + 2, 2)
        size = buf.u(addr + 8, 4)
        blocks = [(addr + 16, size)]
        while blocks and len(self.msgs) < nmsg:
            p, n = blocks.pop(0)
            end = p + n
            while p + 8 <= end and len(self.msgs) < nmsg:
                mtype, msize, _flags = struct.unpack_from("<HHB", buf.bytes(p, 5))
                body = bytes(buf.bytes(p + 8, msize))
                p += 8 + msize
                if mtype == 0x0010:
                    blocks.append((int.from_bytes(body[:f.osz], "little"), int.from_bytes(body[f.osz:f.osz + f.lsz], "little")))
                self.msgs.append((mtype, body))

    def _v2(self, addr):
        buf, f = self.f.buf, self.f
        if buf.u(addr + 4, 1) != 2:
            raise HDF5FormatError("object header version is not 2")
        flags = buf.u(addr + 5, 1)
        p = addr + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        w = 1 << (flags & 3)
        size = buf.u(p, w)
        p += w
        blocks = [(p, size)]
        with_order = bool(flags & 0x04)
        n_blocks = 0
        while blocks:
            p, n = blocks.pop(0)
            end = p + n
            while p + 4 <= end:
                mtype = buf.u(p, 1)
                msize = buf.u(p + 1, 2)
                p += 4 + (2 if with_order else 0)
                body = bytes(buf.bytes(p, msize))
                p += msize
                if mtype == 0x10:
                    a = int.from_bytes(body[:f.osz], "little")
                    ln = int.from_bytes(body[f.osz:f.osz + f.lsz], "little")
                    if buf.bytes(a, 4) != b"OCHK":
                        raise HDF5FormatError("object header continuation block without its signature")
                    blocks.append((a + 4, ln - 8))  # without signature and checksum
                    n_blocks += 1
                    if n_blocks > 4096:
                        raise HDF5FormatError("object header continuation blocks do not end")
                elif mtype != 0:
                    self.msgs.append((mtype, body))

    def find(self, mtype):
        for t, b in self.msgs:
            if t == mtype:
                return b
        return None

    def all(self, mtype):
        return [b for t, b in self.msgs if t == mtype]


class AttributeManager:
    """`.attrs` of a group or dataset: a read-only mapping name -> NumPy value (np.void for opaque bytes, as h5py gives it)."""

    def __init__(self, f, header):
        self._d = {}
        info = header.find(0x0015)
        if info is not None:
            flags = info[1]
            p = 2 + (2 if flags & 1 else 0)
            if int.from_bytes(info[p:p + f.osz], "little") != f.undef:
                raise _unsupported("dense attribute storage")
        for b in header.all(0x000C):
            ver = b[0]
            nsz, tsz, ssz = struct.unpack_from("<HHH", b, 2)
            if ver == 1:
                p = 8
                name = bytes(b[p:p + nsz]); p += _pad8(nsz)
                tb = b[p:p + tsz]; p += _pad8(tsz)
                sb = b[p:p + ssz]; p += _pad8(ssz)
            elif ver in (2, 3):
                if b[1] & 3:
                    raise _unsupported("shared attribute datatypes / dataspaces")
                p = 8 if ver == 2 else 9
                name = bytes(b[p:p + nsz]); p += nsz
                tb = b[p:p + tsz]; p += tsz
                sb = b[p:p + ssz]; p += ssz
            else:
                raise _unsupported(f"attribute message version {ver}")
            name = name.split(b"\0", 1)[0].decode("utf-8")
            t = _parse_datatype(tb, f.lsz)
            shape = _parse_dataspace(sb, f.lsz)
            if shape is None:
                self._d[name] = None
                continue
            n = int(np.prod(shape, dtype=np.int64))
            arr = np.frombuffer(bytes(b[p:p + n * t.size]), dtype=t.dtype, count=n).reshape(shape)
            self._d[name] = _finish(arr, t)

    def get(self, name, default=None):
        return self._d.get(name, default)

    def __getitem__(self, name):
        return self._d[name]

    def __contains__(self, name):
        return name in self._d

    def __iter__(self):
        return iter(self._d)

    def __len__(self):
        return len(self._d)

    def keys(self):
        return self._d.keys()

    def items(self):
        return self._d.items()


def _finish(arr, t):
    """Stored elements -> what h5py hands back: native byte order, numpy bool for the bool enum, a scalar for rank 0."""
    if t.as_bool:
        arr = arr.astype(np.bool_)
    elif arr.dtype.byteorder == ">":
        arr = arr.astype(arr.dtype.newbyteorder("="))
    return arr[()] if arr.ndim == 0 else arr


class Dataset:
    def __init__(self, f, header, name):
        self._f = f
        self._h = header
        self.name = name
        self._t = _parse_datatype(header.find(0x0003), f.lsz)
        sp = header.find(0x0001)
        self.shape = _parse_dataspace(sp, f.lsz) if sp is not None else ()
        if self.shape is None:
            raise _unsupported("a dataset with a null dataspace")
        self.dtype = np.dtype(np.bool_) if self._t.as_bool else self._t.dtype.newbyteorder("=") if self._t.dtype.kind in "iuf" else self._t.dtype
        self._attrs = None

    @property
    def attrs(self):
        if self._attrs is None:
            self._attrs = AttributeManager(self._f, self._h)
        return self._attrs

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def __len__(self):
        if not self.shape:
            raise TypeError("Attempt to take len() of scalar dataset")
        return self.shape[0]

    def _read(self):
        f, h, t = self._f, self._h, self._t
        lay = _parse_layout(h.find(0x0008), f.osz, f.lsz)
        n = self.size
        fill = None
        for mt in (0x0005, 0x0004):
            fb = h.find(mt)
            if fb is not None:
                fill = _parse_fill(fb, mt)
                if fill is not None:
                    break
        if lay.kind == "compact":
            flat = np.frombuffer(lay.data, dtype=t.dtype, count=n)
        elif lay.kind == "contiguous":
            if lay.addr == f.undef or n == 0:
                flat = self._filled(n, fill)
            else:
                flat = np.frombuffer(f.buf.bytes(f.base + lay.addr, n * t.size), dtype=t.dtype, count=n)
        else:
            flat = self._read_chunks(lay, fill)
        out = _finish(np.asarray(flat).reshape(self.shape), t)
        if isinstance(out, np.ndarray) and not out.flags.writeable:
            out = out.copy()  # (views of the file's bytes are read-only; h5py hands out fresh, writable arrays)
        return out

    def _filled(self, n, fill):
        out = np.zeros(n, dtype=self._t.dtype)
        if fill is not None and len(fill) == self._t.size and any(fill):
            out[:] = np.frombuffer(fill, dtype=self._t.dtype, count=1)[0]
        return out

    def _read_chunks(self, lay, fill):
        f, t = self._f, self._t
        rank = len(self.shape)
        if len(lay.chunk) != rank + 1 or lay.chunk[-1] != t.size:
            raise HDF5FormatError("chunk dimensions disagree with the dataspace / datatype")
        cdims = tuple(lay.chunk[:-1])
        filters = []
        fb = self._h.find(0x000B)
        if fb is not None:
            filters = _parse_filters(fb)
            for fid, _, _ in filters:
                if fid not in (1, 2, 3):
                    raise _unsupported(f"filter {fid}" + {4: " (szip)", 32000: " (lzf)", 32001: " (blosc)"}.get(fid, ""))
        out = self._filled(self.size, fill).reshape(self.shape)
        if lay.addr == f.undef or self.size == 0:
            return out
        cn = int(np.prod(cdims, dtype=np.int64))
        for size, mask, offs, addr in f.chunk_leaves(lay.addr, rank):
            raw = f.buf.bytes(f.base + addr, size)
            for k in range(len(filters) - 1, -1, -1):
                if mask >> k & 1:
                    continue
                fid, _, cd = filters[k]
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 3:
                    raw = raw[:-4]  # the checksum trails the data; it is not verified
                else:  # shuffle: byte b of every element stored together
                    es = cd[0] if cd else t.size
                    m = len(raw) // es
                    a = np.frombuffer(raw, np.uint8, count=m * es).reshape(es, m).T
                    raw = a.tobytes() + bytes(raw[m * es:])
            if len(raw) < cn * t.size:
                raise HDF5FormatError("a chunk is shorter than its dimensions say")
            c = np.frombuffer(raw, dtype=t.dtype, count=cn).reshape(cdims)
            sel_o, sel_c = [], []
            for d in range(rank):
                lo = offs[d]
                hi = min(lo + cdims[d], self.shape[d])
                if lo >= self.shape[d]:
                    break
                sel_o.append(slice(lo, hi))
                sel_c.append(slice(0, hi - lo))
            else:
                out[tuple(sel_o)] = c[tuple(sel_c)]
        return out

    def __array__(self, dtype=None, copy=None):
        a = np.asarray(self._read())
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, key):
        a = self._read()  # (the whole dataset, every time: nothing is cached, as with h5py)
        if isinstance(key, tuple) and key == ():
            return a
        if not isinstance(a, np.ndarray):
            raise ValueError("Illegal slicing argument for scalar dataspace")
        return a[key]

    def __iter__(self):
        return iter(self._read())

    def __repr__(self):
        return f'<HDF5 dataset "{self.name}": shape {self.shape}, type "{self.dtype.str}">'


class Group:
    def __init__(self, f, header, name):
        self._f = f
        self._h = header
        self.name = name
        self._attrs = None
        self._links = None

    @property
    def attrs(self):
        if self._attrs is None:
            self._attrs = AttributeManager(self._f, self._h)
        return self._attrs

    def _load_links(self):
        if self._links is not None:
            return self._links
        f, h = self._f, self._h
        links = {}
        st = h.find(0x0011)
        if st is not None:
            btree = int.from_bytes(st[:f.osz], "little")
            heap = int.from_bytes(st[f.osz:2 * f.osz], "little")
            for nm, addr in f.symbol_table(btree, heap):
                links[nm] = addr
        info = h.find(0x0002)
        if info is not None:
            flags = info[1]
            p = 2 + (8 if flags & 1 else 0)
            if int.from_bytes(info[p:p + f.osz], "little") != f.undef:
                raise _unsupported("dense link storage (a group with many members in a libver='latest' file)")
        for b in h.all(0x0006):
            if b[0] != 1:
                raise _unsupported(f"link message version {b[0]}")
            flags = b[1]
            p = 2
            ltype = 0
            if flags & 0x08:
                ltype = b[p]; p += 1
            if flags & 0x04:
                p += 8
            if flags & 0x10:
                p += 1
            w = 1 << (flags & 3)
            ln = int.from_bytes(b[p:p + w], "little"); p += w
            nm = bytes(b[p:p + ln]).decode("utf-8"); p += ln
            if ltype != 0:
                raise _unsupported("soft / external links")
            links[nm] = int.from_bytes(b[p:p + f.osz], "little")
        self._links = dict(sorted(links.items()))
        return self._links

    def keys(self):
        return self._load_links().keys()

    def __iter__(self):
        return iter(self._load_links())

    def __len__(self):
        return len(self._load_links())

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, name):
        node = self
        if name.startswith("/"):
            node = self._f
        for part in [p for p in name.split("/") if p]:
            if not isinstance(node, Group):
                raise KeyError(f"Unable to open object (component of {name!r} is not a group)")
            links = node._load_links()
            if part not in links:
                raise KeyError(f"Unable to open object (object {part!r} doesn't exist)")
            node = node._f.open(links[part], (node.name.rstrip("/") + "/" + part))
        return node

    def get(self, name, default=None):
        try:
            return self[name]
        except KeyError:
            return default

    def items(self):
        return [(k, self[k]) for k in self]

    def values(self):
        return [self[k] for k in self]

    def visititems(self, fn):
        """h5py's Group.visititems: fn(path relative to this group, object) for every member, recursively; stops at a non-None return."""
        def walk(g, prefix):
            for k in g:
                obj = g[k]
                r = fn(prefix + k, obj)
                if r is not None:
                    return r
                if isinstance(obj, Group):
                    r = walk(obj, prefix + k + "/")
                    if r is not None:
                        return r
            return None
        return walk(self, "")

    def __repr__(self):
        return f'<HDF5 group "{self.name}" ({len(self)} members)>'


class File(Group):
    """`File(path)` (read-only; the mode argument is accepted for h5py compatibility and must be 'r')."""

    def __init__(self, path, mode="r"):
        if mode != "r":
            raise ValueError("this reader opens files read-only")
        # the file is mapped, not read: a big log costs its arrays once (contiguous datasets are copied out of the map, chunks inflated)
        self._fh = open(path, "rb")
        try:
            self._map = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        except ValueError:  # an empty file cannot be mapped
            self._fh.close()
            raise HDF5FormatError(f"{path}: no HDF5 signature (not an HDF5 file)") from None
        data = memoryview(self._map)
        self.buf = _Buf(data)
        self.filename = str(path)
        self._objects = {}
        base = 0
        while True:
            if base + 8 > len(data):
                self.close()
                raise HDF5FormatError(f"{path}: no HDF5 signature (not an HDF5 file)")
            if bytes(data[base:base + 8]) == _SIG:
                break
            base = 512 if base == 0 else base * 2
        buf = self.buf
        ver = buf.u(base + 8, 1)
        if ver in (0, 1):
            self.osz, self.lsz = buf.u(base + 13, 1), buf.u(base + 14, 1)
            p = base + 24 + (4 if ver == 1 else 0)
            self.base = buf.u(p, self.osz)
            root_entry = p + 4 * self.osz
            root_addr = buf.u(root_entry + self.osz, self.osz)
        elif ver in (2, 3):
            self.osz, self.lsz = buf.u(base + 9, 1), buf.u(base + 10, 1)
            self.base = buf.u(base + 12, self.osz)
            root_addr = buf.u(base + 12 + 3 * self.osz, self.osz)
        else:
            raise _unsupported(f"superblock version {ver}")
        if self.osz not in (4, 8) or self.lsz not in (4, 8):
            raise _unsupported(f"{self.osz}-byte offsets / {self.lsz}-byte lengths")
        self.undef = (1 << (8 * self.osz)) - 1
        if self.base == self.undef:
            self.base = 0
        Group.__init__(self, self, _Header(self, self.base + root_addr), "/")
        self._f = self

    def open(self, addr, name):
        obj = self._objects.get(addr)
        if obj is None:
            h = _Header(self, self.base + addr)
            obj = Dataset(self, h, name) if h.find(0x0008) is not None and h.find(0x0003) is not None else Group(self, h, name)
            self._objects[addr] = obj
        return obj

    # ---- version-1 B-trees ----
    def _tree(self, addr, want_type):
        buf = self.buf
        a = self.base + addr
        if buf.bytes(a, 4) != b"TREE":
            raise HDF5FormatError(f"no B-tree node at offset {a}")
        ntype, level, used = buf.u(a + 4, 1), buf.u(a + 5, 1), buf.u(a + 6, 2)
        if ntype != want_type:
            raise HDF5FormatError("B-tree node of the wrong type")
        return a + 8 + 2 * self.osz, level, used

    def symbol_table(self, btree, heap):
        buf = self.buf
        ha = self.base + heap
        if buf.bytes(ha, 4) != b"HEAP":
            raise HDF5FormatError("no local heap where the symbol table points")
        heap_data = self.base + buf.u(ha + 8 + 2 * self.lsz, self.osz)
        out = []

        def name_at(off):
            p = heap_data + off
            e = p
            while buf.u(e, 1):
                e += 1
            return bytes(buf.bytes(p, e - p)).decode("utf-8")

        def walk(addr):
            p, level, used = self._tree(addr, 0)
            for i in range(used):
                child = buf.u(p + self.lsz + i * (self.lsz + self.osz), self.osz)
                if level:
                    walk(child)
                    continue
                s = self.base + child
                if buf.bytes(s, 4) != b"SNOD":
                    raise HDF5FormatError("no symbol table node where the B-tree points")
                esz = 2 * self.osz + 24
                for k in range(buf.u(s + 6, 2)):
                    e = s + 8 + k * esz
                    out.append((name_at(buf.u(e, self.osz)), buf.u(e + self.osz, self.osz)))
        walk(btree)
        return out

    def chunk_leaves(self, btree, rank):
        """(stored size, filter mask, offsets, address) of every chunk, in B-tree order."""
        buf = self.buf
        ksz = 8 + 8 * (rank + 1)

        def walk(addr):
            p, level, used = self._tree(addr, 1)
            for i in range(used):
                k = p + i * (ksz + self.osz)
                child = buf.u(k + ksz, self.osz)
                if level:
                    yield from walk(child)
                else:
                    size, mask = struct.unpack_from("<II", buf.bytes(k, 8))
                    offs = struct.unpack_from(f"<{rank}Q", buf.bytes(k + 8, 8 * rank))
                    yield size, mask, offs, child
        yield from walk(btree)

    def close(self):
        """Unmaps the file.  Arrays already handed out are copies and stay valid; groups and datasets of a closed file are not."""
        if self._map is not None:
            self.buf = _Buf(b"")
            try:
                self._map.close()
            except BufferError:  # (a view of the map is still referenced somewhere: the map goes with its last reference)
                pass
            self._map = None
            self._fh.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
