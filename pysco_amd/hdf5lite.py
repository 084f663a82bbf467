"""A minimal HDF5 reader / writer for particle snapshots, used when h5py is not installed.

Covers exactly the layout of the reference's snapshots (iostream.py:229-265, written through h5py with default
settings): a version-0 superblock, an old-style root group (symbol table: v1 B-tree + local heap), contiguous
little-endian datasets and scalar / 1-D attributes in version-1 object headers.  Written from the HDF5 File Format
Specification (version 1.1 structures); files written here open with h5py / libhdf5 and vice versa
(tests/test_hdf5lite.py cross-checks against the genuine h5py of this image's conda environment).

API: the subset of h5py the snapshot code uses -- File(name, "r" | "w") as a context manager, create_dataset(name,
data=...), file[name][:], file.attrs[key] (get / set), file.attrs.keys().
"""
from __future__ import annotations

import struct

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF
_LEAF_K, _INTERNAL_K = 4, 16


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


# ------------------------------------------------------------------------------------------------
# datatype / dataspace messages


def _dtype_message(dt: np.dtype) -> bytes:
    """Datatype message (version 1) of a NumPy dtype: little-endian ints, floats, bool (h5py's FALSE/TRUE enum),
    fixed-length byte strings."""
    dt = np.dtype(dt)
    if dt.kind == "b":
        base = _dtype_message(np.dtype("i1"))
        return (struct.pack("<BBBBI", 0x18, 2, 0, 0, 1) + base + _pad8(b"FALSE\0") + _pad8(b"TRUE\0") + b"\x00\x01")
    if dt.kind in "iu":
        return struct.pack("<BBBBIHH", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, dt.itemsize, 0, 8 * dt.itemsize)
    if dt.kind == "f":
        exp_bits, mant_bits = {2: (5, 10), 4: (8, 23), 8: (11, 52)}[dt.itemsize]
        return struct.pack("<BBBBIHHBBBBI", 0x11, 0x20, 8 * dt.itemsize - 1, 0, dt.itemsize, 0, 8 * dt.itemsize,
                           mant_bits, exp_bits, 0, mant_bits, (1 << (exp_bits - 1)) - 1)
    if dt.kind == "S":
        return struct.pack("<BBBBI", 0x13, 0x10, 0, 0, dt.itemsize)  # null-terminated, UTF-8
    raise TypeError(f"hdf5lite cannot store dtype {dt}")


def _dataspace_message(shape) -> bytes:
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", int(s)) for s in shape)


class _VlenStr:
    """A Python str attribute: stored like h5py stores it, as a variable-length UTF-8 string (a 16-byte reference
    into a global heap collection), so that h5py -- and the reference's restart, which calls .casefold() on what it
    reads back (initial_conditions.py:99) -- get a str again, not bytes."""
    dtype_message = struct.pack("<BBBBI", 0x19, 0x01, 0x01, 0, 16) + struct.pack("<BBBBIHH", 0x10, 0, 0, 0, 1, 0, 8)

    def __init__(self, text):
        self.raw = text.encode("utf-8")


def _storable(value):
    """(array in its stored dtype) for an attribute value: Python / NumPy scalars, strings, small arrays."""
    if isinstance(value, (bool, np.bool_)):
        return np.asarray(value, dtype=np.bool_)
    if isinstance(value, str):
        return _VlenStr(value)
    if isinstance(value, bytes):
        return np.asarray(value, dtype=f"S{len(value) + 1}")
    a = np.asarray(value)
    if a.dtype.kind == "U":
        raw = [s.encode("utf-8") for s in a.reshape(-1)]
        return np.asarray(raw, dtype=f"S{max(len(r) for r in raw) + 1}").reshape(a.shape)
    if a.dtype.kind == "O":
        raise TypeError(f"hdf5lite cannot store attribute value {value!r}")
    return a.astype(a.dtype.newbyteorder("<")) if a.dtype.byteorder == ">" else a


def _message(mtype: int, body: bytes, flags: int = 0) -> bytes:
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _object_header(messages) -> bytes:
    data = b"".join(messages)
    return struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(data)) + data


# ------------------------------------------------------------------------------------------------
# writer


class _AttrsW:
    def __init__(self):
        self._items = {}

    def __setitem__(self, key, value):
        self._items[str(key)] = _storable(value)

    def keys(self):
        return self._items.keys()

    def _messages(self, gcol_addr=0):
        out, index = [], 0
        for name, a in self._items.items():
            nm = name.encode("utf-8") + b"\0"
            if isinstance(a, _VlenStr):
                index += 1
                dtm, dsm = a.dtype_message, _dataspace_message(())
                data = struct.pack("<IQI", len(a.raw), gcol_addr, index)
            else:
                dtm, dsm = _dtype_message(a.dtype), _dataspace_message(a.shape)
                data = np.ascontiguousarray(a).tobytes()
            body = struct.pack("<BBHHH", 1, 0, len(nm), len(dtm), len(dsm)) + _pad8(nm) + _pad8(dtm) + _pad8(dsm) + data
            out.append(_message(0x000C, body))
        return out

    def _global_heap(self) -> bytes:
        """The collection the string attributes point into: objects 1.. in attribute order, then the free space."""
        strings = [a.raw for a in self._items.values() if isinstance(a, _VlenStr)]
        if not strings:
            return b""
        body = b"".join(struct.pack("<HHIQ", i + 1, 0, 0, len(r)) + _pad8(r) for i, r in enumerate(strings))
        total = max(4096, 16 + len(body) + 16)
        total += -total % 8
        free = total - 16 - len(body)
        return (b"GCOL" + struct.pack("<B3xQ", 1, total) + body + struct.pack("<HHIQ", 0, 0, 0, free)).ljust(total, b"\0")


class _FileW:
    def __init__(self, name):
        self._name = name
        self.attrs = _AttrsW()
        self._datasets = {}

    def create_dataset(self, name, data):
        a = np.ascontiguousarray(data)
        if a.dtype.byteorder == ">":
            a = a.astype(a.dtype.newbyteorder("<"))
        if "/" in name or not name:
            raise ValueError("hdf5lite writes datasets in the root group only")
        self._datasets[str(name)] = a

    def close(self):
        names = sorted(self._datasets)  # symbol table nodes are ordered by name
        if len(names) > 2 * _LEAF_K:
            raise ValueError(f"hdf5lite writes at most {2 * _LEAF_K} datasets")
        # local heap data: "" at offset 0, then the link names
        heap, offsets = bytearray(_pad8(b"\0")), {}
        for n in names:
            offsets[n] = len(heap)
            heap += _pad8(n.encode("utf-8") + b"\0")
        heap += struct.pack("<QQ", 1, 16)  # one free block closing the segment (next: none, size 16)
        free_off = len(heap) - 16
        # layout of the file
        root_msgs_len = len(b"".join(self.attrs._messages())) + 8 + 16
        pos = 96
        root_addr = pos
        pos += 16 + root_msgs_len
        btree_addr = pos
        pos += 24 + (2 * _INTERNAL_K + 1) * 8 + 2 * _INTERNAL_K * 8
        heap_addr = pos
        pos += 32
        heap_data_addr = pos
        pos += len(heap)
        snod_addr = pos
        pos += 8 + 2 * _LEAF_K * 40
        ds_headers, ds_addr = {}, {}
        for n in names:
            a = self._datasets[n]
            ds_addr[n] = pos
            msgs = [_message(0x0001, _dataspace_message(a.shape)), _message(0x0003, _dtype_message(a.dtype), flags=1),
                    _message(0x0005, struct.pack("<BBBB", 2, 2, 2, 0)), None]
            ds_headers[n] = msgs
            pos += 16 + sum(len(m) for m in msgs[:3]) + 8 + 24
        data_addr = {}
        for n in names:
            pos += -pos % 8
            data_addr[n] = pos
            pos += self._datasets[n].nbytes
        pos += -pos % 8
        gcol_addr, gcol = pos, self.attrs._global_heap()
        eof = pos + len(gcol)
        with open(self._name, "wb") as fh:
            fh.write(_SIG + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, _LEAF_K, _INTERNAL_K, 0)
                     + struct.pack("<QQQQ", 0, _UNDEF, eof, _UNDEF)
                     + struct.pack("<QQII", 0, root_addr, 1, 0) + struct.pack("<QQ", btree_addr, heap_addr))
            fh.write(_object_header(self.attrs._messages(gcol_addr)
                                    + [_message(0x0011, struct.pack("<QQ", btree_addr, heap_addr))]))
            assert fh.tell() == btree_addr
            node = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if names else 0, _UNDEF, _UNDEF)
            node += struct.pack("<QQQ", 0, snod_addr, offsets[names[-1]] if names else 0)
            fh.write(node.ljust(24 + (2 * _INTERNAL_K + 1) * 8 + 2 * _INTERNAL_K * 8, b"\0"))
            fh.write(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free_off, heap_data_addr) + bytes(heap))
            snod = b"SNOD" + struct.pack("<BBH", 1, 0, len(names))
            for n in names:
                snod += struct.pack("<QQII16x", offsets[n], ds_addr[n], 0, 0)
            fh.write(snod.ljust(8 + 2 * _LEAF_K * 40, b"\0"))
            for n in names:
                assert fh.tell() == ds_addr[n]
                msgs = ds_headers[n]
                msgs[3] = _message(0x0008, struct.pack("<BBQQ", 3, 1, data_addr[n], self._datasets[n].nbytes))
                fh.write(_object_header(msgs))
            for n in names:
                fh.write(b"\0" * (data_addr[n] - fh.tell()))
                fh.write(self._datasets[n].tobytes())
            fh.write(b"\0" * (gcol_addr - fh.tell()) + gcol)
            assert fh.tell() == eof


# ------------------------------------------------------------------------------------------------
# reader


class _Reader:
    def __init__(self, name):
        self.buf = np.memmap(name, dtype=np.uint8, mode="r") if _size(name) else np.zeros(0, np.uint8)
        b = self.bytes(0, 8)
        if b != _SIG:
            raise OSError(f"{name}: not an HDF5 file (or one with a user block)")
        ver = self.buf[8]
        if ver not in (0, 1):
            raise NotImplementedError(f"{name}: superblock version {ver} (hdf5lite reads the default, version-0 layout)")
        if self.buf[13] != 8 or self.buf[14] != 8:
            raise NotImplementedError("hdf5lite reads files with 8-byte offsets and lengths")
        entry = 24 + (4 if ver == 1 else 0) + 32
        self.root = self.u64(entry + 8)

    def bytes(self, off, n) -> bytes:
        return self.buf[off:off + n].tobytes()

    def u64(self, off):
        return struct.unpack_from("<Q", self.buf, off)[0]

    def messages(self, addr):
        """[(type, flags, body bytes)] of a version-1 object header, continuation blocks included."""
        ver, _, nmsg, _, size = struct.unpack_from("<BBHII", self.buf, addr)
        if ver != 1:
            raise NotImplementedError("hdf5lite reads version-1 object headers (h5py's default libver)")
        out, blocks = [], [(addr + 16, size)]
        while blocks and len(out) < nmsg:
            off, left = blocks.pop(0)
            while left >= 8 and len(out) < nmsg:
                mtype, msize, flags = struct.unpack_from("<HHB", self.buf, off)
                body = self.bytes(off + 8, msize)
                if mtype == 0x0010:
                    blocks.append(struct.unpack("<QQ", body[:16]))
                out.append((mtype, flags, body))
                off += 8 + msize
                left -= 8 + msize
        return out

    def links(self, addr):
        """{name: object header address} of an old-style group."""
        sym = [b for t, f, b in self.messages(addr) if t == 0x0011]
        if not sym:
            raise NotImplementedError("hdf5lite reads old-style groups (symbol tables) only")
        btree, heap = struct.unpack("<QQ", sym[0][:16])
        if self.bytes(heap, 4) != b"HEAP":
            raise OSError("bad local heap")
        heap_data = self.u64(heap + 24)
        out = {}

        def walk(node):
            if self.bytes(node, 4) != b"TREE":
                raise OSError("bad B-tree node")
            _, level, used = struct.unpack_from("<BBH", self.buf, node + 4)
            for i in range(used):
                child = self.u64(node + 24 + 8 + i * 16)
                if level > 0:
                    walk(child)
                    continue
                if self.bytes(child, 4) != b"SNOD":
                    raise OSError("bad symbol table node")
                nsym = struct.unpack_from("<H", self.buf, child + 6)[0]
                for s in range(nsym):
                    name_off, obj = struct.unpack_from("<QQ", self.buf, child + 8 + s * 40)
                    p = heap_data + name_off
                    end = p
                    while self.buf[end] != 0:
                        end += 1
                    out[self.bytes(p, end - p).decode("utf-8")] = obj

        walk(btree)
        return out

    # ---- datatypes
    def dtype(self, b):
        """(numpy dtype or ("vlen_str",) or ("enum_bool", base), consumed bytes)"""
        cls, ver = b[0] & 0x0F, b[0] >> 4
        bits0 = b[1]
        size = struct.unpack_from("<I", b, 4)[0]
        order = ">" if bits0 & 1 else "<"
        if cls == 0:
            return np.dtype(f"{order}{'i' if bits0 & 0x08 else 'u'}{size}"), 12
        if cls == 1:
            return np.dtype(f"{order}f{size}"), 20
        if cls == 3:
            return np.dtype(f"S{size}"), 8
        if cls == 9:
            if (bits0 & 0x0F) != 1:
                raise NotImplementedError("variable-length sequences")
            return ("vlen_str",), 8
        if cls == 8:
            nmemb = b[1] | (b[2] << 8)
            base, used = self.dtype(b[8:])
            off = 8 + used
            names = []
            for _ in range(nmemb):
                end = b.index(b"\0", off)
                names.append(b[off:end].decode())
                off = end + 1 if ver >= 3 else off + ((end - off) // 8 + 1) * 8
            values = np.frombuffer(b, dtype=base, count=nmemb, offset=off)
            if sorted(names) == ["FALSE", "TRUE"] and sorted(values.tolist()) == [0, 1]:
                return ("enum_bool", base), off + nmemb * base.itemsize
            return base, off + nmemb * base.itemsize
        raise NotImplementedError(f"HDF5 datatype class {cls}")

    @staticmethod
    def shape(b):
        ver, rank = b[0], b[1]
        off = 8 if ver == 1 else 4
        return tuple(struct.unpack_from("<Q", b, off + 8 * i)[0] for i in range(rank))

    def vlen_string(self, ref: bytes) -> str:
        length, gcol, index = struct.unpack("<IQI", ref)
        if self.bytes(gcol, 4) != b"GCOL":
            raise OSError("bad global heap collection")
        total = self.u64(gcol + 8)
        off = gcol + 16
        while off < gcol + total:
            idx, _, _, osize = struct.unpack_from("<HHIQ", self.buf, off)
            if idx == index:
                return self.bytes(off + 16, length).decode("utf-8")
            if idx == 0:
                break
            off += 16 + osize + (-osize % 8)
        raise OSError("global heap object not found")

    def value(self, dt, shape, raw: bytes):
        count = int(np.prod(shape)) if shape else 1
        if isinstance(dt, tuple) and dt[0] == "vlen_str":
            vals = [self.vlen_string(raw[16 * i:16 * i + 16]) for i in range(count)]
            return vals[0] if not shape else np.array(vals, dtype=object).reshape(shape)
        if isinstance(dt, tuple):  # enum_bool
            a = np.frombuffer(raw, dtype=dt[1], count=count).astype(np.bool_)
        else:
            a = np.frombuffer(raw, dtype=dt, count=count)
            if a.dtype.kind == "S":
                vals = [x.split(b"\0")[0].decode("utf-8") for x in a.tolist()]
                return vals[0] if not shape else np.array(vals, dtype=object).reshape(shape)
            if a.dtype.byteorder == ">":
                a = a.astype(a.dtype.newbyteorder("="))
        return a.reshape(shape).copy() if shape else a[0]

    def attributes(self, addr):
        out = {}
        for mtype, _, b in self.messages(addr):
            if mtype != 0x000C:
                continue
            ver = b[0]
            nlen, tlen, slen = struct.unpack_from("<HHH", b, 2)
            off = 8 if ver < 3 else 9
            pad = (lambda n: n + (-n % 8)) if ver == 1 else (lambda n: n)
            name = b[off:off + nlen].split(b"\0")[0].decode("utf-8")
            off += pad(nlen)
            dt, _ = self.dtype(b[off:off + tlen])
            off += pad(tlen)
            shape = self.shape(b[off:off + slen])
            off += pad(slen)
            out[name] = self.value(dt, shape, b[off:])
        return out

    def dataset(self, addr):
        dt = shape = layout = None
        for mtype, _, b in self.messages(addr):
            if mtype == 0x0003:
                dt, _ = self.dtype(b)
            elif mtype == 0x0001:
                shape = self.shape(b)
            elif mtype == 0x0008:
                layout = b
        if dt is None or shape is None or layout is None or isinstance(dt, tuple):
            raise NotImplementedError("hdf5lite reads numeric datasets")
        ver, cls = layout[0], layout[1]
        if ver != 3 or cls not in (0, 1):
            raise NotImplementedError("hdf5lite reads contiguous or compact datasets (layout message version 3)")
        count = int(np.prod(shape)) if shape else 1
        if cls == 1:
            address, _ = struct.unpack_from("<QQ", layout, 2)
            if address == _UNDEF:
                return np.zeros(shape, dtype=dt)
            a = np.frombuffer(self.buf, dtype=dt, count=count, offset=address)
        else:
            size = struct.unpack_from("<H", layout, 2)[0]
            a = np.frombuffer(layout[4:4 + size], dtype=dt, count=count)
        return a.reshape(shape)


def _size(name):
    import os

    return os.path.getsize(name)


class _DatasetR:
    def __init__(self, array):
        self._a = array
        self.shape, self.dtype = array.shape, array.dtype

    def __getitem__(self, key):
        return np.array(self._a[key])


class _FileR:
    def __init__(self, name):
        self._r = _Reader(name)
        self._links = self._r.links(self._r.root)
        self.attrs = self._r.attributes(self._r.root)

    def __getitem__(self, name):
        return _DatasetR(self._r.dataset(self._links[name.strip("/")]))

    def keys(self):
        return self._links.keys()

    def close(self):
        self._r = None


class File:
    """h5py.File look-alike for the snapshot layout (see the module docstring)."""

    def __init__(self, name, mode="r"):
        if mode not in ("r", "w"):
            raise ValueError("hdf5lite.File modes: 'r' or 'w'")
        self._impl = _FileR(name) if mode == "r" else _FileW(name)
        self.attrs = self._impl.attrs

    def create_dataset(self, name, data):
        return self._impl.create_dataset(name, data)

    def __getitem__(self, name):
        return self._impl[name]

    def keys(self):
        return self._impl.keys()

    def close(self):
        self._impl.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
