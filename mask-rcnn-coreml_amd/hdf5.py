"""Read-only HDF5 subset reader for Keras ``weights.h5`` files.

The reference's converter takes its weights as a Keras HDF5 file (``ConvertCommand.swift:10,52-60``
mounts it as ``model/weights.h5``; ``Conversion/task.py:166`` hands it to the third-party model
class).  This image has no h5py for the system interpreter, and the importer must not depend on one,
so this module parses the part of the HDF5 file format that ``keras.Model.save_weights`` /
``h5py`` (default ``libver='earliest'``) emit:

  * superblock version 0/1,
  * old-style groups: symbol-table message (0x0011) → v1 B-tree ("TREE") → symbol nodes ("SNOD")
    with names in a local heap ("HEAP"),
  * version-1 object headers with continuation blocks (0x0010),
  * dataspace (0x0001) v1/v2, datatype (0x0003) classes 0 (integer), 1 (IEEE float), 3 (fixed
    string), data layout (0x0008) v3 compact / contiguous (and v1/v2 contiguous),
  * attribute messages (0x000C) v1–v3 with the same datatypes (variable-length strings are returned
    via the global heap, "GCOL").

Anything else (new-style groups / v2 object headers from ``libver='latest'``, chunked or filtered
datasets) raises ``HDF5FormatError`` naming the feature — there is no silent partial read.
Format reference: the published "HDF5 File Format Specification Version 2.0" (The HDF Group).
"""
from __future__ import annotations

import struct
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"


class HDF5FormatError(ValueError):
    pass


class _Datatype:
    def __init__(self, np_dtype: Optional[np.dtype], size: int, vlen_str: bool = False):
        self.np_dtype, self.size, self.vlen_str = np_dtype, size, vlen_str


class Dataset:
    def __init__(self, f: "File", name: str, shape: Tuple[int, ...], dt: _Datatype, layout):
        self._f, self.name, self.shape, self._dt, self._layout = f, name, shape, dt, layout

    @property
    def dtype(self) -> np.dtype:
        return self._dt.np_dtype

    def read(self) -> np.ndarray:
        if self._dt.np_dtype is None:
            raise HDF5FormatError(f"{self.name}: dataset datatype not supported")
        n = int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1
        nbytes = n * self._dt.size
        kind, a, b = self._layout
        if kind == "compact":
            raw = a
        else:
            addr, size = a, b
            if addr == self._f._undef:              # never written: HDF5 fill value (zeros)
                return np.zeros(self.shape, dtype=self._dt.np_dtype)
            raw = self._f._bytes(addr, nbytes)
        if len(raw) < nbytes:
            raise HDF5FormatError(f"{self.name}: {len(raw)} bytes stored, {nbytes} expected")
        return np.frombuffer(raw, dtype=self._dt.np_dtype, count=n).reshape(self.shape).copy()


class Group:
    def __init__(self, f: "File", name: str, links: Dict[str, int], attrs: Dict[str, object]):
        self._f, self.name, self._links, self.attrs = f, name, links, attrs

    def keys(self) -> List[str]:
        return list(self._links)

    def __contains__(self, k: str) -> bool:
        return k in self._links

    def __getitem__(self, path: str):
        if path.count("/") > 64:
            raise KeyError(path)
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group) or part not in node._links:
                raise KeyError(f"{path!r} not found under {self.name!r}")
            child = (node.name.rstrip("/") + "/" + part)
            node = node._f._object(node._links[part], child)
        return node

    def visit_datasets(self, _depth: int = 0) -> Iterator[Dataset]:
        if _depth > 32:
            raise HDF5FormatError(f"{self.name}: groups nested deeper than 32 levels (link cycle in a corrupt file?)")
        for k in self._links:
            o = self[k]
            if isinstance(o, Group):
                yield from o.visit_datasets(_depth + 1)
            else:
                yield o


class File(Group):
    def __init__(self, path: str):
        with open(path, "rb") as fh:
            self._buf = fh.read()
        b = self._buf
        if b[:8] != SIGNATURE:
            raise HDF5FormatError(f"{path}: not an HDF5 file (signature at offset 0 missing)")
        ver = b[8]
        if ver not in (0, 1):
            raise HDF5FormatError(f"{path}: superblock version {ver} (libver='latest' files) not supported; "
                                  "re-save with h5py's default libver")
        self._O, self._L = b[13], b[14]
        if self._O not in (4, 8) or self._L not in (4, 8):
            raise HDF5FormatError(f"{path}: offset/length sizes {self._O}/{self._L} not supported")
        self._undef = (1 << (8 * self._O)) - 1
        p = 24 + (4 if ver == 1 else 0)
        self._base = self._off(p)
        p += 4 * self._O                       # base, free-space, end-of-file, driver-info addresses
        root_header = self._off(p + self._O)   # symbol table entry: link-name offset, object header address
        self._cache: Dict[int, object] = {}
        root = self._object(root_header, "/")
        if not isinstance(root, Group):
            raise HDF5FormatError(f"{path}: root object is not a group")
        Group.__init__(self, self, "/", root._links, root.attrs)

    # ---- primitive reads ----------------------------------------------------------------------
    def _bytes(self, addr: int, n: int) -> bytes:
        a = self._base + addr
        if a < 0 or a + n > len(self._buf):
            raise HDF5FormatError(f"address {addr:#x}+{n} beyond end of file")
        return self._buf[a:a + n]

    def _uint(self, p: int, n: int) -> int:
        return int.from_bytes(self._buf[p:p + n], "little")

    def _off(self, p: int) -> int:
        return self._uint(p, self._O)

    def _len(self, p: int) -> int:
        return self._uint(p, self._L)

    # ---- object headers -----------------------------------------------------------------------
    def _messages(self, addr: int) -> List[Tuple[int, int, int]]:
        """[(type, absolute file position of the body, size)] of a version-1 object header."""
        p = self._base + addr
        if p < 0 or p + 16 > len(self._buf):
            raise HDF5FormatError(f"object header address {addr:#x} beyond end of file")
        if self._buf[p:p + 4] == b"OHDR":
            raise HDF5FormatError("version-2 object header (libver='latest') not supported")
        if self._buf[p] != 1:
            raise HDF5FormatError(f"object header version {self._buf[p]} at {addr:#x} not supported")
        n_msgs = self._uint(p + 2, 2)
        hdr_size = self._uint(p + 8, 4)
        blocks = [(p + 16, hdr_size)]
        out = []
        seen_blocks = 0
        while blocks and len(out) < n_msgs:
            seen_blocks += 1
            if seen_blocks > 4096:
                raise HDF5FormatError("object header continuation chain too long (corrupt file?)")
            q, size = blocks.pop(0)
            end = q + size
            while q + 8 <= end and len(out) < n_msgs:
                mtype, msize = self._uint(q, 2), self._uint(q + 2, 2)
                body = q + 8
                if mtype == 0x0010:
                    blocks.append((self._base + self._off(body), self._len(body + self._O)))
                out.append((mtype, body, msize))
                q = body + msize
        return out

    @staticmethod
    def _find(msgs, mtype: int, name: str) -> int:
        for t, body, _ in msgs:
            if t == mtype:
                return body
        raise HDF5FormatError(f"{name}: object header lacks message {mtype:#06x}")

    def _object(self, addr: int, name: str):
        if addr in self._cache:
            return self._cache[addr]
        msgs = self._messages(addr)
        types = {t for t, _, _ in msgs}
        if 0x000B in types:
            raise HDF5FormatError(f"{name}: filtered (compressed) datasets are not supported")
        if 0x0002 in types or 0x0006 in types:
            raise HDF5FormatError(f"{name}: new-style group (link messages) not supported")
        attrs: Dict[str, object] = {}
        for t, body, size in msgs:
            if t == 0x000C:
                k, v = self._attribute(body)
                attrs[k] = v
        if 0x0011 in types:
            body = self._find(msgs, 0x0011, name)
            links = self._group_links(self._off(body), self._off(body + self._O))
            obj: object = Group(self, name, links, attrs)
        elif 0x0008 in types:
            shape = self._dataspace(self._find(msgs, 0x0001, name))
            dt = self._datatype(self._find(msgs, 0x0003, name))[0]
            layout = self._layout(self._find(msgs, 0x0008, name), name)
            obj = Dataset(self, name, shape, dt, layout)
            obj.attrs = attrs
        else:
            raise HDF5FormatError(f"{name}: object is neither an old-style group nor a dataset")
        self._cache[addr] = obj
        return obj

    # ---- groups -------------------------------------------------------------------------------
    def _heap_string(self, heap_data: int, off: int) -> str:
        a = self._base + heap_data + off
        e = self._buf.index(b"\0", a)
        return self._buf[a:e].decode("utf-8")

    def _group_links(self, btree: int, heap: int) -> Dict[str, int]:
        h = self._base + heap
        if self._buf[h:h + 4] != b"HEAP":
            raise HDF5FormatError(f"local heap signature missing at {heap:#x}")
        heap_data = self._off(h + 8 + 2 * self._L)
        links: Dict[str, int] = {}
        self._btree_walk(btree, heap_data, links)
        return links

    def _btree_walk(self, addr: int, heap_data: int, links: Dict[str, int], depth: int = 0) -> None:
        if depth > 16:
            raise HDF5FormatError("group B-tree deeper than 16 levels (corrupt file?)")
        p = self._base + addr
        if p < 0 or p + 8 > len(self._buf):
            raise HDF5FormatError(f"B-tree node address {addr:#x} beyond end of file")
        sig = self._buf[p:p + 4]
        if sig == b"TREE":
            if self._buf[p + 4] != 0:
                raise HDF5FormatError("B-tree node is not a group node")
            n = self._uint(p + 6, 2)
            q = p + 8 + 2 * self._O
            for i in range(n):
                q += self._L                                  # key i
                self._btree_walk(self._off(q), heap_data, links, depth + 1)
                q += self._O
        elif sig == b"SNOD":
            n = self._uint(p + 6, 2)
            q = p + 8
            esz = 2 * self._O + 24
            for i in range(n):
                links[self._heap_string(heap_data, self._off(q))] = self._off(q + self._O)
                q += esz
        else:
            raise HDF5FormatError(f"unexpected node signature {sig!r} at {addr:#x}")

    # ---- dataset description messages -----------------------------------------------------------
    def _dataspace(self, p: int) -> Tuple[int, ...]:
        ver, rank, flags = self._buf[p], self._buf[p + 1], self._buf[p + 2]
        if ver == 1:
            q = p + 8
        elif ver == 2:
            q = p + 4
        else:
            raise HDF5FormatError(f"dataspace message version {ver} not supported")
        return tuple(self._len(q + i * self._L) for i in range(rank))

    def _datatype(self, p: int) -> Tuple[_Datatype, int]:
        """→ (datatype, number of bytes the message occupies)."""
        cls, ver = self._buf[p] & 0x0F, self._buf[p] >> 4
        bits0 = self._buf[p + 1]
        size = self._uint(p + 4, 4)
        bo = ">" if bits0 & 1 else "<"
        if cls == 0:                                             # fixed-point
            signed = "i" if bits0 & 0x08 else "u"
            return _Datatype(np.dtype(f"{bo}{signed}{size}") if size in (1, 2, 4, 8) else None, size), 8 + 4
        if cls == 1:                                             # floating-point (IEEE layouts only)
            return _Datatype(np.dtype(f"{bo}f{size}") if size in (2, 4, 8) else None, size), 8 + 12
        if cls == 3:                                             # fixed-length string
            return _Datatype(np.dtype(f"S{size}"), size), 8
        if cls == 9:                                             # variable-length
            is_str = (bits0 & 0x0F) == 1
            _, inner = self._datatype(p + 8)
            return _Datatype(None, size, vlen_str=is_str), 8 + inner
        return _Datatype(None, size), 8

    def _layout(self, p: int, name: str):
        ver = self._buf[p]
        if ver == 3:
            cls = self._buf[p + 1]
            if cls == 0:
                n = self._uint(p + 2, 2)
                return ("compact", self._buf[p + 4:p + 4 + n], n)
            if cls == 1:
                return ("contiguous", self._off(p + 2), self._len(p + 2 + self._O))
            raise HDF5FormatError(f"{name}: chunked datasets are not supported (Keras writes contiguous ones)")
        if ver in (1, 2):
            rank, cls = self._buf[p + 1], self._buf[p + 2]
            if cls != 1:
                raise HDF5FormatError(f"{name}: layout class {cls} (message v{ver}) not supported")
            return ("contiguous", self._off(p + 8), -1)
        raise HDF5FormatError(f"{name}: data layout message version {ver} not supported")

    # ---- attributes ---------------------------------------------------------------------------
    def _attribute(self, p: int) -> Tuple[str, object]:
        ver = self._buf[p]
        if ver not in (1, 2, 3):
            raise HDF5FormatError(f"attribute message version {ver} not supported")
        nsz, tsz, ssz = self._uint(p + 2, 2), self._uint(p + 4, 2), self._uint(p + 6, 2)
        q = p + 8 + (1 if ver == 3 else 0)
        pad = (lambda n: (n + 7) & ~7) if ver == 1 else (lambda n: n)
        name = self._buf[q:q + nsz].split(b"\0", 1)[0].decode("utf-8")
        q += pad(nsz)
        dt, _ = self._datatype(q)
        q += pad(tsz)
        shape = self._dataspace(q) if ssz >= 4 else ()
        if ssz >= 4 and self._buf[q] == 2 and self._buf[q + 3] == 2:      # null dataspace
            return name, None
        q += pad(ssz)
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if dt.vlen_str:
            vals = []
            for i in range(n):
                e = q + i * (4 + self._O + 4)
                vals.append(self._global_heap_object(self._off(e + 4), self._uint(e + 4 + self._O, 4)))
            arr = np.array(vals, dtype=object).reshape(shape) if shape else vals[0]
            return name, arr
        if dt.np_dtype is None:
            return name, None
        arr = np.frombuffer(self._buf, dtype=dt.np_dtype, count=n, offset=q).reshape(shape).copy()
        return name, (arr if shape else arr.reshape(())[()])

    def _global_heap_object(self, addr: int, index: int) -> bytes:
        p = self._base + addr
        if self._buf[p:p + 4] != b"GCOL":
            raise HDF5FormatError(f"global heap signature missing at {addr:#x}")
        end = p + self._len(p + 8)
        q = p + 8 + self._L
        while q + 8 + self._L <= end:
            idx = self._uint(q, 2)
            size = self._len(q + 8)
            if idx == 0:
                break
            if idx == index:
                return self._buf[q + 8 + self._L:q + 8 + self._L + size]
            q += 8 + self._L + ((size + 7) & ~7)
        raise HDF5FormatError(f"global heap object {index} not found in collection at {addr:#x}")


def read_keras_weights(path: str) -> Dict[str, np.ndarray]:
    """All datasets of a Keras ``save_weights`` file keyed ``<layer>/<weight>`` (the last two path
    components with the TensorFlow ``:0`` suffix dropped), whatever the nesting: plain layers are
    stored at ``/<layer>/<layer>/kernel:0``, layers of a nested model (``rpn_model``) at
    ``/rpn_model/<layer>/kernel:0``."""
    out: Dict[str, np.ndarray] = {}
    for ds in File(path).visit_datasets():
        parts = [p for p in ds.name.split("/") if p]
        if len(parts) < 2:
            continue
        key = parts[-2] + "/" + parts[-1].split(":")[0]
        if key in out:
            raise HDF5FormatError(f"{path}: weight {key} appears twice ({ds.name})")
        out[key] = ds.read()
    return out
