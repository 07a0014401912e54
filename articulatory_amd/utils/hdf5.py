"""Minimal HDF5 reader / writer for the reference's feature dumps and statistics files (``read_hdf5`` / ``write_hdf5`` of
articulatory/utils/utils.py:83-143: plain datasets under the root group — "feats", "wave", "mean", "scale" — written by
``h5py.File.create_dataset(name, data=array)``).  h5py / libhdf5 are not dependencies of this package; when h5py is importable it is used,
otherwise this module parses the file format itself (HDF5 File Format Specification 3.0):

  read   superblock version 0 / 1, version-1 object headers (with continuation blocks), old-style groups (symbol table: v1 B-tree +
         local heap + symbol-table nodes; nested groups by path), dataspace v1 / v2, little-endian IEEE float and fixed-point integer
         datatypes, compact / contiguous / chunked (v1 B-tree index, no filters) layouts
  write  the same subset: superblock 0, one root group, contiguous float32 / float64 / int32 / int64 datasets

NOT covered (raises): superblock 2 / 3 files whose groups use link messages and fractal heaps (libver="latest"), compressed / filtered
chunks, strings and compound types.  No h5py-written file exists in this image to test against (none ships with the reference and the
library is absent): the parser follows the published specification and is exercised on files of its own writer — see
tests/test_hdf5.py; that compatibility claim is unverified against libhdf5.
"""
import os
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIG = b"\x89HDF\r\n\x1a\n"


class HDF5Error(RuntimeError):
    pass


# ------------------------------------------------------------------------------------------------ reading
class _File:
    def __init__(self, path):
        with open(path, "rb") as f:
            self.b = f.read()
        if self.b[:8] != SIG:
            raise HDF5Error(f"{path}: not an HDF5 file (signature at offset 0 expected)")
        ver = self.b[8]
        if ver not in (0, 1):
            raise HDF5Error(f"{path}: superblock version {ver} (new-style groups, libver='latest') is not supported by the built-in reader")
        if self.b[13] != 8 or self.b[14] != 8:
            raise HDF5Error("only 8-byte offsets and lengths are supported")
        off = 24 + (4 if ver == 1 else 0)
        self.base, _free, _eof, _drv = struct.unpack_from("<4Q", self.b, off)
        ste = off + 32
        # root symbol table entry: link name offset, object header address, cache type, reserved, scratch (b-tree, heap)
        _, self.root_header, cache, _, btree, heap = struct.unpack_from("<QQIIQQ", self.b, ste)
        self.root = (btree, heap) if cache == 1 else self._group_of(self.root_header)

    def messages(self, addr):
        """[(type, flags, bytes)] of a version-1 object header at ``addr``, continuation blocks followed."""
        a = self.base + addr
        if self.b[a:a + 4] == b"OHDR":
            raise HDF5Error("version-2 object headers are not supported by the built-in reader")
        ver, _, nmsg, _ref, size = struct.unpack_from("<BBHII", self.b, a)
        if ver != 1:
            raise HDF5Error(f"object header version {ver} at {addr}")
        blocks = [(a + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            p, left = blocks.pop(0)
            while left >= 8 and len(out) < nmsg:
                mtype, msize, flags = struct.unpack_from("<HHB", self.b, p)
                data = self.b[p + 8:p + 8 + msize]
                if mtype == 0x0010:  # continuation
                    o, n = struct.unpack_from("<QQ", data)
                    blocks.append((self.base + o, n))
                out.append((mtype, flags, data))
                p += 8 + msize
                left -= 8 + msize
        return out

    def _group_of(self, header_addr):
        for mtype, _, data in self.messages(header_addr):
            if mtype == 0x0011:
                return struct.unpack_from("<QQ", data)
        raise HDF5Error("not an old-style group (no symbol table message): files written with libver='latest' are not supported")

    def _heap_name(self, heap, off):
        a = self.base + heap
        if self.b[a:a + 4] != b"HEAP":
            raise HDF5Error("local heap signature missing")
        (seg,) = struct.unpack_from("<Q", self.b, a + 24)
        s = self.base + seg + off
        return self.b[s:self.b.index(b"\0", s)].decode()

    def entries(self, group):
        """{name: object header address} of an old-style group."""
        btree, heap = group
        found = {}

        def walk(addr):
            a = self.base + addr
            sig = self.b[a:a + 4]
            if sig == b"TREE":
                ntype, _level, used = struct.unpack_from("<BBH", self.b, a + 4)
                if ntype != 0:
                    raise HDF5Error("group B-tree expected")
                for i in range(used):
                    (child,) = struct.unpack_from("<Q", self.b, a + 24 + 8 + i * 16)
                    walk(child)
            elif sig == b"SNOD":
                (n,) = struct.unpack_from("<H", self.b, a + 6)
                for i in range(n):
                    name_off, hdr = struct.unpack_from("<QQ", self.b, a + 8 + i * 40)
                    found[self._heap_name(heap, name_off)] = hdr
            else:
                raise HDF5Error(f"unexpected node signature {sig!r} in a group")

        walk(btree)
        return found

    def find(self, path):
        group, parts = self.root, [p for p in path.split("/") if p]
        for i, part in enumerate(parts):
            ent = self.entries(group)
            if part not in ent:
                return None
            if i + 1 == len(parts):
                return ent[part]
            group = self._group_of(ent[part])
        return None

    def dataset(self, header_addr):
        shape = dtype = layout = None
        for mtype, _, d in self.messages(header_addr):
            if mtype == 0x0001:  # dataspace
                ver, rank, flags = struct.unpack_from("<BBB", d)
                o = 8 if ver == 1 else 4
                shape = struct.unpack_from(f"<{rank}Q", d, o)
            elif mtype == 0x0003:  # datatype
                cls, b0 = d[0] & 0x0F, d[1]
                (size,) = struct.unpack_from("<I", d, 4)
                if b0 & 1:
                    raise HDF5Error("big-endian datasets are not supported")
                if cls == 1:
                    dtype = {2: np.float16, 4: np.float32, 8: np.float64}.get(size)
                elif cls == 0:
                    dtype = np.dtype(("i" if b0 & 8 else "u") + str(size))
                if dtype is None:
                    raise HDF5Error(f"datatype class {cls} size {size} is not supported")
            elif mtype == 0x0008:
                layout = d
            elif mtype == 0x000B:
                raise HDF5Error("filtered (compressed) datasets are not supported by the built-in reader")
        if shape is None or dtype is None or layout is None:
            raise HDF5Error("not a dataset (dataspace / datatype / layout message missing)")
        dtype = np.dtype(dtype).newbyteorder("<")
        n = int(np.prod(shape)) if len(shape) else 1
        if layout[0] != 3:
            raise HDF5Error(f"data layout message version {layout[0]} is not supported")
        cls = layout[1]
        if cls == 0:  # compact
            (sz,) = struct.unpack_from("<H", layout, 2)
            return np.frombuffer(layout[4:4 + sz], dtype, n).reshape(shape).copy()
        if cls == 1:  # contiguous
            addr, sz = struct.unpack_from("<QQ", layout, 2)
            if addr == UNDEF:
                return np.zeros(shape, dtype)
            return np.frombuffer(self.b, dtype, n, self.base + addr).reshape(shape).copy()
        if cls == 2:  # chunked, v1 B-tree
            rank1 = layout[2]
            (btree,) = struct.unpack_from("<Q", layout, 3)
            cdims = struct.unpack_from(f"<{rank1}I", layout, 11)[:-1]
            out = np.zeros(shape, dtype)

            def walk(addr):
                a = self.base + addr
                if self.b[a:a + 4] != b"TREE":
                    raise HDF5Error("chunk B-tree signature missing")
                _ntype, level, used = struct.unpack_from("<BBH", self.b, a + 4)
                ksz = 8 + 8 * rank1
                p = a + 24
                for _ in range(used):
                    csize, _mask = struct.unpack_from("<II", self.b, p)
                    offs = struct.unpack_from(f"<{rank1}Q", self.b, p + 8)[:-1]
                    (child,) = struct.unpack_from("<Q", self.b, p + ksz)
                    if level > 0:
                        walk(child)
                    else:
                        chunk = np.frombuffer(self.b, dtype, csize // dtype.itemsize, self.base + child).reshape(cdims)
                        sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, shape))
                        out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]
                    p += ksz + 8

            if btree != UNDEF:
                walk(btree)
            return out
        raise HDF5Error(f"layout class {cls} is not supported")


def _h5py():
    try:
        import h5py

        return h5py
    except ImportError:
        return None


def read_hdf5(hdf5_name, hdf5_path):
    """Dataset ``hdf5_path`` of file ``hdf5_name`` as a numpy array (reference: utils.py:83-109)."""
    if not os.path.exists(hdf5_name):
        raise FileNotFoundError(f"There is no such a hdf5 file ({hdf5_name}).")
    h5py = _h5py()
    if h5py is not None:
        with h5py.File(hdf5_name, "r") as f:
            if hdf5_path not in f:
                raise KeyError(f"There is no such a data in hdf5 file. ({hdf5_path})")
            return f[hdf5_path][()]
    f = _File(hdf5_name)
    addr = f.find(hdf5_path)
    if addr is None:
        raise KeyError(f"There is no such a data in hdf5 file. ({hdf5_path})")
    return f.dataset(addr)


def list_hdf5(hdf5_name):
    """Names under the root group."""
    h5py = _h5py()
    if h5py is not None:
        with h5py.File(hdf5_name, "r") as f:
            return sorted(f.keys())
    f = _File(hdf5_name)
    return sorted(f.entries(f.root))


# ------------------------------------------------------------------------------------------------ writing
def _msg(mtype, data, flags=0):
    data = data + b"\0" * (-len(data) % 8)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _datatype(dt):
    dt = np.dtype(dt)
    if dt.kind == "f":
        e, m = {4: (8, 23), 8: (11, 52)}[dt.itemsize]
        bits = dt.itemsize * 8
        return struct.pack("<BBBBI", 0x11, 0x20, bits - 1, 0, dt.itemsize) + struct.pack("<HHBBBBI", 0, bits, m, e, 0, m, (1 << (e - 1)) - 1)
    if dt.kind in "iu":
        return struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, dt.itemsize) + struct.pack("<HH", 0, dt.itemsize * 8)
    raise HDF5Error(f"cannot write dtype {dt}")


def _object_header(msgs):
    body = b"".join(msgs)
    return struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body


def write_hdf5(hdf5_name, hdf5_path, write_data, is_overwrite=True):
    """Add (or replace) dataset ``hdf5_path`` in file ``hdf5_name`` (reference: utils.py:112-143).  Without h5py the whole file is
    rewritten from its datasets (root-level names only)."""
    write_data = np.asarray(write_data)
    folder = os.path.dirname(hdf5_name)
    if folder:
        os.makedirs(folder, exist_ok=True)
    h5py = _h5py()
    if h5py is not None:
        with h5py.File(hdf5_name, "a") as f:
            if hdf5_path in f:
                if not is_overwrite:
                    raise KeyError("Dataset in hdf5 file already exists. if you want to overwrite, please set is_overwrite = True.")
                del f[hdf5_path]
            f.create_dataset(hdf5_path, data=write_data)
        return
    name = hdf5_path.strip("/")
    if "/" in name:
        raise HDF5Error("the built-in writer stores root-level datasets only")
    data = {}
    if os.path.exists(hdf5_name):
        for k in list_hdf5(hdf5_name):
            data[k] = read_hdf5(hdf5_name, k)
        if name in data and not is_overwrite:
            raise KeyError("Dataset in hdf5 file already exists. if you want to overwrite, please set is_overwrite = True.")
    if write_data.dtype == np.float16 or write_data.dtype.kind not in "fiu":
        raise HDF5Error(f"cannot write dtype {write_data.dtype}")
    data[name] = np.ascontiguousarray(write_data).astype(write_data.dtype.newbyteorder("<"))
    write_file(hdf5_name, data)


def write_file(path, datasets):
    """One file from {root-level name: array}: superblock 0, symbol-table root group, contiguous little-endian datasets."""
    names = sorted(datasets, key=lambda s: s.encode())
    leaf_k = max(4, (len(names) + 1) // 2)
    # local heap data: "" at offset 0, then the names, each null-terminated and padded to 8 bytes
    heap, name_off = bytearray(8), {}
    for nm in names:
        name_off[nm] = len(heap)
        raw = nm.encode() + b"\0"
        heap += raw + b"\0" * (-len(raw) % 8)
    pos = 96
    root_hdr_at = pos
    pos += 16 + 24                       # root object header + symbol table message
    btree_at = pos
    pos += 24 + (2 * 16 + 1) * 8 + 2 * 16 * 8
    heap_at = pos
    pos += 32
    heap_data_at = pos
    pos += len(heap)
    snod_at = pos
    pos += 8 + 2 * leaf_k * 40
    headers, blobs = {}, {}
    for nm in names:
        arr = np.asarray(datasets[nm])
        msgs_len = sum(len(m) for m in (_msg(1, b"\0" * (8 + 8 * arr.ndim)), _msg(3, _datatype(arr.dtype)), _msg(8, b"\0" * 18)))
        headers[nm] = pos
        pos += 16 + msgs_len
    for nm in names:
        pos += -pos % 8
        blobs[nm] = pos
        pos += np.asarray(datasets[nm]).nbytes
    eof = pos
    out = bytearray(eof)
    out[0:8] = SIG
    struct.pack_into("<8B", out, 8, 0, 0, 0, 0, 0, 8, 8, 0)
    struct.pack_into("<HHI", out, 16, leaf_k, 16, 0)
    struct.pack_into("<4Q", out, 24, 0, UNDEF, eof, UNDEF)
    struct.pack_into("<QQIIQQ", out, 56, 0, root_hdr_at, 1, 0, btree_at, heap_at)
    hdr = _object_header([_msg(0x0011, struct.pack("<QQ", btree_at, heap_at))])
    out[root_hdr_at:root_hdr_at + len(hdr)] = hdr
    # B-tree: one leaf-level node with one child (the symbol node); keys: "" and the largest name
    struct.pack_into("<4sBBHQQ", out, btree_at, b"TREE", 0, 0, 1 if names else 0, UNDEF, UNDEF)
    struct.pack_into("<QQQ", out, btree_at + 24, 0, snod_at, name_off[names[-1]] if names else 0)
    struct.pack_into("<4sB3xQQQ", out, heap_at, b"HEAP", 0, len(heap), 1, heap_data_at)  # free-list head 1 = H5HL_FREE_NULL (none)
    out[heap_data_at:heap_data_at + len(heap)] = heap
    struct.pack_into("<4sBBH", out, snod_at, b"SNOD", 1, 0, len(names))
    for i, nm in enumerate(names):
        struct.pack_into("<QQII16x", out, snod_at + 8 + i * 40, name_off[nm], headers[nm], 0, 0)
        arr = np.ascontiguousarray(np.asarray(datasets[nm]))
        arr = arr.astype(arr.dtype.newbyteorder("<"))
        space = struct.pack("<BBB5x", 1, arr.ndim, 0) + struct.pack(f"<{arr.ndim}Q", *arr.shape)
        layout = struct.pack("<BBQQ", 3, 1, blobs[nm], arr.nbytes)
        hdr = _object_header([_msg(1, space), _msg(3, _datatype(arr.dtype)), _msg(8, layout)])
        out[headers[nm]:headers[nm] + len(hdr)] = hdr
        out[blobs[nm]:blobs[nm] + arr.nbytes] = arr.tobytes()
    with open(path, "wb") as f:
        f.write(out)
