"""Minimal HDF5 reader / writer for Keras weight files (reference FRVSR.py:88,193: ``load_weights`` /
``save_weights`` on ``.h5`` files), used when h5py is not installed.

Only the subset of the HDF5 file format that ``h5py`` with its default ("earliest") format writes
for ``Model.save_weights`` is handled, straight from the HDF5 File Format Specification v1/v2:

* superblock version 0 (searched at offsets 0, 512, 1024, ... as the specification prescribes;
  addresses are relative to its base address);
* "old style" groups: version-1 object headers with a symbol-table message, version-1 B-tree
  nodes (``TREE``), symbol-table nodes (``SNOD``) and a local heap (``HEAP``) for the link names;
* datasets with a version-1/2 dataspace, fixed-point / IEEE-float / fixed-length-string datatypes
  and a layout message of class contiguous (versions 1-3) or compact (version 3); chunked,
  compressed or variable-length data raise ``NotImplementedError``;
* attribute messages of versions 1-3 (fixed-size values; variable-length ones such as Keras's
  ``backend`` / ``keras_version`` strings are skipped, the loader does not need them), header
  continuation blocks included.

The reader is checked against a libhdf5-written file that ships with scipy's test data (a MATLAB
v7.3 file) and against files made by the writer below; no Keras- or h5py-written file is available
in this environment, so compatibility with those is PARITY UNPINNED and stated as such in
weights_io.py.  The API mirrors the slice of h5py that weights_io uses: ``File(path)``,
``obj.attrs`` (dict), ``group[name]`` with ``/``-separated paths, ``group.keys()``,
``dataset[()]`` / ``dataset.shape`` / ``dataset.dtype``.
"""
import struct

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(ValueError):
    pass


def _pad8(n):
    return (n + 7) & ~7


# =============================================================================== reader
class _Reader:
    def __init__(self, data):
        self.b = data
        pos = 0
        while True:                                   # spec: superblock at 0, 512, 1024, 2048, ...
            if self.b[pos:pos + 8] == SIGNATURE:
                break
            pos = 512 if pos == 0 else pos * 2
            if pos + 8 > len(self.b):
                raise H5Error("not an HDF5 file (signature not found)")
        sb = self.b[pos:]
        if sb[8] != 0:
            raise NotImplementedError(f"HDF5 superblock version {sb[8]} (only version 0, h5py's default, is supported)")
        if sb[13] != 8 or sb[14] != 8:
            raise NotImplementedError("only 8-byte offsets and lengths are supported")
        self.base = struct.unpack_from("<Q", sb, 24)[0]
        # root group symbol table entry follows the four addresses
        self.root_header = struct.unpack_from("<Q", sb, 56 + 8)[0]

    # -- low level ---------------------------------------------------------------
    def at(self, addr, n):
        a = self.base + addr
        if addr == UNDEF or a + n > len(self.b):
            raise H5Error("address outside the file")
        return self.b[a:a + n]

    def messages(self, addr):
        """[(type, flags, payload bytes)] of a version-1 object header, continuations followed."""
        head = self.at(addr, 16)
        if head[0] != 1:
            raise NotImplementedError(f"object header version {head[0]} (only version 1 is supported)")
        nmsg, = struct.unpack_from("<H", head, 2)
        size, = struct.unpack_from("<I", head, 8)
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            baddr, bsize = blocks.pop(0)
            blk = self.at(baddr, bsize)
            p = 0
            while p + 8 <= bsize and len(out) < nmsg:
                mtype, msize, mflags = struct.unpack_from("<HHB", blk, p)
                body = blk[p + 8:p + 8 + msize]
                p += 8 + msize
                if mtype == 0x0010:                   # continuation
                    coff, clen = struct.unpack_from("<QQ", body, 0)
                    blocks.append((coff, clen))
                out.append((mtype, mflags, body))
        return out

    def heap_name(self, heap_addr, offset):
        h = self.at(heap_addr, 32)
        if h[:4] != b"HEAP":
            raise H5Error("bad local heap signature")
        seg_size, _, seg_addr = struct.unpack_from("<QQQ", h, 8)
        seg = self.at(seg_addr, seg_size)
        end = seg.index(b"\x00", offset)
        return seg[offset:end].decode("utf-8")

    def group_links(self, btree_addr, heap_addr):
        """{name: object header address} of an old-style group."""
        links = {}

        def walk(addr):
            node = self.at(addr, 24)
            if node[:4] == b"SNOD":
                nsym, = struct.unpack_from("<H", node, 6)
                ent = self.at(addr + 8, nsym * 40)
                for i in range(nsym):
                    name_off, ohdr = struct.unpack_from("<QQ", ent, 40 * i)
                    links[self.heap_name(heap_addr, name_off)] = ohdr
                return
            if node[:4] != b"TREE":
                raise H5Error("bad B-tree node signature")
            if node[4] != 0:
                raise H5Error("not a group B-tree")
            used, = struct.unpack_from("<H", node, 6)
            body = self.at(addr + 24, 8 + used * 16)      # key0, child0, key1, child1, ..., key_n
            for i in range(used):
                child, = struct.unpack_from("<Q", body, 8 + 16 * i)
                walk(child)

        walk(btree_addr)
        return links

    # -- message decoding ----------------------------------------------------------
    @staticmethod
    def dataspace(body):
        ver, rank, flags = body[0], body[1], body[2]
        if ver == 1:
            p = 8
        elif ver == 2:
            p = 4
        else:
            raise NotImplementedError(f"dataspace version {ver}")
        return tuple(struct.unpack_from("<" + "Q" * rank, body, p)) if rank else ()

    @staticmethod
    def datatype(body):
        """numpy dtype of a datatype message, or None for classes that are skipped (variable length...)."""
        cls, bits0 = body[0] & 0x0F, body[1]
        size, = struct.unpack_from("<I", body, 4)
        order = ">" if bits0 & 1 else "<"
        if cls == 0:                                  # fixed point
            signed = (bits0 >> 3) & 1
            return np.dtype(f"{order}{'i' if signed else 'u'}{size}")
        if cls == 1:                                  # floating point
            if size not in (2, 4, 8):
                raise NotImplementedError(f"{size}-byte float")
            return np.dtype(f"{order}f{size}")
        if cls == 3:                                  # fixed-length string
            return np.dtype(f"S{size}")
        return None

    def layout_data(self, body, nbytes):
        ver, cls = body[0], body[1]
        if ver in (1, 2):                             # libhdf5 <= 1.6: version, rank, class, 5 reserved, address, dims
            cls = body[2]
            if cls != 1:
                raise NotImplementedError("only contiguous data in layout messages of version 1 / 2")
            addr, = struct.unpack_from("<Q", body, 8)
            return bytes(nbytes) if addr == UNDEF else self.at(addr, nbytes)
        if ver != 3:
            raise NotImplementedError(f"data layout message version {ver}")
        if cls == 0:                                  # compact
            size, = struct.unpack_from("<H", body, 2)
            return body[4:4 + size]
        if cls == 1:                                  # contiguous
            addr, size = struct.unpack_from("<QQ", body, 2)
            if addr == UNDEF:                         # never written: fill value (zeros)
                return bytes(nbytes)
            return self.at(addr, size)
        raise NotImplementedError("chunked datasets are not supported (Keras save_weights writes contiguous ones)")

    def attribute(self, body):
        ver = body[0]
        if ver not in (1, 2, 3):
            raise NotImplementedError(f"attribute message version {ver}")
        if ver >= 2 and body[1] & 3:
            raise NotImplementedError("attributes with shared datatype / dataspace messages")
        nlen, tlen, slen = struct.unpack_from("<HHH", body, 2)
        p = 8 if ver < 3 else 9                       # version 3 adds the name's character-set byte
        pad = _pad8 if ver == 1 else (lambda n: n)    # only version 1 pads the three parts to 8 bytes
        name = body[p:p + nlen].split(b"\x00")[0].decode("utf-8")
        p += pad(nlen)
        dt = self.datatype(body[p:p + tlen])
        p += pad(tlen)
        shape = self.dataspace(body[p:p + slen])
        p += pad(slen)
        if dt is None:
            return name, None
        n = int(np.prod(shape)) if shape else 1
        arr = np.frombuffer(body, dt, n, p).reshape(shape)
        return name, (arr.copy() if shape else arr[()])


class _Object:
    def __init__(self, reader, addr, name):
        self._r, self._addr, self.name = reader, addr, name
        self._msgs = reader.messages(addr)
        self.attrs = {}
        for mtype, _, body in self._msgs:
            if mtype == 0x000C:
                try:
                    k, v = reader.attribute(body)
                except NotImplementedError:
                    continue
                if v is not None:
                    self.attrs[k] = v

    def _msg(self, mtype):
        for t, _, body in self._msgs:
            if t == mtype:
                return body
        return None


class Dataset(_Object):
    def __init__(self, reader, addr, name):
        super().__init__(reader, addr, name)
        self.shape = reader.dataspace(self._msg(0x0001))
        self.dtype = reader.datatype(self._msg(0x0003))
        if self.dtype is None:
            raise NotImplementedError(f"{name}: unsupported datatype class")
        for t, _, body in self._msgs:
            if t == 0x000B:
                raise NotImplementedError(f"{name}: filtered (compressed) datasets are not supported")

    def __getitem__(self, key):
        n = int(np.prod(self.shape)) if self.shape else 1
        raw = self._r.layout_data(self._msg(0x0008), n * self.dtype.itemsize)
        arr = np.frombuffer(raw, self.dtype, n).reshape(self.shape).copy()
        return arr if key == () or key is Ellipsis else arr[key]

    @property
    def value(self):
        return self[()]


class Group(_Object):
    def __init__(self, reader, addr, name):
        super().__init__(reader, addr, name)
        st = self._msg(0x0011)
        if st is None:
            raise NotImplementedError(f"{name}: new-style (link message / fractal heap) groups are not supported")
        btree, heap = struct.unpack_from("<QQ", st, 0)
        self._links = reader.group_links(btree, heap)

    def keys(self):
        return sorted(self._links)

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        obj = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(obj, Group) or part not in obj._links:
                raise KeyError(path)
            addr = obj._links[part]
            kinds = [t for t, _, _ in obj._r.messages(addr)]
            cls = Dataset if 0x0008 in kinds else Group
            obj = cls(obj._r, addr, (obj.name.rstrip("/") + "/" + part))
        return obj


class File(Group):
    """Read-only view of an HDF5 file (the whole file is read into memory: weight files are ~10 MB)."""

    def __init__(self, path, mode="r"):
        if mode != "r":
            raise ValueError("h5lite.File is read-only; use h5lite.write_file to create files")
        with open(path, "rb") as f:
            reader = _Reader(f.read())
        super().__init__(reader, reader.root_header, "/")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def close(self):
        pass


# =============================================================================== writer
def _dtype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind == "f":
        size = dt.itemsize
        exp_bits, mant_bits = {2: (5, 10), 4: (8, 23), 8: (11, 52)}[size]
        bias = (1 << (exp_bits - 1)) - 1
        head = struct.pack("<BBBBI", 0x11, 0x20, 8 * size - 1, 0, size)          # version 1, class 1; implied msb; sign bit
        return head + struct.pack("<HHBBBBI", 0, 8 * size, mant_bits, exp_bits, 0, mant_bits, bias)
    if dt.kind in "iu":
        head = struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0, 0, 0, dt.itemsize)
        return head + struct.pack("<HH", 0, 8 * dt.itemsize)
    if dt.kind == "S":
        return struct.pack("<BBBBI", 0x13, 0x01, 0, 0, dt.itemsize)              # null padded, ASCII
    raise TypeError(f"h5lite cannot store dtype {dt}")


def _space_msg(shape):
    return struct.pack("<BBBB4x", 1, len(shape), 0, 0) + b"".join(struct.pack("<Q", d) for d in shape)


def _message(mtype, body, flags=0):
    body = body + bytes(_pad8(len(body)) - len(body))
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _attr_msg(name, value):
    arr = np.asarray(value)
    if arr.dtype.kind == "U":
        arr = np.char.encode(arr, "utf-8")
    if arr.dtype.kind not in "fiuS":
        raise TypeError(f"attribute {name}: unsupported dtype {arr.dtype}")
    if arr.dtype.kind != "S":
        arr = arr.astype(arr.dtype.newbyteorder("<"))
    nm = name.encode("utf-8") + b"\x00"
    dt, sp = _dtype_msg(arr.dtype), _space_msg(arr.shape)
    body = struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(sp))
    for part in (nm, dt, sp):
        body += part + bytes(_pad8(len(part)) - len(part))
    return _message(0x000C, body + arr.tobytes())


class _Writer:
    def __init__(self):
        self.buf = bytearray(96)                      # superblock, patched at the end

    def alloc(self, data):
        while len(self.buf) % 8:
            self.buf.append(0)
        addr = len(self.buf)
        self.buf += data
        return addr

    def header(self, messages):
        body = b"".join(messages)
        return self.alloc(struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(body)) + body)

    def dataset(self, arr, attrs):
        arr = np.ascontiguousarray(arr)
        if arr.dtype.kind != "S":
            arr = arr.astype(arr.dtype.newbyteorder("<"))
        data_addr = self.alloc(arr.tobytes()) if arr.size else UNDEF
        msgs = [_message(0x0001, _space_msg(arr.shape)), _message(0x0003, _dtype_msg(arr.dtype), flags=1),
                _message(0x0008, struct.pack("<BBQQ", 3, 1, data_addr, arr.nbytes))]
        msgs += [_attr_msg(k, v) for k, v in attrs.items()]
        return self.header(msgs)

    def group(self, node):
        """node = {"attrs": {...}, "children": {name: node | ndarray | (ndarray, attrs)}} -> object header address."""
        children = {}
        for name, child in node.get("children", {}).items():
            if isinstance(child, dict):
                children[name] = self.group(child)
            elif isinstance(child, tuple):
                children[name] = self.dataset(child[0], child[1])
            else:
                children[name] = self.dataset(child, {})
        names = sorted(children, key=lambda s: s.encode("utf-8"))       # B-tree order = strcmp order
        # local heap: offset 0 is the empty string (key 0 of the B-tree)
        heap, offs = bytearray(8), {}
        for n in names:
            offs[n] = len(heap)
            heap += n.encode("utf-8") + b"\x00"
            while len(heap) % 8:
                heap.append(0)
        seg_addr = self.alloc(bytes(heap) if len(heap) else bytes(8))
        heap_addr = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), UNDEF, seg_addr))
        # symbol nodes of at most 2K = 8 entries under one level-0 B-tree node (2K = 32 children)
        if len(names) > 8 * 32:
            raise ValueError("h5lite: more than 256 links in one group")
        snods = []
        for i in range(0, max(1, len(names)), 8):
            part = names[i:i + 8]
            ent = b"".join(struct.pack("<QQII16x", offs[n], children[n], 0, 0) for n in part)
            ent += bytes(40 * (8 - len(part)))
            snods.append((self.alloc(b"SNOD" + struct.pack("<BBH", 1, 0, len(part)) + ent), part))
        body = struct.pack("<Q", 0)
        for addr, part in snods:
            body += struct.pack("<QQ", addr, offs[part[-1]] if part else 0)
        body += bytes(8 + 16 * 32 - len(body))
        btree_addr = self.alloc(b"TREE" + struct.pack("<BBHQQ", 0, 0, len(snods), UNDEF, UNDEF) + body)
        msgs = [_message(0x0011, struct.pack("<QQ", btree_addr, heap_addr))]
        msgs += [_attr_msg(k, v) for k, v in node.get("attrs", {}).items()]
        hdr = self.header(msgs)
        self._last = (btree_addr, heap_addr)
        return hdr

    def finish(self, root):
        root_hdr = self.group(root)
        btree_addr, heap_addr = self._last
        eof = len(self.buf)
        sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQII", 0, root_hdr, 1, 0) + struct.pack("<QQ", btree_addr, heap_addr)
        self.buf[:96] = sb
        return bytes(self.buf)


def write_file(path, root):
    """Write a tree ``{"attrs": {...}, "children": {name: subtree | ndarray}}`` as an HDF5 file with the
    structures listed in the module docstring."""
    data = _Writer().finish(root)
    with open(path, "wb") as f:
        f.write(data)
