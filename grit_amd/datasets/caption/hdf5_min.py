"""Minimal HDF5 reader / writer for the cached-feature files of GRIT (SURVEY next-row N3) -- no h5py, no libhdf5.

The reference writes its caches with h5py (tools/extract_features.py:66-155: `create_dataset(name, shape, dtype)` without
chunking, int64 / float32 / bool) and reads single rows back (`h5py.File(path)['gri_feat'][idx]`, datasets/caption/field.py:47-63).
h5py's defaults (libver 'earliest') give exactly one on-disk shape, restated here from the published HDF5 File Format
Specification (version 2.0, sections III.A superblock v0, III.A.1 / IV.A.1 object header v1, III.B B-tree v1, III.C symbol table
nodes, III.D local heaps, IV.A.2 messages 0x0001 dataspace, 0x0003 datatype, 0x0005 fill value, 0x0008 layout, 0x0010
continuation, 0x0011 symbol table):

    superblock v0 -> root group (old-style: B-tree v1 of symbol-table nodes + local heap of names) -> one object header v1 per
    dataset -> contiguous raw data.

`H5File(path)` parses that subset (B-trees of any depth, header continuation blocks, dataspace v1/v2, fixed-point / float / enum
datatypes, contiguous or compact layout) and returns the datasets as numpy memory maps -- a row read touches only that row, as in
the reference.  `create(path, datasets)` writes a file of the same shape (one symbol-table node: up to 8 datasets) that libhdf5
opens: tests/golden/make_golden.py checks both directions against the HDF5 library that happens to sit in this image's
/opt/conda (h5dump / a C generator), the committed fixture tests/golden/features_ref.h5 was written by that library.
numpy bool <-> the enum over int8 {FALSE = 0, TRUE = 1} that h5py uses.  Chunked / compressed datasets raise."""
import struct

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K, INTERNAL_K = 4, 16           # library defaults (superblock fields)
OH_MESSAGE_BYTES = 256               # message area of the object headers we write (NIL-padded, like the library's)


class H5FormatError(RuntimeError):
    pass


# ----------------------------------------------------------------------------------------------------------------- reading
def _u(buf, off, n):
    return int.from_bytes(buf[off:off + n], 'little')


def _parse_datatype(b):
    """Datatype message body -> (numpy dtype, is_bool, consumed bytes)."""
    cls, ver = b[0] & 0x0F, b[0] >> 4
    bits = b[1] | (b[2] << 8) | (b[3] << 16)
    size = _u(b, 4, 4)
    if cls == 0:  # fixed point
        order = '>' if bits & 1 else '<'
        signed = bool(bits & 8)
        return np.dtype('%s%s%d' % (order, 'i' if signed else 'u', size)), False, 12
    if cls == 1:  # floating point (IEEE layouts only)
        order = '>' if bits & 1 else '<'
        if size not in (2, 4, 8):
            raise H5FormatError("unsupported float size %d" % size)
        return np.dtype('%sf%d' % (order, size)), False, 20
    if cls == 8:  # enumeration: base type, names, values
        n = bits & 0xFFFF
        base, _, used = _parse_datatype(b[8:])
        off = 8 + used
        names = []
        for _ in range(n):
            end = b.index(b'\0', off)
            names.append(bytes(b[off:end]).decode())
            ln = end - off + 1
            off += ln if ver >= 3 else (ln + 7) // 8 * 8
        values = [int.from_bytes(b[off + i * base.itemsize: off + (i + 1) * base.itemsize], 'little') for i in range(n)]
        is_bool = base.itemsize == 1 and dict(zip(names, values)) == {'FALSE': 0, 'TRUE': 1}
        return base, is_bool, off + n * base.itemsize
    raise H5FormatError("datatype class %d is not supported (int / float / bool-enum only)" % cls)


class H5File(object):
    """Read (mode 'r') or update in place (mode 'r+') the datasets of a GRIT feature file."""

    def __init__(self, path, mode='r'):
        self.path, self.mode = path, mode
        with open(path, 'rb') as f:
            head = f.read(96)
            if head[:8] != SIGNATURE:
                raise H5FormatError("%s is not an HDF5 file" % path)
            if head[8] != 0:
                raise H5FormatError("superblock version %d (only version 0, h5py's default, is read)" % head[8])
            if head[13] != 8 or head[14] != 8:
                raise H5FormatError("only 8-byte offsets / lengths are supported")
            self._base = _u(head, 24, 8)
            self._f = f
            root_oh = _u(head, 64, 8)
            btree, heap = None, None
            if _u(head, 72, 4) == 1:  # cached: B-tree and heap addresses in the scratch pad
                btree, heap = _u(head, 80, 8), _u(head, 88, 8)
            else:
                for mtype, body in self._messages(root_oh):
                    if mtype == 0x11:
                        btree, heap = _u(body, 0, 8), _u(body, 8, 8)
            if btree is None:
                raise H5FormatError("root group without a symbol table (new-style groups are not supported)")
            heap_data = self._heap(heap)
            self.datasets = {}
            for name_off, oh in self._walk(btree):
                name = heap_data[name_off:heap_data.index(b'\0', name_off)].decode()
                info = self._dataset(oh)
                if info is not None:
                    self.datasets[name] = info
        self._f = None
        self._maps = {}

    # -- low level
    def _read(self, addr, n):
        self._f.seek(self._base + addr)
        return self._f.read(n)

    def _heap(self, addr):
        h = self._read(addr, 32)
        if h[:4] != b'HEAP':
            raise H5FormatError("bad local heap")
        return self._read(_u(h, 24, 8), _u(h, 8, 8))

    def _walk(self, addr):
        node = self._read(addr, 24)
        if node[:4] != b'TREE' or node[4] != 0:
            raise H5FormatError("bad group B-tree node")
        level, used = node[5], _u(node, 6, 2)
        body = self._read(addr + 24, (2 * used + 1) * 8)
        for i in range(used):
            child = _u(body, 8 + 16 * i, 8)
            if level > 0:
                for e in self._walk(child):
                    yield e
            else:
                sn = self._read(child, 8)
                if sn[:4] != b'SNOD':
                    raise H5FormatError("bad symbol table node")
                n = _u(sn, 6, 2)
                ent = self._read(child + 8, 40 * n)
                for j in range(n):
                    yield _u(ent, 40 * j, 8), _u(ent, 40 * j + 8, 8)

    def _messages(self, addr):
        h = self._read(addr, 16)
        if h[0] != 1:
            raise H5FormatError("object header version %d (only version 1 is read)" % h[0])
        nmsg, size = _u(h, 2, 2), _u(h, 8, 4)
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            baddr, bsize = blocks.pop(0)
            buf = self._read(baddr, bsize)
            off = 0
            while off + 8 <= bsize and len(out) < nmsg:
                mtype, msize = _u(buf, off, 2), _u(buf, off + 2, 2)
                body = buf[off + 8: off + 8 + msize]
                if mtype == 0x10:  # continuation
                    blocks.append((_u(body, 0, 8), _u(body, 8, 8)))
                out.append((mtype, body))
                off += 8 + msize
        return out

    def _dataset(self, oh):
        shape = dtype = layout = None
        is_bool = False
        for mtype, b in self._messages(oh):
            if mtype == 0x01:
                ver, rank = b[0], b[1]
                off = 8 if ver == 1 else 4
                shape = tuple(_u(b, off + 8 * i, 8) for i in range(rank))
            elif mtype == 0x03:
                dtype, is_bool, _ = _parse_datatype(b)
            elif mtype == 0x08:
                if b[0] != 3:
                    raise H5FormatError("data layout message version %d" % b[0])
                if b[1] == 1:
                    layout = ('contiguous', _u(b, 2, 8), _u(b, 10, 8))
                elif b[1] == 0:
                    layout = ('compact', bytes(b[4:4 + _u(b, 2, 2)]))
                else:
                    layout = ('chunked',)
        if shape is None or dtype is None or layout is None:
            return None  # not a dataset (sub-group, named datatype)
        return {'shape': shape, 'dtype': dtype, 'bool': is_bool, 'layout': layout}

    # -- access
    def keys(self):
        return self.datasets.keys()

    def __contains__(self, name):
        return name in self.datasets

    def __getitem__(self, name):
        if name in self._maps:
            return self._maps[name]
        d = self.datasets[name]
        kind = d['layout'][0]
        if kind == 'chunked':
            raise H5FormatError("dataset %r is chunked / filtered: only contiguous storage (h5py's default for fixed shapes) "
                                "is supported" % name)
        if kind == 'compact':
            arr = np.frombuffer(d['layout'][1], dtype=d['dtype']).reshape(d['shape'])
        else:
            addr = d['layout'][1]
            if addr == UNDEF:  # never written: libhdf5 allocates late; reads give the fill value
                arr = np.zeros(d['shape'], d['dtype'])
            else:
                arr = np.memmap(self.path, dtype=d['dtype'], mode='r' if self.mode == 'r' else 'r+', offset=self._base + addr,
                                shape=d['shape'])
        if d['bool']:
            arr = arr.view(np.bool_)
        self._maps[name] = arr
        return arr

    def flush(self):
        for a in self._maps.values():
            if hasattr(a, 'flush'):
                a.flush()

    def close(self):
        self.flush()
        self._maps = {}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


# ----------------------------------------------------------------------------------------------------------------- writing
def _pad8(b):
    return b + b'\0' * (-len(b) % 8)


def _datatype_message(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.bool_:  # h5py: enum over int8
        base = _datatype_message(np.int8)
        body = bytes([0x18, 2, 0, 0]) + struct.pack('<I', 1) + base[:12] + _pad8(b'FALSE\0') + _pad8(b'TRUE\0') + bytes([0, 1])
        return _pad8(body)
    if dtype.kind in 'iu':
        bits = 0x08 if dtype.kind == 'i' else 0x00
        return _pad8(bytes([0x10, bits, 0, 0]) + struct.pack('<IHH', dtype.itemsize, 0, 8 * dtype.itemsize))
    if dtype.kind == 'f' and dtype.itemsize in (4, 8):
        if dtype.itemsize == 4:
            props = struct.pack('<HHBBBBI', 0, 32, 23, 8, 0, 23, 127)
            sign = 31
        else:
            props = struct.pack('<HHBBBBI', 0, 64, 52, 11, 0, 52, 1023)
            sign = 63
        return _pad8(bytes([0x11, 0x20, sign, 0]) + struct.pack('<I', dtype.itemsize) + props)
    raise H5FormatError("dtype %s cannot be written (int / uint / float32 / float64 / bool)" % dtype)


def _message(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack('<HHB3x', mtype, len(body), flags) + body


def _dataset_header(shape, dtype, data_addr, nbytes):
    rank = len(shape)
    space = bytes([1, rank, 1, 0, 0, 0, 0, 0]) + b''.join(struct.pack('<Q', s) for s in shape) * 2  # dims, then max dims
    msgs = _message(0x01, space) + _message(0x03, _datatype_message(dtype), flags=1)
    msgs += _message(0x05, bytes([2, 2, 2, 1]) + struct.pack('<I', 0), flags=1)  # fill value v2: late alloc, defined, size 0
    msgs += _message(0x08, bytes([3, 1]) + struct.pack('<QQ', data_addr, nbytes))
    n = 4
    if len(msgs) + 8 > OH_MESSAGE_BYTES:
        raise H5FormatError("dataset rank too large for the fixed-size header")
    nil = OH_MESSAGE_BYTES - len(msgs) - 8
    msgs += struct.pack('<HHB3x', 0, nil, 0) + b'\0' * nil
    return struct.pack('<BBHII4x', 1, 0, n + 1, 1, OH_MESSAGE_BYTES) + msgs


def create(path, datasets, align=4096):
    """Write an HDF5 file with the given datasets {name: (shape, dtype)}, zero-filled, and return H5File(path, 'r+').
    The raw data of every dataset starts at a multiple of `align` (page-aligned memory maps)."""
    names = sorted(datasets)  # a symbol-table node lists its entries in name order
    if not 0 < len(names) <= 2 * LEAF_K:
        raise H5FormatError("this writer emits a single symbol-table node: 1..%d datasets" % (2 * LEAF_K))
    # local heap data: offset 0 = the empty name, then the names, each padded to a multiple of 8
    heap, name_off = b'\0' * 8, {}
    for n in names:
        name_off[n] = len(heap)
        heap += _pad8(n.encode() + b'\0')
    root_oh = 96
    btree = root_oh + 16 + 24
    heap_hdr = btree + 24 + (2 * (2 * INTERNAL_K) + 1) * 8
    heap_data = heap_hdr + 32
    snod = heap_data + len(heap)
    oh0 = snod + 8 + 2 * LEAF_K * 40
    oh_size = 16 + OH_MESSAGE_BYTES
    data = oh0 + oh_size * len(names)
    addrs, sizes = {}, {}
    for n in names:
        shape, dtype = datasets[n]
        dtype = np.dtype(dtype)
        data = (data + align - 1) // align * align
        addrs[n] = data
        sizes[n] = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        data += sizes[n]
    eof = data
    out = bytearray()
    # superblock v0 + root symbol-table entry (cache type 1: B-tree / heap addresses in the scratch pad)
    out += SIGNATURE + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack('<HHI', LEAF_K, INTERNAL_K, 0)
    out += struct.pack('<QQQQ', 0, UNDEF, eof, UNDEF)
    out += struct.pack('<QQII', 0, root_oh, 1, 0) + struct.pack('<QQ', btree, heap_hdr)
    # root object header: one symbol-table message
    out += struct.pack('<BBHII4x', 1, 0, 1, 1, 24) + _message(0x11, struct.pack('<QQ', btree, heap_hdr))
    # B-tree: one leaf entry -> the symbol-table node; key 0 = "" (offset 0), key 1 = the largest name
    tree = b'TREE' + bytes([0, 0]) + struct.pack('<HQQ', 1, UNDEF, UNDEF)
    tree += struct.pack('<QQQ', 0, snod, name_off[names[-1]])
    out += tree + b'\0' * (heap_hdr - btree - len(tree))
    out += b'HEAP' + bytes(4) + struct.pack('<QQQ', len(heap), 1, heap_data) + heap  # free-list head 1 = no free block
    entries = b''.join(struct.pack('<QQII16x', name_off[n], oh0 + i * oh_size, 0, 0) for i, n in enumerate(names))
    out += b'SNOD' + bytes([1, 0]) + struct.pack('<H', len(names)) + entries + b'\0' * (40 * (2 * LEAF_K - len(names)))
    for n in names:
        shape, dtype = datasets[n]
        out += _dataset_header(tuple(int(s) for s in shape), dtype, addrs[n], sizes[n])
    assert len(out) == oh0 + oh_size * len(names)
    with open(path, 'wb') as f:
        f.write(out)
        f.truncate(eof)  # sparse: the raw data regions read as zeros until written
    return H5File(path, 'r+')
