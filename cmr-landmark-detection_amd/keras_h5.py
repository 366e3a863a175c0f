"""Keras-HDF5 weights files (``model.h5``) without h5py: the layout ``Model.save_weights`` of tf.keras 2.3 writes and
``Model.load_weights`` reads (the reference checkpoints with ``ModelCheckpoint(save_weights_only=True)``,
src/utils/KerasCallbacks.py:54-61, and restores with ``model.load_weights(.../model.h5)``, src/models/predict_model.py:75-76,
predict_4d_on_seg.py:59-61, notebooks/Predict/Prediction.ipynb:513-514).

Keras layout (``tensorflow/python/keras/saving/hdf5_format.py`` ``save_weights_to_hdf5_group`` -- third-party, restated):

    /                       attrs  layer_names   fixed-length byte strings [n_layers]  (EVERY layer, also weight-less ones)
                                   backend       b'tensorflow'      keras_version  b'2.4.0' (tf.keras 2.3.0 reports 2.4.0)
    /<layer>                attrs  weight_names  byte strings [n_weights] ('<layer>/kernel:0' ...; an EMPTY list is stored by
                                   h5py as a float64 attribute of shape (0,))
    /<layer>/<layer>/kernel:0      contiguous little-endian float32 dataset (the '/' in a weight name makes the sub-group)

HDF5 subset written (HDF5 File Format Specification 1.x, what h5py's default ``libver='earliest'`` produces): superblock
version 0 with 8-byte offsets / lengths, old-style groups (version-1 B-tree of symbol-table nodes + local heap), version-1
object headers, version-1 attribute / dataspace / datatype messages, version-3 contiguous data layout.  The reader accepts
more than the writer emits: superblock 0-3, version-1 and version-2 ("OHDR") object headers with continuation blocks,
attribute messages 1-3, dataspace 1-2, compact + contiguous layouts, IEEE floats / integers of either byte order,
fixed-length and variable-length (global-heap) strings, multi-level group B-trees.  Chunked / filtered datasets and
new-style (link-message / fractal-heap) groups raise: Keras weight files have neither.

Checked against the real library where one is installed (``tests/test_keras_h5.py``: libhdf5 1.10.6 reads the files this
module writes, and this module reads ``tests/golden/keras_ref_libhdf5.h5`` which that library wrote).
"""
from __future__ import annotations

import struct
from collections import OrderedDict

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K, INTERNAL_K = 4, 16                    # symbol-table node holds <= 2*LEAF_K entries, B-tree node <= 2*INTERNAL_K children
SNOD_SIZE = 8 + 2 * LEAF_K * 40
TREE_SIZE = 24 + (2 * INTERNAL_K + 1) * 8 + 2 * INTERNAL_K * 8
MSG_DATASPACE, MSG_LINKINFO, MSG_DATATYPE, MSG_FILL_OLD, MSG_FILL, MSG_LINK, MSG_LAYOUT = 0x1, 0x2, 0x3, 0x4, 0x5, 0x6, 0x8
MSG_GROUPINFO, MSG_FILTER, MSG_ATTRIBUTE, MSG_CONT, MSG_STAB, MSG_MTIME, MSG_ATTRINFO = 0xA, 0xB, 0xC, 0x10, 0x11, 0x12, 0x15


class H5FormatError(ValueError):
    pass


def _pad8(n):
    return (n + 7) & ~7


# =====================================================================================================================
# writer
# =====================================================================================================================
def _dtype_message(dt):
    """Datatype message (version 1) of a NumPy dtype: IEEE little-endian floats, little-endian integers, fixed strings."""
    dt = np.dtype(dt)
    if dt.kind == 'S':                                                     # class 3, null-padded ASCII (h5py's mapping of 'S')
        return struct.pack('<BBBBI', 0x13, 0x01, 0, 0, max(dt.itemsize, 1))
    if dt.kind == 'f' and dt.itemsize in (4, 8):
        size = dt.itemsize
        exp_bits, man_bits, bias = (8, 23, 127) if size == 4 else (11, 52, 1023)
        # bit field: byte order LE (0), mantissa normalisation 2 (msb implied) in bits 4-5, sign bit position in byte 1
        return struct.pack('<BBBBI', 0x11, 0x20, size * 8 - 1, 0, size) + struct.pack('<HHBBBBI', 0, size * 8, man_bits, exp_bits, 0, man_bits, bias)
    if dt.kind in 'iu':
        return struct.pack('<BBBBI', 0x10, 0x08 if dt.kind == 'i' else 0x00, 0, 0, dt.itemsize) + struct.pack('<HH', 0, dt.itemsize * 8)
    raise TypeError('dtype %r is not written' % (dt,))


def _dataspace_message(shape):
    """Dataspace message version 1; simple dataspaces carry their maximum dimensions as libhdf5 writes them."""
    if shape is None or len(shape) == 0:
        return struct.pack('<BBBBI', 1, 0, 0, 0, 0)
    dims = b''.join(struct.pack('<Q', int(d)) for d in shape)
    return struct.pack('<BBBBI', 1, len(shape), 1, 0, 0) + dims + dims


def _message(mtype, data, flags=0):
    data = data + b'\0' * (_pad8(len(data)) - len(data))
    return struct.pack('<HHBBBB', mtype, len(data), flags, 0, 0, 0) + data


def _attribute_message(name, value):
    """Attribute message version 1 (name, datatype and dataspace each padded to 8 bytes)."""
    arr = np.asarray(value)
    if arr.dtype.kind == 'U':
        arr = np.char.encode(arr, 'utf8')
    if arr.dtype.kind not in 'Sfiu':
        raise TypeError('attribute %r: dtype %r' % (name, arr.dtype))
    if arr.dtype.kind != 'S':
        arr = arr.astype(arr.dtype.newbyteorder('<'))
    nm = name.encode('utf8') + b'\0'
    dt, ds = _dtype_message(arr.dtype), _dataspace_message(arr.shape)
    body = struct.pack('<BBHHH', 1, 0, len(nm), len(dt), len(ds))
    for part in (nm, dt, ds):
        body += part + b'\0' * (_pad8(len(part)) - len(part))
    return _message(MSG_ATTRIBUTE, body + np.ascontiguousarray(arr).tobytes())


class _Node:
    def __init__(self):
        self.attrs = OrderedDict()
        self.children = OrderedDict()          # name -> _Node (group) | np.ndarray (dataset)


class H5Writer:
    """Build a file in memory: ``create_group`` / ``create_dataset`` / ``attrs`` of a tiny h5py-like tree, then ``tobytes``."""

    def __init__(self):
        self.root = _Node()

    def group(self, path):
        node = self.root
        for part in [p for p in path.split('/') if p]:
            nxt = node.children.get(part)
            if nxt is None:
                nxt = node.children[part] = _Node()
            if not isinstance(nxt, _Node):
                raise ValueError('%r is a dataset' % part)
            node = nxt
        return node

    def create_dataset(self, path, array):
        parts = [p for p in path.split('/') if p]
        parent = self.group('/'.join(parts[:-1]))
        if parts[-1] in parent.children:
            raise ValueError('%r exists' % path)
        a = np.asarray(array)
        parent.children[parts[-1]] = np.ascontiguousarray(a, a.dtype.newbyteorder('<'))

    def set_attr(self, path, name, value):
        self.group(path).attrs[name] = value

    # -- serialisation ------------------------------------------------------------------------------------------------
    def tobytes(self):
        return bytes(self.serialise())

    def serialise(self):
        """the file image as a bytearray (no further copy: save_keras_weights hands it to write())"""
        self._buf = bytearray(96)                                  # superblock (56 bytes) + root symbol-table entry (40)
        root_oh, root_bt, root_hp = self._write_group(self.root)
        eof = len(self._buf)
        sb = SIGNATURE + struct.pack('<BBBBBBBBHHI', 0, 0, 0, 0, 0, 8, 8, 0, LEAF_K, INTERNAL_K, 0)
        sb += struct.pack('<QQQQ', 0, UNDEF, eof, UNDEF)
        sb += struct.pack('<QQII', 0, root_oh, 1, 0) + struct.pack('<QQ', root_bt, root_hp)
        assert len(sb) == 96
        self._buf[0:96] = sb
        return self._buf

    def _alloc(self, data):
        while len(self._buf) % 8:
            self._buf.append(0)
        addr = len(self._buf)
        self._buf += data
        return addr

    def _object_header(self, messages):
        body = b''.join(messages)
        return self._alloc(struct.pack('<BBHII', 1, 0, len(messages), 1, len(body)) + b'\0' * 4 + body)

    def _write_dataset(self, arr):
        raw = memoryview(arr.reshape(-1)).cast('B') if arr.size else b''      # one copy, into the file image (create_dataset made it contiguous little-endian)
        addr = self._alloc(raw) if len(raw) else UNDEF
        msgs = [_message(MSG_DATASPACE, _dataspace_message(arr.shape)),
                _message(MSG_DATATYPE, _dtype_message(arr.dtype), flags=1),                      # constant message, as libhdf5 marks it
                _message(MSG_FILL, struct.pack('<BBBBI', 2, 2, 2, 1, 0), flags=1),               # v2: late allocation, fill if set, default (size 0) value
                _message(MSG_LAYOUT, struct.pack('<BBQQ', 3, 1, addr, len(raw)))]
        return self._object_header(msgs)

    def _write_group(self, node):
        # children first (their object-header addresses go into this group's symbol-table nodes)
        names = sorted(node.children, key=lambda s: s.encode('utf8'))                        # libhdf5 orders by strcmp
        entries = []
        heap = bytearray(8)                                                                    # offset 0: the empty name
        for nm in names:
            child = node.children[nm]
            off = len(heap)
            b = nm.encode('utf8') + b'\0'
            heap += b + b'\0' * (_pad8(len(b)) - len(b))
            if isinstance(child, _Node):
                oh, bt, hp = self._write_group(child)
                entries.append((off, struct.pack('<QQII', off, oh, 1, 0) + struct.pack('<QQ', bt, hp)))
            else:
                entries.append((off, struct.pack('<QQII', off, self._write_dataset(child), 0, 0) + b'\0' * 16))
        # leaf level: symbol-table nodes of <= 2*LEAF_K entries
        level = []                                                                            # (address, last key) per node of the current level
        for i in range(0, len(entries), 2 * LEAF_K):
            chunk = entries[i:i + 2 * LEAF_K]
            snod = b'SNOD' + struct.pack('<BBH', 1, 0, len(chunk)) + b''.join(e[1] for e in chunk)
            level.append((self._alloc(snod + b'\0' * (SNOD_SIZE - len(snod))), chunk[-1][0]))
        lvl = 0
        while True:
            nodes = []
            groups = [level[i:i + 2 * INTERNAL_K] for i in range(0, len(level), 2 * INTERNAL_K)] or [[]]
            first_key = 0
            addrs = []
            for gi, kids in enumerate(groups):
                body = struct.pack('<Q', first_key)
                for addr, key in kids:
                    body += struct.pack('<QQ', addr, key)
                node_b = b'TREE' + struct.pack('<BBH', 0, lvl, len(kids)) + struct.pack('<QQ', UNDEF, UNDEF) + body
                addrs.append(self._alloc(node_b + b'\0' * (TREE_SIZE - len(node_b))))
                nodes.append((addrs[-1], kids[-1][1] if kids else 0))
                first_key = kids[-1][1] if kids else 0
            if len(nodes) > 1:                                     # sibling links of one level
                for i, a in enumerate(addrs):
                    left = addrs[i - 1] if i else UNDEF
                    right = addrs[i + 1] if i + 1 < len(addrs) else UNDEF
                    self._buf[a + 8:a + 24] = struct.pack('<QQ', left, right)
            if len(nodes) == 1:
                btree = nodes[0][0]
                break
            level, lvl = nodes, lvl + 1
        hdata = self._alloc(bytes(heap))
        hp = self._alloc(b'HEAP' + struct.pack('<BBBB', 0, 0, 0, 0) + struct.pack('<QQQ', len(heap), 1, hdata))   # free list: 1 = none
        msgs = [_message(MSG_STAB, struct.pack('<QQ', btree, hp))]
        msgs += [_attribute_message(k, v) for k, v in node.attrs.items()]
        return self._object_header(msgs), btree, hp


# =====================================================================================================================
# reader
# =====================================================================================================================
class Dataset:
    def __init__(self, f, shape, dtype, layout):
        self._f, self.shape, self.dtype, self._layout = f, tuple(shape), dtype, layout
        self.attrs = OrderedDict()

    def read(self):
        kind, a, b = self._layout
        n = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        if n == 0:
            return np.zeros(self.shape, self.dtype.newbyteorder('='))
        if kind == 'compact':
            raw = a
        else:
            if a == UNDEF:
                raise H5FormatError('dataset storage was never allocated')
            raw = self._f._bytes(a, n)
        if len(raw) < n:
            raise H5FormatError('dataset is shorter than its dataspace')
        arr = np.frombuffer(raw[:n], self.dtype).reshape(self.shape)
        return arr.astype(self.dtype.newbyteorder('='))


class Group:
    def __init__(self):
        self.attrs = OrderedDict()
        self.children = OrderedDict()

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split('/') if p]:
            node = node.children[part]
        return node

    def __contains__(self, path):
        try:
            self[path]
            return True
        except (KeyError, AttributeError):
            return False

    def visit(self, prefix=''):
        for k, v in self.children.items():
            yield prefix + '/' + k, v
            if isinstance(v, Group):
                yield from v.visit(prefix + '/' + k)


class H5Reader:
    def __init__(self, data):
        self.data = data if isinstance(data, (bytes, bytearray, memoryview)) else open(data, 'rb').read()
        self.base = 0
        self._parse_superblock()

    # -- primitives --------------------------------------------------------------------------------------------------
    def _bytes(self, addr, n):
        addr += self.base
        if addr < 0 or addr + n > len(self.data):
            raise H5FormatError('address %d + %d beyond the end of the file (%d)' % (addr, n, len(self.data)))
        return bytes(self.data[addr:addr + n])

    def _off(self, buf, pos):
        return int.from_bytes(buf[pos:pos + self.so], 'little')

    def _len(self, buf, pos):
        return int.from_bytes(buf[pos:pos + self.sl], 'little')

    def _undef(self, v, size=None):
        return v == (1 << (8 * (size or self.so))) - 1

    def _parse_superblock(self):
        pos = None
        for cand in [0] + [512 << i for i in range(24)]:           # the signature may follow a user block
            if cand + 8 <= len(self.data) and bytes(self.data[cand:cand + 8]) == SIGNATURE:
                pos = cand
                break
        if pos is None:
            raise H5FormatError('not an HDF5 file (no signature)')
        d = self.data
        self.sb_version = v = d[pos + 8]
        if v in (0, 1):
            self.so, self.sl = d[pos + 13], d[pos + 14]
            self.leaf_k, self.internal_k = struct.unpack_from('<HH', d, pos + 16)
            p = pos + 24 + (4 if v == 1 else 0)
            base = int.from_bytes(d[p:p + self.so], 'little')
            self.eof = int.from_bytes(d[p + 2 * self.so:p + 3 * self.so], 'little')
            p += 4 * self.so
            self.base = base
            self.root_oh = int.from_bytes(d[p + self.so:p + 2 * self.so], 'little')   # root symbol-table entry: name offset, header address
        elif v in (2, 3):
            self.so, self.sl = d[pos + 9], d[pos + 10]
            p = pos + 12
            base = int.from_bytes(d[p:p + self.so], 'little')
            self.eof = int.from_bytes(d[p + 2 * self.so:p + 3 * self.so], 'little')
            self.root_oh = int.from_bytes(d[p + 3 * self.so:p + 4 * self.so], 'little')
            self.base = base
            self.leaf_k, self.internal_k = LEAF_K, INTERNAL_K
        else:
            raise H5FormatError('superblock version %d' % v)
        if self.so not in (2, 4, 8) or self.sl not in (2, 4, 8):
            raise H5FormatError('size of offsets / lengths %d / %d' % (self.so, self.sl))

    # -- object headers -------------------------------------------------------------------------------------------------
    def _messages(self, addr):
        """[(type, flags, data)] of an object header, continuation blocks followed."""
        head = self._bytes(addr, 16)
        out = []
        if head[:4] == b'OHDR':                                   # version 2
            flags = head[5]
            p = 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            szlen = 1 << (flags & 3)
            head = self._bytes(addr, p + szlen)
            chunk0 = int.from_bytes(head[p:p + szlen], 'little')
            blocks = [(addr + p + szlen, chunk0)]
            track_order = bool(flags & 0x04)
            while blocks:
                a, n = blocks.pop(0)
                buf = self._bytes(a, n)
                q = 0
                while q + 4 <= n:
                    mtype, msize, mflags = buf[q], int.from_bytes(buf[q + 1:q + 3], 'little'), buf[q + 3]
                    q += 4 + (2 if track_order else 0)
                    data = buf[q:q + msize]
                    q += msize
                    if mtype == MSG_CONT:
                        ca, cl = self._off(data, 0), self._len(data, self.so)
                        blocks.append((ca + 4, cl - 8))           # 'OCHK' signature in front, checksum behind
                    elif mtype != 0:
                        out.append((mtype, mflags, data))
            return out
        version, _, nmsg, _refs, size = struct.unpack_from('<BBHII', head, 0)
        if version != 1:
            raise H5FormatError('object header version %d at %d' % (version, addr))
        blocks = [(addr + 16, size)]
        while blocks and len(out) < nmsg + 64:
            a, n = blocks.pop(0)
            buf = self._bytes(a, n)
            q = 0
            while q + 8 <= n:
                mtype, msize, mflags = struct.unpack_from('<HHB', buf, q)
                data = buf[q + 8:q + 8 + msize]
                q += 8 + msize
                if mtype == MSG_CONT:
                    blocks.append((self._off(data, 0), self._len(data, self.so)))
                elif mtype != 0:
                    if mflags & 0x02:
                        raise H5FormatError('shared object-header messages are not read')
                    out.append((mtype, mflags, data))
        return out

    def _parse_dataspace(self, d):
        v = d[0]
        if v == 1:
            rank, flags = d[1], d[2]
            p = 8
        elif v == 2:
            rank, flags, stype = d[1], d[2], d[3]
            p = 4
            if stype == 2:
                return None                                       # null dataspace
        else:
            raise H5FormatError('dataspace version %d' % v)
        return tuple(self._len(d, p + i * self.sl) for i in range(rank))

    def _parse_datatype(self, d):
        """-> (numpy dtype | ('vlen_str',) , encoded size)"""
        cls, ver = d[0] & 0x0F, d[0] >> 4
        b0, b1 = d[1], d[2]
        size = struct.unpack_from('<I', d, 4)[0]
        order = '>' if b0 & 1 else '<'
        if cls == 0:
            return np.dtype('%s%s%d' % (order, 'i' if b0 & 0x08 else 'u', size)), 12
        if cls == 1:
            if size not in (2, 4, 8):
                raise H5FormatError('float of %d bytes' % size)
            return np.dtype('%sf%d' % (order, size)), 20
        if cls == 3:
            return np.dtype('S%d' % size), 8
        if cls == 9:
            vtype = b0 & 0x0F
            if vtype != 1:
                raise H5FormatError('variable-length sequences are not read')
            return ('vlen_str',), 8 + self._parse_datatype(d[8:])[1]
        raise H5FormatError('datatype class %d (version %d) is not read' % (cls, ver))

    def _global_heap_object(self, addr, index):
        head = self._bytes(addr, 8 + self.sl)
        if head[:4] != b'GCOL':
            raise H5FormatError('no global heap collection at %d' % addr)
        size = self._len(head, 8)
        buf = self._bytes(addr, size)
        q = 8 + self.sl
        while q + 8 + self.sl <= size:
            idx = int.from_bytes(buf[q:q + 2], 'little')
            osz = self._len(buf, q + 8)
            if idx == 0:
                break
            if idx == index:
                return buf[q + 8 + self.sl:q + 8 + self.sl + osz]
            q += 8 + self.sl + _pad8(osz)
        raise H5FormatError('global heap object %d not found' % index)

    def _decode_values(self, dtype, shape, raw):
        count = 1 if shape is None else int(np.prod(shape, dtype=np.int64))
        if dtype == ('vlen_str',):
            vals, step = [], 4 + self.so + 4
            for i in range(count):
                rec = raw[i * step:(i + 1) * step]
                ln = int.from_bytes(rec[:4], 'little')
                ga, gi = self._off(rec, 4), int.from_bytes(rec[4 + self.so:8 + self.so], 'little')
                vals.append(self._global_heap_object(ga, gi)[:ln] if ln else b'')
            arr = np.array(vals, dtype=object).reshape(shape or ())
            return arr if shape else arr[()]
        if shape is None:
            return np.zeros((0,), dtype)
        arr = np.frombuffer(raw[:count * dtype.itemsize], dtype).reshape(shape)
        if dtype.kind != 'S':
            arr = arr.astype(dtype.newbyteorder('='))
        return arr.copy() if shape else arr[()]

    def _parse_attribute(self, d):
        v = d[0]
        nsz, tsz, ssz = struct.unpack_from('<HHH', d, 2)
        if v == 1:
            p = 8
            name = d[p:p + nsz]; p += _pad8(nsz)
            tmsg = d[p:p + tsz]; p += _pad8(tsz)
            smsg = d[p:p + ssz]; p += _pad8(ssz)
        elif v in (2, 3):
            if d[1] & 0x03:
                raise H5FormatError('attributes with shared datatype / dataspace are not read')
            p = 8 + (1 if v == 3 else 0)
            name = d[p:p + nsz]; p += nsz
            tmsg = d[p:p + tsz]; p += tsz
            smsg = d[p:p + ssz]; p += ssz
        else:
            raise H5FormatError('attribute message version %d' % v)
        dtype, _ = self._parse_datatype(tmsg)
        shape = self._parse_dataspace(smsg)
        return name.split(b'\0')[0].decode('utf8'), self._decode_values(dtype, shape, d[p:])

    # -- groups ------------------------------------------------------------------------------------------------------
    def _heap(self, addr):
        h = self._bytes(addr, 8 + 2 * self.sl + self.so)
        if h[:4] != b'HEAP':
            raise H5FormatError('no local heap at %d' % addr)
        size = self._len(h, 8)
        return self._bytes(self._off(h, 8 + 2 * self.sl), size)

    def _symbols(self, btree, heap):
        """name -> (object header address) of an old-style group, in B-tree (= name) order."""
        out = OrderedDict()
        if self._undef(btree):
            return out
        head = self._bytes(btree, 8 + 2 * self.so)
        if head[:4] == b'SNOD':
            n = struct.unpack_from('<H', head, 6)[0]
            esz = 2 * self.so + 24
            buf = self._bytes(btree + 8, n * esz)
            for i in range(n):
                e = buf[i * esz:(i + 1) * esz]
                noff = self._off(e, 0)
                name = heap[noff:heap.index(b'\0', noff)].decode('utf8')
                out[name] = self._off(e, self.so)
            return out
        if head[:4] != b'TREE' or head[4] != 0:
            raise H5FormatError('no group B-tree node at %d' % btree)
        used = struct.unpack_from('<H', head, 6)[0]
        buf = self._bytes(btree + 8 + 2 * self.so, (2 * used + 1) * max(self.so, self.sl))
        for i in range(used):
            child = self._off(buf, self.sl + i * (self.sl + self.so))
            out.update(self._symbols(child, heap))
        return out

    def _load(self, addr, depth=0):
        if depth > 64:
            raise H5FormatError('group nesting too deep (cycle?)')
        msgs = self._messages(addr)
        kinds = {m[0] for m in msgs}
        attrs = OrderedDict()
        for mtype, _, data in msgs:
            if mtype == MSG_ATTRIBUTE:
                k, v = self._parse_attribute(data)
                attrs[k] = v
        if MSG_ATTRINFO in kinds:
            for mtype, _, data in msgs:
                if mtype == MSG_ATTRINFO:
                    flags = data[1]
                    p = 2 + (2 if flags & 1 else 0)
                    if not self._undef(self._off(data, p)):
                        raise H5FormatError('densely stored attributes (fractal heap) are not read')
        if MSG_LAYOUT in kinds:
            shape = dtype = layout = None
            for mtype, _, data in msgs:
                if mtype == MSG_DATASPACE:
                    shape = self._parse_dataspace(data)
                elif mtype == MSG_DATATYPE:
                    dtype, _ = self._parse_datatype(data)
                elif mtype == MSG_FILTER:
                    raise H5FormatError('filtered (compressed) datasets are not read')
                elif mtype == MSG_LAYOUT:
                    if data[0] != 3:
                        raise H5FormatError('data layout message version %d' % data[0])
                    if data[1] == 0:
                        n = struct.unpack_from('<H', data, 2)[0]
                        layout = ('compact', data[4:4 + n], n)
                    elif data[1] == 1:
                        layout = ('contiguous', self._off(data, 2), self._len(data, 2 + self.so))
                    else:
                        raise H5FormatError('chunked datasets are not read (Keras weight files are contiguous)')
            if shape is None:
                shape = (0,)
            if not isinstance(dtype, np.dtype):
                raise H5FormatError('dataset datatype is not numeric')
            ds = Dataset(self, shape, dtype, layout)
            ds.attrs = attrs
            return ds
        g = Group()
        g.attrs = attrs
        if MSG_STAB in kinds:
            for mtype, _, data in msgs:
                if mtype == MSG_STAB:
                    heap = self._heap(self._off(data, self.so))
                    for name, oh in self._symbols(self._off(data, 0), heap).items():
                        g.children[name] = self._load(oh, depth + 1)
        elif MSG_LINK in kinds or MSG_LINKINFO in kinds:
            for mtype, _, data in msgs:
                if mtype == MSG_LINKINFO:
                    flags = data[1]
                    p = 2 + (8 if flags & 1 else 0)
                    if not self._undef(self._off(data, p)):
                        raise H5FormatError('densely stored links (fractal heap) are not read: rewrite the file with libver="earliest"')
                elif mtype == MSG_LINK:
                    flags = data[1]
                    p = 2
                    ltype = 0
                    if flags & 0x08:
                        ltype = data[p]; p += 1
                    if flags & 0x04:
                        p += 8
                    if flags & 0x10:
                        p += 1
                    lsz = 1 << (flags & 3)
                    nlen = int.from_bytes(data[p:p + lsz], 'little'); p += lsz
                    name = data[p:p + nlen].decode('utf8'); p += nlen
                    if ltype == 0:
                        g.children[name] = self._load(self._off(data, p), depth + 1)
        return g

    def root(self):
        return self._load(self.root_oh)


# =====================================================================================================================
# Keras layer on top
# =====================================================================================================================
def _as_str_list(v):
    if v is None:
        return []
    arr = np.atleast_1d(v)
    if arr.dtype.kind not in 'SOU':
        return []                                                  # an empty list is stored as float64 (0,)
    return [x.decode('utf8') if isinstance(x, (bytes, np.bytes_)) else str(x) for x in arr.tolist()]


def _split_chunks(group_attrs, name):
    """Keras' load_attributes_from_hdf5_group: ``name`` or, when too large for one attribute, ``name0``, ``name1`` ..."""
    if name in group_attrs:
        return _as_str_list(group_attrs[name])
    out, i = [], 0
    while '%s%d' % (name, i) in group_attrs:
        out += _as_str_list(group_attrs['%s%d' % (name, i)])
        i += 1
    return out


def save_keras_weights(path, layers, backend='tensorflow', keras_version='2.4.0'):
    """``layers``: [(layer_name, [(weight_name, ndarray), ...])] in model.layers order, weight-less layers included."""
    w = H5Writer()
    names = [ln for ln, _ in layers]
    w.set_attr('/', 'layer_names', np.array([n.encode('utf8') for n in names], dtype='S') if names else np.zeros((0,), np.float64))
    w.set_attr('/', 'backend', np.bytes_(backend.encode('utf8')))
    w.set_attr('/', 'keras_version', np.bytes_(keras_version.encode('utf8')))
    for ln, ws in layers:
        w.group(ln)
        wn = [n.encode('utf8') for n, _ in ws]
        w.set_attr(ln, 'weight_names', np.array(wn, dtype='S') if wn else np.zeros((0,), np.float64))
        for n, arr in ws:
            w.create_dataset(ln + '/' + n, np.asarray(arr, np.float32))
    data = w.serialise()
    with open(path, 'wb') as f:
        f.write(data)
    return len(data)


def load_keras_weights(path):
    """-> (OrderedDict layer_name -> [(weight_name, float32 ndarray)], root attrs dict).  Accepts a weights-only file and a
    full-model file (``model.save``: the weights live under /model_weights)."""
    root = H5Reader(path).root()
    attrs = root.attrs
    if 'layer_names' not in attrs and 'layer_names0' not in attrs and 'model_weights' in root.children:
        root = root.children['model_weights']
        attrs = OrderedDict(list(attrs.items()) + list(root.attrs.items()))
    if 'layer_names' not in root.attrs and 'layer_names0' not in root.attrs:
        raise H5FormatError('%s: no layer_names attribute -- not a Keras weights file' % path)
    out = OrderedDict()
    for ln in _split_chunks(root.attrs, 'layer_names'):
        g = root.children.get(ln)
        if g is None:
            raise H5FormatError('layer group %r is missing' % ln)
        ws = []
        for wn in _split_chunks(g.attrs, 'weight_names'):
            try:
                ds = g[wn]
            except KeyError:
                raise H5FormatError('dataset %s/%s is missing' % (ln, wn))
            ws.append((wn, np.asarray(ds.read(), np.float32)))
        out[ln] = ws
    meta = {k: (v.decode('utf8') if isinstance(v, (bytes, np.bytes_)) else v) for k, v in attrs.items()
            if k in ('backend', 'keras_version')}
    return out, meta
