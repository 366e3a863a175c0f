"""Keras-HDF5 weights interop: the in-tree HDF5 subset of cmr-landmark-detection_amd/keras_h5.py behind
``Model.save_weights('model.h5')`` / ``load_weights`` (reference: KerasCallbacks.py:54-61 writes, predict_model.py:75-76 reads).

Pins: (1) the READER against tests/golden/keras_ref_libhdf5.h5, written by the real HDF5 library (libhdf5 1.10.6 through ctypes,
tests/golden/make_keras_h5_fixture.py) in the Keras layout; (2) the WRITER against the HDF5 file-format fields it emits, byte by
byte, and -- where libhdf5's command-line tools exist (the build container's /opt/conda) -- against the library itself: h5ls lists
the same objects and h5diff finds no difference to a file libhdf5 wrote from the same arrays.  No GPU."""
import importlib
import os
import shutil
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rvip = importlib.import_module('cmr-landmark-detection_amd')
H5 = importlib.import_module('cmr-landmark-detection_amd.keras_h5')
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_keras_h5_fixture as FX   # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden', 'keras_ref_libhdf5.h5')
H5LS = shutil.which('h5ls') or ('/opt/conda/bin/h5ls' if os.path.exists('/opt/conda/bin/h5ls') else None)
H5DIFF = shutil.which('h5diff') or ('/opt/conda/bin/h5diff' if os.path.exists('/opt/conda/bin/h5diff') else None)
HAVE_LIB = os.path.exists(FX.LIBHDF5)


def _cfg(**kw):
    c = dict(DIM=[32, 32], FILTERS=8, DEPTH=2, BATCH_NORMALISATION=True, ACTIVATION='relu', MASK_CLASSES=2,
             LOSS_FUNCTION=rvip.Loss_and_metrics.mse)
    c.update(kw)
    return c


def test_reader_on_file_written_by_libhdf5():
    layers, meta = H5.load_keras_weights(GOLDEN)
    want = FX.expected_weights()
    assert meta == {'backend': 'tensorflow', 'keras_version': '2.4.0'}
    assert list(layers) == [ln for ln, _ in want]                                # layer_names order = model.layers order
    for (ln, ws), (_, got) in zip(want, layers.items()):
        assert [n for n, _ in got] == [n for n, _ in ws]
        for (_, a), (_, b) in zip(ws, got):
            assert b.dtype == np.float32 and a.shape == b.shape
            np.testing.assert_array_equal(a, b)
    root = H5.H5Reader(GOLDEN).root()
    assert root.attrs['layer_names'].dtype.kind == 'S' and root.attrs['backend'] == b'tensorflow'
    assert root['input_1'].attrs['weight_names'].shape == (0,)                   # h5py's rendering of []
    assert root['conv2d/conv2d/kernel:0'].shape == (3, 3, 1, 4)
    # ... and straight into a model of the same config (Keras load_weights by topology)
    m = rvip.create_unet(dict(FX.CFG, LOSS_FUNCTION=rvip.Loss_and_metrics.mse))
    m.load_weights(GOLDEN)
    flat = [a for _, ws in want for _, a in ws]
    for a, b in zip(flat, m.get_weights()):
        np.testing.assert_array_equal(a, b)


def test_model_h5_round_trip_and_keras_layout(tmp_path):
    m = rvip.create_unet(_cfg())
    w = m.get_weights()
    p = str(tmp_path / 'model.h5')
    m.save_weights(p)
    root = H5.H5Reader(p).root()
    names = [n.decode() for n in root.attrs['layer_names']]
    assert names == [l.name for l in m.plan.layers]                              # EVERY layer, Keras auto-names
    assert names[:4] == ['input_1', 'conv2d', 'batch_normalization', 'dropout'] and names[-1] == 'unet'
    assert root.attrs['backend'] == b'tensorflow' and root.attrs['keras_version'] == b'2.4.0'
    assert [n.decode() for n in root['batch_normalization'].attrs['weight_names']] == [
        'batch_normalization/gamma:0', 'batch_normalization/beta:0', 'batch_normalization/moving_mean:0',
        'batch_normalization/moving_variance:0']
    assert root['dropout'].attrs['weight_names'].shape == (0,) and root['dropout'].attrs['weight_names'].dtype == np.float64
    ds = root['conv2d_1/conv2d_1/kernel:0']
    assert ds.shape == (3, 3, 8, 8) and ds.dtype == np.dtype('<f4') and ds._layout[0] == 'contiguous'
    m2 = rvip.create_unet(_cfg(SEED=7))
    assert not np.array_equal(m2.get_weights()[0], w[0])
    m2.load_weights(p)
    for a, b in zip(m2.get_weights(), w):
        np.testing.assert_array_equal(a, b)
    # by_name, extension dispatch, overwrite guard, the .npz container still works
    m3 = rvip.create_unet(_cfg(SEED=9))
    m3.load_weights(p, by_name=True)
    np.testing.assert_array_equal(m3.get_weights()[5], w[5])
    with pytest.raises(FileExistsError):
        m.save_weights(p, overwrite=False)
    q = str(tmp_path / 'model.npz')
    m.save_weights(q)
    m3 = rvip.create_unet(_cfg(SEED=11))
    m3.load_weights(q)
    np.testing.assert_array_equal(m3.get_weights()[0], w[0])
    # a checkpoint of another architecture is refused the way Keras refuses it
    with pytest.raises(ValueError, match='layers'):
        rvip.create_unet(_cfg(DEPTH=1)).load_weights(p)
    with pytest.raises(ValueError, match='shape'):
        rvip.create_unet(_cfg(FILTERS=16)).load_weights(p)
    # Conv2DTranspose decoder (USE_UPSAMPLE=False): kernel stored [kh,kw,Cout,Cin] under conv2d_transpose
    mt = rvip.create_unet(_cfg(USE_UPSAMPLE=False))
    pt = str(tmp_path / 't.h5')
    mt.save_weights(pt)
    rt = H5.H5Reader(pt).root()
    assert rt['conv2d_transpose/conv2d_transpose/kernel:0'].shape == (3, 3, 16, 32)
    mt2 = rvip.create_unet(_cfg(USE_UPSAMPLE=False, SEED=3))
    mt2.load_weights(pt)
    for a, b in zip(mt2.get_weights(), mt.get_weights()):
        np.testing.assert_array_equal(a, b)


def test_writer_bytes_follow_the_hdf5_format_spec(tmp_path):
    """Every structure the writer emits, decoded by hand from the bytes (HDF5 File Format Specification, superblock v0 /
    object header v1 / group B-tree v1 + SNOD + local heap / messages 0x1 0x3 0x5 0x8 0xC 0x11)."""
    arr = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
    p = str(tmp_path / 'w.h5')
    H5.save_keras_weights(p, [('input_1', []), ('conv2d', [('conv2d/kernel:0', arr), ('conv2d/bias:0', arr[0, 0])])])
    b = open(p, 'rb').read()
    u8 = lambda o: struct.unpack_from('<Q', b, o)[0]                                     # noqa: E731
    # -- superblock version 0 (56 bytes) + root symbol-table entry (40 bytes)
    assert b[:8] == b'\x89HDF\r\n\x1a\n'
    assert tuple(b[8:16]) == (0, 0, 0, 0, 0, 8, 8, 0)                                     # versions 0, size of offsets / lengths 8
    assert struct.unpack_from('<HHI', b, 16) == (4, 16, 0)                               # leaf K, internal K, consistency flags
    assert u8(24) == 0 and u8(32) == H5.UNDEF and u8(40) == len(b) and u8(48) == H5.UNDEF  # base, free-space, END OF FILE, driver
    assert u8(56) == 0 and struct.unpack_from('<II', b, 72) == (1, 0)                     # root entry: name offset 0, cache type 1 (stab)
    root_oh, root_bt, root_hp = u8(64), u8(80), u8(88)
    assert root_oh % 8 == 0 and root_bt % 8 == 0 and root_hp % 8 == 0
    # -- root object header, version 1: 16-byte prefix, messages 8-byte aligned
    ver, _, nmsg, refs, size = struct.unpack_from('<BBHII', b, root_oh)
    assert (ver, refs) == (1, 1) and nmsg == 4                                            # stab + 3 attributes
    q, kinds = root_oh + 16, []
    while q < root_oh + 16 + size:
        mtype, msize, mflags = struct.unpack_from('<HHB', b, q)
        assert msize % 8 == 0
        kinds.append(mtype)
        if mtype == 0x11:
            assert (u8(q + 8), u8(q + 16)) == (root_bt, root_hp)                          # symbol-table message = the cached scratch pad
        if mtype == 0xC and b[q + 16:q + 27] == b'layer_names':
            v, _, nsz, tsz, ssz = struct.unpack_from('<BBHHH', b, q + 8)
            assert (v, nsz, tsz, ssz) == (1, 12, 8, 24)                                   # attribute v1: name incl. NUL, datatype, dataspace sizes
            t = q + 16 + 16                                                               # name padded to 16
            assert b[t] == 0x13 and b[t + 1] == 0x01 and struct.unpack_from('<I', b, t + 4)[0] == 7   # string class v1, null-pad ASCII, len('input_1')
            s = t + 8
            assert tuple(b[s:s + 4]) == (1, 1, 1, 0) and u8(s + 8) == 2 and u8(s + 16) == 2   # dataspace v1, rank 1, max dims present
            assert b[s + 24:s + 24 + 14] == b'input_1conv2d\0'
        q += 8 + msize
    assert kinds == [0x11, 0xC, 0xC, 0xC] and q == root_oh + 16 + size
    # -- group B-tree node (544 bytes for K = 16), symbol-table node (328 bytes), local heap
    assert b[root_bt:root_bt + 4] == b'TREE' and tuple(b[root_bt + 4:root_bt + 6]) == (0, 0)
    assert struct.unpack_from('<H', b, root_bt + 6)[0] == 1 and u8(root_bt + 8) == H5.UNDEF and u8(root_bt + 16) == H5.UNDEF
    key0, snod, key1 = u8(root_bt + 24), u8(root_bt + 32), u8(root_bt + 40)
    assert b[snod:snod + 4] == b'SNOD' and b[snod + 4] == 1 and struct.unpack_from('<H', b, snod + 6)[0] == 2
    assert b[root_hp:root_hp + 4] == b'HEAP' and b[root_hp + 4] == 0
    hsize, hfree, hdata = u8(root_hp + 8), u8(root_hp + 16), u8(root_hp + 24)
    assert hfree == 1 and hsize % 8 == 0                                                  # H5HL_FREE_NULL: no free block
    heap = b[hdata:hdata + hsize]
    assert heap[:8] == b'\0' * 8 and key0 == 0                                            # the empty name sits at offset 0 = left-most key
    e0, e1 = snod + 8, snod + 48
    n0, n1 = u8(e0), u8(e1)
    assert heap[n0:n0 + 7] == b'conv2d\0' and heap[n1:n1 + 8] == b'input_1\0' and key1 == n1   # entries sorted by name, right key = last name
    assert struct.unpack_from('<I', b, e0 + 16)[0] == 1                                   # child groups cache their B-tree / heap
    assert len(b) >= snod + 328 and len(b) >= root_bt + 544                               # nodes are allocated at full size
    # -- the dataset: dataspace v1, IEEE f32 LE datatype, fill v2, contiguous layout v3 pointing at the raw values
    rd = H5.H5Reader(p)
    msgs = rd._messages(rd._symbols(*(lambda d: (rd._off(d, 0), rd._heap(rd._off(d, 8))))(
        [m for m in rd._messages(rd._symbols(*(lambda d: (rd._off(d, 0), rd._heap(rd._off(d, 8))))(
            [m for m in rd._messages(rd._symbols(root_bt, heap)['conv2d']) if m[0] == 0x11][0][2]))['conv2d']) if m[0] == 0x11][0][2]))['kernel:0'])
    by = {m[0]: m for m in msgs}
    assert sorted(by) == [0x1, 0x3, 0x5, 0x8]
    assert by[0x1][2][:8] == bytes([1, 3, 1, 0, 0, 0, 0, 0]) and struct.unpack_from('<6Q', by[0x1][2], 8) == (2, 3, 4, 2, 3, 4)
    assert by[0x3][2][:20].hex() == '11201f00040000000000200017080017' + '7f000000'        # == the bytes libhdf5 writes for H5T_IEEE_F32LE
    assert by[0x5][2][:8].hex() == '0202020100000000'                                     # == libhdf5's default fill-value message
    lv, lc, addr, nbytes = struct.unpack_from('<BBQQ', by[0x8][2], 0)
    assert (lv, lc, nbytes) == (3, 1, 96) and b[addr:addr + 96] == arr.tobytes()


def test_many_layers_need_a_two_level_group_btree(tmp_path):
    layers = [('layer_%03d' % i, [('layer_%03d/w:0' % i, np.full((2,), i, np.float32))]) for i in range(300)]   # > 32 SNODs of 8
    p = str(tmp_path / 'big.h5')
    H5.save_keras_weights(p, layers)
    got, _ = H5.load_keras_weights(p)
    assert list(got) == [ln for ln, _ in layers]
    assert all(float(ws[0][1][0]) == i for i, (_, ws) in enumerate(got.items()))
    b = open(p, 'rb').read()
    root_bt = struct.unpack_from('<Q', b, 80)[0]
    assert b[root_bt:root_bt + 4] == b'TREE' and b[root_bt + 5] == 1                      # root node at level 1
    if H5LS:
        out = subprocess.run([H5LS, p], capture_output=True, text=True, check=True).stdout.split('\n')
        assert len([l for l in out if l.strip()]) == 300


def test_reader_rejects_what_it_does_not_read(tmp_path):
    with pytest.raises(H5.H5FormatError, match='signature'):
        H5.H5Reader(b'not an hdf5 file' * 8)
    p = str(tmp_path / 'x.h5')
    H5.save_keras_weights(p, [('a', [('a/w:0', np.ones(3, np.float32))])])
    b = bytearray(open(p, 'rb').read())
    H5.H5Reader(bytes(b)).root()
    b[8] = 7                                                                              # unknown superblock version
    with pytest.raises(H5.H5FormatError, match='superblock'):
        H5.H5Reader(bytes(b))
    with pytest.raises(H5.H5FormatError, match='beyond the end'):
        H5.H5Reader(open(p, 'rb').read()[:200]).root()                                    # truncated file
    w = H5.H5Writer()
    w.create_dataset('d', np.ones(2, np.float32))
    q = str(tmp_path / 'nokeras.h5')
    open(q, 'wb').write(w.tobytes())
    with pytest.raises(H5.H5FormatError, match='layer_names'):
        H5.load_keras_weights(q)


@pytest.mark.skipif(not (HAVE_LIB and H5LS and H5DIFF), reason='libhdf5 and its tools are only in the build container (/opt/conda)')
def test_libhdf5_reads_what_the_writer_wrote(tmp_path):
    m = rvip.create_unet(_cfg())
    mine, theirs = str(tmp_path / 'mine.h5'), str(tmp_path / 'lib.h5')
    m.save_weights(mine)
    w = m.get_weights()
    layers = [(ln, [(wn, w[i]) for wn, i in ws]) for ln, ws in m._layers_with_weights()]
    FX.write_with_libhdf5(theirs, layers)
    ls = lambda f: subprocess.run([H5LS, '-r', f], capture_output=True, text=True, check=True).stdout      # noqa: E731
    assert ls(mine) == ls(theirs) and 'conv2d_1/conv2d_1/kernel:0 Dataset {3, 3, 8, 8}' in ls(mine)
    r = subprocess.run([H5DIFF, '-c', theirs, mine], capture_output=True, text=True)      # objects AND attributes, values compared
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'not comparable' not in r.stdout.lower() and 'differences' not in r.stdout.lower(), r.stdout
    # the committed golden file is what the fixture script produces today
    again = str(tmp_path / 'again.h5')
    FX.write_with_libhdf5(again, FX.expected_weights())
    assert subprocess.run([H5DIFF, '-c', again, GOLDEN], capture_output=True, text=True).returncode == 0
