"""Writes tests/golden/keras_ref_libhdf5.h5 with the REAL HDF5 library (libhdf5 1.10.6 from /opt/conda, driven through ctypes --
h5py is not installed anywhere in the build container), in the layout tf.keras 2.3 ``Model.save_weights`` produces through h5py
(``save_weights_to_hdf5_group``: root attrs layer_names / backend / keras_version, one group per layer with weight_names, datasets
'<layer>/<weight>:0' created with intermediate groups; an empty weight_names list is a float64 attribute of shape (0,), which is
what h5py makes of ``[]``).  The model is this package's layer table for DIM 16x16, FILTERS 4, DEPTH 1 with BN (Keras auto-names);
the values are seeded, so the test regenerates the expected arrays.  The fixture pins the in-tree READER to a file no code of this
repository wrote; tests/test_keras_h5.py pins the WRITER the other way round (libhdf5's h5ls / h5diff read its output) where the
tools exist.

    python tests/golden/make_keras_h5_fixture.py [out.h5]            (build container only; the GPU box uses the committed file)
"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIBHDF5 = os.environ.get('RVIP_LIBHDF5', '/opt/conda/lib/libhdf5.so.103')
CFG = dict(DIM=[16, 16], FILTERS=4, DEPTH=1, BATCH_NORMALISATION=True, ACTIVATION='relu', MASK_CLASSES=2)
SEED = 20261004


def expected_weights():
    """[(layer, [(weight name, float32 array)])] for every layer of the table, seeded."""
    rvip = importlib.import_module('cmr-landmark-detection_amd')
    plan = rvip.Unets.UnetPlan(CFG)
    rng = np.random.default_rng(SEED)
    by_layer = {l.name: [] for l in plan.layers}
    for (ln, wn, shape, _, _) in plan.weight_specs():
        by_layer[ln].append(('%s/%s:0' % (ln, wn), rng.standard_normal(shape).astype(np.float32)))
    return [(l.name, by_layer[l.name]) for l in plan.layers]


def write_with_libhdf5(path, layers, backend=b'tensorflow', keras_version=b'2.4.0'):
    L = C.CDLL(LIBHDF5)
    hid = C.c_int64
    L.H5open()
    g = lambda name: hid.in_dll(L, name).value                                                      # noqa: E731
    for fn, res, args in [('H5Fcreate', hid, [C.c_char_p, C.c_uint, hid, hid]), ('H5Gcreate2', hid, [hid, C.c_char_p, hid, hid, hid]),
                          ('H5Screate_simple', hid, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]), ('H5Screate', hid, [C.c_int]),
                          ('H5Tcopy', hid, [hid]), ('H5Tset_size', C.c_int, [hid, C.c_size_t]), ('H5Tset_strpad', C.c_int, [hid, C.c_int]),
                          ('H5Acreate2', hid, [hid, C.c_char_p, hid, hid, hid, hid]), ('H5Awrite', C.c_int, [hid, hid, C.c_void_p]),
                          ('H5Dcreate2', hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]),
                          ('H5Dwrite', C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]), ('H5Pcreate', hid, [hid]),
                          ('H5Pset_create_intermediate_group', C.c_int, [hid, C.c_uint])] + \
                         [(f, C.c_int, [hid]) for f in ('H5Fclose', 'H5Gclose', 'H5Sclose', 'H5Tclose', 'H5Aclose', 'H5Dclose', 'H5Pclose')]:
        getattr(L, fn).restype, getattr(L, fn).argtypes = res, args
    F32, F64, S1 = g('H5T_NATIVE_FLOAT_g'), g('H5T_NATIVE_DOUBLE_g'), g('H5T_C_S1_g')

    def str_attr(loc, name, values, scalar=False):
        if not values:                                                    # h5py: np.asarray([]) -> float64, shape (0,)
            d = (C.c_uint64 * 1)(0)
            s = L.H5Screate_simple(1, d, None)
            a = L.H5Acreate2(loc, name, F64, s, 0, 0)
            assert a >= 0
            L.H5Aclose(a); L.H5Sclose(s)
            return
        n = max(len(v) for v in values)                                   # numpy 'S' array -> fixed-length, null-padded
        t = L.H5Tcopy(S1); L.H5Tset_size(t, n); L.H5Tset_strpad(t, 1)
        s = L.H5Screate(0) if scalar else L.H5Screate_simple(1, (C.c_uint64 * 1)(len(values)), None)
        a = L.H5Acreate2(loc, name, t, s, 0, 0)
        assert a >= 0
        L.H5Awrite(a, t, C.c_char_p(b''.join(v.ljust(n, b'\0') for v in values)))
        L.H5Aclose(a); L.H5Sclose(s); L.H5Tclose(t)

    f = L.H5Fcreate(path.encode(), 2, 0, 0)                               # H5F_ACC_TRUNC, default (libver earliest) property lists
    assert f >= 0
    str_attr(f, b'layer_names', [ln.encode() for ln, _ in layers])
    str_attr(f, b'backend', [backend], scalar=True)
    str_attr(f, b'keras_version', [keras_version], scalar=True)
    lcpl = L.H5Pcreate(g('H5P_CLS_LINK_CREATE_ID_g'))
    L.H5Pset_create_intermediate_group(lcpl, 1)                           # what h5py does for 'conv2d/kernel:0'
    for ln, ws in layers:
        grp = L.H5Gcreate2(f, ln.encode(), 0, 0, 0)
        assert grp >= 0
        str_attr(grp, b'weight_names', [wn.encode() for wn, _ in ws])
        for wn, arr in ws:
            arr = np.ascontiguousarray(arr, np.float32)
            s = L.H5Screate_simple(arr.ndim, (C.c_uint64 * arr.ndim)(*arr.shape), None) if arr.ndim else L.H5Screate(0)
            ds = L.H5Dcreate2(grp, wn.encode(), F32, s, lcpl, 0, 0)
            assert ds >= 0
            L.H5Dwrite(ds, F32, 0, 0, 0, arr.ctypes.data_as(C.c_void_p))
            L.H5Dclose(ds); L.H5Sclose(s)
        L.H5Gclose(grp)
    L.H5Pclose(lcpl)
    L.H5Fclose(f)


if __name__ == '__main__':
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), 'keras_ref_libhdf5.h5')
    write_with_libhdf5(out, expected_weights())
    print(out, os.path.getsize(out), 'bytes')
