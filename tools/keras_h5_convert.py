"""model.h5 (Keras `save_weights` / ModelCheckpoint(save_weights_only=True), KerasCallbacks.py:54-61) <-> the .npz
container this package reads and writes (`<layer>/<weight>:0` keys in Keras get_weights() order).

Needs h5py, which is NOT in the build image: run it on a machine that has the reference's environment.  It was written
against the documented Keras HDF5 weight layout (file attrs `layer_names`; per layer group attrs `weight_names`; one
dataset per weight) and is unverified here for that reason - SURVEY.md 8(f) row 1.

    python tools/keras_h5_convert.py to-npz model.h5 model.npz
    python tools/keras_h5_convert.py to-h5  model.npz model.h5
"""
import sys

import numpy as np


def _s(b):
    return b.decode('utf8') if isinstance(b, bytes) else str(b)


def h5_to_npz(src, dst):
    import h5py
    out = {}
    with h5py.File(src, 'r') as f:
        g = f['model_weights'] if 'model_weights' in f else f          # full-model files nest the weights
        for lname in (_s(n) for n in g.attrs['layer_names']):
            for wname in (_s(n) for n in g[lname].attrs['weight_names']):
                out[wname if wname.startswith(lname + '/') else lname + '/' + wname] = np.asarray(g[lname][wname])
    np.savez(dst, **out)
    return len(out)


def npz_to_h5(src, dst):
    import h5py
    data = np.load(src)
    layers = []
    for key in data.files:
        lname = key.split('/')[0]
        if lname not in layers:
            layers.append(lname)
    with h5py.File(dst, 'w') as f:
        f.attrs['layer_names'] = np.array([l.encode('utf8') for l in layers])
        f.attrs['backend'] = b'tensorflow'
        f.attrs['keras_version'] = b'2.4.0'
        for lname in layers:
            g = f.create_group(lname)
            names = [k for k in data.files if k.split('/')[0] == lname]
            g.attrs['weight_names'] = np.array([n.encode('utf8') for n in names])
            for n in names:
                g.create_dataset(n, data=np.asarray(data[n], np.float32))
    return len(data.files)


if __name__ == '__main__':
    if len(sys.argv) != 4 or sys.argv[1] not in ('to-npz', 'to-h5'):
        raise SystemExit(__doc__)
    n = h5_to_npz(sys.argv[2], sys.argv[3]) if sys.argv[1] == 'to-npz' else npz_to_h5(sys.argv[2], sys.argv[3])
    print('%d arrays' % n)
