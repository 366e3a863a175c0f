"""Host-side pre-processing of the reference's file generator (src/data/Preprocess.py), restated on NumPy / SciPy.

The step BEFORE the hot path: CPU, I/O-bound.  The reference does it with SimpleITK, OpenCV and
albumentations, none of which exist in this image; what is kept is the arithmetic each function documents:

  * ``calc_resampled_size``  Preprocess.py:123-134   new size = round(size * spacing / target spacing)
  * ``resample``             Preprocess.py:182-227   sitk.ResampleImageFilter with the input's origin and direction: output
                                                     voxel i sits at physical i * target_spacing, i.e. at input index
                                                     i * target / spacing; linear (images) or nearest (masks), 0 outside
  * ``clip_quantile``        Preprocess.py:458-468
  * ``pad_and_crop``         Preprocess.py:494-541   centre pad / crop, odd differences: pad (floor, floor+1), crop (floor+1, floor)
  * ``augment``              Preprocess.py:382-422   the configured albumentations subset - RandomRotate90(p=0.2),
                                                     ShiftScaleRotate(shift 0.025, no scale / rotation), GridDistortion -
                                                     applied identically to image and mask, constant border 0.  The random
                                                     stream is NumPy's: augmented samples are not bit-comparable with
                                                     albumentations' (nor is that reproducible across its versions).
  * ``read_image``           NRRD (.nrrd; raw / gzip) and NIfTI-1 (.nii, .nii.gz) readers -> (array in z,y,x order, spacing z,y,x)
"""
from __future__ import annotations

import gzip
import io
import os
import struct

import numpy as np

from .Generators import normalise_image, transform_to_binary_mask  # noqa: F401  (re-exported: same module in the reference)


# ------------------------------------------------------------------------------------------------------------------
# file formats
# ------------------------------------------------------------------------------------------------------------------
_NRRD_TYPES = {'signed char': 'i1', 'int8': 'i1', 'int8_t': 'i1', 'uchar': 'u1', 'unsigned char': 'u1', 'uint8': 'u1', 'uint8_t': 'u1',
               'short': 'i2', 'short int': 'i2', 'signed short': 'i2', 'int16': 'i2', 'int16_t': 'i2',
               'ushort': 'u2', 'unsigned short': 'u2', 'uint16': 'u2', 'uint16_t': 'u2',
               'int': 'i4', 'signed int': 'i4', 'int32': 'i4', 'int32_t': 'i4', 'uint': 'u4', 'unsigned int': 'u4', 'uint32': 'u4',
               'longlong': 'i8', 'long long': 'i8', 'int64': 'i8', 'int64_t': 'i8', 'ulonglong': 'u8', 'uint64': 'u8',
               'float': 'f4', 'double': 'f8'}


def read_nrrd(path):
    """-> (ndarray with the LAST file axis first, i.e. NumPy z,y,x order as sitk.GetArrayFromImage, spacing in the same order)."""
    with open(path, 'rb') as f:
        raw = f.read()
    if not raw.startswith(b'NRRD'):
        raise ValueError('%s: not an NRRD file' % path)
    end = raw.find(b'\n\n')
    sep = 2
    if end < 0 or (0 <= raw.find(b'\r\n\r\n') < end):
        end, sep = raw.find(b'\r\n\r\n'), 4
    header = raw[:end].decode('ascii', 'replace').splitlines()[1:]
    fields = {}
    for line in header:
        if line.startswith('#') or ':' not in line:
            continue
        k, v = line.split(':', 1)
        fields[k.strip().lower()] = v.lstrip('=').strip()
    if 'data file' in fields or 'datafile' in fields:
        raise NotImplementedError('detached NRRD headers are not supported')
    sizes = [int(s) for s in fields['sizes'].split()]
    dt = np.dtype(_NRRD_TYPES[fields['type'].lower()])
    if dt.itemsize > 1:
        dt = dt.newbyteorder('>' if fields.get('endian', 'little').lower() == 'big' else '<')
    enc = fields.get('encoding', 'raw').lower()
    body = raw[end + sep:]
    if enc in ('gzip', 'gz'):
        body = gzip.decompress(body)
    elif enc != 'raw':
        raise NotImplementedError('NRRD encoding %r' % enc)
    n = int(np.prod(sizes))
    arr = np.frombuffer(body, dtype=dt, count=n).reshape(sizes[::-1])          # fastest file axis = last NumPy axis
    spacing = None
    if 'space directions' in fields:
        vecs = [v for v in fields['space directions'].replace('none', '').split(')') if '(' in v]
        spacing = [float(np.linalg.norm([float(t) for t in v.split('(')[1].split(',')])) for v in vecs]
    elif 'spacings' in fields:
        spacing = [float(t) for t in fields['spacings'].split()]
    if spacing is None or len(spacing) != len(sizes):
        spacing = [1.0] * len(sizes)
    return np.ascontiguousarray(arr.astype(dt.newbyteorder('='))), tuple(reversed(spacing))


def write_nrrd(path, arr, spacing=None, gz=True):
    """Minimal writer (tests, exports): arr in NumPy z,y,x order, spacing in the same order."""
    arr = np.ascontiguousarray(arr)
    names = {'i1': 'int8', 'u1': 'uint8', 'i2': 'short', 'u2': 'ushort', 'i4': 'int', 'u4': 'uint', 'i8': 'longlong', 'f4': 'float', 'f8': 'double'}
    key = arr.dtype.kind + str(arr.dtype.itemsize)
    spacing = list(spacing) if spacing is not None else [1.0] * arr.ndim
    hdr = ['NRRD0004', 'type: %s' % names[key], 'dimension: %d' % arr.ndim, 'sizes: %s' % ' '.join(str(s) for s in arr.shape[::-1]),
           'spacings: %s' % ' '.join(repr(float(s)) for s in spacing[::-1]), 'endian: little', 'encoding: %s' % ('gzip' if gz else 'raw')]
    body = arr.astype(arr.dtype.newbyteorder('<')).tobytes()
    with open(path, 'wb') as f:
        f.write(('\n'.join(hdr) + '\n\n').encode('ascii'))
        f.write(gzip.compress(body) if gz else body)


_NIFTI_TYPES = {2: 'u1', 4: 'i2', 8: 'i4', 16: 'f4', 64: 'f8', 256: 'i1', 512: 'u2', 768: 'u4'}


def read_nifti(path):
    """NIfTI-1 single file (.nii / .nii.gz) -> (array z,y,x[,...] order, spacing in that order); scl_slope / inter applied."""
    op = gzip.open if str(path).endswith('.gz') else open
    with op(path, 'rb') as f:
        raw = f.read()
    little = struct.unpack('<i', raw[:4])[0] == 348
    e = '<' if little else '>'
    if struct.unpack(e + 'i', raw[:4])[0] != 348:
        raise ValueError('%s: not a NIfTI-1 file' % path)
    dim = struct.unpack(e + '8h', raw[40:56])
    datatype = struct.unpack(e + 'h', raw[70:72])[0]
    pixdim = struct.unpack(e + '8f', raw[76:108])
    vox_offset = int(struct.unpack(e + 'f', raw[108:112])[0])
    slope, inter = struct.unpack(e + '2f', raw[112:120])
    nd = dim[0]
    shape = [int(d) for d in dim[1:1 + nd]]
    while len(shape) > 3 and shape[-1] == 1:
        shape.pop()
    dt = np.dtype(_NIFTI_TYPES[datatype]).newbyteorder(e)
    arr = np.frombuffer(raw, dtype=dt, count=int(np.prod(shape)), offset=max(vox_offset, 352)).reshape(shape[::-1])
    arr = arr.astype(dt.newbyteorder('='))
    if slope not in (0.0, 1.0) or inter != 0.0:
        arr = arr.astype(np.float32) * (slope if slope != 0.0 else 1.0) + inter
    return np.ascontiguousarray(arr), tuple(float(p) for p in reversed(pixdim[1:1 + len(shape)]))


def read_image(path):
    p = str(path).lower()
    if p.endswith('.nrrd'):
        return read_nrrd(path)
    if p.endswith('.nii') or p.endswith('.nii.gz'):
        return read_nifti(path)
    if p.endswith('.npy'):
        return np.load(path), None
    raise NotImplementedError('unsupported image file: %s' % os.path.basename(str(path)))


# ------------------------------------------------------------------------------------------------------------------
# geometry / intensity
# ------------------------------------------------------------------------------------------------------------------
def calc_resampled_size(size, spacing, target_spacing):
    """Preprocess.py:123-134 (all three in the same axis order)."""
    return [int(v) for v in np.around(np.asarray(size, float) * np.asarray(spacing, float) / np.asarray(target_spacing, float))]


def resample(nda, spacing, target_spacing, size=None, order=1):
    """Preprocess.py:182-227 on an array: same origin and axes, output index i reads input index i * target / spacing;
    order 1 = sitkLinear, 0 = sitkNearestNeighbor; samples outside the input are 0 (the filter's default pixel value)."""
    import scipy.ndimage
    nda = np.asarray(nda)
    spacing, target_spacing = np.asarray(spacing, float), np.asarray(target_spacing, float)
    if size is None:
        size = calc_resampled_size(nda.shape, spacing, target_spacing)
    step = target_spacing / spacing
    coords = np.meshgrid(*[np.arange(s) * st for s, st in zip(size, step)], indexing='ij')
    out = scipy.ndimage.map_coordinates(nda.astype(np.float64 if order else nda.dtype), coords, order=order, mode='constant', cval=0.0,
                                        prefilter=False)
    if order:
        # scipy extrapolates linearly inside the last half voxel; the ITK interpolator is only defined up to the last index
        for ax, (c, n) in enumerate(zip(coords, nda.shape)):
            out = np.where(c > n - 1, 0.0, out)
        return out.astype(np.float32)
    return out.astype(nda.dtype)


def clip_quantile(img_nda, upper_quantile=.999, lower_boundary=0):
    """Preprocess.py:458-468."""
    return np.clip(img_nda, lower_boundary, np.quantile(np.asarray(img_nda).flatten(), upper_quantile))


def pad_and_crop(ndarray, target_shape=(10, 10, 10)):
    """Preprocess.py:494-541: centre pad (zeros) / crop per axis.  Odd differences: the reference takes floor(x / 2) of the SIGNED
    difference, so both padding (x < 0: |floor(x/2)| in front, one less behind) and cropping (floor(x/2) + 1 in front) put the
    extra element in FRONT -- pinned by tests/golden/ref_numpy_fixtures.npz, which the reference's own function produced."""
    ndarray = np.asarray(ndarray)
    out = np.zeros(tuple(target_shape), dtype=np.float64)
    src, dst = [], []
    for n, t in zip(ndarray.shape, target_shape):
        diff = n - t
        if diff > 0:                                   # crop: (floor+1, floor) for odd differences
            front = diff // 2 + (diff % 2)
            src.append(slice(front, front + t)); dst.append(slice(0, t))
        elif diff < 0:                                 # pad: (|floor(diff / 2)|, |floor(diff / 2) + 1|)
            front = (-diff + 1) // 2
            src.append(slice(0, n)); dst.append(slice(front, front + n))
        else:
            src.append(slice(0, n)); dst.append(slice(0, t))
    out[tuple(dst)] = ndarray[tuple(src)]
    return out


# ------------------------------------------------------------------------------------------------------------------
# augmentation (the configured albumentations subset, Preprocess.py:382-422), image and mask transformed identically
# ------------------------------------------------------------------------------------------------------------------
def _remap(img, yy, xx, order):
    import scipy.ndimage
    return scipy.ndimage.map_coordinates(img, [yy, xx], order=order, mode='constant', cval=0.0, prefilter=False)


def augment(img, mask, config=None, rng=None, probability=None):
    """img, mask: 2-D arrays (a 3-D stack is transformed slice-wise with ONE set of parameters, like the reference's
    replay of a single albumentations draw, Preprocess.py:230-350).  Returns (img, mask)."""
    config = config or {}
    rng = rng or np.random.default_rng()
    if img.ndim == 3:
        state = rng.bit_generator.state
        out = []
        for a, b in zip(img, mask):
            rng.bit_generator.state = state                       # same draw for every slice
            out.append(augment(a, b, config, rng, probability))
        return np.stack([o[0] for o in out]), np.stack([o[1] for o in out])
    p_all = config.get('AUGMENT_PROB', 0.8) if probability is None else probability
    prob = config.get('AUGMENT_PROB', 0.8)
    if rng.random() >= p_all:
        return img, mask
    h, w = img.shape
    if config.get('RANDOMROTATE', False) and rng.random() < 0.2:
        k = int(rng.integers(0, 4))
        img, mask = np.rot90(img, k), np.rot90(mask, k)
        if img.shape != (h, w):                                     # non-square: keep the frame (albumentations would change it)
            img, mask = pad_and_crop(img, (h, w)), pad_and_crop(mask, (h, w)).astype(mask.dtype)
    if config.get('SHIFTSCALEROTATE', False) and rng.random() < prob:
        dy, dx = rng.uniform(-0.025, 0.025, 2) * (h, w)
        yy, xx = np.meshgrid(np.arange(h) - dy, np.arange(w) - dx, indexing='ij')
        img, mask = _remap(img, yy, xx, 1), _remap(mask, yy, xx, 0)
    if config.get('GRIDDISTORTION', False) and rng.random() < prob:
        steps, limit = 5, 0.3

        def axis_map(n):
            cell = n / steps
            ratios = 1.0 + rng.uniform(-limit, limit, steps + 1)
            src = np.concatenate([[0.0], np.cumsum(cell * ratios)])[:steps + 1]
            src = src / src[-1] * (n - 1) if src[-1] > 0 else np.linspace(0, n - 1, steps + 1)
            return np.interp(np.arange(n), np.linspace(0, n - 1, steps + 1), src)
        yy, xx = np.meshgrid(axis_map(h), axis_map(w), indexing='ij')
        img, mask = _remap(img, yy, xx, 1), _remap(mask, yy, xx, 0)
    return img, mask
