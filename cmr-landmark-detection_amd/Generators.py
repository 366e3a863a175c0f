"""Data-generator protocol of the reference (src/data/Generators.py:26-232 ``BaseGenerator``, :234-398
``DataGenerator``), without its file I/O.

Contract kept: ``__len__`` = floor(N / BATCHSIZE) (:142); ``__getitem__(i)`` ->
``(x float32 [B,*DIM,1] min-max normalised, y float32 [B,*DIM,len(MASK_VALUES)])`` (:97-98,228,379);
``on_epoch_end`` reshuffles the index list when SHUFFLE (:164-173); iteration ``for x, y in gen`` works
(predict_model.py:136-139).  With ``GAUS`` the targets are per-channel Gaussian-filtered one-hot masks, min-max
normalised over all channels together (:385-391) -- the heat-map regression targets of the RVIP model.

``SyntheticSAXGenerator`` produces slices of the same contract from a seed (no ACDC data is available to this
build), ``ArrayGenerator`` wraps arrays already in memory, ``DataGenerator`` is the file-based generator
(:234-398) on the NumPy / SciPy restatement of the reference's pre-processing in ``Preprocess.py`` (SURVEY 8(f)
row 4: CPU, I/O-bound, outside the hot path).
"""
from __future__ import annotations

import sys
import threading

import numpy as np


def normalise_image(img_nda, normaliser='minmax'):
    """src/data/Preprocess.py:471-491 (min-max / standard branches)."""
    normaliser = normaliser.lower()
    if normaliser == 'standard':
        return (img_nda - np.mean(img_nda)) / (np.std(img_nda) + sys.float_info.epsilon)
    if normaliser == 'robust':
        raise NotImplementedError("SCALER='Robust' needs scikit-learn's RobustScaler; not on the synthetic path")
    return (img_nda - img_nda.min()) / (img_nda.max() - img_nda.min() + sys.float_info.epsilon)


def transform_to_binary_mask(mask_nda, mask_values=(0, 1, 2, 3)):
    """src/data/Preprocess.py:425-437: label image -> one channel per mask value."""
    mask = np.zeros((*mask_nda.shape, len(mask_values)), dtype=bool)
    for ix, mask_value in enumerate(mask_values):
        mask[..., ix] = mask_nda == mask_value
    return mask


def gaussian_heatmaps(mask_onehot, sigma):
    """GAUS branch of DataGenerator (Generators.py:385-391)."""
    import scipy.ndimage
    g = np.stack([scipy.ndimage.gaussian_filter(mask_onehot[..., c].astype(np.float32), sigma)
                  for c in range(mask_onehot.shape[-1])], axis=-1)
    return normalise_image(g, normaliser='minmax')


class BaseGenerator:
    """keras.utils.Sequence protocol with the reference's index bookkeeping (Generators.py:136-173)."""

    def __init__(self, n_samples, config=None):
        config = config or {}
        self.config = config
        self.SCALER = config.get('SCALER', 'MinMax')
        self.SHUFFLE = config.get('SHUFFLE', True)
        self.SEED = config.get('SEED', 42)
        self.DIM = list(config.get('DIM', [256, 256]))
        self.BATCHSIZE = config.get('BATCHSIZE', 32)
        self.MASK_VALUES = config.get('MASK_VALUES', [0, 1, 2, 3])
        self.N_CLASSES = len(self.MASK_VALUES)
        self.GAUS = config.get('GAUS', False)
        self.SIGMA = config.get('SIGMA', 1)
        self.INDICES = list(range(n_samples))
        self._epochs_seen = 0
        self.samples_generated = 0          # samples this process has produced (data-parallel tests: B / world per step and rank)
        self._count_lock = threading.Lock() # __getitem__ may run on several pool threads (fit(workers=))
        self._tls = threading.local()       # the reshuffle count a fetch started under (augmentation seed), per fetching thread
        self.on_epoch_end()

    def __len__(self):
        return int(np.floor(len(self.INDICES) / self.BATCHSIZE))

    def __getitem__(self, index):
        epoch, order = self._order          # ONE read: a fetch that overlaps on_epoch_end() (stager thread) keeps its epoch's order and seed
        return self._generate(order[index * self.BATCHSIZE:(index + 1) * self.BATCHSIZE], epoch)

    def batch_slice(self, index, lo, hi):
        """Samples lo..hi-1 of batch `index` only.  The reference is ONE process under MirroredStrategy (Unets.py:70-75): every
        sample of the global batch is generated once (Generators.py:175-228) and Keras splits the batch over the replicas.  Here
        every replica is a process, all of which agree on INDICES (seeded shuffles), so each asks for its own slice of the global
        batch instead of generating all of it and discarding (world - 1) / world (Model.fit)."""
        if not (0 <= lo <= hi <= self.BATCHSIZE):
            raise IndexError('slice %d:%d of a batch of %d' % (lo, hi, self.BATCHSIZE))
        base = index * self.BATCHSIZE
        epoch, order = self._order
        return self._generate(order[base + lo:base + hi], epoch)

    def _generate(self, idxs, epoch):
        """__data_generation__ (the reference's name and signature, which subclasses override) under the reshuffle count the fetch
        started with: per-sample seeds read it from the fetching thread's slot, not from the generator's mutable state."""
        self._tls.epoch = epoch
        try:
            return self.__data_generation__(idxs)
        finally:
            del self._tls.epoch

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def on_epoch_end(self):
        self.INDICES = np.arange(len(self.INDICES))
        if self.SHUFFLE:
            # The reference draws from the process-global NumPy RNG (:172; its SEED key is read and never used, :90).  Here the
            # permutation comes from (SEED, number of reshuffles): every data-parallel rank -- a process of its own, unlike the
            # one-process MirroredStrategy -- must see the same order to slice the same global batch.
            self.INDICES = np.random.default_rng([int(self.SEED), self._epochs_seen]).permutation(len(self.INDICES))
        self._epochs_seen += 1
        self._order = (self._epochs_seen, self.INDICES)      # published together (tuple assignment is atomic)

    def __data_generation__(self, idxs):
        x = np.empty((len(idxs), *self.DIM, 1), dtype=np.float32)
        y = np.empty((len(idxs), *self.DIM, self.N_CLASSES), dtype=np.float32)
        for i, ID in enumerate(idxs):
            x[i], y[i] = self.__preprocess_one_image__(i, int(ID))
        with self._count_lock:
            self.samples_generated += len(idxs)
        return x, y

    def __preprocess_one_image__(self, i, ID):
        raise NotImplementedError


class SyntheticSAXGenerator(BaseGenerator):
    """Seeded SAX-like slices: low-pass noise image, two RVIP landmarks per slice; targets one-hot points (GAUS
    False) or Gaussian heat-maps (GAUS True, SIGMA).  Sample ID fully determines the sample."""

    def __init__(self, n_samples, config=None, in_memory=False):
        config = dict(config or {})
        config.setdefault('MASK_VALUES', [1, 2])
        super().__init__(n_samples, config)
        self.IN_MEMORY = in_memory
        self._cache = {}

    def __preprocess_one_image__(self, i, ID):
        if self.IN_MEMORY and ID in self._cache:
            return self._cache[ID]
        rng = np.random.default_rng([self.SEED, ID])
        if len(self.DIM) == 3:                       # cine volume [T,H,W]: every frame drawn like a slice
            frames = [self._slice(rng, self.DIM[1], self.DIM[2]) for _ in range(self.DIM[0])]
            out = (np.stack([f[0] for f in frames]), np.stack([f[1] for f in frames]))
        else:
            out = self._slice(rng, self.DIM[0], self.DIM[1])
        if self.IN_MEMORY:
            self._cache[ID] = out
        return out

    def _slice(self, rng, h, w):
        import scipy.ndimage
        img = scipy.ndimage.gaussian_filter(rng.random((h, w)), min(8.0, h / 8.0))
        img = normalise_image(img, self.SCALER).astype(np.float32)
        lab = np.zeros((h, w), np.int32)
        m = min(16, h // 4, w // 4)
        for v in self.MASK_VALUES:
            lab[int(rng.integers(m, h - m)), int(rng.integers(m, w - m))] = v
        mask = transform_to_binary_mask(lab, self.MASK_VALUES)
        mask = gaussian_heatmaps(mask, self.SIGMA) if self.GAUS else mask.astype(np.float32)
        return img[..., None], mask.astype(np.float32)


class ArrayGenerator(BaseGenerator):
    """Sequence over arrays already in memory (x [N,*DIM,1], y [N,*DIM,C])."""

    def __init__(self, x, y, batch_size, shuffle=False):
        self._x = np.asarray(x, np.float32)
        self._y = None if y is None else np.asarray(y, np.float32)
        cfg = dict(BATCHSIZE=batch_size, DIM=list(self._x.shape[1:-1]), SHUFFLE=shuffle,
                   MASK_VALUES=list(range(1 if y is None else self._y.shape[-1])))
        super().__init__(self._x.shape[0], cfg)

    def __data_generation__(self, idxs):
        idxs = np.asarray(idxs)
        with self._count_lock:
            self.samples_generated += len(idxs)
        return self._x[idxs], (None if self._y is None else self._y[idxs])


class DataGenerator(BaseGenerator):
    """File-based generator of the reference (Generators.py:234-398) on the NumPy / SciPy restatement of its
    pre-processing (``Preprocess.py`` of this package): ``x`` / ``y`` are lists of image / label files (.nrrd, .nii(.gz),
    .npy), one 2-D slice (or one 3-D volume when ``DIM`` has three entries) per file.

    ``__fix_preprocessing__`` (:283-337): read, resample to ``SPACING`` (linear for the image, nearest for the labels)
    when ``RESAMPLE``, clip to the 0.999 quantile, normalise.  ``__preprocess_one_image__`` (:339-398): augmentation
    (``AUGMENT``), centre pad / crop to ``DIM``, normalise again, labels -> one channel per ``MASK_VALUES`` entry and,
    with ``GAUS``, Gaussian heat-maps min-max normalised over all channels.  Not built: ``MASKING_IMAGE``,
    ``HIST_MATCHING`` (both off in the reference's template config)."""

    def __init__(self, x=None, y=None, config=None, in_memory=False):
        config = dict(config or {})
        if config.get('MASKING_IMAGE', False) or config.get('HIST_MATCHING', False):
            raise NotImplementedError('MASKING_IMAGE / HIST_MATCHING are not built')
        self.IMAGES = list(x or [])
        self.LABELS = list(y) if y is not None else None
        if self.LABELS is not None and len(self.LABELS) != len(self.IMAGES):
            raise ValueError('x and y must list the same number of files')
        self.MASKS = self.LABELS is not None
        self.RESAMPLE = config.get('RESAMPLE', False)
        self.SPACING = list(config.get('SPACING', [1.25, 1.25]))
        self.AUGMENT = config.get('AUGMENT', False)
        self.AUGMENT_PROB = config.get('AUGMENT_PROB', 0.8)
        self.IN_MEMORY = in_memory
        super().__init__(len(self.IMAGES), config)
        if not self.MASKS:
            self.N_CLASSES = 1                                   # the image is yielded twice (Generators.py:393-395)
        self._processed = {}
        if self.IN_MEMORY:
            for i in range(len(self.IMAGES)):
                self._processed[i] = self.__fix_preprocessing__(i)

    def __fix_preprocessing__(self, ID):
        from . import Preprocess as pp
        img, sp = pp.read_image(self.IMAGES[ID])
        if self.MASKS:
            msk, _ = pp.read_image(self.LABELS[ID])
        else:
            msk = img
        nd = len(self.DIM)
        if img.ndim != nd or msk.ndim != nd:
            raise ValueError('%s: %d-D file for a %d-D DIM' % (self.IMAGES[ID], img.ndim, nd))
        if self.RESAMPLE:
            if sp is None:
                raise ValueError('%s carries no spacing: cannot RESAMPLE' % self.IMAGES[ID])
            size = pp.calc_resampled_size(img.shape, sp, self.SPACING[-nd:])
            img = pp.resample(img, sp, self.SPACING[-nd:], size, order=1)
            msk = pp.resample(msk, sp, self.SPACING[-nd:], size, order=0 if self.MASKS else 1)
        img = normalise_image(pp.clip_quantile(img.astype(np.float64), .999), self.SCALER)
        if not self.MASKS:
            msk = normalise_image(pp.clip_quantile(msk.astype(np.float64), .999), self.SCALER)
        return img, msk

    def __preprocess_one_image__(self, i, ID):
        from . import Preprocess as pp
        img, msk = self._processed[ID] if ID in self._processed else self.__fix_preprocessing__(ID)
        if self.AUGMENT:
            # The reference draws from albumentations' process-global stream, one sample after the other in its single process.  Here
            # the draws of a sample come from (SEED, reshuffles so far, sample ID): the same sample gets the same augmentation whichever
            # rank, pool thread or batch slice produces it (a shared sequential stream would hand every data-parallel rank the SAME
            # parameter sequence for DIFFERENT samples, and is not safe under fit(workers > 1)).
            rng = np.random.default_rng([int(self.SEED), int(getattr(self._tls, 'epoch', self._epochs_seen)), int(ID)])
            img, msk = pp.augment(img, msk, self.config, rng, self.AUGMENT_PROB)
        img = normalise_image(pp.pad_and_crop(img, self.DIM), self.SCALER)
        msk = pp.pad_and_crop(msk, self.DIM)
        if self.MASKS:
            msk = transform_to_binary_mask(msk, self.MASK_VALUES)
            msk = gaussian_heatmaps(msk, self.SIGMA) if self.GAUS else msk
        else:
            msk = normalise_image(msk, self.SCALER)[..., None]
        return img[..., None].astype(np.float32), np.asarray(msk, np.float32)
