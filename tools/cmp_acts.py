import numpy as np, sys
a=np.load('/tmp/acts_%s.npz' % sys.argv[1]); b=np.load('/tmp/acts_%s.npz' % sys.argv[2])
for k in a.files:
    d=np.abs(a[k].astype(np.float64)-b[k].astype(np.float64))
    bad=np.argwhere(d>1e-3)
    print('%-28s %s max %.4g mean %.4g nbad %d first %s' % (k, a[k].shape, np.nanmax(d), np.nanmean(d), len(bad), bad[:3].tolist()))
