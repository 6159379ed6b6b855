"""Oracle: similarity metrics between the label maps of two exits (CPU, numpy).

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.

The reference computes them with scikit-image (sim_metrics.py:41-120: ``mean_squared_error``,
``normalized_mutual_information``, ``variation_of_information``; ``structural_similarity`` at :15-37), which is NOT installed here and has no
fixture in the reference: **parity unpinned**.  What follows restates the published scikit-image 0.19+
algorithms on integer label maps:
  * MSE  = mean((a - b)^2) over pixels, float64;
  * NMI  = (H(A) + H(B)) / H(A,B), natural log, from the joint histogram with 100 bins per axis - for integer
    labels spanning fewer than 100 values every label gets its own bin, i.e. the exact contingency table;
  * VI   = [H(B|A), H(A|B)] in bits from the contingency table normalised over the pixels whose FIRST-image
    label is not in ``ignore_labels`` (those pixels get weight 0); ``VI`` sums the two (sim_metrics.py:99),
    ``Seg_comp`` picks one (:120; index int(x_y)).
"""
import numpy as np


def label_maps(y_a, y_b):
    """sim_metrics.py:42-46: 4-D scores -> argmax label maps (softmax is monotone)."""
    y_a, y_b = np.asarray(y_a), np.asarray(y_b)
    if y_a.ndim == 4:
        y_a, y_b = y_a.argmax(axis=1).squeeze(0), y_b.argmax(axis=1).squeeze(0)
    return y_a.astype(np.int64), y_b.astype(np.int64)


def mse(a, b):
    a, b = label_maps(a, b)
    return float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))


def _entropy(p):
    p = np.asarray(p, dtype=np.float64).reshape(-1)
    p = p / p.sum()
    nz = p[p > 0]
    return float(-(nz * np.log(nz)).sum())


def nmi(a, b):
    a, b = label_maps(a, b)
    n = int(max(a.max(), b.max())) + 1
    table = np.zeros((n, n), dtype=np.float64)
    np.add.at(table, (a.reshape(-1), b.reshape(-1)), 1.0)
    h01 = _entropy(table)
    if h01 == 0.0:
        return float("nan")
    return (_entropy(table.sum(axis=1)) + _entropy(table.sum(axis=0))) / h01


def vi_pair(a, b, ignore_labels=()):
    """-> (H(B|A)... as scikit-image orders them: [h1g0.sum(), h0g1.sum()] = [H(A|B), H(B|A)])."""
    a, b = label_maps(a, b)
    n = int(max(a.max(), b.max())) + 1
    w = (~np.isin(a.reshape(-1), list(ignore_labels))).astype(np.float64)
    if w.sum() > 0:
        w /= np.count_nonzero(w)
    pxy = np.zeros((n, n), dtype=np.float64)
    np.add.at(pxy, (a.reshape(-1), b.reshape(-1)), w)
    px, py = pxy.sum(axis=1), pxy.sum(axis=0)

    def xlogx(x):
        y = x.copy()
        nz = y > 0
        y[nz] *= np.log2(y[nz])
        return y

    inv = lambda v: np.where(v > 0, 1.0 / np.where(v > 0, v, 1.0), 0.0)
    hygx = -(px * xlogx(pxy * inv(px)[:, None]).sum(axis=1)).sum()      # H(B|A)
    hxgy = -(xlogx(pxy * inv(py)[None, :]).sum(axis=0) * py).sum()      # H(A|B)
    return np.array([hygx, hxgy])           # variation_of_information returns [h1g0.sum(), h0g1.sum()]


def vi(a, b, ignore_labels=()):
    return float(np.sum(vi_pair(a, b, ignore_labels)))


def ssim(a, b, data_range, win=7, k1=0.01, k2=0.03):
    """sim_metrics.py:15-37 -> skimage.metrics.structural_similarity(im1, im2, data_range=dr) on integer label maps,
    restated from the published algorithm (Wang et al. 2004 as implemented by scikit-image: uniform 7x7 window,
    use_sample_covariance=True, float64, mean over the map cropped by (win-1)//2 per side).  **parity unpinned**
    (scikit-image absent, no fixture in the reference).  Window means via a summed-area table: in the cropped
    interior the 7x7 window never touches the border, so the filter's boundary mode does not matter."""
    a, b = label_maps(a, b)
    a, b = a.astype(np.float64), b.astype(np.float64)

    def box(x):
        s = np.zeros((x.shape[0] + 1, x.shape[1] + 1))
        s[1:, 1:] = x.cumsum(0).cumsum(1)
        return (s[win:, win:] - s[:-win, win:] - s[win:, :-win] + s[:-win, :-win]) / (win * win)

    ux, uy = box(a), box(b)
    cov = win * win / (win * win - 1.0)
    vx, vy, vxy = cov * (box(a * a) - ux * ux), cov * (box(b * b) - uy * uy), cov * (box(a * b) - ux * uy)
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    return float(s.mean())
