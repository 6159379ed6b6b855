"""Oracle: mIoU accumulator and the entropy gate metric (CPU, numpy).

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.
"""
import numpy as np


def compute_basics(y_pred, targets):
    """seg_metrics.py:13-28 as integer counts.

    y_pred [N,C,H,W] scores, targets [N,H,W] / [N,1,H,W] ints (label >= C is
    void).  Returns TP, FP, FN as [N,C] arrays.  Quirk B-9: a void pixel is an
    FP of whatever class was predicted there (its one-hot row is cut off).
    softmax is monotone so argmax(softmax(x)) == argmax(x) (first max wins).
    """
    y_pred = np.asarray(y_pred)
    N, C = y_pred.shape[:2]
    pred = y_pred.reshape(N, C, -1).argmax(axis=1)
    tgt = np.asarray(targets).reshape(N, -1).astype(np.int64)
    TP = np.zeros((N, C)); FP = np.zeros((N, C)); FN = np.zeros((N, C))
    for n in range(N):
        for c in range(C):
            p = pred[n] == c
            t = tgt[n] == c
            TP[n, c] = np.sum(p & t)
            FP[n, c] = np.sum(p & ~t)
            FN[n, c] = np.sum(~p & t)
    return TP, FP, FN


class mIoU:
    """compute_mIoU.py:7-36 - streaming [3,C] accumulator.  Quirk B-8: a class
    absent from prediction and target gives 0/0 = NaN and is NOT replaced."""

    def __init__(self, n_classes):
        self.C = n_classes
        self.acc = np.zeros((3, n_classes), dtype=np.float32)

    def __call__(self, y_pred, targets):
        TP, FP, FN = compute_basics(y_pred, targets)
        self.acc[0] += TP.sum(0).astype(np.float32)
        self.acc[1] += FP.sum(0).astype(np.float32)
        self.acc[2] += FN.sum(0).astype(np.float32)

    def compute(self):
        with np.errstate(invalid="ignore", divide="ignore"):
            ciou = self.acc[0] / self.acc.sum(0)
        return np.float32(ciou.sum() / self.C)


class img_mIoU:
    """compute_mIoU.py:38-63 - per-image mean IoU over the classes PRESENT IN THE TARGET, averaged over images.

    One image per call.  `unique(target)` includes the void label when void pixels exist: that "class" is never
    predicted, so it contributes IoU 0 and still counts in the denominator (reproduced).  union counts every pixel
    that is target-i or predicted-i, so a void pixel predicted i enlarges class i's union."""

    def __init__(self):
        self.acc = [0.0, 0]

    def __call__(self, y_pred, target):
        y_pred = np.asarray(y_pred)
        if y_pred.ndim == 4:
            y_pred = y_pred.argmax(axis=1).squeeze()
        target = np.asarray(target).squeeze()
        classes = np.unique(target.reshape(-1))
        s = 0.0
        for i in classes:
            gt, pr = target == i, y_pred == i
            s += np.float32(np.sum(gt & pr)) / np.float32(np.sum(gt | pr))
        self.acc[0] += float(np.float32(s) / np.float32(classes.shape[0]))
        self.acc[1] += 1

    def compute(self):
        return float("nan") if self.acc[1] <= 0 else self.acc[0] / self.acc[1]


def block_reduce(a, size, func):
    """skimage.measure.block_reduce semantics (documented behaviour; skimage is
    absent -> PARITY UNPINNED): pad with 0 up to a multiple of the block, then
    reduce each non-overlapping block."""
    s0, s1 = size
    H, W = a.shape
    Hp, Wp = -(-H // s0) * s0, -(-W // s1) * s1
    p = np.zeros((Hp, Wp), dtype=a.dtype)
    p[:H, :W] = a
    return func(p.reshape(Hp // s0, s0, Wp // s1, s1), axis=(1, 3))


def img_norm_entropy(probs, n_classes, pool_min=False, s=1):
    """eval_br_ent.py:19-36.  probs [C,H,W] softmax output (fp32).
    scipy.stats.entropy(p, base=C, axis=0) == -sum(p*log(p))/log(C) after
    normalising p to sum 1 (already true for a softmax)."""
    p = np.asarray(probs, dtype=np.float32)
    assert p.ndim == 3
    pn = p / p.sum(axis=0, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        plogp = np.where(pn > 0, pn * np.log(pn), 0.0).astype(np.float32)
    ent = (-plogp.sum(axis=0) / np.float32(np.log(n_classes))).astype(np.float32)
    if s != 1:
        return np.mean(block_reduce(ent, (s, s), np.min if pool_min else np.max))
    return np.mean(ent)


def softmax_np(z, axis=0):
    z = np.asarray(z, dtype=np.float32)
    z = z - z.max(axis=axis, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=axis, keepdims=True)
