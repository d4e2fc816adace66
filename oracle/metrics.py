"""CPU restatement of the reference's headline accuracy metric (TEST INFRASTRUCTURE).

"mIoU" in PyLC = sklearn.metrics.jaccard_score(y_true, y_pred, average='weighted') on flattened class-index arrays
(utils/metrics.py:69-72), after Evaluator.validate() overwrites the first n_classes pixels of BOTH arrays with
0..n_classes-1 to force every class to appear (utils/evaluate.py:171-174)."""
import numpy as np


def weighted_jaccard(y_true, y_pred, n_classes, force_coverage=True):
    yt = np.asarray(y_true).reshape(-1).astype(np.int64).copy()
    yp = np.asarray(y_pred).reshape(-1).astype(np.int64).copy()
    if force_coverage:
        idx = np.arange(n_classes)
        yt[idx] = idx          # evaluate.py:171-174
        yp[idx] = idx
    cm = np.zeros((n_classes, n_classes), np.int64)
    np.add.at(cm, (yt, yp), 1)
    tp = np.diag(cm).astype(np.float64)
    support = cm.sum(1).astype(np.float64)
    denom = support + cm.sum(0) - tp
    iou = np.where(denom > 0, tp / np.maximum(denom, 1), 0.0)
    present = (support + cm.sum(0)) > 0            # sklearn: labels = union of labels in y_true and y_pred
    w = np.where(present, support, 0.0)
    return float((iou * w).sum() / w.sum())
