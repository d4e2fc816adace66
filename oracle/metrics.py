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


def confusion_matrix(y_true, y_pred, n_classes, force_coverage=True):
    """Raw counts cm[t, p] after the Evaluator coverage overwrite (utils/evaluate.py:171-174)."""
    yt = np.asarray(y_true).reshape(-1).astype(np.int64).copy()
    yp = np.asarray(y_pred).reshape(-1).astype(np.int64).copy()
    if force_coverage:
        idx = np.arange(n_classes)
        yt[idx] = idx
        yp[idx] = idx
    cm = np.zeros((n_classes, n_classes), np.int64)
    np.add.at(cm, (yt, yp), 1)
    return cm


def scores_from_confusion(cm):
    """The reference's evaluation scores (utils/metrics.py:64-88) from a confusion matrix of counts:
    weighted F1 (zero_division=0), weighted Jaccard, Matthews correlation, row-normalised confusion matrix."""
    cm = np.asarray(cm, np.float64)
    tp = np.diag(cm)
    support = cm.sum(1)
    predicted = cm.sum(0)
    n = cm.sum()
    present = (support + predicted) > 0
    w = np.where(present, support, 0.0)
    f1_den = support + predicted
    f1 = np.where(f1_den > 0, 2 * tp / np.maximum(f1_den, 1), 0.0)
    iou_den = support + predicted - tp
    iou = np.where(iou_den > 0, tp / np.maximum(iou_den, 1), 0.0)
    cov_ytyp = tp.sum() * n - (support * predicted).sum()
    cov_ypyp = n * n - (predicted * predicted).sum()
    cov_ytyt = n * n - (support * support).sum()
    mcc = 0.0 if cov_ypyp * cov_ytyt == 0 else cov_ytyp / np.sqrt(cov_ytyt * cov_ypyp)
    norm = np.divide(cm, support[:, None], out=np.zeros_like(cm), where=support[:, None] > 0)
    return {'f1': float((f1 * w).sum() / w.sum()), 'iou': float((iou * w).sum() / w.sum()), 'mcc': float(mcc), 'cmatrix': norm}
