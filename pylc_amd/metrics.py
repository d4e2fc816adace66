"""Evaluation scores on the GPU: one confusion-matrix kernel over the class-index masks, then the reference's scores
(utils/metrics.py:64-88 via sklearn: weighted F1, weighted Jaccard = the headline "mIoU", Matthews correlation,
row-normalised confusion matrix) as closed forms of the n_cls x n_cls count matrix -- instead of several sklearn passes
over ~1e7-pixel flattened host arrays (utils/evaluate.py:131-176)."""
import numpy as np
import torch

from . import lib as L
from .lib import lib, check, ptr, stream


def confusion_matrix(y_true, y_pred, n_classes, force_coverage=True):
    """int64 [n_classes, n_classes] counts (device) for uint8 / int64 masks of equal size."""
    L.init()
    yt, yp = y_true.contiguous().reshape(-1), y_pred.contiguous().reshape(-1)
    if yt.numel() != yp.numel():
        raise ValueError('mask sizes differ: %d vs %d' % (yt.numel(), yp.numel()))
    for t in (yt, yp):
        if t.dtype not in (torch.uint8, torch.int64):
            raise TypeError('masks must be uint8 or int64 class indices, got %s' % t.dtype)
    cm = torch.zeros(n_classes * n_classes, device=yt.device, dtype=torch.int64)
    check(lib.pylc_confusion_matrix(ptr(yt), yt.element_size(), ptr(yp), yp.element_size(), yt.numel(), n_classes,
                                    int(force_coverage), ptr(cm), stream()))
    return cm.view(n_classes, n_classes)


def scores(cm):
    """{'f1', 'iou', 'mcc', 'cmatrix'} from a count matrix (host float64 arithmetic on n_cls^2 numbers)."""
    cm = np.asarray(cm.cpu() if torch.is_tensor(cm) else cm, np.float64)
    tp, support, predicted, n = np.diag(cm), cm.sum(1), cm.sum(0), cm.sum()
    w = np.where((support + predicted) > 0, support, 0.0)
    f1 = np.where(support + predicted > 0, 2 * tp / np.maximum(support + predicted, 1), 0.0)           # zero_division=0
    iou = np.where(support + predicted - tp > 0, tp / np.maximum(support + predicted - tp, 1), 0.0)
    c_tp, c_pp, c_tt = tp.sum() * n - (support * predicted).sum(), n * n - (predicted ** 2).sum(), n * n - (support ** 2).sum()
    mcc = 0.0 if c_pp * c_tt == 0 else c_tp / np.sqrt(c_tt * c_pp)
    norm = np.divide(cm, support[:, None], out=np.zeros_like(cm), where=support[:, None] > 0)
    return {'f1': float((f1 * w).sum() / w.sum()), 'iou': float((iou * w).sum() / w.sum()), 'mcc': float(mcc), 'cmatrix': norm}


def evaluate(y_true, y_pred, n_classes):
    """Evaluator.evaluate() (utils/evaluate.py:131-148) minus the plotting / report printing."""
    return scores(confusion_matrix(y_true, y_pred, n_classes, force_coverage=True))
