"""Deterministic synthetic inputs shared by the golden generator and the tests.

numpy's legacy RandomState (MT19937) stream is frozen across numpy versions, so the same
seeds give the same tiles here and on the GPU box (SURVEY.md section 8d 'Synthetic inputs').
"""
import numpy as np
import torch


def tiles(seed, b, c, h, w):
    """uint8-valued float32 tiles [B,C,H,W] in 0..255 (the loader's dtype/range, db/buffer.py:62)."""
    return torch.from_numpy(np.random.RandomState(seed).randint(0, 256, (b, c, h, w)).astype(np.float32))


def masks(seed, b, h, w, n_classes):
    """int64 class-index masks [B,H,W] (db/buffer.py:63)."""
    return torch.from_numpy(np.random.RandomState(seed).randint(0, n_classes, (b, h, w)).astype(np.int64))


def blob_masks(seed, b, h, w, n_classes, cell=16):
    """Piecewise-constant masks (cell x cell blobs) so that class statistics are non-uniform."""
    rs = np.random.RandomState(seed)
    gh, gw = (h + cell - 1) // cell, (w + cell - 1) // cell
    g = rs.randint(0, n_classes, (b, gh, gw))
    m = np.repeat(np.repeat(g, cell, 1), cell, 2)[:, :h, :w]
    return torch.from_numpy(np.ascontiguousarray(m).astype(np.int64))


def class_weights(n_classes, seed=7):
    """utils/profile.py:129-130: w = 1/ln(1.02 + p), normalised, for a fixed synthetic p."""
    p = np.random.RandomState(seed).dirichlet(np.ones(n_classes))
    w = 1.0 / np.log(1.02 + p)
    return (w / w.sum()).astype(np.float32)


def digest(t):
    """(sum, abs-sum, l2) of a tensor in float64 -- the per-tensor checksum stored in fixtures."""
    d = t.detach().double()
    return [float(d.sum()), float(d.abs().sum()), float(d.pow(2).sum().sqrt())]


def learnable_tiles(seed, b, hw, n_classes, cell=8, noise=12.0):
    """Synthetic tiles whose masks are LEARNABLE from the pixels: every cell x cell blob has a
    class, and the blob's colour is that class's palette colour plus Gaussian noise (uint8-valued float32)."""
    rs = np.random.RandomState(seed)
    palette = np.random.RandomState(99).randint(30, 226, (n_classes, 3)).astype(np.float32)
    g = (hw + cell - 1) // cell
    cls = rs.randint(0, n_classes, (b, g, g))
    mask = np.repeat(np.repeat(cls, cell, 1), cell, 2)[:, :hw, :hw]
    img = palette[mask].transpose(0, 3, 1, 2) + noise * rs.standard_normal((b, 3, hw, hw)).astype(np.float32)
    img = np.clip(np.rint(img), 0, 255).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(img)), torch.from_numpy(np.ascontiguousarray(mask).astype(np.int64))
