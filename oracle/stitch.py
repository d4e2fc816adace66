"""CPU restatement of the reference's sliding-window tiling and tile stitching (TEST INFRASTRUCTURE).

split_tiles   <- Extractor.__split utils/extract.py:279-310 (row-major unfold of a fitted image)
stitch_classes <- utils/tools.py:209-319 reconstruct(), up to and including the argmax (the palette lookup of
                  colourize :322-358 and the INTER_NEAREST resize are separate, below).

Quirks reproduced on purpose: inside horizontal overlaps the strip holds the AVERAGE OF
SOFTMAX PROBABILITIES of the two tiles while tile interiors keep RAW LOGITS; the vertical merge then applies softmax
to whatever the strip holds (so overlap corners are soft-maxed twice); the final class is the argmax of that mixture.
Only the two strides the reference uses are supported: stride == tile (training extraction) and stride == tile/2
(test.py:63)."""
import numpy as np


def _softmax0(a):
    e = np.exp(a - a.max(axis=0, keepdims=True))
    return (e / e.sum(axis=0, keepdims=True)).astype(np.float32)


def split_tiles(img_chw, tile, stride):
    """[C,H,W] -> [n_rows*n_cols, C, tile, tile] in row-major tile order; H, W multiples of stride."""
    c, h, w = img_chw.shape
    rows = (h - tile) // stride + 1
    cols = (w - tile) // stride + 1
    out = np.empty((rows * cols, c, tile, tile), img_chw.dtype)
    for i in range(rows):
        for j in range(cols):
            out[i * cols + j] = img_chw[:, i * stride:i * stride + tile, j * stride:j * stride + tile]
    return out, rows, cols


def stitch_scores(tiles, rows, cols, tile, stride):
    """[rows*cols, C, tile, tile] logits -> the reference's full-size score map [C, h, w] (before argmax)."""
    tiles = np.asarray(tiles, np.float32)
    n, c = tiles.shape[0], tiles.shape[1]
    assert n == rows * cols and stride in (tile, tile // 2) and tile % 2 == 0
    olap = tile - stride
    w = cols * stride + olap
    h = rows * stride + olap
    full = np.empty((c, h, w), np.float32)
    prev_bottom = None
    y = 0
    for i in range(rows):
        # --- horizontal pass: one strip of height `tile` -------------------------------------------------------------
        strip = np.empty((c, tile, w), np.float32)
        x = 0
        for j in range(cols):
            t = tiles[i * cols + j]
            left = t[:, :, :olap]                 # overlaps the previous tile
            body = t[:, :, olap:]                 # new columns; its last `olap` columns overlap the next tile
            if j == 0:
                strip[:, :, :olap] = left
                x = olap
            seg = body.copy()
            if j < cols - 1 and olap:
                nxt_left = tiles[i * cols + j + 1][:, :, :olap]
                seg[:, :, stride - olap:] = (_softmax0(body[:, :, stride - olap:]) + _softmax0(nxt_left)) / 2
            strip[:, :, x:x + stride] = seg
            x += stride
        # --- vertical pass ---------------------------------------------------------------------------------------------
        bottom = strip[:, tile - olap:, :].copy() if olap else None
        if i > 0 and olap:
            strip[:, :olap, :] = (_softmax0(strip[:, :olap, :]) + _softmax0(prev_bottom)) / 2
        keep = tile if i == rows - 1 else tile - olap
        full[:, y:y + keep, :] = strip[:, :keep, :]
        y += keep
        prev_bottom = bottom
    return full


def stitch_classes(tiles, rows, cols, tile, stride):
    return stitch_scores(tiles, rows, cols, tile, stride).argmax(axis=0).astype(np.uint8)


def colourize_resize(mask, palette_rgb, out_h, out_w):
    """Palette lookup (tools.py:322-358) + cv2.INTER_NEAREST resize (tools.py:315-317): src = floor(dst * src/dst)."""
    h, w = mask.shape
    ys = np.minimum((np.arange(out_h) * (h / out_h)).astype(np.int64), h - 1)
    xs = np.minimum((np.arange(out_w) * (w / out_w)).astype(np.int64), w - 1)
    return np.asarray(palette_rgb, np.uint8)[mask[ys][:, xs]]
