"""CPU restatement of the reference's mask post-processing (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Follows utils/postprocess.py:7-61 and predict.py:286-315.  OpenCV is absent in this image, so
the INTER_NEAREST index rule is restated from OpenCV's published resizeNN
(sx = min(floor(dx * (1/fx)), ws-1), fx = wd/ws evaluated in double): parity unpinned for that rule.
"""
import numpy as np

_COLOURS = {1: (0, 255, 0), 2: (255, 0, 0), 3: (0, 0, 255), 4: (255, 255, 255), 5: (255, 0, 255),
            6: (0, 255, 255), 7: (255, 255, 0)}


def onehot_to_image(masks, n_classes=4):
    """utils/postprocess.py:21-61."""
    if n_classes not in (4, 7, 8):
        raise NotImplementedError
    if masks.ndim == 2:
        masks = masks[None]
    rgb = np.zeros(masks.shape + (3,), dtype=np.uint8)
    for k in range(1, n_classes):
        rgb[masks == k] = _COLOURS[k]
    return rgb


def resize_nearest(img, out_size):
    """cv2.resize(img, (wd, hd), interpolation=cv2.INTER_NEAREST) on an (H,W[,C]) array."""
    wd, hd = out_size
    hs, ws = img.shape[:2]
    ifx, ify = 1.0 / (wd / ws), 1.0 / (hd / hs)
    sx = np.minimum(np.floor(np.arange(wd) * ifx).astype(np.int64), ws - 1)
    sy = np.minimum(np.floor(np.arange(hd) * ify).astype(np.int64), hs - 1)
    return img[sy][:, sx]


def resize_area_int(img, k, ky=None):
    """cv2.resize(img, (W // k, H // ky), interpolation=cv2.INTER_AREA) (ky = k unless given) on a uint8 (H,W[,C]) array whose size is an exact
    multiple of (ky, k) (utils/dataset.py:312-316: the video path's downscale).  Restated from OpenCV's published
    imgproc/resize.cpp: k = 2 is ResizeAreaFastVec's (a + b + c + d + 2) >> 2; any other integer factor is
    resizeAreaFast_: the k x k block summed in int, saturate_cast<uchar>(sum * scale) with scale = 1.f / (k * k) in
    float, i.e. the float product rounded half to even.  OpenCV is absent here: parity unpinned for this rule."""
    ky = k if ky is None else ky
    h, w = img.shape[0] // ky, img.shape[1] // k
    assert img.dtype == np.uint8 and img.shape[0] == h * ky and img.shape[1] == w * k
    blocks = img.reshape((h, ky, w, k) + img.shape[2:]).astype(np.int32).sum(axis=(1, 3))
    if k == 2 and ky == 2:
        return ((blocks + 2) >> 2).astype(np.uint8)
    scale = np.float32(1.0) / np.float32(k * ky)
    return np.clip(np.rint(blocks.astype(np.float32) * scale), 0, 255).astype(np.uint8)


def _area_tab(ssize, dsize):
    """computeResizeAreaTab of OpenCV's published imgproc/resize.cpp: [(di, si, alpha)] with scale = 1 / (dsize / ssize) in
    double, alpha rounded to float."""
    import math
    scale = 1.0 / (dsize / ssize)
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def resize_area(img, out_size):
    """cv2.resize(img, (wd, hd), interpolation=cv2.INTER_AREA) on a uint8 (H,W[,C]) array for a DOWNSCALE whose factors are not
    both integers (utils/dataset.py:312-316 with e.g. 1920x1080 frames and a 1024x576 target).  Restated from OpenCV's published
    generic path: ResizeArea_<uchar, float> - per source row of a destination row buf[dx] += S[sx] * alpha over the x table
    (float, in table order), sum[dx] = beta * buf[dx] for the first row and += for the others, saturate_cast<uchar> (round
    half to even) at the end.  OpenCV is absent here: parity unpinned for this rule."""
    wd, hd = out_size
    hs, ws = img.shape[:2]
    assert img.dtype == np.uint8 and wd <= ws and hd <= hs
    im = img.reshape(hs, ws, -1).astype(np.float32)
    xtab, ytab = _area_tab(ws, wd), _area_tab(hs, hd)
    out = np.zeros((hd, wd, im.shape[2]), np.float32)
    started = np.zeros(hd, bool)
    for dy, sy, beta in ytab:
        buf = np.zeros((wd, im.shape[2]), np.float32)
        for dx, sx, alpha in xtab:
            buf[dx] = buf[dx] + im[sy, sx] * alpha              # float32 product, float32 sum, in table order
        if not started[dy]:
            out[dy] = beta * buf
            started[dy] = True
        else:
            out[dy] = out[dy] + beta * buf
    res = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return res.reshape((hd, wd) + img.shape[2:])


def format_masks(ids, mask_type, n_classes, out_size):
    """predict.py:286-315 on a uint8 id mask batch (B,H,W)."""
    if mask_type == "rgb":
        m = onehot_to_image(ids, n_classes)
    elif mask_type == "bin":
        m = ((ids > 0) * 255).astype(np.uint8)
    elif mask_type == "gray":
        m = ids
    else:
        raise NotImplementedError
    return np.stack([resize_nearest(x, out_size) for x in m], axis=0)
