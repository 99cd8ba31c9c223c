"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of OpenCV's filled-circle scan conversion, the
rasteriser behind `cv2.circle(img, center, radius, color, thickness=-1)` as the reference calls it
(MToV/tools/dataloader_sample.py:167-169: radius 3 -- radius 6 in the commented-out line above it -- lineType LINE_8,
shift 0, so cv::circle dispatches to the static Circle() of modules/imgproc/src/drawing.cpp, OpenCV 4.x).

OpenCV is a third-party dependency that is NOT vendored in /root/reference and NOT installed in this image or on the
GPU box (`import cv2` fails), and the reference holds no test or fixture for its output: this restatement follows the
published algorithm of drawing.cpp Circle() -- the eight-way midpoint walk with its `inside` fast path and its clipped
path -- statement for statement, on a numpy bitmap.  **Parity unpinned against cv2 itself**: the golden bitmaps in
tests/golden/circle_*.txt were produced by THIS file (tests/golden/make_golden_circle.py) and checked by hand against
the walk for r = 3 (rows +-0..3 half-widths 3,2,2,0: 29 pixels) and r = 6 (6,5,5,5,4,3,0: 113 pixels).
The product's rasteriser is moditalker_amd/pipeline.py:_disc_rows / landmarks_to_images, written differently (a row
table); tests/test_pipeline_host.py compares the two on every position class, including discs cut by each border.
"""
import numpy as np


def _hline(img, y, x0, x1, color):
    img[y, x0:x1 + 1] = color            # ICV_HLINE: inclusive span


def circle_filled(img: np.ndarray, center, radius: int, color=255) -> np.ndarray:
    """drawing.cpp Circle(img, center, radius, color, fill=1).  img [H, W] or [H, W, C] uint8, modified in place."""
    h, w = img.shape[:2]
    cx, cy = int(center[0]), int(center[1])
    err, dx, dy, plus, minus = 0, radius, 0, 1, (radius << 1) - 1
    inside = cx >= radius and cx < w - radius and cy >= radius and cy < h - radius
    while dx >= dy:
        y11, y12, y21, y22 = cy - dy, cy + dy, cy - dx, cy + dx
        x11, x12, x21, x22 = cx - dx, cx + dx, cx - dy, cx + dy
        if inside:
            _hline(img, y11, x11, x12, color)
            _hline(img, y12, x11, x12, color)
            _hline(img, y21, x21, x22, color)
            _hline(img, y22, x21, x22, color)
        elif x11 < w and x12 >= 0 and y21 < h and y22 >= 0:
            x11 = max(x11, 0)
            x12 = min(x12, w - 1)
            if 0 <= y11 < h:
                _hline(img, y11, x11, x12, color)
            if 0 <= y12 < h:
                _hline(img, y12, x11, x12, color)
            if x21 < w and x22 >= 0:
                x21 = max(x21, 0)
                x22 = min(x22, w - 1)
                if 0 <= y21 < h:
                    _hline(img, y21, x21, x22, color)
                if 0 <= y22 < h:
                    _hline(img, y22, x21, x22, color)
        dy += 1
        err += plus
        plus += 2
        mask = (1 if err <= 0 else 0) - 1          # 0 or -1 (all ones)
        err -= minus & mask
        dx += mask
        minus -= mask & 2
    return img


def landmarks_to_images(lm: np.ndarray, WH: int = 256, flip: bool = False, radius: int = 3) -> np.ndarray:
    """dataloader_sample.py:153-179 `_change_np_img_size` with cv2.circle / cv2.flip(img, 0) replaced by the
    restatements: lm [T, 68, 2|3] -> uint8 [T, 256, 256, 3]."""
    lm = np.asarray(lm)
    if lm.shape[-1] == 3:
        lm2d = (lm * WH / 2 + WH / 2).astype(int)[:, :, :2]
    else:
        lm2d = lm.astype(int)
    img = np.zeros([lm.shape[0], 256, 256, 3], dtype=np.uint8)
    for b in range(len(lm2d)):
        for i in range(len(lm2d[b])):
            x, y = lm2d[b][i]
            circle_filled(img[b], (int(x / WH * 256.0), int(y / WH * 256.0)), radius, (255, 255, 255))
    if flip:
        img = np.stack([im[::-1] for im in img], axis=0)       # cv2.flip(img, 0): around the x axis
    return img
