"""Writes tests/golden/circle_r3.txt / circle_r6.txt / circle_clipped.txt from oracle/ref_circle.py (the restatement of
OpenCV's drawing.cpp Circle(); cv2 itself is not installed anywhere this repo runs -- see that file's header).
One bitmap per block: a title line, then rows of '.' / 'X'."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.ref_circle import circle_filled  # noqa: E402


def block(title, img):
    return "# " + title + "\n" + "\n".join("".join("X" if v else "." for v in row) for row in img) + "\n"


def main():
    for r in (3, 6):
        img = np.zeros((2 * r + 1, 2 * r + 1), np.uint8)
        circle_filled(img, (r, r), r)
        open(os.path.join(HERE, f"circle_r{r}.txt"), "w").write(block(f"radius {r}, centre ({r},{r}), {2*r+1}x{2*r+1}", img))
    out = []
    # radius-3 discs cut by every border / corner of a 10 x 8 (w x h) image, and one fully outside
    for cx, cy in [(0, 0), (9, 0), (0, 7), (9, 7), (1, 4), (8, 2), (4, 1), (5, 6), (-3, 4), (12, 3), (4, -3), (4, 10), (-4, 4), (2, 2), (3, 3)]:
        img = np.zeros((8, 10), np.uint8)
        circle_filled(img, (cx, cy), 3)
        out.append(block(f"radius 3, centre ({cx},{cy}), 10x8", img))
    open(os.path.join(HERE, "circle_clipped.txt"), "w").write("".join(out))


if __name__ == "__main__":
    main()
