"""Builds tests/golden/kat_cells.npz from the reference's recorded data (run in the build
container only -- /root/reference does not exist on the GPU box).

Inputs (data files, not source):
  k-means-color-clustering/OutImgs/601_bad_bounce_3/<frame>/<cell>.png   51x51 RGB PNG cells
  k-means-color-clustering/OutCSV/601_bad_bounce_3.csv                   KAT-B: k=1 hue per cell
  k-means-color-clustering/601_bad_bounce_3.mp4_rgb_values.csv           KAT-A: mean-colour hue per cell
See SURVEY.md section 4 for what each pair pins.
"""
import csv
import os

import numpy as np
from PIL import Image

REF = "/root/reference/k-means-color-clustering"
FRAMES = [2, 3, 10, 19]          # folder names; CSV row index = frame - 2
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_cells.npz")


def main():
    cells = np.zeros((len(FRAMES), 350, 51, 51, 3), np.uint8)       # RGB as PIL decodes them
    for i, fr in enumerate(FRAMES):
        for c in range(350):
            im = Image.open(f"{REF}/OutImgs/601_bad_bounce_3/{fr}/{c + 1}.png").convert("RGB")
            a = np.asarray(im)
            assert a.shape == (51, 51, 3), a.shape
            cells[i, c] = a
    with open(f"{REF}/OutCSV/601_bad_bounce_3.csv") as f:
        rows_b = list(csv.reader(f))
    with open(f"{REF}/601_bad_bounce_3.mp4_rgb_values.csv") as f:
        rows_a = list(csv.reader(f))
    assert rows_b[0][0] == "cell_0" and rows_a[0][0] == "cell_0"
    kat_b = np.array([[int(v) for v in rows_b[1 + fr - 2]] for fr in FRAMES], np.int32)
    kat_a = np.array([[float(v) for v in rows_a[1 + fr - 2]] for fr in FRAMES], np.float64)
    np.savez_compressed(OUT, frames=np.array(FRAMES), cells_rgb=cells, hue_kmeans_k1=kat_b,
                        hue_mean=kat_a, csv_header=np.array(rows_b[0]))
    print(OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
