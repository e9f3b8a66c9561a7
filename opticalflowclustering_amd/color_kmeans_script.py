"""Drop-in for k-means-color-clustering/color_kmeans_script.sh (and .ps1): `<images_dir> <csv_file>` -- the reference's
batch driver, which starts `python3 color_kmeans.py -i <file> -c 1 -f <csv>` once per file of the directory
(color_kmeans_script.sh:16-19).  Same rows in the same (shell-glob = sorted) order, but all images go to the GPU in
ONE launch of the batched Lloyd kernel (ofc_kmeans_fit_batched) instead of one Python process and one sklearn fit per
image.  `-c` other than the script's hard-wired 1 is accepted as an extension."""
import argparse
import csv
import os
import sys

import numpy as np

from .color_kmeans import bgr2hsv_pixel, preprocess_image, read_image
from .vis import kmeans_fit_batched


def run(images_dir, csv_file, n_clusters=1, device=0):
    if not images_dir:
        print("Error: Please provide the path to the image directory as the first argument.")      # .sh:4-8
        return 1
    names = sorted(n for n in os.listdir(images_dir) if not n.startswith("."))                    # "$IMAGES_DIR"/*
    paths = [os.path.join(images_dir, n) for n in names]
    rows = [preprocess_image(read_image(p).copy(), device).reshape(-1, 4) for p in paths]
    offsets = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    centers, counts, _, _ = kmeans_fit_batched(np.concatenate(rows), offsets, n_clusters, None, device=device)
    with open(csv_file, "a", newline="") as f:
        w = csv.writer(f)
        for i, p in enumerate(paths):
            if os.stat(csv_file).st_size == 0 and i == 0:                                          # color_kmeans.py:108-110
                w.writerow(["File name", "Cluster 1", "HSV Cluster 1", "Hue 0"])
            dom = int(np.argmax(counts[i]))                    # stable: first maximum, as sorted(..., reverse=True)
            c0 = np.rint(centers[i, dom])
            hsv0 = bgr2hsv_pixel(c0[:3], device)
            w.writerow([os.path.basename(p), c0, hsv0, hsv0[0][0][0]])
            f.flush()
    return 0


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("images_dir", nargs="?", default="")
    ap.add_argument("csv_file", nargs="?", default="cluster_centers.csv")
    ap.add_argument("-c", "--clusters", type=int, default=1)
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    return run(a.images_dir, a.csv_file, a.clusters, a.device)


if __name__ == "__main__":
    sys.exit(main())
