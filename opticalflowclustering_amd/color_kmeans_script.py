"""Drop-in for k-means-color-clustering/color_kmeans_script.sh (and .ps1): `<images_dir> <csv_file>` -- the reference's
batch driver, which starts `python3 color_kmeans.py -i <file> -c 1 -f <csv>` once per file of the directory
(color_kmeans_script.sh:16-19).  Same rows in the same (shell-glob = sorted) order, but the images go to the GPU in ONE
launch of the batched Lloyd kernel (ofc_kmeans_fit_batched) instead of one Python process and one sklearn fit per image.

Like the shell loop, one bad entry costs only its own row: a directory entry that is not a readable image (the
reference's image folders contain a `cropped/` sub-directory, which `"$IMAGES_DIR"/*` also globs) is reported on stderr
and skipped, and an image too large for the LDS-resident batched kernel (more than 24 576 pixels: the reference's
601_3_cropped_*_OF folders hold 270x232 ... 370x280 crops) takes the streaming fit color_kmeans.py itself uses.
`-c` other than the script's hard-wired 1 is accepted as an extension; every image is then seeded exactly as
`color_kmeans.py -c k` seeds it ('seeded-rows', seed 0, unless --init / --seed say otherwise)."""
import argparse
import csv
import os
import sys

import numpy as np

from . import seeding
from .color_kmeans import bgr2hsv_pixel, dominant_cluster, preprocess_image, read_image
from .vis import kmeans_fit_batched

BATCHED_MAX_POINTS = 24576          # lloyd_batched.hip: points + labels of one problem live in LDS


def run(images_dir, csv_file, n_clusters=1, device=0, init="seeded-rows", seed=0):
    if not images_dir:
        print("Error: Please provide the path to the image directory as the first argument.")      # .sh:4-8
        return 1
    if init == "maximin" and n_clusters > 1:
        print("note: --init maximin seeds on the device; rows may differ from color_kmeans.py -c k", file=sys.stderr)
    names = sorted(n for n in os.listdir(images_dir) if not n.startswith("."))                    # "$IMAGES_DIR"/*
    results = {}                       # name -> (rint'ed dominant centre, hsv 1x1x3)
    small = []                         # (name, rows) that fit the batched kernel
    for n in names:
        p = os.path.join(images_dir, n)
        try:
            if os.path.isdir(p):
                raise IsADirectoryError(p)
            rgba = preprocess_image(read_image(p).copy(), device)
        except Exception as e:         # the .sh loop: that one python3 process fails, the loop goes on
            print(f"color_kmeans_script: skipping {n!r}: {e.__class__.__name__}: {e}", file=sys.stderr)
            continue
        rows = rgba.reshape(-1, 4)
        if len(rows) > BATCHED_MAX_POINTS or len(rows) < n_clusters:
            try:
                c0, hsv0, _ = dominant_cluster(rgba, n_clusters, device,
                                               "seeded-rows" if init == "maximin" else init, seed)
            except Exception as e:
                print(f"color_kmeans_script: skipping {n!r}: {e.__class__.__name__}: {e}", file=sys.stderr)
                continue
            results[n] = (c0, hsv0)
        else:
            small.append((n, rows))
    if small:
        offsets = np.concatenate([[0], np.cumsum([len(r) for _, r in small])]).astype(np.int64)
        init_arr = seeding.batched_init([r for _, r in small], n_clusters, init, seed, device)
        centers, counts, _, _ = kmeans_fit_batched(np.concatenate([r for _, r in small]), offsets, n_clusters, init_arr,
                                                   device=device)
        for i, (n, _) in enumerate(small):
            dom = int(np.argmax(counts[i]))                    # stable: first maximum, as sorted(..., reverse=True)
            c0 = np.rint(centers[i, dom]) + 0.0
            results[n] = (c0, bgr2hsv_pixel(c0[:3], device))
    with open(csv_file, "a", newline="") as f:
        w = csv.writer(f)
        for n in names:
            if n not in results:
                continue
            if os.stat(csv_file).st_size == 0 and f.tell() == 0:                                   # color_kmeans.py:108-110
                w.writerow(["File name", "Cluster 1", "HSV Cluster 1", "Hue 0"])
            c0, hsv0 = results[n]
            w.writerow([n, c0, hsv0, hsv0[0][0][0]])
            f.flush()
    return 0


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("images_dir", nargs="?", default="")
    ap.add_argument("csv_file", nargs="?", default="cluster_centers.csv")
    ap.add_argument("-c", "--clusters", type=int, default=1)
    seeding.add_arguments(ap, "seeded-rows")
    a = ap.parse_args(argv)
    return run(a.images_dir, a.csv_file, a.clusters, a.device, a.init, a.seed)


if __name__ == "__main__":
    sys.exit(main())
