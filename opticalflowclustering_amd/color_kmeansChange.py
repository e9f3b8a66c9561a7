"""Drop-in for k-means-color-clustering/color_kmeansChange.py
(`python color_kmeansChange.py -d OutImgs/<video> -c k -f csv`): walks dir/<frame>/<cell>.png in numeric
order and appends one row per cell: `<frame>/<cell>.png,[c0 c1 c2 c3],[[[h s v]]],hue`
(reference lines: color_kmeansChange.py:14-172; this is what produced the recorded addnew.csv and, through an
earlier KmeanGrids revision, OutCSV/601_bad_bounce_3.csv).  All cells of a frame folder go to the GPU in ONE
batched launch (ofc_kmeans_fit_batched) instead of one sklearn fit per PNG."""
import argparse
import csv
import os

import numpy as np

from . import seeding
from .color_kmeans import bgr2hsv_pixel, read_image
from .frameio import get_number
from .vis import kmeans_fit_batched


def parse_arguments(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-d", "--dir", required=True, help="Path to the image")
    ap.add_argument("-c", "--clusters", required=True, type=int, help="# of clusters")
    ap.add_argument("-f", "--csv", required=True, type=str, help="# of clusters")
    seeding.add_arguments(ap, "maximin")
    return vars(ap.parse_args(argv))


def preprocess_rows(image_rgb):
    """preprocess_image (color_kmeansChange.py:36-53) as packed RGBA rows, numpy glue only for the packing;
    the threshold/alpha arithmetic itself is repeated on the device inside the batched kernel's loader when
    cells come from a frame -- here the cells come from disk, so it is applied with the same device routine"""
    from .color_kmeans import preprocess_image
    return preprocess_image(image_rgb.copy()).reshape(-1, 4)


def process_folder(folder, n_clusters, device=0, init="maximin", seed=0):
    """all PNG cells of one frame folder -> list of (name, rint(top centre), hsv 1x1x3)"""
    names = sorted([n for n in os.listdir(folder) if not n.startswith(".")], key=get_number)
    rows = [preprocess_rows(read_image(os.path.join(folder, n))) for n in names]
    offsets = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    init_arr = seeding.batched_init(rows, n_clusters, init, seed, device)        # KMeans(n_clusters=k) per cell (:66)
    centers, counts, _, _ = kmeans_fit_batched(np.concatenate(rows), offsets, n_clusters, init_arr, device=device)
    out = []
    for p, n in enumerate(names):
        dom = int(np.argmax(counts[p]))                     # stable: first maximum, as sorted(..., reverse=True)
        c0 = np.rint(centers[p, dom]) + 0.0
        out.append((n, c0, bgr2hsv_pixel(c0[:3], device)))
    return out


def main(argv=None):
    args = parse_arguments(argv)
    dirs = args["dir"]
    with open(args["csv"], "a", newline="") as file:
        writer = csv.writer(file)
        for contentFolder in sorted([d for d in os.listdir(dirs) if not d.startswith(".")], key=get_number):
            for name, c0, hsv0 in process_folder(os.path.join(dirs, contentFolder), args["clusters"], args["device"],
                                                 args["init"], args["seed"]):
                writer.writerow([contentFolder + "/" + name, c0, hsv0, hsv0[0][0][0]])
            print(contentFolder)


if __name__ == "__main__":
    main()
