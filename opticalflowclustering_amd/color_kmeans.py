"""Drop-in for k-means-color-clustering/color_kmeans.py
(`python color_kmeans.py -i image -c k -f csv`): image -> thresholded RGBA rows -> KMeans(k) -> clusters
ranked by population (predict + bincount) -> rint(top centre) -> BGR2HSV -> one CSV row.
Reference lines: color_kmeans.py:14-145.  All pixel arithmetic runs on the MI355X (libofc).

Conscious deviations:
  * KMeans seeding is deterministic ('seeded-rows', seed 0) instead of unseeded k-means++ (D.8);
    irrelevant for the documented k=1.  `--init k-means++ --seed N` gives sklearn's seeding for RandomState(N).
  * a centre component that rounds to zero from below prints as `0.`, not `-0.` (`np.rint(c) + 0.0`, also in
    color_kmeansChange.py and color_kmeans_script.py).  The reference's recorded CSVs do hold `-0.` entries
    (cluster_centers_copy.csv): there the sign of zero is rounding noise of sklearn's centred float64 means (a centre of
    -1e-16), nothing downstream reads it, and numeric comparison of the column is unaffected.
  * the CSV header is written when the OUTPUT csv is empty; the reference stats a hard-coded
    'cluster_centers.csv' in the CWD and raises if it is absent (D.2)."""
import argparse
import csv
import os

import numpy as np

from ._lib import check, load, ptr
from . import seeding
from .cluster import KMeans
from .frameio import imread_bgr


def parse_arguments(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-i", "--image", required=True, help="Path to the image")
    ap.add_argument("-c", "--clusters", required=True, type=int, help="# of clusters")
    ap.add_argument("-f", "--csv", required=True, type=str, help="# of clusters")
    seeding.add_arguments(ap, "seeded-rows")
    return vars(ap.parse_args(argv))


def read_image(image_path):
    """color_kmeans.py:28-33: imread (BGR) then BGR2RGB -- the data handed on is RGB-ordered"""
    image = imread_bgr(image_path)
    if image is None:
        raise FileNotFoundError(image_path)
    return np.ascontiguousarray(image[..., ::-1])


def preprocess_image(image, device=0):
    """color_kmeans.py:35-52: image[image < 30] = 0 (in place, as the reference), alpha = 255 where the
    grey value is > 0, result HxWx4 with the colour channels in their incoming order"""
    img = np.ascontiguousarray(image, np.uint8)
    out = np.empty(img.shape[:2] + (4,), np.uint8)
    check(load().ofc_preprocess_rgba(device, ptr(img), img.shape[0] * img.shape[1], 30, ptr(out)))
    if isinstance(image, np.ndarray) and image.dtype == np.uint8 and image.flags.writeable:
        image[...] = out[..., :3]                       # the reference thresholds the caller's array in place
    return out


def bgr2hsv_pixel(triple, device=0):
    px = np.array([[triple]], dtype=np.uint8)
    out = np.empty_like(px)
    check(load().ofc_bgr2hsv(device, ptr(px), 1, ptr(out)))
    return out


def dominant_cluster(image_rgba, n_clusters, device=0, init="seeded-rows", random_state=0):
    """color_kmeans.py:65-121 up to the CSV: -> (rint(top centre) [4 floats], hsv 1x1x3 uint8)"""
    flattened_image = image_rgba.reshape(image_rgba.shape[0] * image_rgba.shape[1], 4)
    clt = KMeans(n_clusters=n_clusters, init=init, random_state=random_state, device=device)
    clt.fit(flattened_image)
    labels = clt.predict(flattened_image)                                  # :78
    label_counts = np.bincount(labels, minlength=n_clusters)               # :81
    label_percentages = label_counts.astype(float) / len(flattened_image)  # :88
    label_info = [(label_percentages[i], f"Cluster {i + 1}", centroid) for i, centroid in enumerate(clt.cluster_centers_)]
    label_info = sorted(label_info, key=lambda x: x[0], reverse=True)      # :96 (stable)
    cluster0 = np.rint(label_info[0][2]) + 0.0                             # :112 (+0.0: a centre of -1e-16 prints "0.", not "-0.")
    r0, g0, b0, a0 = cluster0
    hsv0 = bgr2hsv_pixel([r0, g0, b0], device)                             # :117-121 (BGR2HSV on whatever order came in)
    return cluster0, hsv0, clt


def cluster_colors(image, n_clusters, image_path, csv_file, device=0, init="seeded-rows", random_state=0):
    """color_kmeans.py:54-135"""
    if init == "maximin":
        raise ValueError("--init maximin is the batched (grid) kernel's device-side seeding; use seeded-rows or k-means++")
    cluster0, hsv0, _ = dominant_cluster(image, n_clusters, device, init, random_state)
    with open(csv_file, "a", newline="") as file:
        writer = csv.writer(file)
        if os.stat(csv_file).st_size == 0:
            writer.writerow(["File name", "Cluster 1", "HSV Cluster 1", "Hue 0"])
        writer.writerow([os.path.basename(image_path), cluster0, hsv0, hsv0[0][0][0]])
    return None


def main(argv=None):
    args = parse_arguments(argv)
    image = read_image(args["image"])
    print("\n\n\n Image Name", args["image"])
    processed_image = preprocess_image(image, args["device"])
    print("Dimensions", processed_image.ndim)
    cluster_colors(processed_image, args["clusters"], args["image"], args["csv"], args["device"], args["init"], args["seed"])


if __name__ == "__main__":
    main()
