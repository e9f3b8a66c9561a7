"""Drop-in for k-means-color-clustering/findCosineDifferentVectors.py
(`python findCosineDifferentVectors.py small.csv large.csv`): the downstream consumer of the hue CSVs the hot path
writes -- sliding-window cosine similarity of the second CSV column (reference lines :5-61).  The window dot
products and norms run on the device (ofc_sliding_cosine); the argmax bookkeeping keeps the reference's rule
(the LAST offset that attains the maximum wins, because equality updates max_frame)."""
import csv
import sys

import numpy as np

from ._lib import check, load, ptr


def read_hue_column(path):
    """pd.read_csv(path, header=None).iloc[:, 1].values"""
    with open(path, encoding="utf-8-sig", newline="") as f:
        return np.array([float(r[1]) for r in csv.reader(f) if len(r) > 1], np.float64)


def sliding_cosine(small, large, device=0):
    small = np.ascontiguousarray(small, np.float64)
    large = np.ascontiguousarray(large, np.float64)
    if len(large) < len(small):
        return np.zeros(0)
    sims = np.empty(len(large) - len(small) + 1, np.float64)
    check(load().ofc_sliding_cosine(device, ptr(small), len(small), ptr(large), len(large), ptr(sims)))
    return sims


def find_max(small, large, device=0):
    sims = sliding_cosine(small, large, device)
    max_similarity, max_frame = -1, -1                       # :49-51
    for i, similarity in enumerate(sims):
        max_similarity = max(max_similarity, similarity)     # :57
        if similarity == max_similarity:                     # :60-61
            max_frame = i
    return max_similarity, max_frame


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    file1_hue, nobounce_hue = read_hue_column(argv[0]), read_hue_column(argv[1])
    print("Vector sizes are: ", len(file1_hue), len(nobounce_hue))
    max_similarity, max_frame = find_max(file1_hue, nobounce_hue)
    print("Maximum cosine similarity:", max_similarity)
    print("Minimum sum of squared differences:", 0)
    print("Max frame:", max_frame)
    return max_similarity, max_frame


if __name__ == "__main__":
    main()
