"""Seeding of the k-means fits behind the reference's CLIs (`--init`, `--seed`; SURVEY.md section 5 "config / flags").

The reference constructs KMeans(n_clusters=k) (KmeanGrids.py:300, color_kmeans.py:66): sklearn's k-means++ with an
unseeded RandomState, i.e. not reproducible (App. D.8).  The drop-ins accept
    --init k-means++    sklearn's own seeding, numpy-RandomState-compatible draws (cluster.kmeans_plusplus), --seed N
    --init seeded-rows  k distinct rows picked by numpy's default_rng(seed)   (the per-image CLIs' default)
    --init maximin      farthest-point seeding on the device inside the batched kernel (the grid CLIs' default; no seed)
For the documented k = 1 every choice gives the same centre."""
import numpy as np

from . import cluster

CHOICES = ("seeded-rows", "k-means++", "maximin")


def add_arguments(ap, default_init):
    ap.add_argument("--init", choices=CHOICES, default=default_init, help="k-means seeding (see seeding.py)")
    ap.add_argument("--seed", type=int, default=0, help="seed of --init k-means++ / seeded-rows")
    if not any(a.dest == "device" for a in ap._actions):
        ap.add_argument("--device", type=int, default=0, help="GPU ordinal")


def problem_init(rows, k, init, seed, device=0):
    """(k, d) float64 initial centres of ONE problem (rows: (n, d) uint8/float), or None for 'maximin'"""
    if init == "maximin":
        return None
    if init == "k-means++":
        return cluster.kmeans_plusplus(rows, k, random_state=seed, device=device)[0]
    if init == "seeded-rows":
        return cluster.seeded_rows_init(rows, k, seed)
    raise ValueError(f"init should be one of {CHOICES}, got {init!r}")


def batched_init(problems, k, init, seed, device=0):
    """(P, k, d) initial centres for a list of problems (every problem seeded as its own fit with the same seed, which is
    what running the per-image CLI once per file does), or None for 'maximin'"""
    if init == "maximin":
        return None
    return np.stack([problem_init(r, k, init, seed, device) for r in problems]).astype(np.float64)
