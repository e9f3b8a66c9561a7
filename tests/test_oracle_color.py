"""Colour oracle (oracle/color_ref.c) against the reference's two recorded known-answer CSVs
(KAT-A mean-colour hue, KAT-B k=1 k-means hue) and the structural facts the
recorded flow-visualisation PNGs pin (S=255 -> min channel 0, truncation)."""
import os

import numpy as np

from oracle import oracle as O

K = np.load(os.path.join(os.path.dirname(__file__), "golden", "kat_cells.npz"))


def test_kat_b_kmeans_k1_hue():
    """cell PNG (RGB order, color_kmeansChange.py:33) -> preprocess -> KMeans(1) -> rint ->
    BGR2HSV on the RGB-ordered triple (:118-122) == OutCSV/601_bad_bounce_3.csv"""
    cells, want = K["cells_rgb"], K["hue_kmeans_k1"]
    bad = 0
    for i in range(cells.shape[0]):
        for c in range(350):
            X = O.preprocess_rgba(cells[i, c]).reshape(-1, 4)
            cen, _, _, n_iter = O.kmeans_fit(X, X[:1].astype(np.float64))
            assert n_iter == 2
            c0 = np.rint(cen[0])
            hue = O.bgr2hsv(np.array([[c0[:3]]], dtype=np.uint8))[0, 0, 0]
            bad += int(hue != want[i, c])
    assert bad == 0


def test_kat_a_mean_hue_interior_cells():
    """mean -> astype(uint8) -> BGR2HSV hue (drawGridsAndOutputCSVChange.py:86-101), BGR order,
    == 601_bad_bounce_3.mp4_rgb_values.csv on interior cells"""
    cells, want = K["cells_rgb"], K["hue_mean"]
    n = bad = 0
    for i in range(cells.shape[0]):
        for c in range(350):
            cy, cx = divmod(c, 25)
            if cy == 0 or cx == 0:
                continue
            m = np.mean(cells[i, c][..., ::-1], axis=(0, 1)).astype(np.uint8)
            bad += int(float(O.bgr2hsv(m.reshape(1, 1, 3))[0, 0, 0]) != want[i, c])
            n += 1
    assert n == 4 * 13 * 24 and bad == 0


def test_recorded_cells_are_representable_by_truncating_hsv2bgr():
    """every colour in the recorded cells is floor(255*x) of the sector formula for some (H,V)"""
    hv = np.stack(np.meshgrid(np.arange(181), np.arange(256), indexing="ij"), -1).reshape(-1, 2)
    hsv = np.stack([hv[:, 0], np.full(len(hv), 255), hv[:, 1]], 1).astype(np.uint8)
    table = {tuple(t) for t in O.hsv2bgr(hsv.reshape(1, -1, 3))[0][:, ::-1]}   # as RGB
    cells = K["cells_rgb"][:, :, 1:, 1:]          # drop the white grid lines
    seen = np.unique(cells.reshape(-1, 3), axis=0)
    missing = [tuple(t) for t in seen if tuple(t) not in table]
    assert not missing, missing[:5]


def test_hsv_roundtrip_and_gray():
    rng = np.random.default_rng(0)
    bgr = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    hsv = O.bgr2hsv(bgr)
    assert hsv[..., 0].max() < 180
    g = O.bgr2gray(bgr)
    ref = (bgr[..., 0] * 0.114 + bgr[..., 1] * 0.587 + bgr[..., 2] * 0.299)
    assert np.abs(g.astype(float) - ref).max() <= 1.0


def test_flow_to_bgr_structure():
    rng = np.random.default_rng(1)
    flow = rng.standard_normal((48, 64, 2)).astype(np.float32) * 3
    bgr, mm = O.flow_to_bgr(flow)
    nz = bgr.reshape(-1, 3)[bgr.reshape(-1, 3).max(1) > 0]
    assert (nz.min(1) == 0).all()                 # S == 255
    assert bgr.max() in (254, 255) and bgr.min() == 0
    mag = np.sqrt((flow.astype(np.float64) ** 2).sum(-1))
    assert abs(mm - mag.mean()) < 1e-5
