"""GPU parity of the colour / grid / batched-k-means kernels through the C ABI: bit-exact against the
oracle, and against the reference's two recorded known-answer CSVs (KAT-A, KAT-B)."""
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
K = np.load(os.path.join(os.path.dirname(__file__), "golden", "kat_cells.npz"))


@pytest.fixture(scope="module")
def vis():
    from opticalflowclustering_amd import vis
    return vis


def frame_from_cells(cells_rgb):
    """re-assemble a 1275x714 BGR frame from the 350 recorded 51x51 cells (row-major 14x25)"""
    f = np.zeros((14 * 51, 25 * 51, 3), np.uint8)
    for c in range(350):
        cy, cx = divmod(c, 25)
        f[cy * 51:(cy + 1) * 51, cx * 51:(cx + 1) * 51] = cells_rgb[c][..., ::-1]
    return f


@pytest.mark.parametrize("W,H", [(64, 48), (1281, 719), (1920, 1080)])
def test_bgr2gray(vis, W, H):
    rng = np.random.default_rng(W)
    bgr = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    assert np.array_equal(vis.bgr2gray(bgr), O.bgr2gray(bgr))


@pytest.mark.parametrize("W,H", [(64, 48), (481, 271), (1920, 1080)])
def test_flow_to_bgr_bit_exact(vis, W, H):
    rng = np.random.default_rng(H)
    flow = (rng.standard_normal((H, W, 2)) * rng.uniform(0.1, 6)).astype(np.float32)
    flow[0, 0] = 0
    flow[1, 1] = (-3, 0)
    flow[2, 2] = (0, -2)
    got, mm = vis.flow_to_bgr(flow)
    want, wm = O.flow_to_bgr(flow)
    assert np.array_equal(got, want)
    assert abs(mm - wm) <= 1e-6 * abs(wm)


def test_flow_to_bgr_constant_flow(vis):
    flow = np.full((32, 40, 2), 1.25, np.float32)      # max == min -> scale 0 -> V = 0 -> black
    got, _ = vis.flow_to_bgr(flow)
    assert np.array_equal(got, O.flow_to_bgr(flow)[0]) and got.max() == 0


def test_grid_cell_means_oracle_and_kat_a(vis):
    """KAT-A: mean colour -> u8 -> hue == 601_bad_bounce_3.mp4_rgb_values.csv on interior cells"""
    for i in range(K["cells_rgb"].shape[0]):
        frame = frame_from_cells(K["cells_rgb"][i])
        mean, hsv = vis.grid_cell_means(frame)
        om, oh = O.grid_cell_means(frame)
        assert np.array_equal(mean, om) and np.array_equal(hsv, oh)
        for c in range(350):
            cy, cx = divmod(c, 25)
            if cy >= 1 and cx >= 1:
                assert float(hsv[c, 0]) == K["hue_mean"][i, c]


def test_grid_cell_means_1080p(vis):
    rng = np.random.default_rng(4)
    frame = rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
    mean, hsv = vis.grid_cell_means(frame)
    om, oh = O.grid_cell_means(frame)
    assert np.array_equal(mean, om) and np.array_equal(hsv, oh)


def test_grid_kmeans_kat_b(vis):
    """KAT-B: recorded cells -> (RGB order quirk) -> preprocess -> KMeans(1) -> rint -> BGR2HSV hue
    == OutCSV/601_bad_bounce_3.csv, all 350 cells of every stored frame, in one launch per frame"""
    for i in range(K["cells_rgb"].shape[0]):
        frame = frame_from_cells(K["cells_rgb"][i])
        cen, hsv = vis.grid_kmeans(frame, k=1, channel_order=1)
        assert np.array_equal(hsv[:, 0].astype(np.int32), K["hue_kmeans_k1"][i])


def test_all_black_cell_gives_recorded_centre(vis):
    """addnew.csv:1 of the reference: an all-black 51x51 cell -> [10. 10. 10. 10.], hue 0"""
    frame = np.zeros((14 * 51, 25 * 51, 3), np.uint8)
    cen, hsv = vis.grid_kmeans(frame, k=1)
    assert np.array_equal(cen[37], [10.0, 10.0, 10.0, 10.0]) and tuple(hsv[37]) == (0, 0, 10)


@pytest.mark.parametrize("k", [1, 2, 3, 5, 8, 12])
def test_batched_matches_oracle(vis, k):
    rng = np.random.default_rng(k)
    sizes = [2601, 5852, 300, k, 23562, 977]
    Xs, inits = [], []
    for n in sizes:
        X = np.zeros((n, 4), np.uint8)
        m = rng.random(n) < 0.35
        X[m, :3] = rng.integers(30, 256, (m.sum(), 3))
        X[m, rng.integers(0, 3, m.sum())] = 0
        X[m, 3] = 255
        X[0] = 255
        Xs.append(X)
        uniq = np.unique(X, axis=0)
        idx = rng.choice(len(uniq), k, replace=len(uniq) < k)
        inits.append(uniq[idx].astype(np.float64) + np.arange(k)[:, None] * 1e-3)
    offsets = np.concatenate([[0], np.cumsum(sizes)])
    cen, counts, labels, n_iter = vis.kmeans_fit_batched(np.concatenate(Xs), offsets, k, np.stack(inits))
    for p, X in enumerate(Xs):
        oc, ol, _, oi = O.kmeans_fit(X, inits[p])
        assert n_iter[p] == oi, (p, n_iter[p], oi)
        assert np.array_equal(labels[offsets[p]:offsets[p + 1]], ol)
        assert np.abs(cen[p] - oc).max() <= 1e-9
        assert np.array_equal(counts[p], np.bincount(O.kmeans_predict(X, oc), minlength=k))


def test_batched_relocation_and_ties(vis):
    X = np.zeros((400, 4), np.uint8)
    X[:100] = (200, 0, 0, 255)
    X[100:150] = (0, 180, 0, 255)
    init = np.array([[[0, 0, 0, 0], [250, 250, 250, 250], [1, 1, 1, 1.0]]])      # two start empty
    cen, counts, labels, n_iter = vis.kmeans_fit_batched(X, [0, 400], 3, init)
    oc, ol, _, oi = O.kmeans_fit(X, init[0])
    assert n_iter[0] == oi and np.array_equal(labels, ol) and np.abs(cen[0] - oc).max() <= 1e-9


def maximin_init(X, k):
    seeds = [X[0].astype(np.int64)]
    for _ in range(1, k):
        d = np.min([((X.astype(np.int64) - s) ** 2).sum(1) for s in seeds], 0)
        seeds.append(X[int(np.argmax(d))].astype(np.int64))
    return np.array(seeds, np.float64)


def test_grid_kmeans_k3_device_seeding_matches_oracle(vis):
    """k>1 without an explicit init: deterministic maximin seeding on the device (documented deviation
    from the reference's unseeded k-means++); the fit from that seeding must equal the oracle's"""
    frame = frame_from_cells(K["cells_rgb"][1])
    cen, hsv = vis.grid_kmeans(frame, k=3)
    for c in (0, 26, 137, 200, 349):
        cell = O.extract_cell(frame, c)
        X = O.preprocess_rgba(cell).reshape(-1, 4)
        oc, _, _, _ = O.kmeans_fit(X, maximin_init(X, 3))
        counts = np.bincount(O.kmeans_predict(X, oc), minlength=3)
        dom = np.rint(oc[int(np.argmax(counts))])
        assert np.array_equal(cen[c], dom)
        assert np.array_equal(hsv[c], O.bgr2hsv(dom[:3].astype(np.uint8).reshape(1, 1, 3))[0, 0])


def test_batched_errors(vis):
    with pytest.raises(ValueError):
        vis.kmeans_fit_batched(np.zeros((5, 4), np.uint8), [0, 2, 5], 3, np.zeros((2, 3, 4)))   # 2 samples < k
    from opticalflowclustering_amd._lib import OfcError
    with pytest.raises(OfcError):
        vis.kmeans_fit_batched(np.zeros((40000, 4), np.uint8), [0, 40000], 2, np.zeros((1, 2, 4)))  # > LDS
