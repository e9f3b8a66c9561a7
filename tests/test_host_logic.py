"""CPU-side tests: the C-ABI library loads and exports every symbol include/ofc.h declares (no compute
calls), the host glue (sharding, frame I/O, CSV wire formats, CLI parsing) behaves like the reference's."""
import csv
import io
import os
import re
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ofc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ofc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from opticalflowclustering_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ofc.h but not exported by libofc.so"
    assert set(names) == set(_lib.EXPORTS), set(names) ^ set(_lib.EXPORTS)
    assert _lib.load().ofc_version() == 100


def test_no_gpu_means_loud_failure_not_fallback():
    from opticalflowclustering_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    from opticalflowclustering_amd.cluster import KMeans
    with pytest.raises(_lib.OfcError) as e:
        KMeans(n_clusters=2, init=np.zeros((2, 2))).fit(np.zeros((10, 2), np.float32))
    assert e.value.code == _lib.OFC_ENODEV


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "opticalflowclustering_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "libofc_oracle" not in src, f


def test_shard_pairs_contiguous_and_balanced():
    from opticalflowclustering_amd.pipeline import shard_pairs
    for n, w in [(299, 8), (299, 1), (7, 4), (16, 16), (3, 8)]:
        rng = [shard_pairs(n, w, r) for r in range(w)]
        assert rng[0][0] == 0 and rng[-1][1] == n
        assert all(rng[i][1] == rng[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in rng]
        assert max(sizes) - min(sizes) <= 1
    assert [b - a for a, b in [shard_pairs(299, 8, r) for r in range(8)]] == [38, 38, 38, 37, 37, 37, 37, 37]


def test_mjpeg_avi_writer_roundtrip(tmp_path):
    from PIL import Image
    from opticalflowclustering_amd.frameio import MjpegAviWriter
    p = str(tmp_path / "o.mp4")
    w = MjpegAviWriter(p, 25.0, (64, 48))
    frames = [np.full((48, 64, 3), v, np.uint8) for v in (10, 128, 250)]
    for f in frames:
        w.write(f)
    w.release()
    data = open(p, "rb").read()
    assert data[:4] == b"RIFF" and data[8:12] == b"AVI " and b"MJPG" in data[:300]
    assert struct.unpack("<I", data[4:8])[0] == len(data) - 8
    pos, got = data.find(b"movi") + 4, []
    while data[pos:pos + 4] == b"00dc":
        n = struct.unpack("<I", data[pos + 4:pos + 8])[0]
        got.append(np.asarray(Image.open(io.BytesIO(data[pos + 8:pos + 8 + n]))))
        pos += 8 + n + (n & 1)
    assert len(got) == 3
    for g, f in zip(got, frames):
        assert g.shape == (48, 64, 3) and abs(int(g.mean()) - int(f.mean())) <= 2


def test_frame_source_npy_and_dir(tmp_path):
    from PIL import Image
    from opticalflowclustering_amd.frameio import FrameSource, get_number
    stack = np.random.default_rng(0).integers(0, 256, (4, 20, 30, 3), dtype=np.uint8)
    np.save(tmp_path / "v.npy", stack)
    cap = FrameSource(str(tmp_path / "v.npy"))
    assert (cap.width, cap.height, cap.count) == (30, 20, 4)
    got = []
    while True:
        ret, f = cap.read()
        if not ret:
            break
        got.append(f)
    assert np.array_equal(np.stack(got), stack)
    d = tmp_path / "frames"
    d.mkdir()
    for i in (10, 2, 1):
        Image.fromarray(stack[0][..., ::-1] if i == 1 else stack[1][..., ::-1]).save(d / f"f{i}.png")
    cap = FrameSource(str(d))
    ret, f = cap.read()
    assert np.array_equal(f, stack[0])          # numeric order: f1 first, BGR restored
    assert get_number("cell_12.png") == 12 and get_number("abc") is None


def test_csv_row_format_matches_recorded_exemplar(tmp_path):
    """addnew.csv:1 of the reference: `50/348.png,[10. 10. 10. 10.],[[[ 0  0 10]]],0`"""
    p = tmp_path / "x.csv"
    with open(p, "w", newline="") as f:
        csv.writer(f).writerow(["50/348.png", np.array([10.0, 10.0, 10.0, 10.0]), np.array([[[0, 0, 10]]], np.uint8), np.uint8(0)])
    assert open(p).read().strip() == "50/348.png,[10. 10. 10. 10.],[[[ 0  0 10]]],0"


def test_cli_parsers_match_reference_flags():
    from opticalflowclustering_amd import KmeanGrids, color_kmeans, color_kmeansChange
    a = KmeanGrids.parse_arguments(["-d", "OutImgs/v", "-c", "1", "-f", "x.csv", "--noyolo", "--nocontour", "--path", "v.mp4"])
    assert a["dir"] == "OutImgs/v" and a["clusters"] == 1 and a["noyolo"] is False and a["nocontour"] is False
    # the reference's flags, plus the seeding / device extension (SURVEY.md section 5 "config / flags") with defaults that
    # leave the documented commands unchanged
    assert color_kmeans.parse_arguments(["-i", "a.png", "-c", "3", "-f", "o.csv"]) == {
        "image": "a.png", "clusters": 3, "csv": "o.csv", "init": "seeded-rows", "seed": 0, "device": 0}
    assert color_kmeansChange.parse_arguments(["-d", "D", "-c", "1", "-f", "o.csv"]) == {
        "dir": "D", "clusters": 1, "csv": "o.csv", "init": "maximin", "seed": 0, "device": 0}
    assert a["init"] == "maximin" and a["seed"] == 0 and a["device"] == 0
    b = KmeanGrids.parse_arguments(["-d", "OutImgs/v", "-c", "3", "-f", "x.csv", "--path", "v.mp4", "--init", "k-means++",
                                    "--seed", "7", "--device", "1"])
    assert (b["init"], b["seed"], b["device"], b["noyolo"]) == ("k-means++", 7, 1, True)
    with pytest.raises(SystemExit):
        KmeanGrids.parse_arguments(["-d", "x", "-c", "1", "-f", "x.csv", "--path", "v", "--init", "random"])
    with pytest.raises(SystemExit):
        color_kmeans.parse_arguments(["-i", "a.png"])


def test_draw_rectangle_like_cv2():
    from opticalflowclustering_amd.KmeanGrids import draw_rectangle, grid_geometry
    f = np.zeros((10, 12, 3), np.uint8)
    draw_rectangle(f, 2, 3, 6, 8)
    assert f[3, 2:7].min() == 255 and f[8, 2:7].min() == 255 and f[3:9, 2].min() == 255 and f[3:9, 6].min() == 255
    assert f[4:8, 3:6].max() == 0 and f[2].max() == 0 and f[9].max() == 0
    draw_rectangle(f, 8, 8, 12, 10)              # partly outside: clipped, no exception
    assert grid_geometry((1080, 1920, 3)) == (76, 77) and grid_geometry((720, 1280, 3)) == (51, 51)


def test_seeded_rows_init_is_deterministic_and_distinct():
    from opticalflowclustering_amd.cluster import seeded_rows_init
    X = np.random.default_rng(1).integers(0, 4, (500, 4), dtype=np.uint8)
    a, b = seeded_rows_init(X, 5), seeded_rows_init(X, 5)
    assert np.array_equal(a, b) and len(np.unique(a, axis=0)) == 5


def test_draw_grids_overlay_formats(tmp_path, monkeypatch):
    """drawGridsAndOutputCSVChange.overlayGridAndComputeAvgColor: CSV wire format (float hue strings, header only for
    framNum <= 2 which also truncates), cell PNGs = views incl. the neighbours' grid lines, captions inside the frame.
    The device call is replaced by the oracle here (CPU suite)."""
    from oracle import oracle as O
    from opticalflowclustering_amd import drawGridsAndOutputCSVChange as D
    from opticalflowclustering_amd.frameio import imread_bgr
    monkeypatch.setattr(D, "grid_cell_means", lambda frame, rows, cols, device=0: O.grid_cell_means(frame, rows, cols))
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(4)
    frame = rng.integers(0, 200, (140, 250, 3), dtype=np.uint8)
    f0 = frame.copy()
    open("rgb_values.csv", "w").write("stale\n")
    _, hues = D.overlayGridAndComputeAvgColor(2, frame, D.GRID_PARAMS, "rgb_values.csv", "some/dir/clip.mp4")
    D.overlayGridAndComputeAvgColor(3, f0.copy(), D.GRID_PARAMS, "rgb_values.csv", "some/dir/clip.mp4")
    lines = open("rgb_values.csv").read().splitlines()
    assert len(lines) == 3 and lines[0].split(",")[:2] == ["cell_0", "cell_1"] and lines[1] == lines[2]
    _, oh = O.grid_cell_means(f0)
    assert lines[1].split(",") == [str(float(h)) for h in oh[:, 0]] and hues == [float(h) for h in oh[:, 0]]
    cell = imread_bgr("OutImgs/clip/2/27.png")                       # row 1, col 1
    assert np.array_equal(cell, O.extract_cell(f0, 26, 14, 25))
    assert len(os.listdir("OutImgs/clip/3")) == 350
    assert (frame[0, :250] == 255).all() and (frame[:140, 10] == 255).all()      # grid lines
    tw, th = D.get_text_size("(255, 255, 255)")
    canvas = np.zeros((20, 120, 3), np.uint8)
    D.put_text(canvas, "(255, 255, 255)", (3, 15))
    ys, xs = np.nonzero(canvas[..., 0])
    assert xs.min() >= 3 and xs.max() < 3 + tw and ys.max() <= 15 and ys.min() >= 15 - th + 1
    assert set("0123456789(), ") <= set(D._FONT)


def test_batch_schedule_and_auto_batch():
    """how a clip is cut into flow launch sequences (pipeline.batch_schedule, bench.auto_batch): covers every pair once,
    short batch first, 64-pair batches for the 300-frame clip, an even split for a 1/8 shard"""
    import bench
    from opticalflowclustering_amd.pipeline import batch_schedule
    assert batch_schedule(299, 64) == [43, 64, 64, 64, 64]
    assert batch_schedule(38, 19) == [19, 19] and batch_schedule(5, 32) == [5] and batch_schedule(64, 64) == [64]
    assert batch_schedule(10, [3, 7]) == [3, 7]
    with pytest.raises(ValueError):
        batch_schedule(10, [3, 6])
    assert bench.auto_batch(299) == 64 and bench.auto_batch(38) == 19 and bench.auto_batch(150) == 32 and bench.auto_batch(75) == 25
    for n in (1, 2, 37, 38, 75, 149, 150, 255, 256, 299, 999):
        sch = batch_schedule(n, bench.auto_batch(n))
        assert sum(sch) == n and max(sch) <= 64 and min(sch) >= 1
