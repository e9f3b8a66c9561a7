"""Drop-in for k-means-color-clustering/KmeanGrids.py, the "combined" pipeline
(`python KmeanGrids.py -d OutImgs/<name> -c 1 -f x.csv --noyolo --nocontour --path <video>`):
video -> per frame pair Farneback flow -> HSV-coded visualisation -> 14x25 grid -> per cell
preprocess_image + KMeans(k) -> hue of the dominant cluster -> one 350-column row per frame appended to
OutCSV/<name>.csv.  Reference lines: KmeanGrids.py:52-113 (grid), :149-239 (process_video), :269-339
(preprocess_image, cluster_colors), :350-401 (__main__).

Everything numerical runs on the MI355X: compute() keeps its visualisation resident and all 350 cells of
a frame are clustered by ONE launch of the LDS-resident batched Lloyd kernel (ofc_grid_kmeans_dev).

Conscious deviations: the per-frame 30 ms cv2.waitKey sleep, the YOLO/contour overlays
and the never-written VideoWriter are dropped (D.7); frames/cells are enumerated from what was processed
(frame numbers 2..N, cells 1..350) instead of from PNG listings of -d that an earlier tool must have left on
disk (D.4); k-means seeding is deterministic (D.8).  The quirks that change numbers ARE kept: cells carry
cv2.rectangle's white row 0 / column 0 (D.5), the first processed pair is "frame 2" (D.6), np.rint of the centre
and uint8 truncation of the mean (D.10)."""
import argparse
import ctypes as C
import os

import numpy as np

from . import seeding
from ._lib import check, load, ptr
from .computeOpticalFlowModule import ComputeOpticalFLow
from .frameio import FrameSource, get_number  # noqa: F401  (get_number re-exported like the reference)
from .vis import grid_cell_means

GRID_PARAMS = {"rows": 14, "cols": 25, "cell_width": 50, "cell_height": 50}     # KmeanGrids.py:177

image_dict = {}        # reference module global (KmeanGrids.py:13): "<frameNum>/<cell>" -> cell pixels


def grid_geometry(frame_shape, grid_params=GRID_PARAMS):
    height, width = frame_shape[:2]
    return int(width / grid_params["cols"]), int(height / grid_params["rows"])      # :58-59


def draw_rectangle(frame, x1, y1, x2, y2, value=255):
    """cv2.rectangle(frame, (x1,y1), (x2,y2), white, 1): both corner rows/columns inclusive, clipped"""
    H, W = frame.shape[:2]
    xa, xb, ya, yb = max(x1, 0), min(x2, W - 1), max(y1, 0), min(y2, H - 1)
    for yy in (y1, y2):
        if 0 <= yy < H:
            frame[yy, xa:xb + 1] = value
    for xx in (x1, x2):
        if 0 <= xx < W:
            frame[ya:yb + 1, xx] = value


def overlayGridAndComputeAvgColor(framNum, frame, grid_params, csv_file=None, inputVideoFile=None, device=0,
                                  store_cells=True):
    """KmeanGrids.py:52-113: per-cell mean colour -> uint8 -> HSV (returned), white 1-px grid lines drawn on
    `frame` in place, the cells (views of `frame`, lines included) stored in image_dict."""
    rows, cols = grid_params["rows"], grid_params["cols"]
    x_step, y_step = grid_geometry(frame.shape, grid_params)
    avg_bgr, avg_hsv = grid_cell_means(frame, rows, cols, device)          # means as the sequential loop sees them
    height, width = frame.shape[:2]
    for y in range(rows):
        for x in range(cols):
            x1, y1 = x * x_step, y * y_step
            x2, y2 = min(x1 + x_step, width), min(y1 + y_step, height)
            draw_rectangle(frame, x1, y1, x2, y2)                           # :108
            if store_cells:
                image_dict[f"{framNum}/{y * cols + x + 1}"] = frame[y1:y2, x1:x2]   # :113 (a view, as the reference)
    return avg_bgr, avg_hsv


def cell_problems(frame_bgr, grid_params=GRID_PARAMS, device=0):
    """the 350 k-means problems of one visualisation frame as the reference forms them (KmeanGrids.py:108-113, 269-286):
    cells cut from the frame AFTER cv2.rectangle drew the white grid lines, then preprocess_image -> (n, 4) uint8 rows.
    Only needed to seed the fits on the host (--init k-means++ / seeded-rows); the fits themselves read the
    device-resident frame."""
    from .color_kmeans import preprocess_image
    rows, cols = grid_params["rows"], grid_params["cols"]
    x_step, y_step = grid_geometry(frame_bgr.shape, grid_params)
    cells = np.empty((rows * cols, y_step, x_step, 3), np.uint8)
    for y in range(rows):
        for x in range(cols):
            x1, y1 = x * x_step, y * y_step
            cell = cells[y * cols + x]
            cell[...] = frame_bgr[y1:y1 + y_step, x1:x1 + x_step]
            cell[0, :] = 255          # what the stored cell holds (App. D.5): its own top / left grid line; the bottom /
            cell[:, 0] = 255          # right lines are drawn later and belong to the next cells (lloyd_batched.hip loader)
    rgba = preprocess_image(cells.reshape(rows * cols * y_step, x_step, 3), device)      # one launch for all cells
    return list(rgba.reshape(rows * cols, y_step * x_step, 4))


def cluster_frame_cells(compflow, n_clusters, grid_params=GRID_PARAMS, device=0, max_iter=300, tol=1e-4,
                        init="maximin", seed=0, frame_bgr=None):
    """KmeanGrids.py:382-392 for the frame compute() just produced, on the device-resident visualisation:
    -> (rint'ed dominant centres (350,4), hsv (350,3) uint8).  init other than 'maximin' seeds every cell's fit on the
    host from `frame_bgr` (the array compute() returned) the way KMeans(n_clusters=k) would for that seed (:300)."""
    rows, cols = grid_params["rows"], grid_params["cols"]
    nc = rows * cols
    centers = np.empty((nc, 4), np.float64)
    hsv = np.empty((nc, 3), np.uint8)
    init_arr = None
    if init != "maximin" and n_clusters > 1:
        if frame_bgr is None:
            raise ValueError("seeding on the host needs the frame compute() returned")
        init_arr = np.ascontiguousarray(seeding.batched_init(cell_problems(frame_bgr, grid_params, device), n_clusters,
                                                             init, seed, device))
    check(load().ofc_grid_kmeans_dev(device, C.c_void_p(compflow.vis_device_ptr()), compflow.width, compflow.height, 1,
                                     rows, cols, n_clusters, ptr(init_arr), max_iter, tol, 0, ptr(centers), ptr(hsv)))
    return centers, hsv


def process_video(inputVideoFile, n_clusters, out_csv, device=0, quiet=False, store_cells=False, init="maximin", seed=0):
    """KmeanGrids.py:149-239 + :376-401 fused into one streaming loop; returns the list of hue rows"""
    cap = FrameSource(inputVideoFile)
    ret, frame = cap.read()                                                 # :171
    if not ret:
        raise RuntimeError(f"no frames in {inputVideoFile!r}")
    compflow = ComputeOpticalFLow(frame, device=device)                     # :178
    frameNum, rows_out = 1, []
    os.makedirs(os.path.dirname(out_csv) or ".", exist_ok=True)
    header = [f"cell_{i}" for i in range(GRID_PARAMS["rows"] * GRID_PARAMS["cols"])]   # :394
    while cap.isOpened():                                                   # :180
        ret, frame_rgb = cap.read()
        if not ret:
            break
        frame_optical = compflow.compute(frame_rgb)                         # :187
        frameNum += 1                                                       # :189 (first row is "frame 2")
        _, hsv = cluster_frame_cells(compflow, n_clusters, device=device, init=init, seed=seed,
                                     frame_bgr=frame_optical)               # :382-392
        if store_cells:
            overlayGridAndComputeAvgColor(frameNum, frame_optical, GRID_PARAMS, device=device)   # :230-231
        hues = [int(h) for h in hsv[:, 0]]
        rows_out.append(hues)
        with open(out_csv, "a" if len(rows_out) > 1 else "w", newline="") as f:   # :396-399
            if len(rows_out) == 1:
                f.write(",".join(header) + "\n")
            f.write(",".join(str(h) for h in hues) + "\n")
        if not quiet:
            print(frameNum)
    cap.release()
    compflow.close()
    return rows_out


def parse_arguments(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-d", "--dir", required=True, help="Path to the image")
    ap.add_argument("-c", "--clusters", required=True, type=int, help="# of clusters")
    ap.add_argument("-f", "--csv", required=True, type=str, help="# of clusters")
    ap.add_argument("--noyolo", action="store_false", help="do not load yolo bounding boxes")
    ap.add_argument("--nocontour", action="store_false", help="do not use contour detection")
    ap.add_argument("--path", required=True, help="Path to the input video")
    seeding.add_arguments(ap, "maximin")
    return vars(ap.parse_args(argv))


def main(argv=None):
    args = parse_arguments(argv)
    dirs = args["dir"]
    parts = str(dirs).replace("\\", "/").rstrip("/").split("/")
    name = parts[1] if len(parts) > 1 else parts[0]                         # :379 dirs.split('/')[1]
    filepathcsv = os.path.join("OutCSV", name + ".csv")                     # :377-379
    # the reference's cluster_colors opens `-f <csv>` in append mode for every cell and writes nothing to it
    # (:320-330: the writerow calls are commented out), so a run leaves that file behind, created if missing, untouched
    # otherwise; the hues go to OutCSV/<name>.csv
    with open(args["csv"], "a", newline=""):
        pass
    process_video(args["path"], args["clusters"], filepathcsv, device=args["device"], init=args["init"], seed=args["seed"])


if __name__ == "__main__":
    main()
