"""Drop-in for k-means-color-clustering/drawGridsAndOutputCSVChange.py -- stage 1 of the reference's documented
two-step pipeline (`python drawGridsAndOutputCSV.py --noyolo --nocontour --path video_lq.mp4`, then
`color_kmeans.py -d OutImgs/video_lq/ -c 1 -f add.csv`, drawGridsAndOutputCSVChange.py:261-262):

    video -> ComputeOpticalFLow.compute (HSV-coded Farneback flow) -> 14x25 grid:
        per cell  mean BGR -> uint8 -> BGR2HSV hue                      -> one row of rgb_values.csv per frame
                  the cell's pixels (grid lines included)               -> OutImgs/<video>/<frame>/<cell>.png
        white 1-px rectangles + "(b, g, r)" captions on the frame       -> <video>_output.mp4 (MJPG)

Reference lines: :49-143 overlayGridAndComputeAvgColor, :145-240 process_video, :243-259 __main__.

The flow, its visualisation and the 350 cell means run on the MI355X (libofc: ofc_flow_push_bgr,
ofc_grid_cell_means); this module is the host glue and the file formats either side.  What the quirks of the
reference do to NUMBERS is kept: the cell ROI is a view of the frame, so the mean and the PNG of a cell see the
white right/bottom edges its left/upper neighbours drew before it (row 0 / column 0 of the cell) but not its own
rectangle in the mean; the CSV holds the hue as a float string ("60.0"), header only when framNum <= 2 and the file
is then truncated; frame numbering starts at 2.  Conscious deviations: the 30 ms cv2.waitKey per frame and its key
handling are dropped; captions are drawn with a built-in 5x7 stroke font instead of FONT_HERSHEY_SIMPLEX + LINE_AA
(OpenCV's glyph tables are not available here) -- pixels of the caption differ, nothing else reads them; YOLO boxes
are 2-px rectangles, contours are filled through PIL when it is importable."""
import argparse
import os

import numpy as np

from .computeOpticalFlowModule import ComputeOpticalFLow
from .frameio import FrameSource, imwrite_bgr, open_writer
from .KmeanGrids import draw_rectangle
from .vis import grid_cell_means

GRID_PARAMS = {"rows": 14, "cols": 25, "cell_width": 50, "cell_height": 50}     # :173

# 5x7 glyphs of the caption alphabet "(123, 45, 6)": rows top->bottom, bit 4 = leftmost column
_FONT = {
    "0": (14, 17, 19, 21, 25, 17, 14), "1": (4, 12, 4, 4, 4, 4, 14), "2": (14, 17, 1, 2, 4, 8, 31),
    "3": (31, 2, 4, 2, 1, 17, 14), "4": (2, 6, 10, 18, 31, 2, 2), "5": (31, 16, 30, 1, 1, 17, 14),
    "6": (6, 8, 16, 30, 17, 17, 14), "7": (31, 1, 2, 4, 8, 8, 8), "8": (14, 17, 17, 14, 17, 17, 14),
    "9": (14, 17, 17, 15, 1, 2, 12), "(": (2, 4, 8, 8, 8, 4, 2), ")": (8, 4, 2, 2, 2, 4, 8),
    ",": (0, 0, 0, 0, 12, 4, 8), " ": (0, 0, 0, 0, 0, 0, 0), ".": (0, 0, 0, 0, 0, 12, 12),
    "-": (0, 0, 0, 31, 0, 0, 0),
}
_ADVANCE = 6


def get_text_size(text):
    """(width, height) of put_text's rendering -- the role of cv2.getTextSize at :125"""
    return (_ADVANCE * len(text) - 1 if text else 0, 7)


def put_text(frame, text, org, color=(255, 255, 255)):
    """cv2.putText's role at :129: `org` is the bottom-left corner of the text; clipped to the frame"""
    H, W = frame.shape[:2]
    x0, y0 = int(org[0]), int(org[1]) - 6
    for i, ch in enumerate(text):
        glyph = _FONT.get(ch, _FONT[" "])
        for r, bits in enumerate(glyph):
            y = y0 + r
            if not 0 <= y < H:
                continue
            for c in range(5):
                x = x0 + i * _ADVANCE + c
                if bits & (16 >> c) and 0 <= x < W:
                    frame[y, x] = color


def load_yolo_bounding_boxes(yolo_bounding_box_file):
    """:13-20: rows of 11 numbers, rounded to int"""
    data = np.round(np.loadtxt(yolo_bounding_box_file)).astype(np.int32)
    return data.reshape(-1, 11)


def draw_yolo_bounding_box(frame, selectedRow):
    """:23-28: columns 3..6 = x, y, w, h; white rectangle of thickness 2"""
    for row in selectedRow:
        x, y, w, h = (int(v) for v in row[3:7])
        for t in (0, 1):
            draw_rectangle(frame, x - t, y - t, x + w + t, y + h + t)


def load_contours(inputVideoFile, frameNum, frame):
    """:31-46: Contours/<video>/<video>_<frame>.txt, one polygon per line (first number skipped): white 2-px
    outline, then filled black"""
    path = "Contours/" + inputVideoFile + "/" + inputVideoFile + "_" + str(frameNum) + ".txt"
    if not os.path.isfile(path):
        return
    from PIL import Image, ImageDraw
    img = Image.fromarray(frame)
    draw = ImageDraw.Draw(img)
    with open(path) as f:
        for line in f:
            pts = np.array(line.split(), dtype=int)[1:]
            pts = pts[: len(pts) // 2 * 2].reshape(-1, 2)
            if len(pts) > 0:
                poly = [tuple(int(v) for v in p) for p in pts]
                draw.line(poly + poly[:1], fill=(255, 255, 255), width=2)
                draw.polygon(poly, fill=(0, 0, 0))
    frame[...] = np.asarray(img)


def overlayGridAndComputeAvgColor(framNum, frame, grid_params, csv_file, inputVideoFile, device=0, write_cells=True):
    """:49-143.  Draws on `frame` in place, writes the cell PNGs and appends the hue row; returns
    (avg_bgr (cells,3) u8, hue row as floats)."""
    height, width = frame.shape[:2]
    rows, cols = grid_params["rows"], grid_params["cols"]
    x_step, y_step = int(width / cols), int(height / rows)                  # :55-56
    avg_bgr, avg_hsv = grid_cell_means(frame, rows, cols, device)           # :86-92, as the sequential loop sees them
    tm = os.path.basename(inputVideoFile).split(".")[0]                     # :107
    pat = f"OutImgs/{tm}/{framNum}"
    if write_cells:
        os.makedirs(pat, exist_ok=True)
    cell_idx = 0
    for y in range(rows):
        for x in range(cols):
            x1, y1 = x * x_step, y * y_step
            x2, y2 = min(x1 + x_step, width), min(y1 + y_step, height)
            cell_idx += 1
            draw_rectangle(frame, x1, y1, x2, y2)                           # :106
            if write_cells:
                imwrite_bgr(f"{pat}/{cell_idx}.png", frame[y1:y2, x1:x2])   # :109 (after its own rectangle)
    for i in range(rows * cols):                                            # :116-129 captions
        x = (i % cols) * x_step
        y = (i // cols) * y_step + 10
        b, g, r = (int(v) for v in avg_bgr[i])
        text = f"({b:.0f}, {g:.0f}, {r:.0f})"
        tw, th = get_text_size(text)
        put_text(frame, text, (x + (x_step - tw) // 2, y + (y_step - th) // 2 + th))
    hues = [float(h) for h in avg_hsv[:, 0]]                                # :99 avg_hsv_colors (float array)
    header = ",".join(f"cell_{i}" for i in range(rows * cols))              # :136
    line = ",".join(str(h) for h in hues)                                   # :132 str(float) -> "60.0"
    if framNum <= 2:                                                        # :139-142
        with open(csv_file, "w", newline="") as f:
            f.write(header + "\n" + line + "\n")
    else:
        with open(csv_file, "a", newline="") as f:
            f.write(line + "\n")
    return avg_bgr, hues


def process_video(yolo_bounding_box_file, inputVideoFile, loadYoloBoxes=True, loadContours=True, device=0,
                  csv_file="rgb_values.csv", quiet=False, write_cells=True):
    """:145-240.  Returns the number of frames written to <inputVideoFile>_output.mp4."""
    data = load_yolo_bounding_boxes(yolo_bounding_box_file) if loadYoloBoxes else None     # :147-149
    cap = FrameSource(inputVideoFile)                                        # :152
    out = open_writer(inputVideoFile + "_output.mp4", cap.fps, (cap.width, cap.height))   # :155-161
    ret, frame = cap.read()                                                  # :166
    if not ret:
        raise RuntimeError(f"no frames in {inputVideoFile!r}")
    frameNum, written = 1, 0
    compflow = ComputeOpticalFLow(frame, device=device)                      # :174
    while cap.isOpened():                                                    # :176
        ret, frame_rgb = cap.read()
        if not ret:
            break
        frame = compflow.compute(frame_rgb)                                  # :180 (showRGB False: the flow frame)
        frameNum += 1                                                        # :185
        if not quiet:
            print("\n\n frameNum: ", frameNum)                               # :196
        if data is not None:                                                 # :198-205
            sel = data[data[:, 0] == frameNum]
            if np.any(sel):
                draw_yolo_bounding_box(frame, sel)
        if loadContours:                                                     # :207-209
            load_contours(inputVideoFile, frameNum, frame)
        overlayGridAndComputeAvgColor(frameNum, frame, GRID_PARAMS, csv_file, inputVideoFile, device,
                                      write_cells)                           # :226-227 (showOverlay True)
        out.write(frame)                                                     # :229-230
        written += 1
    cap.release()
    out.release()
    compflow.close()
    return written


def main(argv=None):
    ap = argparse.ArgumentParser(description="Example script with argparse")              # :246
    ap.add_argument("--noyolo", action="store_false", help="do not load yolo bounding boxes")
    ap.add_argument("--nocontour", action="store_false", help="do not use contour detection")
    ap.add_argument("--path", required=True, help="Path to the input video")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)
    print("noyolo flag is set" if args.noyolo else "noyolo flag is not set")               # :255-256 (as printed there)
    process_video("yolo_labels.txt", args.path, args.noyolo, args.nocontour, device=args.device)


if __name__ == "__main__":
    main()
