"""Drop-in for k-means-color-clustering/drawGridsAndOutputCSV.py.  The reference keeps two copies of this tool: this one
reads a SECOND, pre-rendered `<name>_optical<ext>` flow video and overlays a 10x10 grid (drawGridsAndOutputCSV.py:
147-148,168,173), `drawGridsAndOutputCSVChange.py` computes the flow itself on a 14x25 grid and also writes the cell
PNGs.  The documented command (`python drawGridsAndOutputCSV.py --noyolo --nocontour --path video_lq.mp4`,
drawGridsAndOutputCSVChange.py:261) only parses with the Change variant's arguments, so this module exposes that
pipeline under both names and keeps the older grid as `process_prerendered`."""
import numpy as np

from .drawGridsAndOutputCSVChange import (GRID_PARAMS, draw_yolo_bounding_box, get_text_size,  # noqa: F401
                                          load_contours, load_yolo_bounding_boxes, main,
                                          overlayGridAndComputeAvgColor, process_video, put_text)
from .frameio import FrameSource, open_writer

GRID_PARAMS_PRERENDERED = {"rows": 10, "cols": 10, "cell_width": 10, "cell_height": 100}    # drawGridsAndOutputCSV.py:168


def process_prerendered(inputVideoFile, inputVideoFileExtension, device=0, csv_file="rgb_values.csv", quiet=True):
    """drawGridsAndOutputCSV.py:139-229 with showRGB False: grid statistics of an already rendered flow video"""
    cap = FrameSource(inputVideoFile + inputVideoFileExtension)
    cap_flow = FrameSource(inputVideoFile + "_optical" + inputVideoFileExtension)
    out = open_writer(inputVideoFile + "_output.mp4", cap.fps, (cap.width, cap.height))
    cap.read()
    frameNum, written = 1, 0
    while cap.isOpened():
        ret, _ = cap.read()
        ret2, frame = cap_flow.read()
        if not ret or not ret2:
            break
        frameNum += 1
        frame = np.ascontiguousarray(frame)
        overlayGridAndComputeAvgColor(frameNum, frame, GRID_PARAMS_PRERENDERED, csv_file, inputVideoFile, device,
                                      write_cells=False)
        out.write(frame)
        written += 1
    cap.release()
    cap_flow.release()
    out.release()
    return written


if __name__ == "__main__":
    main()
