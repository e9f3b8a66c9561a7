"""Drop-in for k-means-color-clustering/computeOpticalFlow.py (`python computeOpticalFlow.py -i video`):
per frame pair Farneback flow -> HSV-coded visualisation, written to <in>onlyOpticalflow.mp4 (MJPG), plus
the per-frame mean flow magnitude in <in>_opticalFlow.csv (pandas layout: index, Frame, Average Magnitude)
and the plot <in>_squares.png.  Reference lines: computeOpticalFlow.py:9-160."""
import argparse
import csv

import numpy as np

from .computeOpticalFlowModule import ComputeOpticalFLow
from .frameio import FrameSource, open_writer


def run(input_path, device=0, quiet=False):
    cap = FrameSource(input_path)                                           # :18
    number_of_videoFrames = cap.count                                       # :19
    output_onlyOpticalFlow = open_writer(input_path + "onlyOpticalflow.mp4", cap.fps, (cap.width, cap.height))  # :31-33
    ret, first_frame = cap.read()                                           # :39
    if not ret:
        raise RuntimeError(f"no frames in {input_path!r}")
    flow = ComputeOpticalFLow(first_frame, device=device)                   # :58 prev_gray
    x_values, y_values = [], []
    frameNum = 0
    while cap.isOpened():                                                   # :74
        ret, frame = cap.read()
        if not ret:
            break
        rgb = flow.compute(frame)                                           # :96-120
        mean_mag = flow.last_mean_magnitude                                 # :114-117 np.mean(magnitude)
        if not quiet:
            print("Average Magnitude of optical flow ", mean_mag)
        x_values.append(frameNum)
        y_values.append(mean_mag)
        output_onlyOpticalFlow.write(rgb)                                   # :129
        frameNum += 1
        if not quiet:
            print("Number of VideoFrames processed", frameNum, "/", number_of_videoFrames)
    output_onlyOpticalFlow.release()
    with open(input_path + "_opticalFlow.csv", "w", newline="") as f:       # :146-149 df.to_csv
        w = csv.writer(f)
        w.writerow(["", "Frame", "Average Magnitude"])
        for i, (x, y) in enumerate(zip(x_values, y_values)):
            w.writerow([i, x, str(np.float32(y))])      # float32 column, shortest round-trip repr as pandas
    try:                                                                    # :152-155
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        plt.figure()
        plt.plot(x_values, y_values, color="black")
        plt.savefig(input_path + "_squares.png")
        plt.close()
    except Exception as e:                                                  # plotting is cosmetic
        print("plot skipped:", e)
    cap.release()
    flow.close()
    return x_values, y_values


def main(argv=None):
    parser = argparse.ArgumentParser(prog="OpticalFlow", description="find optical flow of video")
    parser.add_argument("-i", "--input")
    parser.add_argument("--device", type=int, default=0)
    args = parser.parse_args(argv)
    run(args.input, device=args.device)


if __name__ == "__main__":
    main()
