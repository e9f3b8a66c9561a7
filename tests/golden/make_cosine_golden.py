"""Runs the reference's own findCosineDifferentVectors.py (numpy/pandas only, so it runs in the build container) on
pairs of its recorded hue CSVs and stores inputs + printed results in tests/golden/cosine_kat.json.
Only the vectors travel; the reference script is never copied."""
import csv
import json
import os
import re
import subprocess
import sys

REF = "/root/reference/k-means-color-clustering"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cosine_kat.json")
PAIRS = [("bounce.csv", "cropped_trimmed2.csv"), ("file1.csv", "file2.csv"), ("bounce.csv", "nobounce.csv"),
         ("file1.csv", "cropped_trimmed.csv")]


def column1(path):
    with open(path, encoding="utf-8-sig") as f:
        return [float(r[1]) for r in csv.reader(f) if len(r) > 1]


def main():
    cases = []
    for a, b in PAIRS:
        if not (os.path.exists(f"{REF}/{a}") and os.path.exists(f"{REF}/{b}")):
            continue
        va, vb = column1(f"{REF}/{a}"), column1(f"{REF}/{b}")
        if len(va) > len(vb):
            continue
        p = subprocess.run([sys.executable, "findCosineDifferentVectors.py", a, b], cwd=REF, capture_output=True, text=True)
        if p.returncode != 0:
            print("skipped", a, b, p.stderr[-200:])
            continue
        out = p.stdout
        sim = float(re.search(r"Maximum cosine similarity: ([-0-9.eE+na]+)", out).group(1))
        frame = int(re.search(r"Max frame: (-?\d+)", out).group(1))
        cases.append({"small": a, "large": b, "small_hue": va, "large_hue": vb, "max_similarity": sim,
                      "max_frame": frame, "stdout": out})
        print(a, b, len(va), len(vb), sim, frame)
    json.dump(cases, open(OUT, "w"), indent=0)
    print(OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
