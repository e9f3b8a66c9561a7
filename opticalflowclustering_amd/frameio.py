"""Frame sources and sinks for the drop-in scripts (the step either side of the hot path: the
reference uses cv2.VideoCapture / cv2.VideoWriter, computeOpticalFlow.py:18-33, KmeanGrids.py:156-171).

cv2 is used when it is importable (true drop-in on real videos).  Without it -- as in this image --
a source is a `.npy`/`.npz` stack (T,H,W,3 uint8 BGR) or a directory of images (PIL, numeric order),
and a sink writes a Motion-JPEG AVI stream (the reference asks for fourcc 'MJPG') through PIL's JPEG
encoder, or a .npy stack when the file name ends in .npy."""
import os
import re
import struct

import numpy as np


def _have_cv2():
    try:
        import cv2  # noqa: F401
        return True
    except Exception:
        return False


def get_number(filename):
    """KmeanGrids.py:341-347"""
    m = re.compile(r"(\d+)").search(filename)
    return int(m.group(1)) if m else None


class FrameSource:
    """cap = FrameSource(path); ret, frame = cap.read(); cap.fps, cap.width, cap.height, cap.count"""

    def __init__(self, path):
        self.path = path
        self._i = 0
        self._cap = None
        self.fps = 30.0
        if os.path.isdir(path):
            names = [n for n in os.listdir(path) if n.lower().endswith((".png", ".jpg", ".jpeg", ".bmp"))]
            self._files = [os.path.join(path, n) for n in sorted(names, key=lambda n: (get_number(n) is None, get_number(n), n))]
            self._frames = None
            self.count = len(self._files)
            h, w = self._load(0).shape[:2] if self.count else (0, 0)
        elif path.endswith((".npy", ".npz")):
            arr = np.load(path)
            if hasattr(arr, "files"):
                arr = arr[arr.files[0]]
            if arr.ndim == 3:
                arr = np.repeat(arr[..., None], 3, -1)
            self._frames = np.ascontiguousarray(arr, np.uint8)
            self.count = len(self._frames)
            h, w = self._frames.shape[1:3]
        elif _have_cv2():
            import cv2
            self._cap = cv2.VideoCapture(path)
            self.fps = self._cap.get(cv2.CAP_PROP_FPS) or 30.0
            w, h = int(self._cap.get(3)), int(self._cap.get(4))
            self.count = int(self._cap.get(cv2.CAP_PROP_FRAME_COUNT))
        else:
            raise RuntimeError(f"cannot open {path!r}: cv2 is not installed; give a .npy/.npz stack "
                               "(T,H,W,3 uint8 BGR) or a directory of images")
        self.width, self.height = int(w), int(h)

    def _load(self, i):
        from PIL import Image
        rgb = np.asarray(Image.open(self._files[i]).convert("RGB"))
        return np.ascontiguousarray(rgb[..., ::-1])            # BGR, as cv2.imread gives

    def isOpened(self):
        return self._cap.isOpened() if self._cap is not None else True

    def read(self):
        if self._cap is not None:
            return self._cap.read()
        if self._i >= self.count:
            return False, None
        f = self._frames[self._i].copy() if self._frames is not None else self._load(self._i)
        self._i += 1
        return True, f

    def release(self):
        if self._cap is not None:
            self._cap.release()


class MjpegAviWriter:
    """minimal RIFF/AVI muxer with one 'MJPG' video stream (frames JPEG-encoded by PIL)"""

    def __init__(self, path, fps, size, quality=90):
        self.path, self.fps, self.size, self.quality = path, float(fps or 30.0), (int(size[0]), int(size[1])), quality
        self._f = open(path, "wb")
        self._index = []
        self._f.write(b"\0" * self._header_len())
        self._movi_start = self._f.tell()
        self._f.write(b"LIST\0\0\0\0movi")

    @staticmethod
    def _header_len():
        return 12 + (8 + 4 + (8 + 56) + (8 + 4 + (8 + 56) + (8 + 40)))

    def write(self, frame_bgr):
        import io
        from PIL import Image
        buf = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(frame_bgr[..., ::-1])).save(buf, format="JPEG", quality=self.quality)
        data = buf.getvalue()
        off = self._f.tell() - self._movi_start - 8
        self._f.write(b"00dc" + struct.pack("<I", len(data)) + data + (b"\0" if len(data) & 1 else b""))
        self._index.append((off, len(data)))

    def release(self):
        if self._f is None:
            return
        f, n = self._f, len(self._index)
        movi_end = f.tell()
        f.write(b"idx1" + struct.pack("<I", 16 * n))
        for off, ln in self._index:
            f.write(b"00dc" + struct.pack("<III", 0x10, off, ln))
        end = f.tell()
        w, h = self.size
        usec = int(round(1e6 / self.fps))
        avih = struct.pack("<IIIIIIIIIIIIII", usec, 0, 0, 0x10, n, 0, 1, 0, w, h, 0, 0, 0, 0)
        strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIIIhhhh", 0, 0, 0, 0, 1000, int(round(self.fps * 1000)), 0, n,
                                                 0, 0xFFFFFFFF, 0, 0, 0, w, h)
        strf = struct.pack("<IiiHH4sIiiII", 40, w, h, 1, 24, b"MJPG", w * h * 3, 0, 0, 0, 0)
        strl = b"LIST" + struct.pack("<I", 4 + 8 + len(strh) + 8 + len(strf)) + b"strl" + \
            b"strh" + struct.pack("<I", len(strh)) + strh + b"strf" + struct.pack("<I", len(strf)) + strf
        hdrl = b"LIST" + struct.pack("<I", 4 + 8 + len(avih) + len(strl)) + b"hdrl" + \
            b"avih" + struct.pack("<I", len(avih)) + avih + strl
        head = b"RIFF" + struct.pack("<I", end - 8) + b"AVI " + hdrl
        assert len(head) == self._header_len(), (len(head), self._header_len())
        f.seek(0)
        f.write(head)
        f.seek(self._movi_start + 4)
        f.write(struct.pack("<I", movi_end - self._movi_start - 8))
        f.close()
        self._f = None


class NpyWriter:
    def __init__(self, path):
        self.path, self._frames = path, []

    def write(self, frame):
        self._frames.append(np.array(frame, np.uint8))

    def release(self):
        np.save(self.path, np.stack(self._frames) if self._frames else np.zeros((0,), np.uint8))


def open_writer(path, fps, size):
    """cv2.VideoWriter(path, VideoWriter_fourcc(*'MJPG'), fps, size) or the built-in MJPEG muxer"""
    if path.endswith(".npy"):
        return NpyWriter(path)
    if _have_cv2():
        import cv2
        return cv2.VideoWriter(path, cv2.VideoWriter_fourcc(*"MJPG"), fps, size)
    return MjpegAviWriter(path, fps, size)


def imread_bgr(path):
    """cv2.imread equivalent (BGR uint8)"""
    if _have_cv2():
        import cv2
        return cv2.imread(path)
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[..., ::-1])


def imwrite_bgr(path, img):
    """cv2.imwrite equivalent for 8-bit BGR (or single-channel) images; PNG is lossless either way"""
    if _have_cv2():
        import cv2
        return bool(cv2.imwrite(path, img))
    from PIL import Image
    img = np.ascontiguousarray(img, np.uint8)
    Image.fromarray(img[..., ::-1] if img.ndim == 3 else img).save(path)
    return True
