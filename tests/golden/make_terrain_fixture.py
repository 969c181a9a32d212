#!/usr/bin/env python3
"""Terrain fixture for the N4 parity tests (runs ONLY where /root/reference exists; data-only output).

Takes ONE of the reference's terrain images (model/terrains/*.png -- data files, the assets terrain_random.py chooses from),
decodes it with the product's own PNG reader, cross-checks that reader against an independent inflate+unfilter written with
numpy here, and stores the grey levels as uint8 [512, 512] in tests/golden/terrain_png.npz together with the file name.
"""
import os
import struct
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from cassierl_amd import terrain as T  # noqa: E402

SRC_DIR = "/root/reference/model/terrains"


def independent_decode(path):
    raw = open(path, "rb").read()
    pos, idat = 8, b""
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        if typ == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", raw[pos + 8:pos + 18])
        if typ == b"IDAT":
            idat += raw[pos + 8:pos + 8 + n]
        pos += 12 + n
    ch = {0: 1, 2: 3, 6: 4}[ctype]
    data = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + w * ch)
    img = np.zeros((h, w * ch), dtype=np.int64)
    for r in range(h):
        f, line = int(data[r, 0]), data[r, 1:].astype(np.int64)
        up = img[r - 1] if r else np.zeros(w * ch, dtype=np.int64)
        for i in range(w * ch):
            a = img[r, i - ch] if i >= ch else 0
            b, c = up[i], (up[i - ch] if i >= ch else 0)
            p = a + b - c
            pred = [0, a, b, (a + b) // 2, min((a, b, c), key=lambda v: abs(p - v))][f] if f != 4 else \
                (a if abs(p - a) <= abs(p - b) and abs(p - a) <= abs(p - c) else (b if abs(p - b) <= abs(p - c) else c))
            img[r, i] = (line[i] + pred) & 255
    img = img.reshape(h, w, ch)[:, :, :3].astype(np.float64)
    return img.mean(axis=2) if ch >= 3 else img[:, :, 0]


if __name__ == "__main__":
    name = sorted(os.listdir(SRC_DIR))[0]
    g = T.read_png_gray(os.path.join(SRC_DIR, name))
    ref = independent_decode(os.path.join(SRC_DIR, name))
    assert g.shape == ref.shape == (512, 512) and np.array_equal(g, ref), "PNG readers disagree"
    assert np.array_equal(g, np.round(g)), "grey image (R = G = B) expected"
    np.savez_compressed(os.path.join(HERE, "terrain_png.npz"), gray=g.astype(np.uint8), name=np.array(name))
    print(name, g.shape, g.min(), g.max(), os.path.getsize(os.path.join(HERE, "terrain_png.npz")), "bytes")
