"""Height-field terrain: host-side counterpart of rllab/envs/terrain_random.py.

The reference's script rewrites the MJCF: for NUM_TERRAINS randomly chosen files of model/terrains/*.png it adds
    <hfield name='terrainI' size='10 10 1 0.001' file='../terrains/X.png'/>            (terrain_random.py:60-64)
    <geom type='hfield' hfield='terrainI' pos='0 0 0' condim='3' conaffinity='7'/>       (terrain_random.py:67-70)
and MuJoCo's compiler turns each PNG into a height grid.  Here the same steps end in a float64 grid in metres that
`CassieVecEnv.set_heightfield` uploads to HBM (the model tables are compile-time constants, so the terrain is the one
run-time asset of the model).

MuJoCo semantics restated (documentation of <asset>/<hfield>; not observable without the binary):
  * size = (radius_x, radius_y, elevation_z, base_z): the grid spans [-rx, rx] x [-ry, ry]; elevation data are normalised to
    [0, 1] (min -> 0, max -> 1) and scaled by elevation_z; base_z only thickens the solid below z = 0;
  * a PNG is converted to grey levels (mean of R, G, B) and its rows are flipped so that the top image row is +y;
  * several hfield geoms at the same place act like their pointwise maximum for a body coming from above (`combine`).
No image library is needed: the PNG decoder below handles the non-interlaced 8-bit grey / RGB / RGBA files of the reference.
"""
import os
import struct
import zlib

import numpy as np

DEFAULT_SIZE = (10.0, 10.0, 1.0, 0.001)  # terrain_random.py:41


def read_png_gray(path):
    """Decode a non-interlaced 8-bit PNG (colour type 0, 2, 4 or 6) into a float64 [height, width] grey image (0..255)."""
    raw = open(path, "rb").read()
    if raw[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("%s is not a PNG file" % path)
    pos, idat, ihdr = 8, [], None
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = ihdr
    if depth != 8 or interlace != 0 or ctype not in (0, 2, 4, 6):
        raise ValueError("unsupported PNG layout (depth %d, colour type %d, interlace %d)" % (depth, ctype, interlace))
    ch = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    stride = w * ch
    data = zlib.decompress(b"".join(idat))
    out = np.zeros((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    p = 0
    for r in range(h):
        f = data[p]
        line = np.frombuffer(data, dtype=np.uint8, count=stride, offset=p + 1).astype(np.int32)
        p += 1 + stride
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        else:  # Sub, Average, Paeth depend on the pixel to the left: per channel, left to right
            cur = np.zeros(stride, dtype=np.int32)
            cl, pl, ll = cur.tolist(), prev.tolist(), line.tolist()
            for i in range(stride):
                a = cl[i - ch] if i >= ch else 0
                b = pl[i]
                c = pl[i - ch] if i >= ch else 0
                if f == 1:
                    pred = a
                elif f == 3:
                    pred = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cl[i] = (ll[i] + pred) & 255
            cur = np.array(cl, dtype=np.int32)
        out[r] = cur
        prev = cur
    img = out.reshape(h, w, ch).astype(np.float64)
    if ctype in (2, 6):
        return img[:, :, :3].mean(axis=2)
    return img[:, :, 0]


def hfield_from_gray(gray, size=DEFAULT_SIZE):
    """Grey image [rows, cols] -> heights in metres [nrow, ncol] with row 0 at y = -size_y (image rows flipped), normalised."""
    g = np.asarray(gray, dtype=np.float64)[::-1]
    lo, hi = g.min(), g.max()
    g = (g - lo) / (hi - lo) if hi > lo else np.zeros_like(g)
    return np.ascontiguousarray(g * size[2])


def hfield_from_png(path, size=DEFAULT_SIZE):
    return hfield_from_gray(read_png_gray(path), size)


def choose_terrains(num_terrains, terrain_dir, rng=None):
    """terrain_random.py:20-34: NUM_TERRAINS files drawn (with replacement) from the terrain folder."""
    assert isinstance(num_terrains, int) and num_terrains > 0, "NUM_TERRAINS must be positive."
    files = sorted(os.listdir(terrain_dir))
    assert len(files) >= num_terrains, "Not enough terrains."
    rng = rng or np.random.default_rng()
    return [os.path.join(terrain_dir, files[int(i)]) for i in rng.integers(0, len(files), num_terrains)]


def combine(fields):
    """Several hfield geoms at the same pose: what a body coming from above meets is their pointwise maximum."""
    return np.maximum.reduce([np.asarray(f, dtype=np.float64) for f in fields])


def ramp(nrow=64, ncol=257, size_x=10.0, slope=0.1, x0=0.5):
    """Synthetic terrain for tests: flat (z = 0) up to x0, then a ramp of the given slope along +x (constant along y)."""
    x = np.linspace(-size_x, size_x, ncol)
    z = np.where(x > x0, (x - x0) * slope, 0.0)
    return np.ascontiguousarray(np.tile(z, (nrow, 1)))


def height_at(heights_m, size_x, size_y, x, y):
    """Terrain height under (x, y): the triangle of the grid cell, as the collision stage evaluates it (tests, spawn placement)."""
    hm = np.asarray(heights_m)
    nr, nc = hm.shape
    dx, dy = 2.0 * size_x / (nc - 1), 2.0 * size_y / (nr - 1)
    gx, gy = (x + size_x) / dx, (y + size_y) / dy
    if not (0 <= gx <= nc - 1 and 0 <= gy <= nr - 1):
        return 0.0
    ci, ri = min(int(gx), nc - 2), min(int(gy), nr - 2)
    fx, fy = gx - ci, gy - ri
    z00, z10, z01, z11 = hm[ri, ci], hm[ri, ci + 1], hm[ri + 1, ci], hm[ri + 1, ci + 1]
    if fy <= fx:
        a, b = (z10 - z00) / dx, (z11 - z10) / dy
    else:
        a, b = (z11 - z01) / dx, (z01 - z00) / dy
    return float(z00 + a * (fx * dx) + b * (fy * dy))
