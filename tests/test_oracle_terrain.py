"""N4 height-field terrain on the CPU side: the PNG reader against the committed fixture, MuJoCo's hfield conventions as
restated in cassierl_amd/terrain.py, and the oracle's sphere-vs-terrain test (oracle/cassie_oracle.c: hfield_sphere) against
hand arithmetic on a ramp (rllab/envs/terrain_random.py:38-76 is what puts such a field under the robot)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from cassierl_amd import terrain as T

EQ, LIM, CON = 0, 1, 2


@pytest.fixture(scope="module")
def png():
    d = np.load(os.path.join(GOLDEN, "terrain_png.npz"))
    return d["gray"].astype(np.float64), str(d["name"])


def test_png_reader_against_the_fixture(png, tmp_path):
    gray, name = png
    ref = os.path.join("/root/reference/model/terrains", name)
    if os.path.exists(ref):  # build container only; the fixture itself was cross-checked against a second decoder when it was made
        assert np.array_equal(T.read_png_gray(ref), gray)
    # a PNG written here with every filter type exercises Sub / Up / Average / Paeth without the reference tree
    import struct
    import zlib
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (9, 7, 3), dtype=np.uint8)
    rows = b""
    prev = np.zeros(21, dtype=np.int64)
    for r in range(9):
        f, cur = r % 5, img[r].reshape(-1).astype(np.int64)
        out = np.zeros(21, dtype=np.int64)
        for i in range(21):
            a = cur[i - 3] if i >= 3 else 0
            b, c = prev[i], (prev[i - 3] if i >= 3 else 0)
            p = a + b - c
            pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
            pred = [0, a, b, (a + b) // 2, a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)][f]
            out[i] = (cur[i] - pred) & 255
        rows += bytes([f]) + bytes(out.astype(np.uint8).tolist())
        prev = cur

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b))

    path = tmp_path / "t.png"
    path.write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 7, 9, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(rows)) + chunk(b"IEND", b""))
    assert np.allclose(T.read_png_gray(str(path)), img.astype(np.float64).mean(axis=2))


def test_hfield_conventions(png):
    gray, _ = png
    h = T.hfield_from_gray(gray, (10, 10, 1.0, 0.001))
    assert h.shape == (512, 512) and h.min() == 0.0 and h.max() == 1.0
    assert h[0, 0] == (gray[-1, 0] - gray.min()) / (gray.max() - gray.min())  # image rows flipped: top row = +y
    h2 = T.hfield_from_gray(gray, (10, 10, 0.25, 0.001))
    assert np.allclose(h2, 0.25 * h)
    assert np.array_equal(T.combine([h, 0.5 + 0 * h]), np.maximum(h, 0.5))
    files = T.choose_terrains(3, os.path.dirname(__file__), np.random.default_rng(1))
    assert len(files) == 3 and all(os.path.exists(f) for f in files)
    # height_at: vertices are reproduced, the cell is split along the (c, r)-(c+1, r+1) diagonal
    dx = 20.0 / 511
    assert abs(T.height_at(h, 10, 10, -10 + 5 * dx, -10 + 7 * dx) - h[7, 5]) < 1e-12
    lower = T.height_at(h, 10, 10, -10 + 5.75 * dx, -10 + 7.25 * dx)  # fy <= fx: triangle (c,r), (c+1,r), (c+1,r+1)
    assert abs(lower - (h[7, 5] + 0.75 * (h[7, 6] - h[7, 5]) + 0.25 * (h[8, 6] - h[7, 6]))) < 1e-12
    upper = T.height_at(h, 10, 10, -10 + 5.25 * dx, -10 + 7.75 * dx)  # fy > fx: triangle (c,r), (c+1,r+1), (c,r+1)
    assert abs(upper - (h[7, 5] + 0.25 * (h[8, 6] - h[8, 5]) + 0.75 * (h[8, 5] - h[7, 5]))) < 1e-12
    assert T.height_at(h, 10, 10, 10.5, 0.0) == 0.0  # outside the field: the floor plane


def _place(o, dx=0.0, dz=0.0):
    q, v = o.state()
    q = q.copy()
    q[0] += dx
    q[1] += dz
    o.set_state_raw(q, np.zeros(13), np.zeros(13))
    o.forward()
    return q


def test_zero_field_is_the_flat_floor(oracle_mod):
    a, b = oracle_mod.Oracle(), oracle_mod.Oracle()
    b.set_hfield(np.zeros((16, 33)), 10.0, 10.0)
    rng = np.random.default_rng(3)
    seen = 0
    for t in range(200):
        if t % 10 == 0:
            u = rng.uniform(-1, 1, 6) * np.array([12.0, 12.0, 0.9] * 2)
        a.step_torque(u); b.step_torque(u)
        assert a.ncon == b.ncon
        seen = max(seen, a.ncon)
    (qa, va), (qb, vb) = a.state(), b.state()
    assert seen >= 4
    assert np.abs(qa - qb).max() < 1e-10 and np.abs(va - vb).max() < 1e-8  # same contacts, contact point rounded differently
    b.set_hfield(None, 0, 0)
    a.set_state_raw(qb, vb, b.warmstart())
    a.step_torque(u); b.step_torque(u)
    assert np.array_equal(a.state()[0], b.state()[0])  # back on the plane: bit-identical


def test_ramp_contact_frame_by_hand(oracle_mod):
    """Robot standing where the terrain is a ramp of slope 0.1 starting at x = 0.5: every foot sphere of radius r = 0.02 at
    (cx, cz) must report n = (-0.1, 0, 1)/sqrt(1.01), dist = (cz - 0.1 (cx - 0.5)) / sqrt(1.01) - r, and a contact point
    r + dist/2 behind the centre along n; tangents in the sagittal plane and along y."""
    hm = T.ramp(nrow=64, ncol=2001, size_x=10.0, slope=0.1, x0=0.5)  # dx = 1 cm: the kink at 0.5 falls on a grid line
    o = oracle_mod.Oracle()
    o.set_hfield(hm, 10.0, 10.0)
    q = _place(o, dx=1.0, dz=0.047)  # rear spheres near x = 0.986 (ground 0.0486), front ones near x = 1.144 (ground 0.0644): all four dip in
    c = o.contacts()
    assert o.ncon == 4
    nrm = np.array([-0.1, 0.0, 1.0]) / np.sqrt(1.01)
    for i in range(4):
        fr = c["frame"][i].reshape(3, 3)
        assert np.abs(fr[0] - nrm).max() < 1e-12
        # tangents: one along +-y, one in the sagittal plane, orthonormal
        assert abs(abs(fr[1] @ fr[2]) ) < 1e-12 and abs(np.linalg.norm(fr[1]) - 1) < 1e-12 and abs(np.linalg.norm(fr[2]) - 1) < 1e-12
        assert min(abs(abs(fr[1][1]) - 1), abs(abs(fr[2][1]) - 1)) < 1e-12
        # centre recovered from the contact point: pos = centre - n (r + dist / 2)
        centre = c["pos"][i] + nrm * (0.02 + 0.5 * c["dist"][i])
        expect = (centre[2] - 0.1 * (centre[0] - 0.5)) / np.sqrt(1.01) - 0.02
        assert abs(c["dist"][i] - expect) < 1e-12 and -0.03 < c["dist"][i] < 0
    # the constraint rows: normal row = n . J_point; its velocity for a pure +x slide of the base is n_x
    e = o.efc()
    rows = [i for i in range(len(e["type"])) if e["type"][i] == CON]
    assert len(rows) == 12
    for k in range(0, 12, 3):
        Jn = e["J"][rows[k]]
        assert abs(Jn[0] - nrm[0]) < 1e-12 and abs(Jn[1] - nrm[2]) < 1e-12  # base slides x, z
        assert abs(e["pos"][rows[k]] - c["dist"][k // 3]) < 1e-15
    # on the flat part (x < 0.5) the same robot stands as on the floor plane
    o2 = oracle_mod.Oracle()
    o2.set_hfield(hm, 10.0, 10.0)
    _place(o2, dx=-1.0)
    c2 = o2.contacts()
    assert o2.ncon == 4 and np.abs(c2["frame"][:, :3] - np.array([0, 0, 1.0])).max() == 0


def test_robot_settles_on_real_terrain(oracle_mod, png):
    """One of the reference's terrain images, elevation 0.2 m, shifted so that the spawn area is just under the feet: random
    torques for 2500 substeps; the robot ends up lying ON the terrain (no sphere deeper than a few mm), nothing diverges."""
    gray, _ = png
    hm = T.hfield_from_gray(gray, (10, 10, 0.2, 0.001))
    hm = hm - max(T.height_at(hm, 10, 10, x, y) for x in np.linspace(-0.2, 0.3, 26) for y in (-0.1305, 0.1305)) - 1e-4
    o = oracle_mod.Oracle()
    o.set_hfield(hm, 10.0, 10.0)
    rng = np.random.default_rng(8)
    seen, deepest = 0, 0.0
    for t in range(2500):
        if t % 10 == 0:
            u = rng.uniform(-1, 1, 6) * np.array([12.0, 12.0, 0.9] * 2)
        o.step_torque(u)
        if o.ncon:
            c = o.contacts()
            seen, deepest = max(seen, o.ncon), min(deepest, float(c["dist"].min()))
            assert np.abs(c["frame"][:, 1]).max() == 0.0  # normals stay in the sagittal plane
    q, v = o.state()
    assert np.isfinite(q).all() and np.isfinite(v).all()
    assert seen >= 3 and deepest > -0.03  # it lands ON the terrain: no sphere sinks more than 3 cm during the impacts
    assert q[1] < 0.8 + T.height_at(hm, 10, 10, q[0], 0.0)  # it fell


def _front_toe_over(oracle_mod, slope, past, height):
    """An oracle whose front toe spheres (r = 0.02) sit `past` metres beyond a kink of the terrain (flat up to x0, then `slope`)
    and `height` metres above the flat part.  Returns (oracle, x0 - centre_x offset is `past` by construction)."""
    flat = oracle_mod.Oracle()
    q0, _ = flat.state()
    flat.set_state_raw(q0 - np.array([0, 0.002] + [0] * 11), np.zeros(13), np.zeros(13))   # dip the feet 2 mm into the floor plane
    flat.forward()
    c = flat.contacts()
    assert flat.ncon == 4
    centres = c["pos"] + np.array([0, 0, 1.0]) * (0.02 + 0.5 * c["dist"])[:, None]
    front = centres[:, 0].max()
    cz = centres[np.argmax(centres[:, 0]), 2] + 0.002            # centre height of the front toe sphere in the nominal pose
    ncol = 4001                                                   # dx = 5 mm
    xg = np.linspace(-10.0, 10.0, ncol)
    x0 = xg[np.searchsorted(xg, front - past) - 1]               # a grid line just behind the sphere centre ...
    shift = (x0 + past) - front                                   # ... and the robot moved so that the centre is `past` beyond it
    hm = T.ramp(nrow=8, ncol=ncol, size_x=10.0, slope=slope, x0=x0)
    o = oracle_mod.Oracle()
    o.set_hfield(hm, 10.0, 10.0)
    q = q0.copy()
    q[0] += shift
    q[1] += height - cz
    o.set_state_raw(q, np.zeros(13), np.zeros(13))
    o.forward()
    return o


def test_ridge_and_valley_are_met_through_their_closest_feature(oracle_mod):
    """The terrain test is a CLOSEST-FEATURE test on the sagittal section (oracle hfield_sphere = kernel terrain_sphere, r04), pinned by
    hand arithmetic on the counter-example that r03 documented as wrong.  A convex ridge: flat up to x0, falling with slope -0.5
    beyond it.  A toe sphere (r = 0.02) whose centre sits 2 mm past the ridge and 20.5 mm above the flat part is nearest to the ridge
    EDGE at (x0, 0): distance sqrt(0.002^2 + 0.0205^2) - 0.02 = +0.597 mm -- NO contact (r03's extended downhill plane said
    (0.0205 + 0.5 * 0.002) / sqrt(1.25) - 0.02 = -0.770 mm: a contact 1.37 mm early with the face normal).  Lowered by 0.8 mm the sphere
    touches the edge: dist = sqrt(0.002^2 + 0.0197^2) - 0.02 = -0.199 mm along the edge-to-centre direction.  At a concave kink (slope
    +0.5) a sphere 2 mm BEFORE the kink touches the uphill face first: its distance to that face, (h - 0.5 * (-0.002)) / sqrt(1.25) - r,
    is smaller than the one to the flat part under its centre."""
    o = _front_toe_over(oracle_mod, -0.5, 0.002, 0.0205)
    assert o.ncon == 0 and np.hypot(0.002, 0.0205) - 0.02 > 5.9e-4
    o = _front_toe_over(oracle_mod, -0.5, 0.002, 0.0197)
    con = o.contacts()
    assert o.ncon == 4                                             # the rear spheres stand 0.3 mm into the flat part
    assert np.abs(con["frame"][np.argsort(con["pos"][:, 0])[:2], :3] - np.array([0, 0, 1.0])).max() == 0
    n_edge = np.array([0.002, 0.0, 0.0197]) / np.hypot(0.002, 0.0197)
    for i in np.argsort(con["pos"][:, 0])[2:]:                     # the two front toe spheres: on the ridge edge
        assert np.abs(con["frame"][i].reshape(3, 3)[0] - n_edge).max() < 5e-6      # the placement is good to ~3e-8 m of the 2 mm offset
        assert abs(con["dist"][i] - (np.hypot(0.002, 0.0197) - 0.02)) < 1e-7
    # concave kink: centre 2 mm before the kink, 20.5 mm above the flat part: clear of the flat part (+0.5 mm), but the uphill face
    # beyond the kink is (0.0205 + 0.5 * 0.002) / sqrt(1.25) - 0.02 = -0.770 mm away: contact with THAT face's normal
    o = _front_toe_over(oracle_mod, 0.5, -0.002, 0.0205)
    con = o.contacts()
    assert o.ncon == 2
    n_face = np.array([-0.5, 0.0, 1.0]) / np.sqrt(1.25)
    for i in range(2):
        assert np.abs(con["frame"][i].reshape(3, 3)[0] - n_face).max() < 1e-9
        assert abs(con["dist"][i] - ((0.0205 + 0.5 * 0.002) / np.sqrt(1.25) - 0.02)) < 1e-7
