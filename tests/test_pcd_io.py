"""Product-side PCD reader (lom_pcd_read = pcl::io::loadPCDFile<pcl::PointXYZ>, reference test/test.cpp:194):
host code, runs without a GPU.  Checked on files written here in the layouts PCL writes (ascii, binary,
packed and padded records) and -- in this container, where /root/reference exists -- on the reference's
shipped test/test_data/intersection00056.pcd against the committed arrays of tests/golden/ (which
tests/golden/make_fixtures.py produced with an independent Python parser)."""
import os

import numpy as np
import pytest

REF_PCD = "/root/reference/test/test_data/intersection00056.pcd"


def _header(fields, sizes, types, counts, n, data):
    return ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\n"
            f"FIELDS {' '.join(fields)}\nSIZE {' '.join(map(str, sizes))}\nTYPE {' '.join(types)}\n"
            f"COUNT {' '.join(map(str, counts))}\nWIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA {data}\n")


def test_binary_fixture_layout(lom, tmp_path):
    """FIELDS rgb _ x y z _ / SIZE 4 1 4 4 4 1 / COUNT 1 12 1 1 1 4: the layout of the reference's data file."""
    rng = np.random.default_rng(1)
    n = 1000
    xyz = rng.standard_normal((n, 3)).astype(np.float32) * 30
    xyz[7] = np.nan                                             # loadPCDFile keeps NaN points
    rec = np.zeros(n, np.dtype([("rgb", "<f4"), ("p0", "u1", 12), ("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("p1", "u1", 4)]))
    assert rec.dtype.itemsize == 32
    rec["x"], rec["y"], rec["z"] = xyz.T
    path = tmp_path / "a.pcd"
    with open(path, "wb") as f:
        f.write(_header(["rgb", "_", "x", "y", "z", "_"], [4, 1, 4, 4, 4, 1], list("FUFFFU"), [1, 12, 1, 1, 1, 4], n, "binary").encode())
        f.write(rec.tobytes())
        f.write(bytes(100))                                     # trailing padding, as in the shipped file
    got = lom.loadPCDFile(path)
    assert got.tobytes() == xyz.tobytes()
    info = lom.capi.PcdInfo()
    assert lom.capi.lib().lom_pcd_read(str(path).encode(), None, None, 0, info) == n
    assert (info.points, info.point_step, info.has_normals, info.data_kind) == (n, 32, 0, 1)
    part = np.empty((10, 3), np.float32)                        # cap smaller than the file
    assert lom.capi.lib().lom_pcd_read(str(path).encode(), part.ctypes.data, None, 10, None) == n
    assert part.tobytes() == xyz[:10].tobytes()


def test_ascii_with_normals_and_other_types(lom, tmp_path):
    rng = np.random.default_rng(2)
    n = 257
    xyz = rng.standard_normal((n, 3)).astype(np.float32)
    nrm = rng.standard_normal((n, 3)).astype(np.float32)
    path = tmp_path / "b.pcd"
    with open(path, "w") as f:
        f.write(_header(["x", "y", "z", "intensity", "normal_x", "normal_y", "normal_z", "ring"], [4] * 7 + [2],
                        list("FFFFFFFU"), [1] * 8, n, "ascii"))
        for i in range(n):
            vals = [repr(float(v)) for v in xyz[i]] + ["0.5"] + [repr(float(v)) for v in nrm[i]] + [str(i % 16)]
            f.write(" ".join(vals) + "\n")
    gx, gn = lom.loadPCDFile(path, with_normals=True)
    assert gx.tobytes() == xyz.tobytes() and gn.tobytes() == nrm.tobytes()
    # f64 coordinates in a binary file are narrowed to f32
    path2 = tmp_path / "c.pcd"
    with open(path2, "wb") as f:
        f.write(_header(["x", "y", "z"], [8, 8, 8], list("FFF"), [1, 1, 1], n, "binary").encode())
        f.write(xyz.astype(np.float64).tobytes())
    assert lom.loadPCDFile(path2).tobytes() == xyz.tobytes()


def test_errors(lom, tmp_path):
    with pytest.raises(lom.LomError):
        lom.loadPCDFile(tmp_path / "missing.pcd")
    bad = tmp_path / "d.pcd"
    bad.write_text(_header(["x", "y", "z"], [4, 4, 4], list("FFF"), [1, 1, 1], 5, "binary_compressed"))
    with pytest.raises(lom.LomError) as e:
        lom.loadPCDFile(bad)
    assert "binary_compressed" in str(e.value)
    short = tmp_path / "e.pcd"
    with open(short, "wb") as f:
        f.write(_header(["x", "y", "z"], [4, 4, 4], list("FFF"), [1, 1, 1], 5, "binary").encode())
        f.write(bytes(12 * 3))
    with pytest.raises(lom.LomError):
        lom.loadPCDFile(short)
    # sizes under the file's control: a record of half a terabyte must come back as an error, not as an exception
    # escaping the C ABI (which would abort the process)
    huge = tmp_path / "g.pcd"
    with open(huge, "wb") as f:
        f.write(_header(["x", "y", "z", "blob"], [4, 4, 4, 8], list("FFFU"), [1, 1, 1, 1 << 20], 3, "binary").encode())
        f.write(bytes(64))
    with pytest.raises(lom.LomError) as e:
        lom.loadPCDFile(huge)
    assert "record size" in str(e.value)
    longline = tmp_path / "h.pcd"
    with open(longline, "wb") as f:
        f.write(b"# .PCD v0.7\nFIELDS " + b"x" * 200000 + b"\n")
    with pytest.raises(lom.LomError) as e:
        lom.loadPCDFile(longline)
    assert "too long" in str(e.value)
    nox = tmp_path / "f.pcd"
    nox.write_text(_header(["a", "b"], [4, 4], list("FF"), [1, 1], 1, "ascii") + "1 2\n")
    with pytest.raises(lom.LomError):
        lom.loadPCDFile(nox)


@pytest.mark.skipif(not os.path.exists(REF_PCD), reason="the reference checkout is only present in the build container")
def test_reference_data_file_equals_the_committed_arrays(lom, fixture_cloud):
    xyz, _ = fixture_cloud
    got = lom.loadPCDFile(REF_PCD)
    assert len(got) == 59691
    assert got.tobytes() == xyz.tobytes()
