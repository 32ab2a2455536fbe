"""Host-side volume I/O and orientation (reference: utils.load_input_image / read_dicoms utils.py:132-269,
mask.py:156-164,204-208).  SimpleITK is not installed here, so the checks are format round trips, hand-built
files, physical-position invariants and -- when /root/reference is mounted -- the reference's own DICOM fixtures."""
import gzip
import itertools
import os
import struct

import numpy as np
import pytest

from lungmask_amd import volume_io as vio


def _rot(axis, deg):
    t = np.deg2rad(deg)
    c, s = np.cos(t), np.sin(t)
    m = {0: [[1, 0, 0], [0, c, -s], [0, s, c]], 1: [[c, 0, s], [0, 1, 0], [-s, 0, c]], 2: [[c, -s, 0], [s, c, 0], [0, 0, 1]]}[axis]
    return np.asarray(m, dtype=np.float64)


def test_orientation_codes():
    assert vio.orientation_code(np.eye(3)) == "LPS"
    assert vio.orientation_code(np.diag([-1.0, -1.0, 1.0])) == "RAS"
    assert vio.orientation_code(np.diag([1.0, -1.0, -1.0])) == "LAI"
    d = np.zeros((3, 3))  # x index axis runs towards Superior, y towards Left, z towards Anterior
    d[2, 0], d[0, 1], d[1, 2] = 1, 1, -1
    assert vio.orientation_code(d) == "SLA"
    assert vio.orientation_code(_rot(2, 20) @ _rot(0, -15)) == "LPS"  # oblique but dominated by the identity


def test_every_axis_permutation_and_flip_round_trips():
    rng = np.random.default_rng(0)
    a = rng.integers(-100, 100, (3, 4, 5)).astype(np.int16)
    for perm in itertools.permutations(range(3)):
        for signs in itertools.product((1, -1), repeat=3):
            d = np.zeros((3, 3))
            for c in range(3):
                d[perm[c], c] = signs[c]
            d = _rot(1, 7) @ d  # slightly oblique
            vol = vio.Volume(a, spacing=(0.7, 0.8, 2.5), origin=(10, -20, 30), direction=d)
            axes, flips = vio.lps_transform(d)
            lps = vio.reoriented_geometry(vol, axes, flips)
            assert vio.orientation_code(lps.direction) == "LPS"
            # every voxel keeps its value at its physical position
            for _ in range(10):
                k, j, i = (int(rng.integers(0, n)) for n in lps.array.shape)
                p = lps.index_to_physical((i, j, k))
                ijk = np.linalg.solve(vol.direction * np.asarray(vol.spacing)[None, :], p - np.asarray(vol.origin))
                ii, jj, kk = (int(round(v)) for v in ijk)
                assert lps.array[k, j, i] == a[kk, jj, ii]
            back = vio.apply_transform(lps.array, *vio.inverse_transform(axes, flips))
            assert np.array_equal(back, a)


@pytest.mark.parametrize("ext", [".nii", ".nii.gz", ".mha", ".mhd"])
def test_round_trip_formats(tmp_path, ext):
    rng = np.random.default_rng(1)
    for dt in (np.int16, np.uint8, np.float32):
        a = rng.integers(0, 200, (4, 6, 5)).astype(dt)
        d = _rot(2, 10) @ np.diag([-1.0, 1.0, -1.0])
        vol = vio.Volume(a, (0.5, 0.75, 3.0), (12.5, -7.0, 100.0), d)
        path = str(tmp_path / ("v" + ext))
        vio.save_image(path, vol)
        got = vio.load_input_image(path)
        assert got.array.dtype == dt and np.array_equal(got.array, a)
        np.testing.assert_allclose(got.spacing, vol.spacing, rtol=1e-6)
        np.testing.assert_allclose(got.origin, vol.origin, rtol=1e-6)
        np.testing.assert_allclose(got.direction, d, atol=1e-6)


def _nifti_header(en, shape_xyz, code, bitpix, qform=0, sform=0, quat=(0, 0, 0), qoff=(0, 0, 0), pixdim=(1, 1, 1, 1), srow=None, slope=0.0, inter=0.0):
    h = bytearray(352)
    struct.pack_into(en + "i", h, 0, 348)
    struct.pack_into(en + "8h", h, 40, 3, *shape_xyz, 1, 1, 1, 1)
    struct.pack_into(en + "hh", h, 70, code, bitpix)
    struct.pack_into(en + "8f", h, 76, *pixdim, 0, 0, 0, 0)
    struct.pack_into(en + "3f", h, 108, 352.0, slope, inter)
    struct.pack_into(en + "2h", h, 252, qform, sform)
    struct.pack_into(en + "6f", h, 256, *quat, *qoff)
    if srow is not None:
        struct.pack_into(en + "12f", h, 280, *np.asarray(srow, dtype=np.float64).reshape(-1))
    h[344:348] = b"n+1\0"
    return bytes(h)


def test_nifti_hand_built_headers(tmp_path):
    a = np.arange(2 * 3 * 4, dtype=np.int16).reshape(2, 3, 4)
    # (1) big-endian, no qform/sform: RAS identity -> direction diag(-1,-1,1) in LPS
    p = tmp_path / "be.nii"
    p.write_bytes(_nifti_header(">", (4, 3, 2), 4, 16, pixdim=(1, 0.5, 0.6, 2.0)) + a.astype(">i2").tobytes())
    v = vio.read_nifti(str(p))
    assert np.array_equal(v.array, a) and v.array.dtype == np.int16
    assert vio.orientation_code(v.direction) == "RAS" and v.spacing == pytest.approx((0.5, 0.6, 2.0))
    # (2) qform: 180 degrees about z (b=c=0, d=1) in RAS == identity in LPS; qfac=-1 flips the third axis
    p = tmp_path / "q.nii.gz"
    p.write_bytes(gzip.compress(_nifti_header("<", (4, 3, 2), 4, 16, qform=1, quat=(0, 0, 1), qoff=(5, 6, 7), pixdim=(-1, 1, 1, 1)) + a.tobytes()))
    v = vio.read_nifti(str(p))
    np.testing.assert_allclose(v.direction, np.diag([1.0, 1.0, -1.0]), atol=1e-6)
    assert vio.orientation_code(v.direction) == "LPI" and v.origin == pytest.approx((-5, -6, 7))
    # (3) sform only, with scaling; (4) scl_slope/inter rescale into floating point
    srow = [[-0.7, 0, 0, 90], [0, 0, 2.5, -100], [0, -0.8, 0, 30]]
    p = tmp_path / "s.nii"
    p.write_bytes(_nifti_header("<", (4, 3, 2), 4, 16, sform=2, srow=srow, slope=2.0, inter=-1024.0) + a.tobytes())
    v = vio.read_nifti(str(p))
    assert v.spacing == pytest.approx((0.7, 0.8, 2.5)) and v.origin == pytest.approx((-90, 100, 30))
    assert vio.orientation_code(v.direction) == "LIA"
    assert v.array.dtype == np.float32 and np.array_equal(v.array, a.astype(np.float32) * 2 - 1024)
    with pytest.raises(ValueError):
        bad = tmp_path / "bad.nii"
        bad.write_bytes(b"\0" * 400)
        vio.read_nifti(str(bad))


# ---- DICOM ------------------------------------------------------------------------------------
def _el(tag, vr, value, explicit=True):
    g, e = tag
    if isinstance(value, str):
        value = value.encode("ascii")
    if len(value) % 2:
        value += b"\0" if vr == "UI" else b" "
    if not explicit:
        return struct.pack("<HHI", g, e, len(value)) + value
    if vr in ("OB", "OW", "SQ", "UN", "UT"):
        return struct.pack("<HH2sHI", g, e, vr.encode(), 0, len(value)) + value
    return struct.pack("<HH2sH", g, e, vr.encode(), len(value)) + value


def write_dicom(path, pixels, ipp, series="1.2.3.4", study="1.2.3", image_type="ORIGINAL\\PRIMARY\\AXIAL", explicit=True, intercept=0,
                signed=1, iop="1\\0\\0\\0\\1\\0", with_sequence=False):
    ts = "1.2.840.10008.1.2.1" if explicit else "1.2.840.10008.1.2"
    meta = _el((2, 0x10), "UI", ts)
    meta = _el((2, 0), "UL", struct.pack("<I", len(meta))) + meta
    ds = b""
    if image_type is not None:
        ds += _el((8, 8), "CS", image_type, explicit)
    ds += _el((8, 0x20), "DA", "20240102", explicit)
    if with_sequence:  # an undefined-length sequence with one undefined-length item holding a nested element
        inner = _el((8, 0x100), "SH", "CODE", explicit)
        item = struct.pack("<HHI", 0xFFFE, 0xE000, 0xFFFFFFFF) + inner + struct.pack("<HHI", 0xFFFE, 0xE00D, 0)
        seq = item + struct.pack("<HHI", 0xFFFE, 0xE0DD, 0)
        ds += (struct.pack("<HH2sHI", 8, 0x1140, b"SQ", 0, 0xFFFFFFFF) if explicit else struct.pack("<HHI", 8, 0x1140, 0xFFFFFFFF)) + seq
    ds += _el((0x10, 0x10), "PN", "Doe^Jane", explicit)
    ds += _el((0x20, 0xD), "UI", study, explicit) + _el((0x20, 0xE), "UI", series, explicit)
    ds += _el((0x20, 0x32), "DS", "\\".join(str(v) for v in ipp), explicit) + _el((0x20, 0x37), "DS", iop, explicit)
    ds += _el((0x28, 2), "US", struct.pack("<H", 1), explicit)
    ds += _el((0x28, 0x10), "US", struct.pack("<H", pixels.shape[0]), explicit) + _el((0x28, 0x11), "US", struct.pack("<H", pixels.shape[1]), explicit)
    ds += _el((0x28, 0x30), "DS", "0.8\\0.6", explicit)
    ds += _el((0x28, 0x100), "US", struct.pack("<H", 16), explicit) + _el((0x28, 0x101), "US", struct.pack("<H", 16), explicit)
    ds += _el((0x28, 0x103), "US", struct.pack("<H", signed), explicit)
    ds += _el((0x28, 0x1052), "DS", str(intercept), explicit) + _el((0x28, 0x1053), "DS", "1", explicit)
    ds += _el((0x7FE0, 0x10), "OW", pixels.astype("<i2" if signed else "<u2").tobytes(), explicit)
    with open(path, "wb") as f:
        f.write(b"\0" * 128 + b"DICM" + meta + ds)


@pytest.mark.parametrize("explicit", [True, False])
def test_dicom_series_logic(tmp_path, explicit):
    rng = np.random.default_rng(2)
    sl = [rng.integers(-1000, 1000, (6, 5)).astype(np.int16) for _ in range(4)]
    d = tmp_path / "study"
    (d / "sub").mkdir(parents=True)
    # main series: written out of order, one duplicate under another name, one localizer, one file without ImageType
    for k in (2, 0, 3, 1):
        write_dicom(d / f"im{k}.dcm", sl[k], (1.0, 2.0, 10.0 + 2.5 * k), explicit=explicit, with_sequence=(k == 0))
    write_dicom(d / "sub" / "copy_of_1.dcm", sl[1], (1.0, 2.0, 12.5), explicit=explicit)
    write_dicom(d / "loc.dcm", sl[0], (0, 0, -50.0), image_type="ORIGINAL\\PRIMARY\\LOCALIZER", explicit=explicit)
    write_dicom(d / "notype.dcm", sl[0], (0, 0, -60.0), image_type=None, explicit=explicit)
    # a smaller second series and a non-DICOM file
    write_dicom(d / "other.dcm", sl[0], (0, 0, 0), series="1.2.3.5", explicit=explicit)
    (d / "readme.txt").write_text("not dicom at all, but longer than eight bytes")
    v = vio.load_input_image(str(d))
    assert v.array.shape == (4, 6, 5) and v.array.dtype == np.int16
    for k in range(4):
        assert np.array_equal(v.array[k], sl[k])
    assert v.spacing == pytest.approx((0.6, 0.8, 2.5)) and v.origin == pytest.approx((1.0, 2.0, 10.0))
    assert vio.orientation_code(v.direction) == "LPS"
    assert v.meta["0010|0010"].strip() == "Doe^Jane" and v.meta["0008|0020"] == "20240102"
    assert len(vio.read_dicoms(str(d), primary=False, original=False)) == 2


def test_dicom_rescale_and_feet_first(tmp_path):
    px = np.arange(30, dtype=np.uint16).reshape(6, 5) + 1000
    d = tmp_path / "s"
    d.mkdir()
    for k in range(3):  # z decreases with the instance index; sorting by position restores ascending z
        write_dicom(d / f"{k}.dcm", px + k, (0, 0, -5.0 * k), signed=0, intercept=-1024)
    v = vio.load_input_image(str(d))
    assert v.array.dtype == np.int32  # unsigned 16 bit stored, shifted: does not fit int16 for the full stored range
    assert np.array_equal(v.array[0], px.astype(np.int32) + 2 - 1024) and np.array_equal(v.array[2], px.astype(np.int32) - 1024)
    assert v.origin == pytest.approx((0, 0, -10.0)) and v.spacing[2] == pytest.approx(5.0)
    one = vio.load_input_image(str(d / "1.dcm"))
    assert one.array.shape == (1, 6, 5)


def test_dicom_compressed_is_refused_without_simpleitk(tmp_path):
    p = tmp_path / "c.dcm"
    meta = _el((2, 0x10), "UI", "1.2.840.10008.1.2.4.70")
    meta = _el((2, 0), "UL", struct.pack("<I", len(meta))) + meta
    ds = _el((8, 8), "CS", "ORIGINAL\\PRIMARY") + struct.pack("<HH2sHI", 0x7FE0, 0x10, b"OB", 0, 0xFFFFFFFF)
    p.write_bytes(b"\0" * 128 + b"DICM" + meta + ds)
    with pytest.raises((vio.DicomError, ImportError)):
        vio.load_input_image(str(p))


@pytest.mark.skipif(not os.path.isdir("/root/reference/tests/testdata"), reason="reference fixtures not mounted")
def test_reference_dicom_fixtures():
    """tests/testdata/{0,1}.dcm of the reference (built by tests/test_utils.py:18-55): int16 512x512 each."""
    v = vio.load_input_image("/root/reference/tests/testdata")
    assert v.array.shape == (2, 512, 512) and v.array.dtype == np.int16
    assert v.spacing[:2] == pytest.approx((0.625, 0.625)) and vio.orientation_code(v.direction) == "LPS"
    for k, name in enumerate(sorted(os.listdir("/root/reference/tests/testdata"))):
        raw = np.fromfile(os.path.join("/root/reference/tests/testdata", name), dtype="<i2", count=512 * 512, offset=910).reshape(512, 512)
        assert any(np.array_equal(v.array[j], raw) for j in range(2))


def test_dicom_writer_roundtrip_and_tag_carry_over(tmp_path):
    """__main__.py:119-144 without SimpleITK: the label volume goes out as one explicit-VR little-endian multi-frame file with the
    input's geometry, the carried-over study / patient tags (utils.py:17-30), the original Study Instance UID
    (SetKeepOriginalImageUID), 'Created with lungmask' and the 1 / 2 window; series and instance UIDs are new."""
    from lungmask_amd.__main__ import DICOM_METADATA_TO_KEEP

    rng = np.random.default_rng(3)
    lab = rng.integers(0, 6, (7, 24, 18)).astype(np.uint8)
    direction = np.array([[0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, -1.0]])
    in_meta = {"0008|0020": "20240131", "0010|0010": "DOE^JANE", "0010|0020": "ID-7", "0020|000d": "1.2.826.0.1.3680043.8.498.1",
               "0020|0010": "S1", "0018|5100": "FFS", "0008|0070": "SomeVendor", "0020|000e": "1.2.3.999"}
    image = vio.Volume(np.zeros_like(lab, dtype=np.int16), (0.7, 0.8, 2.5), (10.0, -20.5, 33.25), direction, in_meta)
    keep = {k: v for k, v in image.meta.items() if k in DICOM_METADATA_TO_KEEP}  # as lungmask_amd/__main__.py builds it
    keep.update({"0008|103e": "Created with lungmask", "0028|1050": "1", "0028|1051": "2"})
    out = tmp_path / "mask.dcm"
    vio.save_image(str(out), image.like(lab), keep)
    raw = out.read_bytes()
    assert raw[128:132] == b"DICM"
    tags, pixels, tsuid = vio.parse_dicom(raw)
    assert tsuid == "1.2.840.10008.1.2.1" and len(pixels) == lab.size
    got = vio.load_input_image(str(out))
    assert np.array_equal(got.array, lab) and got.array.shape == lab.shape
    assert np.allclose(got.spacing, image.spacing) and np.allclose(got.origin, image.origin) and np.allclose(got.direction, direction)
    m = {k: v.strip() for k, v in got.meta.items()}
    for k in ("0008|0020", "0010|0010", "0010|0020", "0020|000d", "0020|0010", "0018|5100"):
        assert m[k] == in_meta[k], k
    assert m["0008|103e"] == "Created with lungmask" and m["0028|1050"] == "1" and m["0028|1051"] == "2"
    assert "0008|0070" not in m                       # not in the keep list
    assert m["0020|000e"] != in_meta["0020|000e"]      # a new series
    assert m["0028|0008"] == "7" and m["0008|0016"] == "1.2.840.10008.5.1.4.1.1.7.2"
    # --removemetadata: nothing carried over, a fresh study
    vio.save_image(str(tmp_path / "anon.dcm"), image.like(lab), None)
    m2 = {k: v.strip() for k, v in vio.load_input_image(str(tmp_path / "anon.dcm")).meta.items()}
    assert "0010|0010" not in m2 and m2["0020|000d"] != in_meta["0020|000d"] and m2["0020|000d"].startswith("2.25.")
    # every element has an even length and the tags are in ascending order (PS3.5 7.1)
    keys = [k for k in tags if k[0] > 2]
    assert keys == sorted(keys) and all(len(v[1]) % 2 == 0 for v in tags.values())


def test_dicom_writer_left_handed_volume_keeps_every_voxel_in_place(tmp_path):
    """a multi-frame file advances its frames along +cross(row, column).  A left-handed volume (third direction column
    = -cross: e.g. a NIfTI / MetaImage input written to .dcm) goes out with its frames reversed from the position of its last
    slice, so every voxel reads back at its physical position; an oblique slice axis is written along the in-plane normal with a warning."""
    rng = np.random.default_rng(9)
    lab = rng.integers(0, 4, (5, 6, 7)).astype(np.uint8)
    d = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, -1.0]])  # columns: row dir, column dir, slice dir = -cross(row, column)
    vol = vio.Volume(lab, (0.5, 0.75, 2.0), (3.0, -4.0, 50.0), d)
    out = tmp_path / "lh.dcm"
    vio.save_image(str(out), vol, None)
    got = vio.load_input_image(str(out))
    assert np.array_equal(got.array, lab[::-1])

    def pos(v, z, y, x):
        return np.asarray(v.origin) + np.asarray(v.direction).reshape(3, 3) @ (np.asarray([x, y, z]) * np.asarray(v.spacing))

    for z, y, x in ((0, 0, 0), (4, 5, 6), (2, 1, 3)):
        assert np.allclose(pos(vol, z, y, x), pos(got, lab.shape[0] - 1 - z, y, x))
    # a gantry-tilted series (slice axis 30 degrees off the in-plane normal): the reference's SimpleITK writer accepts it, and the
    # writer runs after the whole inference -- a warning and the projected geometry, not an error that loses the result
    tilted = vio.Volume(lab, (1, 1, 2.0), (0, 0, 0), np.array([[1.0, 0, 0.5], [0, 1.0, 0], [0, 0, 0.8660254]]))
    with pytest.warns(RuntimeWarning, match="not perpendicular"):
        vio.save_image(str(tmp_path / "oblique.dcm"), tilted, None)
    got = vio.load_input_image(str(tmp_path / "oblique.dcm"))
    assert np.array_equal(got.array, lab) and np.allclose(got.spacing, (1, 1, 2.0 * 0.8660254), atol=1e-5)
    assert np.allclose(np.asarray(got.direction).reshape(3, 3), np.eye(3), atol=1e-6)
    #  a slice axis nearer to the image plane than to its normal has no meaningful projected spacing: refused, not written
    flat = vio.Volume(lab, (1, 1, 2.0), (0, 0, 0), np.array([[1.0, 0, 0.9], [0, 1.0, 0], [0, 0, 0.43589]]))
    with pytest.raises(ValueError, match="cannot express"):
        vio.save_image(str(tmp_path / "flat.dcm"), flat, None)
    assert not os.path.exists(tmp_path / "flat.dcm")
    # ... and an essentially perpendicular series (cos > 0.999) keeps the direction it came with, untouched
    d = np.array([[1.0, 0, 0.01], [0, 1.0, 0], [0, 0, 0.99995]])
    vio.save_image(str(tmp_path / "straight.dcm"), vio.Volume(lab, (1, 1, 2.0), (0, 0, 0), d), None)
    got = vio.load_input_image(str(tmp_path / "straight.dcm"))
    assert np.array_equal(got.array, lab) and np.allclose(got.spacing, (1, 1, 2.0), atol=1e-6)


_CONDA_PY = "/opt/conda/bin/python3.9"  # the interpreter oracle/make_golden.py runs the reference's utils.py under (scikit-image 0.18.3, imageio 2.9)


def _have_imageio():
    import subprocess

    if not os.path.exists(_CONDA_PY):
        return False
    return subprocess.run([_CONDA_PY, "-c", "import imageio.plugins.dicom"], capture_output=True).returncode == 0


@pytest.mark.skipif(not _have_imageio(), reason="no second interpreter with imageio's DICOM plugin in this container")
@pytest.mark.parametrize("dtype", [np.uint8, np.int16])
def test_dicom_writer_is_read_by_an_independent_parser(tmp_path, dtype):
    """what write_dicom emits had only ever been read back by this repository's own parser.  imageio's DICOM plugin
    (a pure-Python reader that shares no code with lungmask_amd.volume_io; neither SimpleITK nor pydicom exist in this image) reads
    the multi-frame file: same voxels frame by frame, rows / columns / frames, pixel spacing (row \\ column order), slice spacing,
    position and orientation of frame 0, transfer syntax, SOP class, and the carried-over text tags."""
    import json
    import subprocess

    rng = np.random.default_rng(11)
    lab = rng.integers(0, 6, (7, 24, 18)).astype(dtype)
    if dtype == np.int16:
        lab = (lab.astype(np.int16) * 411 - 1024).astype(np.int16)  # signed 16-bit values incl. negatives
    vol = vio.Volume(lab, (0.7, 0.8, 2.5), (10.0, -20.5, 33.25), np.eye(3), {})
    out = tmp_path / "mask.dcm"
    vio.save_image(str(out), vol, {"0010|0010": "DOE^JANE", "0008|103e": "Created with lungmask", "0020|000d": "1.2.826.0.1.3680043.8.498.1"})
    script = ("import sys, json, numpy as np, imageio\n"
              "v = imageio.volread(sys.argv[1], 'DICOM')\n"
              "np.save(sys.argv[2], np.asarray(v).astype(np.int64))\n"
              "m = v.meta\n"
              "keys = ['TransferSyntaxUID', 'SOPClassUID', 'SeriesDescription', 'PatientName', 'StudyInstanceUID', 'SliceSpacing', 'ImagePositionPatient',\n"
              "        'ImageOrientationPatient', 'NumberOfFrames', 'Rows', 'Columns', 'PixelSpacing', 'BitsAllocated', 'PixelRepresentation', 'sampling']\n"
              "print(json.dumps({k: (list(m[k]) if isinstance(m[k], tuple) else m[k]) for k in keys if k in m}))\n")
    r = subprocess.run([_CONDA_PY, "-W", "ignore", "-c", script, str(out), str(tmp_path / "back.npy")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    m = json.loads(r.stdout.strip().splitlines()[-1])
    back = np.load(tmp_path / "back.npy")
    if dtype == np.uint8:
        back = back & 0xFF  # (imageio hands 8-bit unsigned frames out as int8; the labels are < 128 either way)
    assert back.shape == lab.shape and np.array_equal(back, lab.astype(np.int64))
    assert m["TransferSyntaxUID"] == "1.2.840.10008.1.2.1"
    assert m["SOPClassUID"] == ("1.2.840.10008.5.1.4.1.1.7.2" if dtype == np.uint8 else "1.2.840.10008.5.1.4.1.1.7.3")
    assert (m["NumberOfFrames"], m["Rows"], m["Columns"]) == (7, 24, 18)
    assert m["BitsAllocated"] == (8 if dtype == np.uint8 else 16) and m["PixelRepresentation"] == (0 if dtype == np.uint8 else 1)
    assert m["PixelSpacing"] == pytest.approx([0.8, 0.7]) and m["SliceSpacing"] == pytest.approx(2.5) and m["sampling"] == pytest.approx([2.5, 0.8, 0.7])
    assert m["ImagePositionPatient"] == pytest.approx([10.0, -20.5, 33.25]) and m["ImageOrientationPatient"] == pytest.approx([1, 0, 0, 0, 1, 0])
    assert m["PatientName"] == "DOE^JANE" and m["SeriesDescription"] == "Created with lungmask" and m["StudyInstanceUID"] == "1.2.826.0.1.3680043.8.498.1"
