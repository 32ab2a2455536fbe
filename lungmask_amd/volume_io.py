"""Volume I/O for the command line -- dependency-free readers/writers for the formats the reference
reaches through SimpleITK/pydicom (`utils.load_input_image` / `read_dicoms`, utils.py:132-269; the writer
of `__main__.py:119-144`), and the orientation logic of `mask.py:156-164,204-208` as an index transform.

* `Volume`: voxel array [z][y][x] as stored + the ITK-style geometry (LPS physical space; `direction`
  columns are the physical directions of the x, y, z index axes; `spacing` / `origin` in x, y, z order).
* NIfTI-1 (`.nii`, `.nii.gz`), MetaImage (`.mha`, `.mhd` + raw, zlib) : read and write.
* DICOM: uncompressed little-endian files (explicit or implicit VR), single files and series folders,
  with the reference's series logic (ImageType required, LOCALIZER skipped, duplicates by
  (study, series, position) dropped, slices sorted by the z position, largest series wins).
  Compressed transfer syntaxes need SimpleITK/GDCM (the CLI falls back to it when importable).
* `orientation_code` restates ITK's `DICOMOrientImageFilter::DirectionCosinesToOrientation` (greedy
  dominant-axis assignment); SimpleITK is not installed in the build image, so agreement with ITK is
  by construction, not by test ("parity unpinned" for exotic oblique ties).

All of this is host-side I/O, off the hot path.
"""
from __future__ import annotations

import gzip
import os
import struct
import sys
import zlib
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .logger import logger


class Volume:
    """A 3-D image: `array[z][y][x]` (file order) + geometry in LPS physical space."""

    def __init__(self, array: np.ndarray, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), direction=None, meta: Optional[Dict[str, str]] = None):
        assert array.ndim == 3, f"expected a 3-D volume, got shape {array.shape}"
        self.array = array
        self.spacing = tuple(float(s) for s in spacing)
        self.origin = tuple(float(o) for o in origin)
        self.direction = np.eye(3) if direction is None else np.asarray(direction, dtype=np.float64).reshape(3, 3)
        self.meta = dict(meta or {})

    # the few SimpleITK accessors user code tends to call
    def GetSize(self):
        return tuple(int(s) for s in self.array.shape[::-1])

    def GetSpacing(self):
        return self.spacing

    def GetOrigin(self):
        return self.origin

    def GetDirection(self):
        return tuple(float(v) for v in self.direction.reshape(-1))

    def like(self, array: np.ndarray) -> "Volume":
        """`CopyInformation` (__main__.py:120): same geometry, new voxels."""
        assert tuple(array.shape) == tuple(self.array.shape), (array.shape, self.array.shape)
        return Volume(array, self.spacing, self.origin, self.direction, self.meta)

    def index_to_physical(self, ijk: Sequence[float]) -> np.ndarray:
        return np.asarray(self.origin) + self.direction @ (np.asarray(ijk, dtype=np.float64) * np.asarray(self.spacing))


# ------------------------------------------------------------------------------------------------
# orientation (mask.py:156-164, 204-208)
# ------------------------------------------------------------------------------------------------
_LETTERS = (("R", "L"), ("A", "P"), ("I", "S"))  # physical axis -> (negative, positive) in LPS


def _dominant_axes(direction) -> Tuple[List[int], List[int]]:
    """For each index axis (column of `direction`): its dominant physical axis and sign.
    Greedy on the largest |cosine| like ITK: take the overall maximum, strike its row and column, repeat.
    Ties resolve to the later (row, column) in row-major order (std::multimap::rbegin of equal keys)."""
    d = np.asarray(direction, dtype=np.float64).reshape(3, 3)
    entries = sorted(((abs(d[r, c]), r * 3 + c, r, c) for r in range(3) for c in range(3)))
    phys = [-1, -1, -1]
    sign = [1, 1, 1]
    for _ in range(3):
        _, _, r, c = entries[-1]
        phys[c] = r
        sign[c] = 1 if d[r, c] > 0 else -1
        entries = [e for e in entries if e[2] != r and e[3] != c]
    return phys, sign


def orientation_code(direction) -> str:
    """`sitk.DICOMOrientImageFilter_GetOrientationFromDirectionCosines`: letter k = where index axis k
    increases towards ("LPS" = identity direction)."""
    phys, sign = _dominant_axes(direction)
    return "".join(_LETTERS[phys[c]][1 if sign[c] > 0 else 0] for c in range(3))


def lps_transform(direction) -> Tuple[Tuple[int, int, int], Tuple[bool, bool, bool]]:
    """(axes, flips) in numpy [z][y][x] terms such that `to_lps(a) == flip(a.transpose(axes))`:
    output axis k is input axis axes[k], reversed where flips[k]."""
    phys, sign = _dominant_axes(direction)
    src = [phys.index(p) for p in range(3)]  # ITK output axis p <- ITK input axis src[p]
    axes = tuple(2 - src[2 - k] for k in range(3))
    flips = tuple(sign[src[2 - k]] < 0 for k in range(3))
    return axes, flips


def inverse_transform(axes, flips):
    inv_axes = tuple(int(np.argsort(axes)[k]) for k in range(3))
    inv_flips = tuple(bool(flips[inv_axes[k]]) for k in range(3))
    return inv_axes, inv_flips


def apply_transform(a: np.ndarray, axes, flips) -> np.ndarray:
    """Host version of `lm_reorient_dev` (used by the tests and for tiny inputs)."""
    out = a.transpose(axes)
    for k in range(3):
        if flips[k]:
            out = np.flip(out, axis=k)
    return out


def reoriented_geometry(vol: Volume, axes, flips) -> Volume:
    """Geometry of `apply_transform(vol.array)`: every voxel keeps its physical position."""
    a = apply_transform(vol.array, axes, flips)
    n_in = vol.array.shape
    sp, org, d = list(vol.spacing), np.asarray(vol.origin, dtype=np.float64), vol.direction
    new_sp, new_d = [0.0] * 3, np.zeros((3, 3))
    start = [0.0, 0.0, 0.0]  # input ITK index of output voxel (0,0,0)
    for k in range(3):  # output numpy axis k == output ITK axis 2-k
        ia = axes[k]  # input numpy axis
        itk_in, itk_out = 2 - ia, 2 - k
        new_sp[itk_out] = sp[itk_in]
        new_d[:, itk_out] = d[:, itk_in] * (-1.0 if flips[k] else 1.0)
        if flips[k]:
            start[itk_in] = n_in[ia] - 1
    new_org = org + d @ (np.asarray(start) * np.asarray(sp))
    return Volume(a, new_sp, new_org, new_d, vol.meta)


# ------------------------------------------------------------------------------------------------
# NIfTI-1
# ------------------------------------------------------------------------------------------------
_NIFTI_DTYPES = {2: "u1", 4: "i2", 8: "i4", 16: "f4", 64: "f8", 256: "i1", 512: "u2", 768: "u4", 1024: "i8", 1280: "u8"}
_NIFTI_CODES = {np.dtype(v).str[1:]: k for k, v in _NIFTI_DTYPES.items()}
_RAS_TO_LPS = np.diag([-1.0, -1.0, 1.0])


def _read_all(path: str) -> bytes:
    with open(path, "rb") as f:
        head = f.read(2)
        f.seek(0)
        raw = f.read()
    return gzip.decompress(raw) if head == b"\x1f\x8b" else raw


def read_nifti(path: str) -> Volume:
    buf = _read_all(path)
    if len(buf) < 348:
        raise ValueError(f"{path}: too short for a NIfTI-1 header")
    en = "<" if struct.unpack_from("<i", buf, 0)[0] == 348 else ">"
    if struct.unpack_from(en + "i", buf, 0)[0] != 348:
        raise ValueError(f"{path}: not a NIfTI-1 file (sizeof_hdr != 348)")
    magic = buf[344:348]
    if magic not in (b"n+1\0", b"ni1\0"):
        raise ValueError(f"{path}: bad NIfTI magic {magic!r}")
    dim = struct.unpack_from(en + "8h", buf, 40)
    ndim = dim[0]
    if ndim < 3 or any(d > 1 for d in dim[4 : 1 + ndim]):
        raise ValueError(f"{path}: expected a 3-D image, dim = {dim}")
    nx, ny, nz = dim[1:4]
    datatype = struct.unpack_from(en + "h", buf, 70)[0]
    if datatype not in _NIFTI_DTYPES:
        raise ValueError(f"{path}: unsupported NIfTI datatype {datatype}")
    pixdim = struct.unpack_from(en + "8f", buf, 76)
    vox_offset, slope, inter = struct.unpack_from(en + "3f", buf, 108)
    qform_code, sform_code = struct.unpack_from(en + "2h", buf, 252)
    qb, qc, qd, qx, qy, qz = struct.unpack_from(en + "6f", buf, 256)
    srow = np.asarray(struct.unpack_from(en + "12f", buf, 280), dtype=np.float64).reshape(3, 4)
    if magic == b"ni1\0":
        img = os.path.splitext(path[:-3] if path.endswith(".gz") else path)[0] + ".img"
        data = _read_all(img if os.path.exists(img) else img + ".gz")
        off = 0
    else:
        data, off = buf, int(vox_offset)
    dt = np.dtype(en + _NIFTI_DTYPES[datatype])
    arr = np.frombuffer(data, dtype=dt, count=nx * ny * nz, offset=off).reshape(nz, ny, nx)
    arr = arr.astype(dt.newbyteorder("="))
    if slope != 0 and not (slope == 1 and inter == 0) and np.isfinite(slope):  # ITK rescales into floating point
        arr = arr.astype(np.float32) * np.float32(slope) + np.float32(inter)
    spacing = [abs(pixdim[1]) or 1.0, abs(pixdim[2]) or 1.0, abs(pixdim[3]) or 1.0]
    if qform_code > 0:
        a2 = 1.0 - (qb * qb + qc * qc + qd * qd)
        qa = np.sqrt(a2) if a2 > 1e-7 else 0.0
        if a2 <= 1e-7:  # 180 degree rotation: renormalise (nifti1_io quatern_to_mat44)
            nrm = 1.0 / np.sqrt(qb * qb + qc * qc + qd * qd)
            qb, qc, qd = qb * nrm, qc * nrm, qd * nrm
        R = np.array([[qa * qa + qb * qb - qc * qc - qd * qd, 2 * (qb * qc - qa * qd), 2 * (qb * qd + qa * qc)],
                      [2 * (qb * qc + qa * qd), qa * qa + qc * qc - qb * qb - qd * qd, 2 * (qc * qd - qa * qb)],
                      [2 * (qb * qd - qa * qc), 2 * (qc * qd + qa * qb), qa * qa + qd * qd - qc * qc - qb * qb]])
        if pixdim[0] < 0:
            R[:, 2] = -R[:, 2]
        origin_ras = np.array([qx, qy, qz], dtype=np.float64)
    elif sform_code > 0:
        M = srow[:, :3]
        norms = np.linalg.norm(M, axis=0)
        norms[norms == 0] = 1.0
        R = M / norms
        spacing = [float(v) for v in norms]
        origin_ras = srow[:, 3].copy()
    else:
        R = np.eye(3)
        origin_ras = np.zeros(3)
    return Volume(arr, spacing, _RAS_TO_LPS @ origin_ras, _RAS_TO_LPS @ R)


def write_nifti(path: str, vol: Volume) -> None:
    arr = np.ascontiguousarray(vol.array)
    code = _NIFTI_CODES.get(arr.dtype.str[1:])
    if code is None:
        raise ValueError(f"write_nifti: unsupported dtype {arr.dtype}")
    arr = arr.astype(arr.dtype.newbyteorder("<"))
    nz, ny, nx = arr.shape
    R = _RAS_TO_LPS @ vol.direction
    org = _RAS_TO_LPS @ np.asarray(vol.origin)
    qfac = -1.0 if np.linalg.det(R) < 0 else 1.0
    Rq = R.copy()
    Rq[:, 2] *= qfac
    # rotation -> quaternion (a >= 0), the branchy form of nifti1_io mat44_to_quatern
    tr = Rq[0, 0] + Rq[1, 1] + Rq[2, 2]
    if tr + 1.0 > 0.5:
        a = 0.5 * np.sqrt(tr + 1.0)
        b, c, d = 0.25 * (Rq[2, 1] - Rq[1, 2]) / a, 0.25 * (Rq[0, 2] - Rq[2, 0]) / a, 0.25 * (Rq[1, 0] - Rq[0, 1]) / a
    else:
        xd, yd, zd = 1 + Rq[0, 0] - (Rq[1, 1] + Rq[2, 2]), 1 + Rq[1, 1] - (Rq[0, 0] + Rq[2, 2]), 1 + Rq[2, 2] - (Rq[0, 0] + Rq[1, 1])
        if xd > 1.0:
            b = 0.5 * np.sqrt(xd)
            c, d, a = 0.25 * (Rq[0, 1] + Rq[1, 0]) / b, 0.25 * (Rq[0, 2] + Rq[2, 0]) / b, 0.25 * (Rq[2, 1] - Rq[1, 2]) / b
        elif yd > 1.0:
            c = 0.5 * np.sqrt(yd)
            b, d, a = 0.25 * (Rq[0, 1] + Rq[1, 0]) / c, 0.25 * (Rq[1, 2] + Rq[2, 1]) / c, 0.25 * (Rq[0, 2] - Rq[2, 0]) / c
        else:
            d = 0.5 * np.sqrt(zd)
            b, c, a = 0.25 * (Rq[0, 2] + Rq[2, 0]) / d, 0.25 * (Rq[1, 2] + Rq[2, 1]) / d, 0.25 * (Rq[1, 0] - Rq[0, 1]) / d
        if a < 0:
            a, b, c, d = -a, -b, -c, -d
    hdr = bytearray(352)
    struct.pack_into("<i", hdr, 0, 348)
    struct.pack_into("<8h", hdr, 40, 3, nx, ny, nz, 1, 1, 1, 1)
    struct.pack_into("<hh", hdr, 70, code, arr.dtype.itemsize * 8)
    struct.pack_into("<8f", hdr, 76, qfac, vol.spacing[0], vol.spacing[1], vol.spacing[2], 0, 0, 0, 0)
    struct.pack_into("<3f", hdr, 108, 352.0, 1.0, 0.0)
    hdr[123] = 2  # xyzt_units: mm
    struct.pack_into("<2h", hdr, 252, 1, 1)  # qform/sform: scanner anatomical
    struct.pack_into("<6f", hdr, 256, b, c, d, org[0], org[1], org[2])
    S = R * np.asarray(vol.spacing)[None, :]
    for r in range(3):
        struct.pack_into("<4f", hdr, 280 + 16 * r, S[r, 0], S[r, 1], S[r, 2], org[r])
    hdr[344:348] = b"n+1\0"
    payload = bytes(hdr) + arr.tobytes()
    if path.endswith(".gz"):
        with gzip.open(path, "wb", compresslevel=1) as f:
            f.write(payload)
    else:
        with open(path, "wb") as f:
            f.write(payload)


# ------------------------------------------------------------------------------------------------
# MetaImage (.mha / .mhd)
# ------------------------------------------------------------------------------------------------
_MET_TYPES = {"MET_CHAR": "i1", "MET_UCHAR": "u1", "MET_SHORT": "i2", "MET_USHORT": "u2", "MET_INT": "i4", "MET_UINT": "u4",
              "MET_LONG": "i4", "MET_ULONG": "u4", "MET_LONG_LONG": "i8", "MET_ULONG_LONG": "u8", "MET_FLOAT": "f4", "MET_DOUBLE": "f8"}
_MET_NAMES = {"i1": "MET_CHAR", "u1": "MET_UCHAR", "i2": "MET_SHORT", "u2": "MET_USHORT", "i4": "MET_INT", "u4": "MET_UINT",
              "i8": "MET_LONG_LONG", "u8": "MET_ULONG_LONG", "f4": "MET_FLOAT", "f8": "MET_DOUBLE"}


def read_metaimage(path: str) -> Volume:
    with open(path, "rb") as f:
        buf = f.read()
    fields: Dict[str, str] = {}
    pos = 0
    while True:
        end = buf.find(b"\n", pos)
        if end < 0:
            raise ValueError(f"{path}: ElementDataFile missing from the MetaImage header")
        line = buf[pos:end].decode("latin-1").strip()
        pos = end + 1
        if "=" not in line:
            continue
        k, v = (t.strip() for t in line.split("=", 1))
        fields[k] = v
        if k == "ElementDataFile":
            break
    if int(fields.get("NDims", "3")) != 3:
        raise ValueError(f"{path}: expected NDims = 3")
    if int(fields.get("ElementNumberOfChannels", "1")) != 1:
        raise ValueError(f"{path}: multi-channel images are not supported")
    nx, ny, nz = (int(t) for t in fields["DimSize"].split())
    et = fields["ElementType"]
    if et not in _MET_TYPES:
        raise ValueError(f"{path}: unsupported ElementType {et}")
    msb = (fields.get("BinaryDataByteOrderMSB") or fields.get("ElementByteOrderMSB") or "False").lower() == "true"
    dt = np.dtype((">" if msb else "<") + _MET_TYPES[et])
    edf = fields["ElementDataFile"]
    if edf == "LOCAL":
        data = buf[pos:]
    else:
        with open(os.path.join(os.path.dirname(path), edf), "rb") as f:
            data = f.read()
        hs = int(fields.get("HeaderSize", "0"))
        data = data[len(data) - nx * ny * nz * dt.itemsize:] if hs == -1 else data[hs:]
    if fields.get("CompressedData", "False").lower() == "true":
        data = zlib.decompress(data)
    arr = np.frombuffer(data, dtype=dt, count=nx * ny * nz).reshape(nz, ny, nx).astype(dt.newbyteorder("="))
    spacing = [float(t) for t in (fields.get("ElementSpacing") or fields.get("ElementSize") or "1 1 1").split()]
    origin = [float(t) for t in (fields.get("Offset") or fields.get("Origin") or fields.get("Position") or "0 0 0").split()]
    tm = fields.get("TransformMatrix") or fields.get("Orientation") or fields.get("Rotation")
    direction = np.asarray([float(t) for t in tm.split()]).reshape(3, 3).T if tm else np.eye(3)  # listed axis by axis
    return Volume(arr, spacing, origin, direction)


def write_metaimage(path: str, vol: Volume, compress: bool = True) -> None:
    arr = np.ascontiguousarray(vol.array)
    key = arr.dtype.str[1:]
    if key not in _MET_NAMES:
        raise ValueError(f"write_metaimage: unsupported dtype {arr.dtype}")
    raw = arr.astype(arr.dtype.newbyteorder("<")).tobytes()
    payload = zlib.compress(raw, 1) if compress else raw
    nz, ny, nx = arr.shape
    local = path.lower().endswith(".mha")
    datafile = "LOCAL" if local else os.path.splitext(os.path.basename(path))[0] + (".zraw" if compress else ".raw")
    lines = ["ObjectType = Image", "NDims = 3", "BinaryData = True", "BinaryDataByteOrderMSB = False",
             f"CompressedData = {'True' if compress else 'False'}"]
    if compress:
        lines.append(f"CompressedDataSize = {len(payload)}")
    lines += ["TransformMatrix = " + " ".join(repr(float(v)) for v in vol.direction.T.reshape(-1)),
              "Offset = " + " ".join(repr(float(v)) for v in vol.origin), "CenterOfRotation = 0 0 0", "AnatomicalOrientation = " +
              "".join({"L": "R", "R": "L", "P": "A", "A": "P", "S": "I", "I": "S"}[c] for c in orientation_code(vol.direction)),
              "ElementSpacing = " + " ".join(repr(float(v)) for v in vol.spacing), f"DimSize = {nx} {ny} {nz}",
              f"ElementType = {_MET_NAMES[key]}", f"ElementDataFile = {datafile}"]
    header = ("\n".join(lines) + "\n").encode("latin-1")
    with open(path, "wb") as f:
        f.write(header)
        if local:
            f.write(payload)
    if not local:
        with open(os.path.join(os.path.dirname(path), datafile), "wb") as f:
            f.write(payload)


# ------------------------------------------------------------------------------------------------
# DICOM (uncompressed little endian)
# ------------------------------------------------------------------------------------------------
_LONG_VR = {b"OB", b"OW", b"OF", b"SQ", b"UT", b"UN", b"OD", b"OL", b"UC", b"UR", b"OV", b"SV", b"UV"}
_UNCOMPRESSED = {"1.2.840.10008.1.2": False, "1.2.840.10008.1.2.1": True}  # transfer syntax -> explicit VR
_TEXT_VR = {b"AE", b"AS", b"CS", b"DA", b"DS", b"DT", b"IS", b"LO", b"LT", b"PN", b"SH", b"ST", b"TM", b"UI", b"UT", b"UC", b"UR"}
# implicit-VR files carry no VR: the handful of tags we interpret
_IMPLICIT_VR = {(0x0028, 0x0010): b"US", (0x0028, 0x0011): b"US", (0x0028, 0x0100): b"US", (0x0028, 0x0101): b"US",
                (0x0028, 0x0103): b"US", (0x0028, 0x0002): b"US"}


class DicomError(ValueError):
    pass


def _skip_undefined(buf: bytes, pos: int, explicit: bool) -> int:
    """Skip a sequence / item of undefined length starting at `pos`; returns the position after its delimiter."""
    while pos + 8 <= len(buf):
        g, e = struct.unpack_from("<HH", buf, pos)
        if g == 0xFFFE:  # item, item delimiter, sequence delimiter: always implicit layout
            ln = struct.unpack_from("<I", buf, pos + 4)[0]
            pos += 8
            if e in (0xE00D, 0xE0DD):
                return pos
            if ln == 0xFFFFFFFF:
                pos = _skip_undefined(buf, pos, explicit)
            else:
                pos += ln
            continue
        pos, ln, _ = _element_header(buf, pos, explicit)
        pos = _skip_undefined(buf, pos, explicit) if ln == 0xFFFFFFFF else pos + ln
    raise DicomError("unterminated undefined-length sequence")


def _element_header(buf: bytes, pos: int, explicit: bool):
    """-> (value position, value length, VR or None)."""
    if explicit:
        vr = buf[pos + 4 : pos + 6]
        if vr in _LONG_VR:
            return pos + 12, struct.unpack_from("<I", buf, pos + 8)[0], vr
        return pos + 8, struct.unpack_from("<H", buf, pos + 6)[0], vr
    return pos + 8, struct.unpack_from("<I", buf, pos + 4)[0], None


def parse_dicom(buf: bytes, stop_before_pixels: bool = False):
    """-> (tags {(group, element): (vr, bytes)}, pixel_bytes or None, transfer syntax uid).  Top-level elements only."""
    pos = 132 if buf[128:132] == b"DICM" else 0  # force=True of utils.py:149: accept preamble-less files
    tags: Dict[Tuple[int, int], Tuple[Optional[bytes], bytes]] = {}
    tsuid = "1.2.840.10008.1.2"
    explicit_ds = None
    pixels = None
    n = len(buf)
    while pos + 8 <= n:
        g, e = struct.unpack_from("<HH", buf, pos)
        if g == 0x0002:
            explicit = True  # file meta information is always explicit VR little endian
        else:
            if explicit_ds is None:
                if (0x0002, 0x0010) not in tags:  # no file meta information: sniff the VR field
                    explicit_ds = buf[pos + 4 : pos + 6].isalpha() and buf[pos + 4 : pos + 6].isupper()
                elif tsuid in _UNCOMPRESSED:
                    explicit_ds = _UNCOMPRESSED[tsuid]
                elif stop_before_pixels and tsuid != "1.2.840.10008.1.2.2":
                    explicit_ds = True  # encapsulated syntaxes keep an explicit-VR little-endian header
                else:
                    raise DicomError(f"transfer syntax {tsuid} is not uncompressed little endian (needs SimpleITK/GDCM)")
            explicit = explicit_ds
        vpos, ln, vr = _element_header(buf, pos, explicit)
        if (g, e) == (0x7FE0, 0x0010):
            if stop_before_pixels:
                break
            if ln == 0xFFFFFFFF:
                raise DicomError("encapsulated (compressed) pixel data needs SimpleITK/GDCM")
            pixels = buf[vpos : vpos + ln]
            break
        if ln == 0xFFFFFFFF:
            pos = _skip_undefined(buf, vpos, explicit)
            continue
        if vr is None:
            vr = _IMPLICIT_VR.get((g, e))
        if vr != b"SQ" and ln <= 4096:
            tags[(g, e)] = (vr, buf[vpos : vpos + ln])
            if (g, e) == (0x0002, 0x0010):
                tsuid = buf[vpos : vpos + ln].decode("ascii", "ignore").strip("\0 ")
        pos = vpos + ln
    return tags, pixels, tsuid


def _text(tags, key, default=None):
    v = tags.get(key)
    if v is None:
        return default
    return v[1].decode("latin-1").strip("\0 ")


def _floats(tags, key, default=None):
    t = _text(tags, key)
    if not t:
        return default
    return [float(x) for x in t.split("\\")]


def _us(tags, key, default=None):
    v = tags.get(key)
    if v is None or len(v[1]) < 2:
        return default
    return struct.unpack_from("<H", v[1], 0)[0]


def _meta_strings(tags) -> Dict[str, str]:
    out = {}
    for (g, e), (vr, val) in tags.items():
        if vr in _TEXT_VR or (vr is None and val[:1].isalnum()):
            out[f"{g:04x}|{e:04x}"] = val.decode("latin-1").rstrip("\0")
    return out


def _dicom_pixels(tags, pixels: bytes, path: str) -> np.ndarray:
    rows, cols = _us(tags, (0x0028, 0x0010)), _us(tags, (0x0028, 0x0011))
    bits, signed = _us(tags, (0x0028, 0x0100), 16), _us(tags, (0x0028, 0x0103), 0)
    spp = _us(tags, (0x0028, 0x0002), 1)
    if rows is None or cols is None or spp != 1 or bits not in (8, 16, 32):
        raise DicomError(f"{path}: unsupported pixel layout (rows={rows}, cols={cols}, samples={spp}, bits={bits})")
    dt = np.dtype(("<i" if signed else "<u") + str(bits // 8))
    nfr = len(pixels) // (rows * cols * dt.itemsize)
    if nfr < 1:
        raise DicomError(f"{path}: pixel data shorter than rows x columns")
    px = np.frombuffer(pixels, dtype=dt, count=nfr * rows * cols).reshape(nfr, rows, cols)
    slope = (_floats(tags, (0x0028, 0x1053)) or [1.0])[0]
    inter = (_floats(tags, (0x0028, 0x1052)) or [0.0])[0]
    if slope == 1.0 and float(inter).is_integer():
        stored = _us(tags, (0x0028, 0x0101), bits)
        lo = (-(1 << (stored - 1)) if signed else 0) + int(inter)
        hi = ((1 << (stored - 1)) - 1 if signed else (1 << stored) - 1) + int(inter)
        out_dt = np.int16 if (lo >= -32768 and hi <= 32767) else np.int32 if (lo >= -(2 ** 31) and hi < 2 ** 31) else np.int64
        return (px.astype(np.int64) + int(inter)).astype(out_dt)
    return px.astype(np.float64) * slope + inter


def read_dicom_file(path: str) -> Volume:
    with open(path, "rb") as f:
        buf = f.read()
    tags, pixels, _ = parse_dicom(buf)
    if pixels is None:
        raise DicomError(f"{path}: no pixel data")
    arr = _dicom_pixels(tags, pixels, path)
    geo = _slice_geometry(tags)
    direction = np.stack([geo["row_dir"], geo["col_dir"], np.cross(geo["row_dir"], geo["col_dir"])], axis=1)
    zsp = (_floats(tags, (0x0018, 0x0088)) or _floats(tags, (0x0018, 0x0050)) or [1.0])[0]
    return Volume(arr, (geo["sx"], geo["sy"], abs(zsp) or 1.0), geo["ipp"], direction, _meta_strings(tags))


def _slice_geometry(tags):
    iop = _floats(tags, (0x0020, 0x0037)) or [1, 0, 0, 0, 1, 0]
    ipp = _floats(tags, (0x0020, 0x0032)) or [0, 0, 0]
    ps = _floats(tags, (0x0028, 0x0030)) or [1.0, 1.0]
    return {"row_dir": np.asarray(iop[:3], dtype=np.float64), "col_dir": np.asarray(iop[3:6], dtype=np.float64),
            "ipp": np.asarray(ipp[:3], dtype=np.float64), "sx": float(ps[1] if len(ps) > 1 else ps[0]), "sy": float(ps[0])}


def read_dicoms(path: str, primary: bool = True, original: bool = True) -> List[Volume]:
    """utils.read_dicoms (utils.py:132-230): every series under `path` as one Volume, in series-UID order."""
    allfnames = []
    for d, _, fnames in os.walk(path):
        allfnames += [os.path.join(d, f) for f in fnames]
    infos = []  # (study, series, fname, ipp)
    seen = set()
    for fname in allfnames:
        if os.path.splitext(os.path.basename(fname))[0] == "DICOMDIR":
            continue
        try:
            with open(fname, "rb") as f:
                buf = f.read()
            tags, _, _ = parse_dicom(buf, stop_before_pixels=True)
            image_type = _text(tags, (0x0008, 0x0008))
            if image_type is None:  # utils.py:153: files without ImageType are skipped
                continue
            kinds = image_type.split("\\")
            if (primary and "PRIMARY" not in kinds) or (original and "ORIGINAL" not in kinds) or "LOCALIZER" in kinds:
                continue
            study, series = _text(tags, (0x0020, 0x000D)), _text(tags, (0x0020, 0x000E))
            ipp = tuple(_floats(tags, (0x0020, 0x0032)) or ())
            if study is None or series is None or len(ipp) != 3:
                raise DicomError("missing StudyInstanceUID / SeriesInstanceUID / ImagePositionPatient")
            key = (study, series, ipp)
            if key not in seen:  # utils.py:183-185: duplicates under different names
                seen.add(key)
                infos.append((study, series, fname, ipp))
        except Exception as ex:  # utils.py:187-189
            logger.warning(f"Doesn't seem to be DICOM, will be skipped: {fname} ({ex})")
    series_ids = sorted({i[1] for i in infos})
    logger.info(f"There {'is' if len(series_ids) == 1 else 'are'} {len(series_ids)} volume{'' if len(series_ids) == 1 else 's'} in the study")
    vols = []
    for sid in series_ids:
        items = [i for i in infos if i[1] == sid]
        order = np.argsort(np.asarray([i[3][2] for i in items]), kind="stable")  # utils.py:208-211: by z position
        files = [items[k][2] for k in order]
        try:
            vols.append(_read_series(files))
        except DicomError as ex:
            # e.g. a JPEG-lossless series: the header scan above accepts it, the pixel decoder is the reference's (ITK/GDCM),
            # on the sorted file list -- as utils.py:213-220 reads every series
            logger.info(f"built-in DICOM reader declined the series ({ex}); trying SimpleITK")
            import SimpleITK as sitk

            reader = sitk.ImageSeriesReader()
            reader.SetFileNames(files)
            vols.append(_sitk_to_volume(reader.Execute()))
    return vols


def _read_series(files: List[str]) -> Volume:
    slices, geos, first_tags = [], [], None
    for fname in files:
        with open(fname, "rb") as f:
            tags, pixels, _ = parse_dicom(f.read())
        if pixels is None:
            raise DicomError(f"{fname}: no pixel data")
        px = _dicom_pixels(tags, pixels, fname)
        if px.shape[0] != 1:
            raise DicomError(f"{fname}: multi-frame file inside a series")
        slices.append(px[0])
        geos.append(_slice_geometry(tags))
        first_tags = first_tags or tags
    if any(s.shape != slices[0].shape for s in slices):
        raise DicomError("slices of one series differ in size")
    arr = np.stack(slices).astype(np.result_type(*[s.dtype for s in slices]))
    g0 = geos[0]
    normal = np.cross(g0["row_dir"], g0["col_dir"])
    zdir, zsp = normal, 1.0
    if len(geos) > 1:  # itk::ImageSeriesReader: third axis from the first to the last slice position
        step = (geos[-1]["ipp"] - g0["ipp"]) / (len(geos) - 1)
        dist = float(np.linalg.norm(step))
        if dist > 0:
            zdir, zsp = step / dist, dist
    direction = np.stack([g0["row_dir"], g0["col_dir"], zdir], axis=1)
    return Volume(arr, (g0["sx"], g0["sy"], zsp), g0["ipp"], direction, _meta_strings(first_tags))


# ------------------------------------------------------------------------------------------------
# DICOM writer (__main__.py:119-144 through sitk.ImageFileWriter / GDCM in the reference)
# ------------------------------------------------------------------------------------------------
# VR of the elements this writer emits or carries over (utils.py:17-30's list + what __main__.py:136-141 sets)
_WRITE_VR = {
    (0x0008, 0x0008): b"CS", (0x0008, 0x0016): b"UI", (0x0008, 0x0018): b"UI", (0x0008, 0x0020): b"DA", (0x0008, 0x0030): b"TM",
    (0x0008, 0x0050): b"SH", (0x0008, 0x0060): b"CS", (0x0008, 0x0064): b"CS", (0x0008, 0x0090): b"PN", (0x0008, 0x1030): b"LO",
    (0x0008, 0x103E): b"LO", (0x0010, 0x0010): b"PN", (0x0010, 0x0020): b"LO", (0x0010, 0x0030): b"DA", (0x0010, 0x0040): b"CS",
    (0x0018, 0x0050): b"DS", (0x0018, 0x0088): b"DS", (0x0018, 0x5100): b"CS", (0x0020, 0x000D): b"UI", (0x0020, 0x000E): b"UI",
    (0x0020, 0x0010): b"SH", (0x0020, 0x0011): b"IS", (0x0020, 0x0013): b"IS", (0x0020, 0x0032): b"DS", (0x0020, 0x0037): b"DS",
    (0x0028, 0x0002): b"US", (0x0028, 0x0004): b"CS", (0x0028, 0x0008): b"IS", (0x0028, 0x0010): b"US", (0x0028, 0x0011): b"US",
    (0x0028, 0x0030): b"DS", (0x0028, 0x0100): b"US", (0x0028, 0x0101): b"US", (0x0028, 0x0102): b"US", (0x0028, 0x0103): b"US",
    (0x0028, 0x1050): b"DS", (0x0028, 0x1051): b"DS", (0x0028, 0x1052): b"DS", (0x0028, 0x1053): b"DS",
}


def _new_uid(*parts) -> str:
    """A DICOM UID under the UUID-derived root 2.25 (PS3.5 B.2), deterministic in `parts`."""
    import hashlib

    return "2.25." + str(int.from_bytes(hashlib.sha256("|".join(str(x) for x in parts).encode()).digest()[:16], "big"))


def _ds(x: float) -> str:
    s = repr(float(x))
    return s if len(s) <= 16 else f"{x:.10g}"


def _encode_element(g: int, e: int, vr: bytes, value: bytes) -> bytes:
    if len(value) & 1:  # even length: UIDs are padded with NUL, text with a space
        value += b"\0" if vr == b"UI" else (b" " if vr in _TEXT_VR else b"\0")
    if vr in _LONG_VR:
        return struct.pack("<HH2sHI", g, e, vr, 0, len(value)) + value
    return struct.pack("<HH2sH", g, e, vr, len(value)) + value


def write_dicom(path: str, vol: Volume, keep_meta: Optional[Dict[str, str]] = None) -> None:
    """One explicit-VR little-endian file holding the whole label volume as a multi-frame image (what ITK's GDCM writer
    produces for a 3-D image under a single file name, __main__.py:122-144): 8-bit unsigned frames in file order, the
    geometry of `vol` (ImagePositionPatient of frame 0, ImageOrientationPatient, PixelSpacing, SpacingBetweenSlices), and
    `keep_meta` -- the carried-over study/patient tags plus Series Description and window (keys "gggg|eeee").  The Study
    Instance UID is the input's when it was kept (SetKeepOriginalImageUID), otherwise new; series / instance UIDs are new.
    Integer volumes up to 16 bits are written at their width; the reference writes uint8 label volumes."""
    a = np.ascontiguousarray(vol.array)
    if a.dtype == np.uint8 or a.dtype == np.bool_:
        a, bits, signed = a.astype(np.uint8), 8, 0
    elif a.dtype.kind in "iu" and a.dtype.itemsize <= 2:
        a, bits, signed = a.astype("<i2" if a.dtype.kind == "i" else "<u2"), 16, int(a.dtype.kind == "i")
    else:
        raise DicomError(f"write_dicom: {a.dtype} volumes are not supported (label volumes are uint8)")
    n, rows, cols = a.shape
    # A multi-frame file carries the slice direction only implicitly (frames advance along +cross(row, column) by
    # SpacingBetweenSlices).  A volume whose third direction column points the other way (a left-handed NIfTI / MetaImage input)
    # is therefore written with its frames in reverse order from the position of its last slice: the same voxels at the same
    # physical positions (the reader returns the right-handed form).
    d0 = np.asarray(vol.direction, dtype=np.float64).reshape(3, 3)
    nrm = np.cross(d0[:, 0], d0[:, 1])
    origin = np.asarray(vol.origin, dtype=np.float64)
    spacing = tuple(float(v) for v in vol.spacing)
    cosz = float(np.dot(nrm, d0[:, 2])) if n > 1 else 1.0
    if cosz < 0:
        origin = origin + d0[:, 2] * spacing[2] * (n - 1)
        a = np.ascontiguousarray(a[::-1])
        cosz = -cosz
        d0 = np.column_stack([d0[:, 0], d0[:, 1], -d0[:, 2]])  # (the frames now advance the other way)
    if cosz < 0.5:
        #  a slice axis that lies nearer to the image plane than to its normal: the projected spacing tends to zero and the
        # file would be geometrically meaningless -- no series a scanner produces looks like this; refuse instead of writing it
        raise ValueError("write_dicom: the slice axis is %.1f degrees off the image plane's normal -- a multi-frame file cannot express that geometry"
                         % np.degrees(np.arccos(max(0.0, cosz))))
    tilted = cosz < 0.999
    if tilted:
        # a gantry-tilted or sheared series (`read_dicom_series` takes the slice direction from the first-to-last slice position):
        # the file format cannot say so.  The reference's SimpleITK writer accepts such images, and this runs AFTER the whole
        # inference: the frames are written along the in-plane normal with the slice distance projected onto it,
        # with a warning, instead of raising and losing the result.
        import warnings as _w

        _w.warn("write_dicom: the slice axis is not perpendicular to the image plane (%.1f degrees off); a multi-frame file cannot express "
                "that -- writing the frames along the in-plane normal with the projected spacing" % np.degrees(np.arccos(min(1.0, cosz))), RuntimeWarning)
        spacing = (spacing[0], spacing[1], spacing[2] * cosz)
    if tilted or n == 1:
        d0 = np.column_stack([d0[:, 0], d0[:, 1], nrm])  # (an essentially perpendicular series keeps the direction it came with)
    vol = Volume(a, spacing, tuple(float(v) for v in origin), d0, getattr(vol, "meta", None))
    meta = {k.lower(): v for k, v in (keep_meta or {}).items()}
    study_uid = (meta.get("0020|000d") or "").strip("\0 ") or _new_uid("study", vol.origin, a.shape)
    series_uid = _new_uid("series", study_uid, os.path.abspath(path))
    sop_uid = _new_uid("instance", series_uid)
    sop_class = "1.2.840.10008.5.1.4.1.1.7.2" if bits == 8 else "1.2.840.10008.5.1.4.1.1.7.3"  # multi-frame grayscale byte / word SC
    d = np.asarray(vol.direction, dtype=np.float64).reshape(3, 3)
    el: Dict[Tuple[int, int], bytes] = {
        (0x0008, 0x0008): b"DERIVED\\SECONDARY", (0x0008, 0x0016): sop_class.encode(), (0x0008, 0x0018): sop_uid.encode(),
        (0x0008, 0x0060): b"OT", (0x0008, 0x0064): b"WSD", (0x0020, 0x000D): study_uid.encode(), (0x0020, 0x000E): series_uid.encode(),
        (0x0020, 0x0011): b"1", (0x0020, 0x0013): b"1",
        (0x0020, 0x0032): "\\".join(_ds(v) for v in vol.origin).encode(),
        (0x0020, 0x0037): "\\".join(_ds(v) for v in list(d[:, 0]) + list(d[:, 1])).encode(),
        (0x0018, 0x0088): _ds(vol.spacing[2]).encode(), (0x0018, 0x0050): _ds(vol.spacing[2]).encode(),
        (0x0028, 0x0002): struct.pack("<H", 1), (0x0028, 0x0004): b"MONOCHROME2", (0x0028, 0x0008): str(n).encode(),
        (0x0028, 0x0010): struct.pack("<H", rows), (0x0028, 0x0011): struct.pack("<H", cols),
        (0x0028, 0x0030): (_ds(vol.spacing[1]) + "\\" + _ds(vol.spacing[0])).encode(),  # row spacing \ column spacing
        (0x0028, 0x0100): struct.pack("<H", bits), (0x0028, 0x0101): struct.pack("<H", bits), (0x0028, 0x0102): struct.pack("<H", bits - 1),
        (0x0028, 0x0103): struct.pack("<H", signed), (0x0028, 0x1052): b"0", (0x0028, 0x1053): b"1",
    }
    for key, val in meta.items():  # carried-over / caller-set tags win over the defaults above, except the image description
        try:
            g, e = (int(x, 16) for x in key.split("|"))
        except ValueError:
            continue
        if (g, e) in _WRITE_VR and _WRITE_VR[(g, e)] in _TEXT_VR and (g, e) not in ((0x0008, 0x0016), (0x0008, 0x0018), (0x0020, 0x000E)):
            el[(g, e)] = str(val).rstrip("\0").encode("latin-1", "replace")
    body = b"".join(_encode_element(g, e, _WRITE_VR[(g, e)], v) for (g, e), v in sorted(el.items()))
    pix = a.tobytes()
    body += struct.pack("<HH2sHI", 0x7FE0, 0x0010, b"OB" if bits == 8 else b"OW", 0, len(pix) + (len(pix) & 1)) + pix + (b"\0" if len(pix) & 1 else b"")
    fm = b"".join([
        _encode_element(0x0002, 0x0001, b"OB", b"\x00\x01"), _encode_element(0x0002, 0x0002, b"UI", sop_class.encode()),
        _encode_element(0x0002, 0x0003, b"UI", sop_uid.encode()), _encode_element(0x0002, 0x0010, b"UI", b"1.2.840.10008.1.2.1"),
        _encode_element(0x0002, 0x0012, b"UI", _new_uid("lungmask_amd implementation").encode()),
    ])
    with open(path, "wb") as f:
        f.write(b"\0" * 128 + b"DICM" + _encode_element(0x0002, 0x0000, b"UL", struct.pack("<I", len(fm))) + fm + body)


# ------------------------------------------------------------------------------------------------
# front door (utils.load_input_image, utils.py:233-269; writer of __main__.py:119-144)
# ------------------------------------------------------------------------------------------------
def _kind(path: str) -> str:
    p = path.lower()
    for ext, kind in ((".nii.gz", "nifti"), (".nii", "nifti"), (".hdr", "nifti"), (".mha", "meta"), (".mhd", "meta"), (".npy", "npy"), (".npz", "npz"),
                      (".dcm", "dicom"), (".ima", "dicom")):
        if p.endswith(ext):
            return kind
    return "other"


def _sitk_to_volume(img) -> Volume:
    import SimpleITK as sitk

    meta = {k: img.GetMetaData(k) for k in img.GetMetaDataKeys()}
    return Volume(sitk.GetArrayFromImage(img), img.GetSpacing(), img.GetOrigin(), np.asarray(img.GetDirection()).reshape(3, 3), meta)


def load_input_image(path: str) -> Volume:
    """A file is read by extension; a folder is searched for DICOM series and the largest one is taken."""
    if os.path.isfile(path):
        logger.info(f"Read input: {path}")
        kind = _kind(path)
        if kind == "npy":
            return Volume(np.load(path))
        if kind == "npz":
            z = np.load(path)
            return Volume(z[z.files[0]])
        if kind == "nifti":
            return read_nifti(path)
        if kind == "meta":
            return read_metaimage(path)
        try:
            with open(path, "rb") as f:
                is_dicom = kind == "dicom" or f.read(132)[128:] == b"DICM"
            if is_dicom:
                return read_dicom_file(path)
        except DicomError as ex:
            logger.info(f"built-in DICOM reader declined ({ex}); trying SimpleITK")
        import SimpleITK as sitk  # everything else (NRRD, compressed DICOM, ...) as in the reference

        return _sitk_to_volume(sitk.ReadImage(path))
    logger.info(f"Looking for dicoms in {path}")
    vols = read_dicoms(path, original=False, primary=False)
    if len(vols) < 1:
        sys.exit("No dicoms found!")  # utils.py:260-261
    if len(vols) > 1:
        logger.warning("There are more than one volume in the path, will take the largest one")
    return vols[int(np.argmax([np.prod(v.GetSize()) for v in vols]))]


def save_image(path: str, vol: Volume, keep_meta: Dict[str, str] = None) -> None:
    kind = _kind(path)
    if kind == "npy":
        np.save(path, vol.array)
    elif kind == "npz":
        np.savez_compressed(path, mask=vol.array)
    elif kind == "nifti" and not path.lower().endswith(".hdr"):
        write_nifti(path, vol)
    elif kind == "meta":
        write_metaimage(path, vol)
    elif kind == "dicom":
        write_dicom(path, vol, keep_meta)
    else:  # NRRD, ...: the reference's writer
        import SimpleITK as sitk

        out = sitk.GetImageFromArray(vol.array)
        out.SetSpacing(vol.spacing)
        out.SetOrigin(vol.origin)
        out.SetDirection(tuple(float(v) for v in vol.direction.reshape(-1)))
        writer = sitk.ImageFileWriter()
        writer.SetFileName(path)
        if keep_meta is not None:
            writer.SetKeepOriginalImageUID(True)
            for k, v in keep_meta.items():
                out.SetMetaData(k, v)
        writer.Execute(out)
