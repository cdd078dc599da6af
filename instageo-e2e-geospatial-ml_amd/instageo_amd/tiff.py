"""Minimal baseline-TIFF / GeoTIFF codec for the tile-inference I/O of the path.

The reference reads chips and writes ``prediction_*.tif`` through rasterio/GDAL (``instageo/model/dataloader.py:672-704``,
``infer_utils.py:37-54,103-128``), which are absent here.  This module is the host-side stand-in: it reads what GDAL writes by
default for HLS chips (classic little/big-endian TIFF, strips or tiles, compression none / deflate / LZW-free, predictor 1 or 2,
band-interleaved "separate" or pixel-interleaved "contig" planes, 8/16/32-bit integer and 32/64-bit float samples) and writes
strip-organised files (uncompressed or deflate), carrying the source file's georeferencing tags over verbatim so a prediction
raster opens in GIS tools at the chip's location -- the role of ``profile`` in the reference's ``save_prediction``.

Not a general TIFF library: no BigTIFF, no JPEG/LZW codecs, no sub-IFDs.  Unsupported features raise ``TiffError``.
"""
from __future__ import annotations

import struct
import zlib
from typing import Any, Dict, List, Optional, Tuple

import numpy as np


class TiffError(ValueError):
    pass


# tag ids
_W, _H, _BPS, _COMP, _PHOTO, _STRIP_OFF, _SPP, _RPS, _STRIP_CNT = 256, 257, 258, 259, 262, 273, 277, 278, 279
_PLANAR, _PRED, _TILE_W, _TILE_H, _TILE_OFF, _TILE_CNT, _EXTRA, _FMT = 284, 317, 322, 323, 324, 325, 338, 339
# georeferencing / GDAL tags copied from the source profile: ModelPixelScale, ModelTiepoint, ModelTransformation,
# GeoKeyDirectory, GeoDoubleParams, GeoAsciiParams, GDAL_METADATA, GDAL_NODATA
GEO_TAGS = (33550, 33922, 34264, 34735, 34736, 34737, 42112, 42113)

_TYPES = {1: ("B", 1), 2: ("c", 1), 3: ("H", 2), 4: ("I", 4), 5: ("II", 8), 6: ("b", 1), 7: ("B", 1), 8: ("h", 2), 9: ("i", 4),
          10: ("ii", 8), 11: ("f", 4), 12: ("d", 8), 16: ("Q", 8)}  # fmt: skip
_DTYPES = {(1, 8): "u1", (1, 16): "u2", (1, 32): "u4", (2, 8): "i1", (2, 16): "i2", (2, 32): "i4", (3, 32): "f4", (3, 64): "f8"}
_FMT_OF = {"u": 1, "i": 2, "f": 3}


def _read_ifd(buf: bytes, bo: str, off: int) -> Dict[int, Tuple[int, Any]]:
    (n,) = struct.unpack_from(bo + "H", buf, off)
    tags: Dict[int, Tuple[int, Any]] = {}
    for i in range(n):
        tag, typ, cnt, val = struct.unpack_from(bo + "HHI4s", buf, off + 2 + 12 * i)
        if typ not in _TYPES:
            continue
        fmt, size = _TYPES[typ]
        nbytes = size * cnt
        data = val[:nbytes] if nbytes <= 4 else buf[struct.unpack(bo + "I", val)[0] :][:nbytes]
        if typ == 2:
            tags[tag] = (typ, data.rstrip(b"\x00").decode("latin-1"))
        elif typ in (5, 10):
            v = struct.unpack(bo + fmt[0] * (2 * cnt), data)
            tags[tag] = (typ, tuple((v[2 * j], v[2 * j + 1]) for j in range(cnt)))
        else:
            tags[tag] = (typ, struct.unpack(bo + fmt * cnt, data))
    return tags


def _unpredict(a: np.ndarray) -> np.ndarray:
    """Undo horizontal differencing (predictor 2) along the last (x) axis of an integer array (rows, x, samples)."""
    return np.cumsum(a, axis=1, dtype=a.dtype)


def read(path: str) -> Tuple[np.ndarray, Dict[str, Any]]:
    """-> (array (bands, H, W), profile).  ``profile`` holds width/height/count/dtype, ``nodata`` (GDAL_NODATA) and the raw
    georeferencing tags under ``"tags"`` ({tag: (tiff_type, values)}), ready for :func:`write`."""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:2] == b"II":
        bo = "<"
    elif buf[:2] == b"MM":
        bo = ">"
    else:
        raise TiffError(f"{path}: not a TIFF file")
    (magic,) = struct.unpack_from(bo + "H", buf, 2)
    if magic == 43:
        raise TiffError(f"{path}: BigTIFF is not supported")
    if magic != 42:
        raise TiffError(f"{path}: bad TIFF magic {magic}")
    (ifd,) = struct.unpack_from(bo + "I", buf, 4)
    t = _read_ifd(buf, bo, ifd)

    def one(tag, default=None):
        return t[tag][1][0] if tag in t else default

    W, H = one(_W), one(_H)
    spp = one(_SPP, 1)
    bps = t.get(_BPS, (3, (1,)))[1]
    if len(set(bps)) != 1:
        raise TiffError(f"{path}: mixed bits per sample {bps}")
    fmt = t.get(_FMT, (3, (1,)))[1][0]
    key = (fmt, bps[0])
    if key not in _DTYPES:
        raise TiffError(f"{path}: unsupported sample format {key}")
    dt = np.dtype(bo + _DTYPES[key])
    comp = one(_COMP, 1)
    if comp not in (1, 8, 32946):
        raise TiffError(f"{path}: unsupported compression {comp} (supported: none, deflate)")
    pred = one(_PRED, 1)
    if pred not in (1, 2) or (pred == 2 and dt.kind == "f"):
        raise TiffError(f"{path}: unsupported predictor {pred}")
    planar = one(_PLANAR, 1)
    planes = spp if planar == 2 else 1       # separately stored sample planes
    sp = 1 if planar == 2 else spp           # samples per pixel inside one chunk
    out = np.empty((spp, H, W), dtype=dt.newbyteorder("="))

    def chunk(off: int, cnt: int, rows: int, cols: int) -> np.ndarray:
        raw = buf[off : off + cnt]
        if comp != 1:
            raw = zlib.decompress(raw)
        a = np.frombuffer(raw, dtype=dt, count=rows * cols * sp).reshape(rows, cols, sp)
        if pred == 2:
            a = _unpredict(a.astype(dt.newbyteorder("=")))
        return a

    if _TILE_OFF in t:
        tw, th = one(_TILE_W), one(_TILE_H)
        offs, cnts = t[_TILE_OFF][1], t[_TILE_CNT][1]
        nx, ny = -(-W // tw), -(-H // th)
        for p in range(planes):
            for ty in range(ny):
                for tx in range(nx):
                    i = (p * ny + ty) * nx + tx
                    a = chunk(offs[i], cnts[i], th, tw)
                    y0, x0 = ty * th, tx * tw
                    hh, ww = min(th, H - y0), min(tw, W - x0)
                    blk = a[:hh, :ww, :]
                    if planar == 2:
                        out[p, y0 : y0 + hh, x0 : x0 + ww] = blk[:, :, 0]
                    else:
                        out[:, y0 : y0 + hh, x0 : x0 + ww] = blk.transpose(2, 0, 1)
    else:
        rps = min(one(_RPS, H), H)
        offs, cnts = t[_STRIP_OFF][1], t[_STRIP_CNT][1]
        ns = -(-H // rps)
        for p in range(planes):
            for s in range(ns):
                i = p * ns + s
                y0 = s * rps
                hh = min(rps, H - y0)
                a = chunk(offs[i], cnts[i], hh, W)
                if planar == 2:
                    out[p, y0 : y0 + hh] = a[:, :, 0]
                else:
                    out[:, y0 : y0 + hh] = a.transpose(2, 0, 1)
    nodata: Optional[float] = None
    if 42113 in t:
        try:
            nodata = float(t[42113][1])
        except ValueError:
            nodata = None
    profile = {"driver": "GTiff", "width": W, "height": H, "count": spp, "dtype": out.dtype.name, "nodata": nodata,
               "tags": {k: t[k] for k in GEO_TAGS if k in t}}  # fmt: skip
    return out, profile


def read_profile(path: str) -> Dict[str, Any]:
    """The profile only (the reference opens the source chip just for ``src.profile``, infer_utils.py:103-113)."""
    return read(path)[1]


def write(path: str, array: np.ndarray, profile: Optional[Dict[str, Any]] = None, compress: Optional[str] = None) -> None:
    """Write ``array`` ((H, W) or (bands, H, W)) as a little-endian strip TIFF, band-interleaved (PlanarConfiguration 2, what
    GDAL calls INTERLEAVE=BAND); georeferencing tags of ``profile["tags"]`` are copied verbatim.  ``compress``: None | "deflate"."""
    a = np.asarray(array)
    if a.ndim == 2:
        a = a[None]
    if a.ndim != 3:
        raise TiffError("array must be (H, W) or (bands, H, W)")
    if a.dtype == np.bool_:
        a = a.astype(np.uint8)
    if a.dtype.kind not in _FMT_OF or (_FMT_OF[a.dtype.kind], a.dtype.itemsize * 8) not in _DTYPES:
        raise TiffError(f"unsupported dtype {a.dtype}")
    if compress not in (None, "none", "deflate"):
        raise TiffError(f"unsupported compression {compress!r}")
    a = np.ascontiguousarray(a.astype(a.dtype.newbyteorder("<"), copy=False))
    bands, H, W = a.shape
    rps = max(1, min(H, (1 << 16) // max(1, W * a.dtype.itemsize)))  # ~64 KiB strips
    strips: List[bytes] = []
    for b in range(bands):
        for y0 in range(0, H, rps):
            raw = a[b, y0 : y0 + rps].tobytes()
            strips.append(zlib.compress(raw, 6) if compress == "deflate" else raw)
    entries: List[Tuple[int, int, int, bytes]] = []  # (tag, type, count, payload)

    def ent(tag: int, typ: int, values) -> None:
        if typ == 2:
            payload = values.encode("latin-1") + b"\x00"
            cnt = len(payload)
        elif typ in (5, 10):
            flat = [x for pair in values for x in pair]
            payload = struct.pack("<" + _TYPES[typ][0][0] * len(flat), *flat)
            cnt = len(values)
        else:
            payload = struct.pack("<" + _TYPES[typ][0] * len(values), *values)
            cnt = len(values)
        entries.append((tag, typ, cnt, payload))

    ent(_W, 4, (W,)), ent(_H, 4, (H,)), ent(_BPS, 3, (a.dtype.itemsize * 8,) * bands)
    ent(_COMP, 3, (8 if compress == "deflate" else 1,)), ent(_PHOTO, 3, (1,))
    ent(_SPP, 3, (bands,)), ent(_RPS, 4, (rps,)), ent(_PLANAR, 3, (2 if bands > 1 else 1,))
    if bands > 1:
        ent(_EXTRA, 3, (0,) * (bands - 1))
    ent(_FMT, 3, (_FMT_OF[a.dtype.kind],) * bands)
    tags = dict((profile or {}).get("tags", {}))
    nd = (profile or {}).get("nodata")
    if nd is not None and 42113 not in tags:
        tags[42113] = (2, repr(float(nd)) if float(nd) != int(nd) else str(int(nd)))
    for tag, (typ, values) in tags.items():
        ent(int(tag), int(typ), values)
    # layout: header | strips | out-of-line values | IFD
    pos = 8
    offs = []
    for sdata in strips:
        offs.append(pos)
        pos += len(sdata) + (len(sdata) & 1)
    ent(_STRIP_OFF, 4, tuple(offs)), ent(_STRIP_CNT, 4, tuple(len(sd) for sd in strips))
    entries.sort(key=lambda e: e[0])
    extra = bytearray()
    recs = []
    for tag, typ, cnt, payload in entries:
        if len(payload) <= 4:
            recs.append(struct.pack("<HHI4s", tag, typ, cnt, payload.ljust(4, b"\x00")))
        else:
            recs.append(struct.pack("<HHII", tag, typ, cnt, pos + len(extra)))
            extra += payload
            if len(extra) & 1:
                extra += b"\x00"
    ifd_off = pos + len(extra)
    if ifd_off + 2 + 12 * len(recs) + 4 >= (1 << 32):
        raise TiffError("raster too large for classic TIFF")
    with open(path, "wb") as f:
        f.write(struct.pack("<2sHI", b"II", 42, ifd_off))
        for sdata in strips:
            f.write(sdata)
            if len(sdata) & 1:
                f.write(b"\x00")
        f.write(bytes(extra))
        f.write(struct.pack("<H", len(recs)))
        for r in recs:
            f.write(r)
        f.write(struct.pack("<I", 0))
