"""Minimal baseline-TIFF / GeoTIFF codec for the tile-inference I/O of the path.

The reference reads chips and writes ``prediction_*.tif`` through rasterio/GDAL (``instageo/model/dataloader.py:672-704``,
``infer_utils.py:37-54,103-128``), which are absent here.  This module is the host-side stand-in: it reads what GDAL writes by
default for HLS chips (classic little/big-endian TIFF, strips or tiles, compression none / deflate / LZW, predictor 1 or 2,
band-interleaved "separate" or pixel-interleaved "contig" planes, 8/16/32-bit integer and 32/64-bit float samples) and writes
strip-organised files (uncompressed or deflate), carrying the source file's georeferencing tags over verbatim so a prediction
raster opens in GIS tools at the chip's location -- the role of ``profile`` in the reference's ``save_prediction``.

Not a general TIFF library: no BigTIFF, no JPEG codec, no sub-IFDs; LZW (GDAL's COMPRESS=LZW, common for HLS derivatives) is
READ only, by a pure-Python decoder (a 256 x 256 x 18 int16 chip takes about a second).  Unsupported features raise ``TiffError``.
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
        if nbytes <= 4:
            data = val[:nbytes]
        else:
            voff = struct.unpack(bo + "I", val)[0]
            if voff + nbytes > len(buf):  # beyond what was read (header-only reads take the first 64 KiB): the caller re-reads the whole file
                raise struct.error(f"tag {tag}: {nbytes} bytes at offset {voff} lie beyond the {len(buf)} bytes read")
            data = buf[voff : voff + nbytes]
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


def _lzw_decode(data: bytes) -> bytes:
    """TIFF 6.0 LZW (compression 5): MSB-first codes of 9..12 bits, ClearCode 256, EndOfInformation 257, the code width grows one
    code EARLY (at 511 / 1023 / 2047 table entries), as libtiff / GDAL write it."""
    out = bytearray()
    table: List[bytes] = []
    base = [bytes((i,)) for i in range(256)] + [b"", b""]
    bitbuf = 0
    nbits = 0
    width = 9
    prev: Optional[bytes] = None
    pos, n = 0, len(data)
    table = list(base)
    while True:
        while nbits < width:
            if pos >= n:
                return bytes(out)
            bitbuf = (bitbuf << 8) | data[pos]
            pos += 1
            nbits += 8
        nbits -= width
        code = (bitbuf >> nbits) & ((1 << width) - 1)
        if code == 257:
            break
        if code == 256:
            table = list(base)
            width = 9
            prev = None
            continue
        if prev is None:
            entry = table[code]
        elif code < len(table):
            entry = table[code]
            table.append(prev + entry[:1])
        elif code == len(table):
            entry = prev + prev[:1]
            table.append(entry)
        else:
            raise TiffError(f"corrupt LZW stream (code {code}, table {len(table)})")
        out += entry
        prev = entry
        ln = len(table)
        if ln >= 511:
            width = 10 if ln < 1023 else 11 if ln < 2047 else 12
    return bytes(out)


class _Header:
    """Parsed first IFD of a classic TIFF (no pixel data touched)."""

    def __init__(self, path: str, buf: bytes):
        self.path, self.buf = path, buf
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
        t = self.t = _read_ifd(buf, bo, ifd)
        self.bo = bo
        self.W, self.H = self.one(_W), self.one(_H)
        self.spp = self.one(_SPP, 1)
        bps = t.get(_BPS, (3, (1,)))[1]
        if len(set(bps)) != 1:
            raise TiffError(f"{path}: mixed bits per sample {bps}")
        fmt = t.get(_FMT, (3, (1,)))[1][0]
        key = (fmt, bps[0])
        if key not in _DTYPES:
            raise TiffError(f"{path}: unsupported sample format {key}")
        self.dt = np.dtype(bo + _DTYPES[key])
        self.comp = self.one(_COMP, 1)
        if self.comp not in (1, 5, 8, 32946):
            raise TiffError(f"{path}: unsupported compression {self.comp} (supported: none, LZW, deflate)")
        self.pred = self.one(_PRED, 1)
        if self.pred not in (1, 2) or (self.pred == 2 and self.dt.kind == "f"):
            raise TiffError(f"{path}: unsupported predictor {self.pred}")
        self.planar = self.one(_PLANAR, 1)

    def one(self, tag, default=None):
        return self.t[tag][1][0] if tag in self.t else default

    def profile(self) -> Dict[str, Any]:
        t = self.t
        nodata: Optional[float] = None
        if 42113 in t:
            try:
                nodata = float(t[42113][1])
            except ValueError:
                nodata = None
        return {"driver": "GTiff", "width": self.W, "height": self.H, "count": self.spp, "dtype": self.dt.newbyteorder("=").name,
                "nodata": nodata, "tags": {k: t[k] for k in GEO_TAGS if k in t}}  # fmt: skip


def _header(path: str, whole: bool) -> _Header:
    with open(path, "rb") as f:
        if whole:
            try:
                return _Header(path, f.read())
            except (struct.error, IndexError) as e:
                raise TiffError(f"{path}: truncated or corrupt TIFF ({e})") from e
        # header only: the IFD and its out-of-line values normally sit in the first or the last kilobytes; fall back to the whole
        # file when an offset points outside what was read
        head = f.read(1 << 16)
        try:
            return _Header(path, head)
        except (struct.error, IndexError):
            f.seek(0)
            try:
                return _Header(path, f.read())
            except (struct.error, IndexError) as e:
                raise TiffError(f"{path}: truncated or corrupt TIFF ({e})") from e


def read(path: str, bands: Optional[List[int]] = None) -> Tuple[np.ndarray, Dict[str, Any]]:
    """-> (array (bands, H, W), profile).  ``profile`` holds width/height/count/dtype, ``nodata`` (GDAL_NODATA) and the raw
    georeferencing tags under ``"tags"`` ({tag: (tiff_type, values)}), ready for :func:`write`.  ``bands`` (0-based) decodes
    only those sample planes of a band-interleaved file (the reference's ``src.read(band)``); the profile still describes the file."""
    h = _header(path, True)
    buf, t, dt, comp, pred, planar = h.buf, h.t, h.dt, h.comp, h.pred, h.planar
    W, H, spp, one = h.W, h.H, h.spp, h.one
    sel = list(range(spp)) if bands is None else [int(b) for b in bands]
    if any(b < 0 or b >= spp for b in sel):
        raise TiffError(f"{path}: band index out of range (file has {spp} bands)")
    planes = spp if planar == 2 else 1       # separately stored sample planes
    sp = 1 if planar == 2 else spp           # samples per pixel inside one chunk
    out = np.empty((spp if planar != 2 else len(sel), H, W), dtype=dt.newbyteorder("="))

    def chunk(off: int, cnt: int, rows: int, cols: int) -> np.ndarray:
        raw = buf[off : off + cnt]
        if comp == 5:
            raw = _lzw_decode(raw)
        elif comp != 1:
            raw = zlib.decompress(raw)
        a = np.frombuffer(raw, dtype=dt, count=rows * cols * sp).reshape(rows, cols, sp)
        if pred == 2:
            a = _unpredict(a.astype(dt.newbyteorder("=")))
        return a

    plane_list = sel if planar == 2 else [0]
    if _TILE_OFF in t:
        tw, th = one(_TILE_W), one(_TILE_H)
        offs, cnts = t[_TILE_OFF][1], t[_TILE_CNT][1]
        nx, ny = -(-W // tw), -(-H // th)
        for oi, p in enumerate(plane_list):
            for ty in range(ny):
                for tx in range(nx):
                    i = (p * ny + ty) * nx + tx
                    a = chunk(offs[i], cnts[i], th, tw)
                    y0, x0 = ty * th, tx * tw
                    hh, ww = min(th, H - y0), min(tw, W - x0)
                    blk = a[:hh, :ww, :]
                    if planar == 2:
                        out[oi, y0 : y0 + hh, x0 : x0 + ww] = blk[:, :, 0]
                    else:
                        out[:, y0 : y0 + hh, x0 : x0 + ww] = blk.transpose(2, 0, 1)
    else:
        rps = min(one(_RPS, H), H)
        offs, cnts = t[_STRIP_OFF][1], t[_STRIP_CNT][1]
        ns = -(-H // rps)
        for oi, p in enumerate(plane_list):
            for s_ in range(ns):
                i = p * ns + s_
                y0 = s_ * rps
                hh = min(rps, H - y0)
                a = chunk(offs[i], cnts[i], hh, W)
                if planar == 2:
                    out[oi, y0 : y0 + hh] = a[:, :, 0]
                else:
                    out[:, y0 : y0 + hh] = a.transpose(2, 0, 1)
    if planar != 2 and bands is not None:
        out = out[sel]
    return out, h.profile()


def read_profile(path: str) -> Dict[str, Any]:
    """The profile only, from the header: no strip is inflated (the reference opens the source chip just for ``src.profile``,
    infer_utils.py:103-113, and ``get_valid_filepaths`` only to see that it opens)."""
    return _header(path, False).profile()


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
