"""CPU tests of the host-side TIFF codec (instageo_amd/tiff.py) against the data files the reference's
own tests hold (tests/golden/tiff/ = /root/reference/tests/data/chip_178_022{,.mask}.tif, GDAL-written) and against Pillow."""
import os

import numpy as np
import pytest

from instageo_amd import tiff

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "tiff")


def test_reads_the_reference_chip_fixture():
    """/root/reference/tests/data_tests/test_create_chips.py:81-82 uses this 18-band (T=3 x 6) int16 HLS chip and its crop mask."""
    chip, prof = tiff.read(os.path.join(GOLD, "chip_178_022.tif"))
    assert chip.shape == (18, 32, 32) and chip.dtype == np.int16 and prof["count"] == 18
    assert int(chip.min()) == -9999 and int(chip.max()) == 6062 and abs(float(chip.astype(np.float64).mean()) + 2259.9460177951387) < 1e-9
    mask, _ = tiff.read(os.path.join(GOLD, "chip_178_022.mask.tif"))
    assert mask.shape == (1, 32, 32) and mask.dtype == np.uint8 and int(mask.min()) == 1 and int(mask.max()) == 13
    # NODATA pixels of the chip are NODATA in every band of a time step (HLS masking is per pixel)
    nd = chip == -9999
    assert nd.any() and all(np.array_equal(nd[6 * t], nd[6 * t + c]) for t in range(3) for c in range(6))


@pytest.mark.parametrize("dtype", ["int16", "uint8", "int8", "float32", "uint16", "int32"])
@pytest.mark.parametrize("compress", [None, "deflate"])
def test_roundtrip_with_geo_tags(tmp_path, dtype, compress):
    rng = np.random.default_rng(0)
    prof = {"tags": {33550: (12, (30.0, 30.0, 0.0)), 33922: (12, (0.0, 0.0, 0.0, 5e5, 4e6, 0.0)), 34737: (2, "WGS 84 / UTM zone 33N|"),
                     34735: (3, (1, 1, 0, 1, 1024, 0, 1, 1))}, "nodata": -9999}
    for shape in ((3, 37, 53), (1, 224, 224), (18, 64, 64), (224, 224)):
        a = (rng.standard_normal(shape) * 1000).astype(dtype)
        p = str(tmp_path / "a.tif")
        tiff.write(p, a, prof, compress=compress)
        b, pr = tiff.read(p)
        assert b.dtype == a.dtype and np.array_equal(a.reshape(b.shape), b)
        assert pr["tags"][33550][1] == (30.0, 30.0, 0.0) and pr["nodata"] == -9999.0 and pr["tags"][34737][1].startswith("WGS")
        assert pr["width"] == shape[-1] and pr["height"] == shape[-2]


def test_interoperates_with_pillow(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(1)
    a = (rng.standard_normal((50, 70)) * 1000).astype("int16")
    p = str(tmp_path / "ours.tif")
    tiff.write(p, a)
    assert np.array_equal(np.array(Image.open(p)), a)  # Pillow reads what we write
    f = rng.standard_normal((50, 70)).astype("float32")
    for kw in ({}, {"compression": "tiff_adobe_deflate"}):
        q = str(tmp_path / "pil.tif")
        Image.fromarray(f).save(q, **kw)
        assert np.array_equal(tiff.read(q)[0][0], f)  # we read what Pillow writes (strips, none / deflate)
    rgb = rng.integers(0, 255, (40, 30, 3)).astype("uint8")
    q = str(tmp_path / "rgb.tif")
    Image.fromarray(rgb).save(q)
    assert np.array_equal(tiff.read(q)[0], rgb.transpose(2, 0, 1))  # pixel-interleaved (contig) planes


def test_rejects_what_it_does_not_support(tmp_path):
    p = str(tmp_path / "x.tif")
    open(p, "wb").write(b"II+\x00" + b"\x00" * 12)
    with pytest.raises(tiff.TiffError):
        tiff.read(p)  # BigTIFF
    open(p, "wb").write(b"not a tiff")
    with pytest.raises(tiff.TiffError):
        tiff.read(p)
    with pytest.raises(tiff.TiffError):
        tiff.write(p, np.zeros((2, 2), dtype=np.complex64))
    with pytest.raises(tiff.TiffError):
        tiff.write(p, np.zeros((2, 2), dtype=np.int16), compress="lzw")


@pytest.mark.parametrize("mode,dtype", [("L", np.uint8), ("RGB", np.uint8), ("I;16", np.uint16), ("F", np.float32)])
def test_lzw_read_matches_pillow(tmp_path, mode, dtype):
    """GDAL's COMPRESS=LZW (common for HLS derivatives): files written by Pillow/libtiff with LZW decode to the source array;
    smooth + noisy content so that the code width grows through 9..12 bits and the table is cleared several times."""
    from PIL import Image

    rng = np.random.default_rng(5)
    H, W = 301, 257
    base = (np.add.outer(np.arange(H), np.arange(W)) % 251).astype(np.float64)
    if mode == "RGB":
        a = np.stack([base, base[::-1], rng.integers(0, 255, (H, W))], -1).astype(dtype)
    elif mode == "F":
        a = (base * 0.25 + rng.normal(size=(H, W))).astype(dtype)
    else:
        a = (base * (200 if dtype == np.uint16 else 1) + rng.integers(0, 3, (H, W))).astype(dtype)
    path = str(tmp_path / f"lzw_{mode.replace(';', '')}.tif")
    Image.fromarray(a, mode=mode).save(path, compression="tiff_lzw")
    got, prof = tiff.read(path)
    want = a.transpose(2, 0, 1) if a.ndim == 3 else a[None]
    assert got.shape == want.shape and got.dtype == want.dtype
    assert np.array_equal(got, want)
    assert tiff.read_profile(path) == prof


def test_read_profile_is_header_only_and_band_subset(tmp_path):
    """read_profile parses the IFD without inflating strips (a truncated pixel section still yields the profile); ``bands`` decodes
    only the requested planes of a band-interleaved file."""
    a = np.arange(6 * 40 * 50, dtype=np.int16).reshape(6, 40, 50)
    path = str(tmp_path / "chip.tif")
    tiff.write(path, a, compress="deflate")
    full, prof = tiff.read(path)
    assert np.array_equal(full, a)
    sub, prof2 = tiff.read(path, bands=[0, 5, 2])
    assert np.array_equal(sub, a[[0, 5, 2]]) and prof2 == prof and prof["count"] == 6
    assert tiff.read_profile(path) == prof
    with pytest.raises(tiff.TiffError):
        tiff.read(path, bands=[6])


def test_read_profile_falls_back_when_tag_values_lie_beyond_the_header_read(tmp_path):
    """read_profile reads the first 64 KiB; an out-of-line ASCII value (GDAL_NODATA) stored beyond them must trigger the
    whole-file fallback instead of coming back truncated; a value beyond the end of the file is an error, not an empty string."""
    import struct

    W = H = 4
    pix = bytes(range(W * H))
    nodata = b"-9999\x00"
    nd_off = 70000  # beyond the 64 KiB header read
    entries = [(256, 3, 1, W), (257, 3, 1, H), (258, 3, 1, 8), (259, 3, 1, 1), (262, 3, 1, 1), (273, 4, 1, 200), (277, 3, 1, 1), (278, 3, 1, H),
               (279, 4, 1, W * H), (339, 3, 1, 1), (42113, 2, len(nodata), nd_off)]
    ifd = struct.pack("<H", len(entries))
    for tag, typ, cnt, val in entries:
        ifd += struct.pack("<HHI", tag, typ, cnt) + (struct.pack("<HH", val, 0) if typ == 3 else struct.pack("<I", val))
    ifd += struct.pack("<I", 0)
    buf = bytearray(nd_off + len(nodata))
    buf[0:8] = b"II" + struct.pack("<HI", 42, 8)
    buf[8 : 8 + len(ifd)] = ifd
    buf[200 : 200 + len(pix)] = pix
    buf[nd_off:] = nodata
    path = str(tmp_path / "far_tag.tif")
    open(path, "wb").write(bytes(buf))
    prof = tiff.read_profile(path)
    assert prof["nodata"] == -9999.0 and prof["width"] == W and prof["height"] == H
    arr, prof2 = tiff.read(path)
    assert prof2 == prof and arr.reshape(-1).tolist() == list(pix)
    open(path, "wb").write(bytes(buf[: nd_off + 2]))  # the value now runs past the end of the file
    with pytest.raises(tiff.TiffError):
        tiff.read_profile(path)
