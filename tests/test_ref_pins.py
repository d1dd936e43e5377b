"""Pins against the REAL reference code (oracle/_ref, built in place from /root/reference by
`make -C oracle ref`): the three self-contained third-party pieces the reference vendors.

  * rgb2spec.c  (src/jakob-and-hanika-2019/rgb2spec.c:60-134)  <- oracle + host JH fetch/eval + file format
  * lodepng.cpp (src/util/lodepng/)                            <- the host's own PNG decoder / encoder
  * spectrum_grid.h (src/meng-et-al.-2015/)                    <- oracle Meng restatement (test_meng_*)

CPU only.  Skipped when oracle/_ref is absent and cannot be built (no /root/reference).
"""
import ctypes as C
import os
import struct
import zlib

import numpy as np
import pytest

import oracle_lib as ol
import ref_lib
from simple_spectral_amd import _capi
from simple_spectral_amd.renderer import Scene

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_or_both_nan(a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    return bool(np.all((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))))


# ------------------------------------------------------------------------------------------------
# Jakob-Hanika: reference loader reads the file the host wrote; fetch/eval bit-identical
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def jh_setup(tmp_path_factory):
    lib = ref_lib.rgb2spec()
    if lib is None:
        pytest.skip("oracle/_ref/libref_rgb2spec.so not built (needs /root/reference)")
    path = str(tmp_path_factory.mktemp("jh") / "model.coeff")
    scene = Scene("cornell-srgb", texture="test-img.png", uplift="jh", jh_res=16, jh_coeff_path=path)
    res, scale, data = scene.jh_model()
    model = lib.rgb2spec_load(path.encode())
    assert bool(model), "the reference's rgb2spec_load rejected the file written by the host"
    orc = ol.Oracle("cornell-srgb", texture="test-img.png", jh=(res, scale, data))
    yield lib, model, orc, (res, scale, data)
    lib.rgb2spec_free(model)


def test_jh_file_written_by_host_loads_in_reference(jh_setup):
    lib, model, _, (res, scale, data) = jh_setup
    m = model.contents
    assert m.res == res                                                      # rgb2spec.c:28-33 header "SPEC", res
    assert np.array_equal(np.ctypeslib.as_array(m.scale, shape=(res,)), scale)
    assert np.array_equal(np.ctypeslib.as_array(m.data, shape=(data.size,)), data)


def test_jh_fetch_and_eval_bit_identical_to_reference(jh_setup):
    lib, model, orc, _ = jh_setup
    rng = np.random.default_rng(7)
    rgbs = [rng.random(3, dtype=np.float32) for _ in range(20000)]
    rgbs += [np.array(v, np.float32) for v in ([1, 1, 1], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0.5, 0.5, 0.5], [0.2, 0.2, 0.7],
                                               [0.7, 0.7, 0.2], [1, 1, 0.999], [1e-6, 1e-7, 1e-8], [0, 0, 0], [0.25, 0.25, 0.25])]
    rgbs += [np.round(rng.random(3) * 255).astype(np.float32) / np.float32(255) for _ in range(5000)]   # texel-like
    fp = C.POINTER(C.c_float)
    lams = np.concatenate([np.linspace(380, 780, 9, dtype=np.float32), rng.uniform(380, 780, 7).astype(np.float32)])
    n_nan = 0
    for rgb in rgbs:
        ref = np.zeros(3, np.float32); got = np.zeros(3, np.float32)
        lib.rgb2spec_fetch(model, rgb.ctypes.data_as(fp), ref.ctypes.data_as(fp))
        orc.lib.orc_jh_fetch(orc.color, rgb.ctypes.data_as(fp), got.ctypes.data_as(fp))
        assert same_or_both_nan(ref, got), (rgb, ref, got)
        n_nan += int(np.isnan(ref).any())
        if np.isnan(ref).any():
            continue
        for lam in lams[:4] if rgb is not rgbs[0] else lams:
            a = lib.rgb2spec_eval_precise(ref.ctypes.data_as(fp), C.c_float(lam))
            b = orc.lib.orc_jh_eval_precise(got.ctypes.data_as(fp), C.c_float(lam))
            assert struct.pack("f", a) == struct.pack("f", b), (rgb, lam, a, b)
    assert n_nan == 1          # only black divides by zero (rgb2spec.c:88)


# ------------------------------------------------------------------------------------------------
# PNG: the host's decoder against the reference's lodepng on every PNG variant
# ------------------------------------------------------------------------------------------------
def _chunk(tag, payload):
    return struct.pack(">I", len(payload)) + tag + payload + struct.pack(">I", zlib.crc32(tag + payload) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def _pack_rows(samples, depth):
    """samples: [h][w][ch] ints < 2^depth -> list of packed scanlines (bytes)"""
    h, w, ch = samples.shape
    rows = []
    for y in range(h):
        flat = samples[y].reshape(-1)
        if depth == 8:
            rows.append(bytes(flat.astype(np.uint8)))
        elif depth == 16:
            rows.append(flat.astype(">u2").tobytes())
        else:
            per = 8 // depth
            pad = (-len(flat)) % per
            v = np.concatenate([flat, np.zeros(pad, flat.dtype)]).reshape(-1, per)
            out = np.zeros(len(v), np.uint32)
            for k in range(per):
                out |= v[:, k].astype(np.uint32) << (8 - depth * (k + 1))
            rows.append(bytes(out.astype(np.uint8)))
    return rows


def _filter_rows(rows, bpp, rng):
    out = bytearray()
    prev = bytes(len(rows[0])) if rows else b""
    for row in rows:
        ft = int(rng.integers(0, 5))
        enc = bytearray(len(row))
        for x in range(len(row)):
            a = row[x - bpp] if x >= bpp else 0
            b = prev[x]
            c = prev[x - bpp] if x >= bpp else 0
            pred = (0, a, b, (a + b) // 2, _paeth(a, b, c))[ft]
            enc[x] = (row[x] - pred) & 0xFF
        out.append(ft)
        out += enc
        prev = row
    return bytes(out)


def make_png(path, w, h, ctype, depth, interlace, rng, palette_size=None):
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    hi = (palette_size if ctype == 3 else (1 << depth))
    samples = rng.integers(0, hi, size=(h, w, ch))
    bpp = max(1, ch * depth // 8)
    raw = b""
    passes = [(0, 0, 1, 1)] if not interlace else [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
    for x0, y0, dx, dy in passes:
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        raw += _filter_rows(_pack_rows(sub, depth), bpp, rng)
    body = _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if ctype == 3:
        body += _chunk(b"PLTE", bytes(rng.integers(0, 256, size=3 * palette_size, dtype=np.uint8)))
        body += _chunk(b"tRNS", bytes(rng.integers(0, 256, size=palette_size // 2, dtype=np.uint8)))
    comp = zlib.compress(raw, 6)
    cut = len(comp) // 3
    body += _chunk(b"IDAT", comp[:cut]) + _chunk(b"IDAT", comp[cut:])       # split stream
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + body + _chunk(b"IEND", b""))


def host_decode(path):
    host = _capi.host_lib()
    ptr = C.POINTER(C.c_uint8)(); w = C.c_uint32(); h = C.c_uint32()
    rc = host.ssh_load_png_rgb8(os.fsencode(path), C.byref(ptr), C.byref(w), C.byref(h))
    if rc != 0:
        return rc, None
    arr = np.ctypeslib.as_array(ptr, shape=(h.value, w.value, 3)).copy()
    host.ssh_free(ptr)
    return 0, arr


@pytest.fixture(scope="module")
def png_ref():
    if ref_lib.lodepng() is None:
        pytest.skip("oracle/_ref/libref_lodepng.so not built (needs /root/reference)")
    return ref_lib


def test_png_decoder_matches_lodepng_on_shipped_textures(png_ref):
    for name in ("test-img.png", "crystal-lizard-512.png"):
        p = os.path.join(DATA, "scenes", name)
        err, ref = png_ref.lodepng_decode(p)
        rc, got = host_decode(p)
        assert err == 0 and rc == 0
        assert np.array_equal(ref, got), name


VARIANTS = [(0, d) for d in (1, 2, 4, 8, 16)] + [(2, 8), (2, 16)] + [(3, d) for d in (1, 2, 4, 8)] + [(4, 8), (4, 16), (6, 8), (6, 16)]


@pytest.mark.parametrize("interlace", [0, 1])
@pytest.mark.parametrize("ctype,depth", VARIANTS)
def test_png_decoder_matches_lodepng_on_every_variant(png_ref, tmp_path, ctype, depth, interlace):
    rng = np.random.default_rng(100 * ctype + depth + 7 * interlace)
    for (w, h) in ((1, 1), (3, 2), (5, 9), (8, 8), (13, 17), (33, 5)):
        p = str(tmp_path / ("v_%d_%d_%d_%dx%d.png" % (ctype, depth, interlace, w, h)))
        make_png(p, w, h, ctype, depth, interlace, rng, palette_size=min(1 << depth, 200) if ctype == 3 else None)
        err, ref = png_ref.lodepng_decode(p)
        rc, got = host_decode(p)
        assert err == 0, "lodepng rejected the generated file (%d)" % err
        assert rc == 0
        assert np.array_equal(ref, got), (ctype, depth, interlace, w, h)


def test_png_decoder_rejects_what_lodepng_rejects(png_ref, tmp_path):
    rng = np.random.default_rng(3)
    good = str(tmp_path / "good.png")
    make_png(good, 9, 7, 2, 8, 0, rng)
    blob = open(good, "rb").read()
    cases = {"sig": b"\x88" + blob[1:], "crc": blob[:40] + bytes([blob[40] ^ 1]) + blob[41:], "trunc": blob[:len(blob) // 2],
             "empty": b"", "zlib": None}
    # corrupt the compressed stream but keep the chunk CRC valid
    ihdr_end = 8 + 25
    ln = struct.unpack(">I", blob[ihdr_end:ihdr_end + 4])[0]
    payload = bytearray(blob[ihdr_end + 8:ihdr_end + 8 + ln]); payload[len(payload) // 2] ^= 0x55; payload[0] ^= 0x0F
    cases["zlib"] = blob[:ihdr_end] + _chunk(b"IDAT", bytes(payload)) + blob[ihdr_end + 12 + ln:]
    for name, data in cases.items():
        p = str(tmp_path / ("bad_%s.png" % name))
        open(p, "wb").write(data)
        err, _ = png_ref.lodepng_decode(p)
        rc, _ = host_decode(p)
        assert err != 0, name
        assert rc == -1, name                      # "Could not load texture" -> SSX_ERR_DATA (src/material.cpp:15-18)


def test_png_writer_round_trips_through_lodepng_and_back(png_ref, tmp_path):
    """Framebuffer::save PNG branch (src/framebuffer.cpp:148-171): clamp, *255, round, flip, RGBA8."""
    rng = np.random.default_rng(11)
    W, H = 37, 21
    fb = rng.uniform(-0.2, 1.3, size=(H, W, 4)).astype(np.float32)
    fb[0, 0] = (0.5 / 255, 1.5 / 255, 2.5 / 255, 1.0)                        # round-half cases
    ours = str(tmp_path / "ours.png")
    assert _capi.host_lib().ssh_save_image(ours.encode(), fb.ctypes.data, W, H) == 0
    err, dec = png_ref.lodepng_decode(ours, rgba=True)
    assert err == 0
    expect = np.floor(np.clip(fb, 0.0, 1.0).astype(np.float32) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)[::-1].copy()
    assert np.array_equal(dec, expect)
    # and a file written by the reference's encoder decodes identically with the host's decoder
    theirs = str(tmp_path / "theirs.png")
    assert png_ref.lodepng().ref_png_encode_rgba8(theirs.encode(), expect.ctypes.data, W, H) == 0
    rc, got = host_decode(theirs)
    assert rc == 0 and np.array_equal(got, expect[..., :3])


# ------------------------------------------------------------------------------------------------
# Meng et al. 2015: oracle restatement against the reference's own spectrum_xyz_to_p
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def meng_setup():
    lib = ref_lib.meng()
    if lib is None:
        pytest.skip("oracle/_ref/libref_meng.so not built (needs /root/reference)")
    table = ref_lib.meng_table()
    orc = ol.Oracle("cornell-srgb", texture="test-img.png", meng=table)
    return lib, table, orc


def test_meng_grid_shape(meng_setup):
    _, t, _ = meng_setup
    assert (t["grid_w"], t["grid_h"], t["n_samples"]) == (12, 14, 81)         # spectra_xyz_5nm_380_780_0.97.h:2-9
    assert (t["sample_min"], t["sample_max"]) == (380.0, 780.0)
    assert t["cells"][:, 1].max() <= 6 and t["cells"][:, 2:].max() < t["n_points"]


def test_meng_xyz_to_p_bit_identical_to_reference(meng_setup):
    lib, t, orc = meng_setup
    rng = np.random.default_rng(5)
    fp = C.POINTER(C.c_float)
    M = np.array([[0.41231515, 0.3576, 0.1805], [0.2126, 0.7152, 0.0722], [0.01932727, 0.1192, 0.95063333]], np.float32) * np.float32(100)
    xyzs = [(M @ rng.random(3, dtype=np.float32)).astype(np.float32) for _ in range(30000)]          # in-gamut colours
    xyzs += [rng.random(3, dtype=np.float32) * np.float32(100) for _ in range(30000)]                # anything, incl. outside the grid
    xyzs += [np.array(v, np.float32) for v in ([0, 0, 0], [1, 1, 1], [100, 0, 0], [0, 100, 0], [0, 0, 100], [1e-30, 1e-30, 1e-30],
                                               [1e-40, 0, 0], [95.047, 100.0, 108.883])]
    lams = [380.0, 780.0, 555.5, 400.0, 779.999]
    n_nonzero = 0; n_fan = 0
    for k, xyz in enumerate(xyzs):
        for lam in (lams if k % 50 == 0 else (float(rng.uniform(380, 780)),)):
            a = lib.ref_meng_xyz_to_p(C.c_float(lam), xyz.ctypes.data_as(fp))
            b = orc.lib.orc_meng_xyz_to_p(orc.color, C.c_float(lam), xyz.ctypes.data_as(fp))
            assert struct.pack("f", a) == struct.pack("f", b), (xyz, lam, a, b)
            n_nonzero += a != 0.0
    assert n_nonzero > 40000          # both the bilinear cells and the boundary fans were exercised


def test_meng_lrgb_to_specrefl_uses_reference_function(meng_setup):
    """color.cpp:175-201: the oracle's per-texel uplift = reference spectrum_xyz_to_p at the 4 hero wavelengths."""
    lib, t, orc = meng_setup
    fp = C.POINTER(C.c_float)
    rng = np.random.default_rng(6)
    orc.lib.orc_lrgb_to_specrefl.argtypes = [C.c_void_p, fp, C.c_float, fp]
    rows = np.array([[0.41231515, 0.3576, 0.1805], [0.2126, 0.7152, 0.0722], [0.01932727, 0.1192, 0.95063333]], np.float32)
    for _ in range(2000):
        lrgb = rng.random(3, dtype=np.float32)
        lam0 = np.float32(rng.uniform(380, 480))
        out = np.zeros(4, np.float32)
        orc.lib.orc_lrgb_to_specrefl(orc.color, lrgb.ctypes.data_as(fp), C.c_float(lam0), out.ctypes.data_as(fp))
        m100 = rows * np.float32(100.0)
        xyz = ((m100[:, 0] * lrgb[0] + m100[:, 1] * lrgb[1]) + m100[:, 2] * lrgb[2]).astype(np.float32)
        for i in range(4):
            lam = np.float32(lam0 + np.float32(i) * np.float32(100.0))
            ref = lib.ref_meng_xyz_to_p(C.c_float(lam), xyz.ctypes.data_as(fp))
            assert struct.pack("f", ref) == struct.pack("f", out[i])


def test_meng_header_converter_file_format_and_host_scene(meng_setup, tmp_path):
    """simple_spectral_amd/meng.py parses the authors' header as text; its result must equal the
    tables as the C compiler sees them (libref_meng.so), survive the SSXMENG1 file, and load in the
    host library; the host's Meng output transform (color.cpp:243-254) equals the oracle's."""
    from simple_spectral_amd import meng
    from simple_spectral_amd.renderer import SsxError
    _, t_ref, orc = meng_setup
    header = "/root/reference/src/meng-et-al.-2015/spectra_xyz_5nm_380_780_0.97.h"
    if os.path.exists(header):
        t_txt = meng.table_from_header(header)
        for k in ("grid_w", "grid_h", "n_points", "n_samples", "sample_min", "sample_max"):
            assert t_txt[k] == t_ref[k], k
        assert np.array_equal(t_txt["cells"], t_ref["cells"])
        assert np.array_equal(bits(t_txt["points"]), bits(t_ref["points"]))
        assert np.array_equal(bits(t_txt["xy_to_uv"]), bits(t_ref["xy_to_uv"]))
    path = str(tmp_path / "grid.bin")
    meng.save_table(path, t_ref)
    back = meng.load_table(path)
    assert np.array_equal(bits(back["points"]), bits(t_ref["points"])) and np.array_equal(back["cells"], t_ref["cells"])
    s = Scene("cornell-srgb", texture="test-img.png", uplift="meng", meng_grid_path=path)
    d = s.desc.contents
    g = d.meng.contents
    assert d.uplift == _capi.SSX_UPLIFT_MENG and (g.grid_w, g.grid_h, g.n_points, g.n_samples) == (12, 14, 186, 81)
    assert np.array_equal(np.ctypeslib.as_array(g.points, shape=(186 * 85,)), t_ref["points"].reshape(-1))
    xyza = np.random.default_rng(1).random((4096, 4)).astype(np.float32) * np.float32(0.5)
    assert np.array_equal(bits(s.xyza_to_srgba(xyza)), bits(orc.to_srgba(xyza)))
    plain = Scene("cornell-srgb", texture="test-img.png")
    assert not np.array_equal(bits(s.xyza_to_srgba(xyza)), bits(plain.xyza_to_srgba(xyza)))
    # error behaviour: missing file -> -1 (data), truncated file -> -1, wrong observer -> -3 (stdafx.hpp:107-109)
    with pytest.raises(SsxError) as e:
        Scene("cornell-srgb", texture="test-img.png", uplift="meng", meng_grid_path=str(tmp_path / "nope.bin"))
    assert e.value.code == -1
    open(str(tmp_path / "cut.bin"), "wb").write(open(path, "rb").read()[:5000])
    with pytest.raises(SsxError) as e:
        Scene("cornell-srgb", texture="test-img.png", uplift="meng", meng_grid_path=str(tmp_path / "cut.bin"))
    assert e.value.code == -1
    with pytest.raises(SsxError) as e:
        Scene("cornell-srgb", texture="test-img.png", uplift="meng", meng_grid_path=path, observer=2006)
    assert e.value.code == -3
