"""Radiance RGBE (.hdr) pictures for the Python host: the environment map the reference loads with
``stbi_loadf("env.hdr", &w, &h, &channels, 4)`` (Application.cpp:225-231 -> ``Sky::hdri_data``, RGBA f32, rows top-down, alpha 1)
and the screenshot it stores with ``stbi_flip_vertically_on_write(true); stbi_write_hdr(..., 4, GetFrame())`` (Image.cpp:71-74).
stb_image / stb_image_write are not vendored by the reference and absent here; this restates their published RGBE arithmetic
(decode: byte * 2^(E - 136), zero for E == 0; encode: frexp of the largest channel, truncation) in numpy.  csrc/hdr_io.hpp is the
C++ twin used by mirt_headless; the tests read each one's files with the other."""
from __future__ import annotations

import numpy as np


def rgbe_to_float(rgbe: np.ndarray) -> np.ndarray:
    """[..., 4] uint8 -> [..., 4] float32 RGBA (alpha 1), stbi__hdr_convert."""
    rgbe = np.asarray(rgbe, dtype=np.uint8)
    e = rgbe[..., 3].astype(np.int32)
    f = np.ldexp(np.float32(1.0), e - 136).astype(np.float32)
    out = np.empty(rgbe.shape, dtype=np.float32)
    out[..., :3] = rgbe[..., :3].astype(np.float32) * f[..., None]
    out[..., :3][e == 0] = 0.0
    out[..., 3] = 1.0
    return out


def float_to_rgbe(rgb: np.ndarray) -> np.ndarray:
    """[..., >=3] float32 -> [..., 4] uint8, stbiw__linear_to_rgbe."""
    rgb = np.asarray(rgb, dtype=np.float32)[..., :3]
    m = rgb.max(axis=-1)
    mant, exp = np.frexp(m.astype(np.float32))
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        norm = (mant.astype(np.float32) * np.float32(256.0) / m).astype(np.float32)
    out = np.zeros(rgb.shape[:-1] + (4,), dtype=np.uint8)
    ok = m >= np.float32(1e-32)
    out[..., :3][ok] = (rgb[ok] * norm[ok][..., None]).astype(np.uint8)
    out[..., 3][ok] = (exp[ok] + 128).astype(np.uint8)
    return out


def read_hdr(path: str) -> np.ndarray:
    """-> [height, width, 4] float32, first row = top of the picture (what stbi_loadf(..., 4) returns)."""
    data = open(path, "rb").read()
    at = 0

    def line():
        nonlocal at
        end = data.index(b"\n", at)
        s = data[at:end]; at = end + 1
        return s.decode("latin-1")
    if line() not in ("#?RADIANCE", "#?RGBE"):
        raise ValueError("not a Radiance picture")
    fmt = False
    while True:
        l = line()
        if l == "":
            break
        fmt |= l == "FORMAT=32-bit_rle_rgbe"
    if not fmt:
        raise ValueError("unsupported FORMAT")
    res = line().split()
    if len(res) != 4 or res[0] != "-Y" or res[2] != "+X":
        raise ValueError("unsupported resolution line")
    h, w = int(res[1]), int(res[3])
    buf = np.frombuffer(data, dtype=np.uint8)
    img = np.empty((h, w, 4), dtype=np.uint8)
    for y in range(h):
        if 8 <= w < 32768 and buf[at] == 2 and buf[at + 1] == 2 and not (buf[at + 2] & 0x80):
            assert (int(buf[at + 2]) << 8 | int(buf[at + 3])) == w
            at += 4
            for k in range(4):
                x = 0
                while x < w:
                    c = int(buf[at]); at += 1
                    if c > 128:
                        c -= 128
                        img[y, x:x + c, k] = buf[at]; at += 1
                    else:
                        img[y, x:x + c, k] = buf[at:at + c]; at += c
                    x += c
        else:
            img[y] = buf[at:at + 4 * w].reshape(w, 4); at += 4 * w
    return rgbe_to_float(img)


def write_hdr(path: str, rgba_bottom_up: np.ndarray) -> None:
    """Image::Store (Image.cpp:71-74): a frame as Renderer.GetFrame() returns it (row 0 = bottom of the picture) -> top-down RGBE file.
    Scanlines are written flat (every reader accepts them; the run-length form is an encoder's choice)."""
    img = np.asarray(rgba_bottom_up, dtype=np.float32)
    h, w = img.shape[:2]
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\n# Written by mirt (Image::Store, Image.cpp:71-74)\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        f.write(float_to_rgbe(img[::-1]).tobytes())
