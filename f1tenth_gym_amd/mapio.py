"""Map files without PIL and PyYAML: a grayscale PNG decoder (stdlib zlib + NumPy) and a reader for the
flat `key: value` yaml that ROS map_server / the reference ship (laser_models.py:397-416 reads exactly
`resolution` and `origin` from it).  PNG decoding is lossless, so the array equals what
`np.array(PIL.Image.open(path))` returns for the same file (checked in tests/test_host_logic.py on every
shipped map); everything downstream (flip, threshold, exact EDT) happens on the device.
"""
import struct
import zlib

import numpy as np

_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def _paeth_row(cur, prev, bpp):
    """filter type 4 for one scanline (serial in x: each byte needs its reconstructed left neighbour)"""
    out = bytearray(len(cur))
    pv = bytes(prev)
    for i, x in enumerate(bytes(cur)):
        a = out[i - bpp] if i >= bpp else 0
        b = pv[i]
        c = pv[i - bpp] if i >= bpp else 0
        p = a + b - c
        pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
        pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
        out[i] = (x + pr) & 255
    return np.frombuffer(bytes(out), dtype=np.uint8)


def _average_row(cur, prev, bpp):
    out = bytearray(len(cur))
    pv = bytes(prev)
    for i, x in enumerate(bytes(cur)):
        a = out[i - bpp] if i >= bpp else 0
        out[i] = (x + ((a + pv[i]) >> 1)) & 255
    return np.frombuffer(bytes(out), dtype=np.uint8)


def read_png_gray(path):
    """-> uint8 [H][W] (8-bit grayscale) or uint16 [H][W] (16-bit), top row first like PIL.
    Colour, palette, alpha, sub-byte and interlaced PNGs raise ValueError: the reference's set_map needs
    a single-channel image (laser_models.py:399-404 thresholds one value per cell)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != _PNG_SIG:
        raise ValueError("%s is not a PNG file" % path)
    pos, idat, hdr = 8, [], None
    while pos + 8 <= len(data):
        n, = struct.unpack(">I", data[pos:pos + 4])
        kind = data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        if zlib.crc32(kind + body) & 0xffffffff != struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]:
            raise ValueError("%s: corrupt PNG chunk %r" % (path, kind))
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
        pos += 12 + n
    if hdr is None or not idat:
        raise ValueError("%s: PNG without IHDR / IDAT" % path)
    w, h, depth, ctype, _, _, interlace = hdr
    if ctype != 0 or depth not in (8, 16) or interlace != 0:
        raise ValueError("map image must be single-channel 8- or 16-bit grayscale, non-interlaced "
                         "(got colour type %d, bit depth %d, interlace %d)" % (ctype, depth, interlace))
    bpp = depth // 8
    stride = w * bpp
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8)
    if raw.size != h * (stride + 1):
        raise ValueError("%s: PNG data size mismatch" % path)
    raw = raw.reshape(h, stride + 1)
    out = np.empty((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.uint8)
    for y in range(h):
        ft, cur = int(raw[y, 0]), raw[y, 1:]
        if ft == 0:
            row = cur
        elif ft == 1:    # Sub: a running sum per byte lane, modulo 256
            row = np.add.accumulate(cur.reshape(-1, bpp), axis=0, dtype=np.uint8).reshape(-1)
        elif ft == 2:    # Up
            row = cur + prev
        elif ft == 3:
            row = _average_row(cur, prev, bpp)
        elif ft == 4:
            row = _paeth_row(cur, prev, bpp)
        else:
            raise ValueError("%s: unknown PNG filter type %d" % (path, ft))
        out[y] = row
        prev = out[y]
    if depth == 16:
        return out.view(">u2").astype(np.uint16)
    return out


def _scalar(tok):
    tok = tok.strip()
    if len(tok) >= 2 and tok[0] == tok[-1] and tok[0] in "'\"":
        return tok[1:-1]
    for conv in (int, float):
        try:
            return conv(tok)
        except ValueError:
            pass
    return tok


def read_map_yaml(path):
    """the flat mapping of a map_server yaml: scalars and one-line [a, b, c] lists; '#' comments.
    Anything nested is outside what map yamls contain and raises ValueError."""
    meta = {}
    with open(path, "r") as f:
        for ln, line in enumerate(f, 1):
            line = line.split("#", 1)[0].rstrip()
            if not line.strip() or line.strip() in ("---", "..."):
                continue
            if line[0] in " \t-" or ":" not in line:
                raise ValueError("%s:%d: only a flat `key: value` mapping is supported" % (path, ln))
            key, val = line.split(":", 1)
            val = val.strip()
            if val.startswith("["):
                if not val.endswith("]"):
                    raise ValueError("%s:%d: list must close on the same line" % (path, ln))
                meta[key.strip()] = [_scalar(t) for t in val[1:-1].split(",") if t.strip()]
            elif val == "":
                raise ValueError("%s:%d: nested values are not supported" % (path, ln))
            else:
                meta[key.strip()] = _scalar(val)
    return meta
