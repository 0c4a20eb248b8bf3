"""Minimal PNG reader for the tests (non-interlaced; grey, RGB, palette and RGBA at 1 to 8 bits per
sample): the encoders under test are this library's own and the reference's lodepng, which picks
the smallest colour type that holds the image. Returns uint8 [h, w, 4] RGBA."""
import struct
import zlib

import numpy as np


def read_png_rgba(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, plte, trns = 8, b"", None, None
    w = h = depth = ctype = 0
    while pos < len(raw):
        n, kind = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        if kind == b"IHDR":
            w, h, depth, ctype, _, _, inter = struct.unpack(">IIBBBBB", body)
            assert inter == 0 and depth <= 8 and ctype in (0, 2, 3, 6)
        elif kind == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif kind == b"tRNS":
            trns = np.frombuffer(body, np.uint8)
        elif kind == b"IDAT":
            idat += body
        pos += 12 + n
    channels = {0: 1, 2: 3, 3: 1, 6: 4}[ctype]
    bits = channels * depth
    stride = (w * bits + 7) // 8
    bpp = max(1, bits // 8)  # the filters' "previous pixel" distance in bytes
    data = zlib.decompress(idat)
    rows = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        f = data[y * (stride + 1)]
        line = np.frombuffer(data, np.uint8, stride, y * (stride + 1) + 1).astype(np.int32)
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        else:  # sub / average / paeth: byte-serial
            cur = np.zeros(stride, np.int32)
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if f == 1:
                    pr = a
                elif f == 3:
                    pr = (a + b) // 2
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pr) & 255
        rows[y] = cur
        prev = cur
    if depth < 8:  # samples packed most significant bit first
        bitsarr = np.unpackbits(rows, axis=1)[:, :w * bits].reshape(h, w * channels, depth)
        samples = (bitsarr * (1 << np.arange(depth - 1, -1, -1))).sum(axis=2).astype(np.uint8)
    else:
        samples = rows[:, :w * channels]
    samples = samples.reshape(h, w, channels)
    out = np.zeros((h, w, 4), np.uint8)
    out[..., 3] = 255
    if ctype == 6:
        out[:] = samples
    elif ctype == 2:
        out[..., :3] = samples
    elif ctype == 3:
        idx = samples[..., 0]
        out[..., :3] = plte[idx]
        if trns is not None:
            alpha = np.full(len(plte), 255, np.uint8)
            alpha[:len(trns)] = trns
            out[..., 3] = alpha[idx]
    else:
        g = samples[..., 0].astype(np.uint16) * (255 // ((1 << depth) - 1))
        out[..., :3] = g[..., None].astype(np.uint8)
    return out
