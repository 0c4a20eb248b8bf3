"""Reader/writer for PTAR, a tiny named-array container.

Layout: magic ``PTAR1\\n`` then records
``u32 name_len | name | u32 dtype | u64 count | raw little-endian data`` with
dtype 0=f64 1=i32 2=f32 3=u8 4=i64.
"""
import struct

import numpy as np

_DTYPES = {0: np.dtype("<f8"), 1: np.dtype("<i4"), 2: np.dtype("<f4"), 3: np.dtype("u1"), 4: np.dtype("<i8")}
_CODES = {v: k for k, v in _DTYPES.items()}


def load(path):
    """Return {name: 1-D numpy array} for every record in the file."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:6] != b"PTAR1\n":
        raise ValueError("%s: not a PTAR file" % path)
    out, p = {}, 6
    while p < len(data):
        (nl,) = struct.unpack_from("<I", data, p)
        p += 4
        name = data[p:p + nl].decode()
        p += nl
        dt, cnt = struct.unpack_from("<IQ", data, p)
        p += 12
        dtype = _DTYPES[dt]
        out[name] = np.frombuffer(data, dtype=dtype, count=cnt, offset=p).copy()
        p += cnt * dtype.itemsize
    return out


def save(path, arrays):
    with open(path, "wb") as f:
        f.write(b"PTAR1\n")
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            code = _CODES[a.dtype.newbyteorder("<") if a.dtype.byteorder == ">" else np.dtype(a.dtype.str.replace("=", "<"))]
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)) + nb + struct.pack("<IQ", code, a.size))
            f.write(a.astype(_DTYPES[code], copy=False).tobytes())
