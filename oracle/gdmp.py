"""Reader for the GDMP1 dump files written by oracle/ref_dump.cpp (TEST INFRASTRUCTURE).

Record layout: u32 name_len, name, u8 dtype ('d' float64 / 'i' int32), u32 ndim, u64 dims[ndim], data.
"""
import struct
import numpy as np


def read_gdmp(path):
    out = {}
    with open(path, "rb") as f:
        magic = f.read(8)
        if magic[:5] != b"GDMP1":
            raise ValueError("%s: not a GDMP1 file" % path)
        while True:
            head = f.read(4)
            if len(head) < 4:
                break
            (nlen,) = struct.unpack("<I", head)
            name = f.read(nlen).decode()
            dt = f.read(1)
            (nd,) = struct.unpack("<I", f.read(4))
            dims = struct.unpack("<%dQ" % nd, f.read(8 * nd))
            count = int(np.prod(dims)) if nd else 1
            if dt == b"d":
                arr = np.frombuffer(f.read(8 * count), dtype="<f8")
            elif dt == b"i":
                arr = np.frombuffer(f.read(4 * count), dtype="<i4")
            else:
                raise ValueError("bad dtype %r" % dt)
            out[name] = arr.reshape(dims).copy()
    return out
