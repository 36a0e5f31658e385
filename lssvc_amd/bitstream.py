"""Bitstream file framing of write_stream=1 -- the reference's wire format (src/utils/stream_helper.py:19-99):
big-endian u32 headers; an I-frame layer file = (height, width, len_y, len_z) + y string + z string, a P-frame
layer file = (len) + one rANS string."""
import os
import struct
import time


def _prof(t0):
    from . import hip_ops
    if hip_ops.STREAM_PROF is not None:
        hip_ops.STREAM_PROF["io_s"] = hip_ops.STREAM_PROF.get("io_s", 0.0) + time.perf_counter() - t0


def get_downsampled_shape(height, width, p, resample_times=1):
    """stream_helper.py:19-23."""
    pad_d = p * resample_times
    new_h = (height + pad_d - 1) // pad_d * pad_d
    new_w = (width + pad_d - 1) // pad_d * pad_d
    return int(new_h / p + 0.5), int(new_w / p + 0.5)


def filesize(path):
    if not os.path.isfile(path):
        raise ValueError('Invalid file "%s".' % path)
    return os.stat(path).st_size


def encode_i(height, width, y_string, z_string, output):
    t0 = time.perf_counter()
    with open(output, "wb") as f:
        f.write(struct.pack(">4I", height, width, len(y_string), len(z_string)))
        f.write(y_string)
        f.write(z_string)
    _prof(t0)


def decode_i(inputpath):
    t0 = time.perf_counter()
    with open(inputpath, "rb") as f:
        height, width, ly, lz = struct.unpack(">4I", f.read(16))
        y_string = f.read(ly)
        z_string = f.read(lz)
    _prof(t0)
    if len(y_string) != ly or len(z_string) != lz:
        raise ValueError('Truncated I-frame stream "%s".' % inputpath)
    return height, width, y_string, z_string


def encode_p(string, output):
    t0 = time.perf_counter()
    with open(output, "wb") as f:
        f.write(struct.pack(">I", len(string)))
        f.write(string)
    _prof(t0)


def decode_p(inputpath):
    t0 = time.perf_counter()
    with open(inputpath, "rb") as f:
        (n,) = struct.unpack(">I", f.read(4))
        string = f.read(n)
    _prof(t0)
    if len(string) != n:
        raise ValueError('Truncated P-frame stream "%s".' % inputpath)
    return string
