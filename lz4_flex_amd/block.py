"""lz4_flex::block, MI355X edition (reference src/block/{compress,decompress,mod}.rs).

Same names, argument meaning and error behaviour as the reference's public block API; every call
runs the HIP kernels through the C ABI.  Errors are exceptions named after the Rust enum variants
(src/block/mod.rs:82-106)."""
import ctypes as C

import numpy as np

from . import _lib as L


class CompressError(Exception):
    """block::CompressError (mod.rs:103-106)"""


class CompressOutputTooSmall(CompressError):
    pass


class DecompressError(Exception):
    """block::DecompressError (mod.rs:82-98)"""


class OutputTooSmall(DecompressError):
    def __init__(self, expected, actual):
        super().__init__("provided output is too small for the decompressed data, actual %d, expected %d"
                         % (actual, expected))
        self.expected, self.actual = expected, actual


class LiteralOutOfBounds(DecompressError):
    pass


class ExpectedAnotherByte(DecompressError):
    pass


class OffsetZero(DecompressError):
    pass


class OffsetOutOfBounds(DecompressError):
    pass


class DeviceError(RuntimeError):
    """HIP/runtime failure or an entry point whose GPU path is not built (no CPU fallback exists)."""


_DECODE_ERRORS = {L.E_LITERAL_OUT_OF_BOUNDS: LiteralOutOfBounds, L.E_EXPECTED_ANOTHER_BYTE: ExpectedAnotherByte,
                  L.E_OFFSET_ZERO: OffsetZero, L.E_OFFSET_OUT_OF_BOUNDS: OffsetOutOfBounds}


def _raise_decode(code, detail):
    if code == L.E_OUTPUT_TOO_SMALL:
        raise OutputTooSmall(detail.expected, detail.actual)
    if code in _DECODE_ERRORS:
        raise _DECODE_ERRORS[code]()
    raise DeviceError("lz4flex error %d: %s" % (code, L.last_error()))


def _buf(b):
    """bytes-like -> (ctypes pointer, length, keepalive)"""
    if isinstance(b, (bytes, bytearray)):
        arr = (C.c_uint8 * max(len(b), 1)).from_buffer_copy(bytes(b) if len(b) else b"\0")
        return C.cast(arr, C.c_void_p), len(b), arr
    a = np.ascontiguousarray(np.frombuffer(memoryview(b), dtype=np.uint8))
    return C.c_void_p(a.ctypes.data if a.size else 0), int(a.size), a


def set_compress_mode(mode, ctx=None):
    """Encoder of this thread's default context (or of `ctx`): "fast" = throughput encoder (own parse: a valid LZ4 block
    that lz4_flex decodes to the input; default), "exact" = lz4_flex's own bytes (src/block/compress.rs:318-489)."""
    v = {"fast": 0, "exact": 1}[mode]
    rc = L.load().lz4flex_set_tuning(ctx, b"compress_mode", v)
    if rc:
        raise DeviceError("lz4flex_set_tuning(compress_mode) failed (%d): %s" % (rc, L.last_error()))


def get_maximum_output_size(input_len):
    """block::get_maximum_output_size (compress.rs:588-590)"""
    return int(L.load().lz4flex_get_maximum_output_size(int(input_len)))


def compress_into(input, output):
    """block::compress_into (compress.rs:599-601): `output` is a writable buffer; returns bytes written."""
    lib = L.load()
    ip, n, _k = _buf(input)
    out = (C.c_uint8 * max(len(output), 1)).from_buffer(output) if len(output) else (C.c_uint8 * 1)()
    r = lib.lz4flex_compress_into(ip, n, C.cast(out, C.c_void_p), len(output))
    if r == -L.E_OUTPUT_TOO_SMALL:
        raise CompressOutputTooSmall("output is too small for the compressed data, use get_maximum_output_size "
                                     "to reserve enough space")
    if r < 0:
        raise DeviceError("lz4flex error %d: %s" % (-r, L.last_error()))
    return int(r)


def compress(input):
    """block::compress (compress.rs:679-681)"""
    out = bytearray(get_maximum_output_size(len(input)))
    n = compress_into(input, out)
    return bytes(out[:n])


def compress_prepend_size(input):
    """block::compress_prepend_size (compress.rs:673-675)"""
    return len(input).to_bytes(4, "little") + compress(input)


class CompressTable:
    """block::CompressTable (compress.rs:710-740): small() / large() / default; reused across compress_into_with_table calls"""

    def __init__(self, large=False):
        self._h = L.load().lz4flex_compress_table_new(1 if large else 0)
        if not self._h:
            raise DeviceError("lz4flex_compress_table_new failed: " + L.last_error())

    @classmethod
    def small(cls):
        return cls(False)

    @classmethod
    def large(cls):
        return cls(True)

    @property
    def is_large(self):
        return bool(L.load().lz4flex_compress_table_is_large(self._h))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                L.load().lz4flex_compress_table_free(h)
            except Exception:
                pass


def compress_into_with_table(input, output, table):
    """block::compress_into_with_table (compress.rs:742-766)"""
    lib = L.load()
    ip, n, _k = _buf(input)
    out = (C.c_uint8 * max(len(output), 1)).from_buffer(output) if len(output) else (C.c_uint8 * 1)()
    r = lib.lz4flex_compress_into_with_table(ip, n, C.cast(out, C.c_void_p), len(output), table._h)
    if r == -L.E_OUTPUT_TOO_SMALL:
        raise CompressOutputTooSmall()
    if r < 0:
        raise DeviceError("lz4flex error %d: %s" % (-r, L.last_error()))
    return int(r)


def compress_prepend_size_with_dict(input, ext_dict):
    """block::compress_prepend_size_with_dict (compress.rs:692-694), through the C entry point"""
    lib = L.load()
    ip, n, _k = _buf(input)
    dp, dn, _k2 = _buf(ext_dict)
    out = bytearray(4 + get_maximum_output_size(n))
    o = (C.c_uint8 * len(out)).from_buffer(out)
    r = lib.lz4flex_compress_prepend_size_with_dict(ip, n, C.cast(o, C.c_void_p), len(out), dp, dn)
    if r < 0:
        raise DeviceError("lz4flex error %d: %s" % (-r, L.last_error()))
    del o
    return bytes(out[:r])


def compress_into_with_dict(input, output, dict_data):
    """block::compress_into_with_dict (compress.rs:610-616)"""
    lib = L.load()
    ip, n, _k = _buf(input)
    dp, dn, _k2 = _buf(dict_data)
    out = (C.c_uint8 * max(len(output), 1)).from_buffer(output) if len(output) else (C.c_uint8 * 1)()
    r = lib.lz4flex_compress_into_with_dict(ip, n, C.cast(out, C.c_void_p), len(output), dp, dn)
    if r == -L.E_OUTPUT_TOO_SMALL:
        raise CompressOutputTooSmall()
    if r < 0:
        raise DeviceError("lz4flex error %d: %s" % (-r, L.last_error()))
    return int(r)


def compress_with_dict(input, ext_dict):
    """block::compress_with_dict (compress.rs:685-687); dicts of <= 3 bytes are ignored (:626-628)"""
    if len(ext_dict) <= 3:
        return compress(input)
    out = bytearray(get_maximum_output_size(len(input)))
    n = compress_into_with_dict(input, out, ext_dict)
    return bytes(out[:n])


def decompress_into(input, output):
    """block::decompress_into (decompress.rs:454-456): returns bytes written."""
    lib = L.load()
    ip, n, _k = _buf(input)
    out = (C.c_uint8 * max(len(output), 1)).from_buffer(output) if len(output) else (C.c_uint8 * 1)()
    d = L.ErrDetail()
    r = lib.lz4flex_decompress_into(ip, n, C.cast(out, C.c_void_p), len(output), C.byref(d))
    if r < 0:
        _raise_decode(int(-r), d)
    return int(r)


def decompress_into_with_dict(input, output, ext_dict):
    """block::decompress_into_with_dict (decompress.rs:462-468)"""
    lib = L.load()
    ip, n, _k = _buf(input)
    dp, dn, _k2 = _buf(ext_dict)
    out = (C.c_uint8 * max(len(output), 1)).from_buffer(output) if len(output) else (C.c_uint8 * 1)()
    d = L.ErrDetail()
    r = lib.lz4flex_decompress_into_with_dict(ip, n, C.cast(out, C.c_void_p), len(output), dp, dn, C.byref(d))
    if r < 0:
        _raise_decode(int(-r), d)
    return int(r)


def decompress(input, min_uncompressed_size):
    """block::decompress (decompress.rs:506-517)"""
    out = bytearray(min_uncompressed_size)
    n = decompress_into(input, out)
    return bytes(out[:n])


def decompress_with_dict(input, min_uncompressed_size, ext_dict):
    """block::decompress_with_dict (decompress.rs:478-489)"""
    out = bytearray(min_uncompressed_size)
    n = decompress_into_with_dict(input, out, ext_dict)
    return bytes(out[:n])


def uncompressed_size(input):
    """block::uncompressed_size (mod.rs:151-157): (size, rest)"""
    if len(input) < 4:
        raise ExpectedAnotherByte()
    return int.from_bytes(bytes(input[:4]), "little"), input[4:]


def decompress_size_prepended(input):
    """block::decompress_size_prepended (decompress.rs:493-496)"""
    size, rest = uncompressed_size(input)
    return decompress(rest, size)


def decompress_size_prepended_with_dict(input, ext_dict):
    """block::decompress_size_prepended_with_dict (decompress.rs:521-527), through the C entry point"""
    lib = L.load()
    size, _rest = uncompressed_size(input)
    ip, n, _k = _buf(input)
    dp, dn, _k2 = _buf(ext_dict)
    out = bytearray(max(size, 1))
    o = (C.c_uint8 * len(out)).from_buffer(out)
    d = L.ErrDetail()
    r = lib.lz4flex_decompress_size_prepended_with_dict(ip, n, C.cast(o, C.c_void_p), size, dp, dn, C.byref(d))
    del o
    if r < 0:
        _raise_decode(int(-r), d)
    return bytes(out[:r])


# ---- batched entry points (host numpy arrays) -----------------------------------------------------
def _np(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    return a, C.c_void_p(a.ctypes.data if a.size else 0)


def compress_batch(in_buf, in_off, in_len, out_buf, out_off, out_cap, flags=None, ctx=None):
    """lz4flex_compress_batch over host buffers: returns (out_len[u32], status[i32])."""
    lib = L.load()
    n = len(in_off)
    in_buf = np.ascontiguousarray(np.frombuffer(memoryview(in_buf), dtype=np.uint8)) if not isinstance(in_buf, np.ndarray) else in_buf
    io, iop = _np(in_off, np.uint64)
    il, ilp = _np(in_len, np.uint32)
    oo, oop = _np(out_off, np.uint64)
    oc, ocp = _np(out_cap, np.uint32)
    fl, flp = (None, None) if flags is None else _np(flags, np.uint32)
    out_len = np.zeros(n, dtype=np.uint32)
    status = np.zeros(n, dtype=np.int32)
    rc = lib.lz4flex_compress_batch(ctx, C.c_void_p(in_buf.ctypes.data if in_buf.size else 0), iop, ilp, flp, n,
                                    C.c_void_p(out_buf.ctypes.data), oop, ocp, C.c_void_p(out_len.ctypes.data),
                                    C.c_void_p(status.ctypes.data), L.MEM_HOST, None)
    if rc:
        raise DeviceError("lz4flex_compress_batch failed (%d): %s" % (rc, L.last_error()))
    return out_len, status


def decompress_batch(in_buf, in_off, in_len, out_buf, out_off, out_cap, ctx=None):
    """lz4flex_decompress_batch over host buffers: returns (out_len[u32], status[i32], detail[n,2] u64)."""
    lib = L.load()
    n = len(in_off)
    io, iop = _np(in_off, np.uint64)
    il, ilp = _np(in_len, np.uint32)
    oo, oop = _np(out_off, np.uint64)
    oc, ocp = _np(out_cap, np.uint32)
    out_len = np.zeros(n, dtype=np.uint32)
    status = np.zeros(n, dtype=np.int32)
    detail = np.zeros((n, 2), dtype=np.uint64)
    rc = lib.lz4flex_decompress_batch(ctx, C.c_void_p(in_buf.ctypes.data if in_buf.size else 0), iop, ilp, n,
                                      C.c_void_p(out_buf.ctypes.data), oop, ocp, C.c_void_p(out_len.ctypes.data),
                                      C.c_void_p(status.ctypes.data), C.c_void_p(detail.ctypes.data), L.MEM_HOST, None)
    if rc:
        raise DeviceError("lz4flex_decompress_batch failed (%d): %s" % (rc, L.last_error()))
    return out_len, status, detail
