"""ctypes view of the CPU oracle (oracle/liblz4flex_oracle.so) and of the system C liblz4 1.9.3.
Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline."""
import ctypes as C
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

ERR_NAMES = {1: "OutputTooSmall", 2: "LiteralOutOfBounds", 3: "ExpectedAnotherByte", 4: "OffsetZero",
             5: "OffsetOutOfBounds"}


class ErrDetail(C.Structure):
    _fields_ = [("expected", C.c_uint64), ("actual", C.c_uint64), ("inner", C.c_int32)]


class FrameInfoO(C.Structure):
    _fields_ = [("has_content_size", C.c_int), ("content_size", C.c_uint64), ("block_size", C.c_int),
                ("block_mode", C.c_int), ("block_checksums", C.c_int), ("content_checksum", C.c_int),
                ("legacy_frame", C.c_int)]


_o = None


def lib():
    global _o
    if _o is None:
        o = C.CDLL(os.path.join(ROOT, "oracle", "liblz4flex_oracle.so"))
        o.lz4o_get_maximum_output_size.restype = C.c_size_t
        o.lz4o_get_maximum_output_size.argtypes = [C.c_size_t]
        for f in ("lz4o_compress_into",):
            getattr(o, f).restype = C.c_int64
            getattr(o, f).argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        o.lz4o_compress_into_with_dict.restype = C.c_int64
        o.lz4o_compress_into_with_dict.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        o.lz4o_compress_frame_block.restype = C.c_int64
        o.lz4o_compress_frame_block.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
        o.lz4o_decompress_into.restype = C.c_int64
        o.lz4o_decompress_into.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(ErrDetail)]
        o.lz4o_decompress_into_with_dict.restype = C.c_int64
        o.lz4o_decompress_into_with_dict.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p,
                                                     C.c_size_t, C.POINTER(ErrDetail)]
        o.lz4o_does_token_fit.restype = C.c_int
        o.lz4o_does_token_fit.argtypes = [C.c_uint8]
        o.lz4o_count_same_bytes.restype = C.c_size_t
        o.lz4o_count_same_bytes.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t, C.c_size_t]
        o.lz4o_xxh32.restype = C.c_uint32
        o.lz4o_xxh32.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
        o.lz4o_frame_info_write.restype = C.c_int64
        o.lz4o_frame_info_write.argtypes = [C.POINTER(FrameInfoO), C.c_char_p, C.c_size_t]
        o.lz4o_frame_compress.restype = C.c_int64
        o.lz4o_frame_compress.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_size_t,
                                          C.POINTER(FrameInfoO), C.c_char_p, C.c_size_t, C.POINTER(ErrDetail)]
        o.lz4o_frame_decompress.restype = C.c_int64
        o.lz4o_frame_decompress.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t),
                                            C.POINTER(ErrDetail)]
        o.lz4o_bench_batch.restype = C.c_double
        o.lz4o_bench_batch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_uint32, C.c_int, C.c_int]
        o.lz4o_bench_batch_fn.restype = C.c_double
        o.lz4o_bench_batch_fn.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_uint32, C.c_int, C.c_int]
        o.lz4o_bench_pool.restype = C.c_double
        o.lz4o_bench_pool.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int)]
        _o = o
    return _o


def max_out(n):
    return int(lib().lz4o_get_maximum_output_size(n))


def compress(data, cap=None):
    data = bytes(data)
    cap = max_out(len(data)) if cap is None else cap
    out = C.create_string_buffer(max(cap, 1))
    n = lib().lz4o_compress_into(data, len(data), out, cap)
    if n < 0:
        raise ValueError(ERR_NAMES.get(-n, str(-n)))
    return out.raw[:n]


def compress_with_dict(data, d):
    data, d = bytes(data), bytes(d)
    cap = max_out(len(data))
    out = C.create_string_buffer(max(cap, 1))
    n = lib().lz4o_compress_into_with_dict(data, len(data), out, cap, d, len(d))
    assert n >= 0
    return out.raw[:n]


def compress_frame_block(data, first_block):
    data = bytes(data)
    cap = max_out(len(data))
    out = C.create_string_buffer(max(cap, 1))
    n = lib().lz4o_compress_frame_block(data, len(data), out, cap, 1 if first_block else 0)
    assert n >= 0
    return out.raw[:n]


def decompress(data, cap, dict_data=None, prefill=None):
    """returns ('ok', bytes) or (ErrName, (expected, actual))"""
    data = bytes(data)
    out = C.create_string_buffer(max(cap, 1))
    if prefill is not None:
        C.memset(out, prefill, max(cap, 1))
    d = ErrDetail()
    if dict_data is None:
        n = lib().lz4o_decompress_into(data, len(data), out, cap, C.byref(d))
    else:
        dd = bytes(dict_data)
        n = lib().lz4o_decompress_into_with_dict(data, len(data), out, cap, dd, len(dd), C.byref(d))
    if n < 0:
        return ERR_NAMES[-n], (int(d.expected), int(d.actual))
    return "ok", out.raw[:n]


def frame_info(content_size=None, block_size=0, block_mode=0, block_checksums=False, content_checksum=False):
    fi = FrameInfoO()
    fi.has_content_size = 0 if content_size is None else 1
    fi.content_size = content_size or 0
    fi.block_size, fi.block_mode = int(block_size), int(block_mode)
    fi.block_checksums, fi.content_checksum = int(block_checksums), int(content_checksum)
    return fi


def frame_compress(data, chunks=None, **kw):
    data = bytes(data)
    fi = frame_info(**kw)
    cap = len(data) + len(data) // 100 + (len(data) // 65536 + 2) * 16 + 64
    out = C.create_string_buffer(cap)
    d = ErrDetail()
    if chunks is None:
        n = lib().lz4o_frame_compress(data, len(data), None, 0, C.byref(fi), out, cap, C.byref(d))
    else:
        arr = (C.c_size_t * len(chunks))(*chunks)
        n = lib().lz4o_frame_compress(data, len(data), arr, len(chunks), C.byref(fi), out, cap, C.byref(d))
    if n < 0:
        return -n, (int(d.expected), int(d.actual))
    return 0, out.raw[:n]


def frame_decompress(data, cap):
    data = bytes(data)
    out = C.create_string_buffer(max(cap, 1))
    consumed = C.c_size_t(0)
    d = ErrDetail()
    n = lib().lz4o_frame_decompress(data, len(data), out, cap, C.byref(consumed), C.byref(d))
    if n < 0:
        return -n, (int(d.expected), int(d.actual), int(d.inner)), int(consumed.value)
    return 0, out.raw[:n], int(consumed.value)


def xxh32(data, seed=0):
    data = bytes(data)
    return int(lib().lz4o_xxh32(data, len(data), seed))


# ---- system C liblz4 1.9.3: the cross-implementation anchor of the reference's tests (tests/tests.rs:25-56)
_lz4 = None


def clz4():
    global _lz4
    if _lz4 is None:
        l = C.CDLL("liblz4.so.1")
        l.LZ4_compress_default.restype = C.c_int
        l.LZ4_compress_default.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        l.LZ4_decompress_safe.restype = C.c_int
        l.LZ4_decompress_safe.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        l.LZ4_compressBound.restype = C.c_int
        l.LZ4_compressBound.argtypes = [C.c_int]
        l.LZ4F_compressFrameBound.restype = C.c_size_t
        l.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.c_void_p]
        l.LZ4F_compressFrame.restype = C.c_size_t
        l.LZ4F_compressFrame.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_void_p]
        l.LZ4F_isError.restype = C.c_uint
        l.LZ4F_isError.argtypes = [C.c_size_t]
        l.LZ4F_createDecompressionContext.restype = C.c_size_t
        l.LZ4F_createDecompressionContext.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        l.LZ4F_freeDecompressionContext.restype = C.c_size_t
        l.LZ4F_freeDecompressionContext.argtypes = [C.c_void_p]
        l.LZ4F_decompress.restype = C.c_size_t
        l.LZ4F_decompress.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_size_t), C.c_char_p, C.POINTER(C.c_size_t), C.c_void_p]
        _lz4 = l
    return _lz4


def c_compress(data):
    data = bytes(data)
    l = clz4()
    cap = l.LZ4_compressBound(len(data))
    out = C.create_string_buffer(max(cap, 1))
    n = l.LZ4_compress_default(data, out, len(data), cap)
    assert n > 0 or len(data) == 0
    return out.raw[:n]


def c_decompress(data, size):
    data = bytes(data)
    out = C.create_string_buffer(max(size, 1))
    n = clz4().LZ4_decompress_safe(data, out, len(data), size)
    if n < 0:
        return None
    return out.raw[:n]


class LZ4FPrefs(C.Structure):
    # LZ4F_preferences_t of lz4 1.9.x: frameInfo{blockSizeID, blockMode, contentChecksumFlag, frameType,
    # contentSize(u64), dictID, blockChecksumFlag}, compressionLevel, autoFlush, favorDecSpeed, reserved[3]
    _fields_ = [("blockSizeID", C.c_uint), ("blockMode", C.c_uint), ("contentChecksumFlag", C.c_uint),
                ("frameType", C.c_uint), ("contentSize", C.c_ulonglong), ("dictID", C.c_uint),
                ("blockChecksumFlag", C.c_uint), ("compressionLevel", C.c_int), ("autoFlush", C.c_uint),
                ("favorDecSpeed", C.c_uint), ("reserved", C.c_uint * 3)]


def c_frame_compress(data, independent=True):
    data = bytes(data)
    l = clz4()
    p = LZ4FPrefs()
    p.blockMode = 1 if independent else 0   # LZ4F_blockLinked = 0, LZ4F_blockIndependent = 1
    cap = l.LZ4F_compressFrameBound(len(data), C.byref(p))
    out = C.create_string_buffer(cap)
    n = l.LZ4F_compressFrame(out, cap, data, len(data), C.byref(p))
    assert not l.LZ4F_isError(n)
    return out.raw[:n]


def c_frame_decompress(data, size):
    data = bytes(data)
    l = clz4()
    ctx = C.c_void_p()
    assert not l.LZ4F_isError(l.LZ4F_createDecompressionContext(C.byref(ctx), 100))
    out = C.create_string_buffer(max(size, 1) + 64)
    src_pos, dst_pos = 0, 0
    try:
        while src_pos < len(data):
            src_sz = C.c_size_t(len(data) - src_pos)
            dst_sz = C.c_size_t(len(out) - dst_pos)
            dst_ptr = C.cast(C.byref(out, dst_pos), C.c_char_p)
            r = l.LZ4F_decompress(ctx, dst_ptr, C.byref(dst_sz), data[src_pos:], C.byref(src_sz), None)
            if l.LZ4F_isError(r):
                return None
            src_pos += src_sz.value
            dst_pos += dst_sz.value
            if r == 0:
                break
            if src_sz.value == 0 and dst_sz.value == 0:
                return None
    finally:
        l.LZ4F_freeDecompressionContext(ctx)
    return out.raw[:dst_pos]


# ---- fixtures ---------------------------------------------------------------------------------
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def golden_block(stem):
    with open(os.path.join(GOLDEN, stem + ".lz4blk"), "rb") as f:
        return f.read()


_plain_cache = {}


def fixture_plain(stem):
    """plain bytes of a reference fixture, decoded from its golden block by the ORACLE (CPU tests)."""
    if stem not in _plain_cache:
        m = manifest()[stem]
        st, data = decompress(golden_block(stem), m["plain_len"])
        assert st == "ok" and hashlib.md5(data).hexdigest() == m["plain_md5"]
        _plain_cache[stem] = data
    return _plain_cache[stem]
