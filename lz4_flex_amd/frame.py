"""lz4_flex::frame, MI355X edition (reference src/frame/{compress,decompress,header,mod}.rs).

FrameEncoder wraps a writer (anything with .write(bytes)), FrameDecoder wraps a reader (anything with
.read(n)); names, defaults and error behaviour follow the reference.  Framing logic is host C++
(lz4_flex_amd/csrc/frame.cpp); block bytes come from the batched HIP kernels."""
import ctypes as C
import enum

from . import _lib as L
from .block import DeviceError


class BlockSize(enum.IntEnum):
    """frame::BlockSize (header.rs:36-55)"""
    Auto = 0
    Max64KB = 4
    Max256KB = 5
    Max1MB = 6
    Max4MB = 7
    Max8MB = 8

    def get_size(self):
        return {4: 64 << 10, 5: 256 << 10, 6: 1 << 20, 7: 4 << 20, 8: 8 << 20}[int(self)]


class BlockMode(enum.IntEnum):
    """frame::BlockMode (header.rs:80-91)"""
    Independent = 0
    Linked = 1


class FrameInfo:
    """frame::FrameInfo (header.rs:128-192); builder-style setters return self like the Rust ones."""

    def __init__(self, content_size=None, block_size=BlockSize.Auto, block_mode=BlockMode.Independent,
                 block_checksums=False, content_checksum=False, legacy_frame=False):
        self.content_size = content_size
        self.block_size = BlockSize(block_size)
        self.block_mode = BlockMode(block_mode)
        self.block_checksums = bool(block_checksums)
        self.content_checksum = bool(content_checksum)
        self.legacy_frame = bool(legacy_frame)

    @classmethod
    def new(cls):
        return cls()

    def _c(self):
        c = L.FrameInfoC()
        c.has_content_size = 0 if self.content_size is None else 1
        c.content_size = 0 if self.content_size is None else int(self.content_size)
        c.block_size = int(self.block_size)
        c.block_mode = int(self.block_mode)
        c.block_checksums = int(self.block_checksums)
        c.content_checksum = int(self.content_checksum)
        c.legacy_frame = int(self.legacy_frame)
        return c

    @classmethod
    def _from_c(cls, c):
        return cls(c.content_size if c.has_content_size else None, BlockSize(c.block_size), BlockMode(c.block_mode),
                   bool(c.block_checksums), bool(c.content_checksum), bool(c.legacy_frame))

    def write(self):
        """FrameInfo::write (header.rs:232-275) -> header bytes"""
        buf = (C.c_uint8 * 19)()
        c = self._c()
        n = L.load().lz4flex_frame_info_write(C.byref(c), C.cast(buf, C.c_void_p), 19)
        if n < 0:
            raise _frame_error(int(-n), L.ErrDetail())
        return bytes(buf[:n])

    @classmethod
    def read(cls, data):
        """FrameInfo::read (header.rs:277-373)"""
        c = L.FrameInfoC()
        d = L.ErrDetail()
        b = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(bytes(data) or b"\0")
        n = L.load().lz4flex_frame_info_read(C.cast(b, C.c_void_p), len(data), C.byref(c), C.byref(d))
        if n < 0:
            raise _frame_error(int(-n), d)
        return cls._from_c(c)

    def __repr__(self):
        return ("FrameInfo(content_size=%r, block_size=%s, block_mode=%s, block_checksums=%r, content_checksum=%r, "
                "legacy_frame=%r)" % (self.content_size, self.block_size.name, self.block_mode.name,
                                      self.block_checksums, self.content_checksum, self.legacy_frame))


class Error(Exception):
    """frame::Error (mod.rs:35-72)"""


class CompressionError(Error): pass
class DecompressionError(Error):
    def __init__(self, inner, expected=0, actual=0):
        super().__init__("DecompressionError(%s)" % inner)
        self.inner, self.expected, self.actual = inner, expected, actual
class IoError(Error): pass
class UnsupportedBlocksize(Error): pass
class UnsupportedVersion(Error): pass
class WrongMagicNumber(Error): pass
class ReservedBitsSet(Error): pass
class InvalidBlockInfo(Error): pass
class BlockTooBig(Error): pass
class HeaderChecksumError(Error): pass
class BlockChecksumError(Error): pass
class ContentChecksumError(Error): pass
class SkippableFrame(Error):
    def __init__(self, length):
        super().__init__("SkippableFrame(%d)" % length)
        self.length = length
class DictionaryNotSupported(Error): pass
class ContentLengthError(Error):
    def __init__(self, expected, actual):
        super().__init__("ContentLengthError { expected: %d, actual: %d }" % (expected, actual))
        self.expected, self.actual = expected, actual


_BLOCK_NAMES = {1: "OutputTooSmall", 2: "LiteralOutOfBounds", 3: "ExpectedAnotherByte", 4: "OffsetZero",
                5: "OffsetOutOfBounds"}
_SIMPLE = {L.FE_COMPRESSION: CompressionError, L.FE_IO: IoError, L.FE_WRONG_MAGIC: WrongMagicNumber,
           L.FE_RESERVED_BITS: ReservedBitsSet, L.FE_INVALID_BLOCK_INFO: InvalidBlockInfo,
           L.FE_BLOCK_TOO_BIG: BlockTooBig, L.FE_HEADER_CHECKSUM: HeaderChecksumError,
           L.FE_BLOCK_CHECKSUM: BlockChecksumError, L.FE_CONTENT_CHECKSUM: ContentChecksumError,
           L.FE_DICTIONARY_NOT_SUPPORTED: DictionaryNotSupported, L.FE_UNSUPPORTED_BLOCKSIZE: UnsupportedBlocksize,
           L.FE_UNSUPPORTED_VERSION: UnsupportedVersion}


def _frame_error(code, d):
    if code == L.FE_DECOMPRESSION:
        return DecompressionError(_BLOCK_NAMES.get(d.inner, str(d.inner)), d.expected, d.actual)
    if code == L.FE_SKIPPABLE_FRAME:
        return SkippableFrame(int(d.expected))
    if code == L.FE_CONTENT_LENGTH:
        return ContentLengthError(int(d.expected), int(d.actual))
    if code in _SIMPLE:
        return _SIMPLE[code]()
    return DeviceError("lz4flex error %d: %s" % (code, L.last_error()))


class FrameEncoder:
    """frame::FrameEncoder<W> (compress.rs:62-206, io::Write :374-404)."""

    def __init__(self, wtr, frame_info=None):
        lib = L.load()
        self._w = wtr
        self._exc = None

        def _cb(_user, buf, n):
            try:
                self._w.write(C.string_at(buf, n))
                return n
            except Exception as e:  # surfaces as frame::Error::IoError
                self._exc = e
                return -1
        self._cb = L.WRITE_FN(_cb)
        fi = (frame_info or FrameInfo())._c()
        self._h = lib.lz4flex_frame_encoder_new(C.byref(fi), self._cb, None)
        if not self._h:
            raise MemoryError("lz4flex_frame_encoder_new")

    @classmethod
    def new(cls, wtr):
        return cls(wtr)

    @classmethod
    def with_frame_info(cls, frame_info, wtr):
        return cls(wtr, frame_info)

    def set_batch_bytes(self, n):
        rc = L.load().lz4flex_frame_encoder_set_batch_bytes(self._h, n)
        if rc:
            raise ValueError("set_batch_bytes")

    def frame_info(self):
        c = L.FrameInfoC()
        L.load().lz4flex_frame_encoder_frame_info(self._h, C.byref(c))
        return FrameInfo._from_c(c)

    def _check(self, rc, d=None):
        if rc < 0:
            if self._exc is not None:
                e, self._exc = self._exc, None
                raise IoError(str(e)) from e
            raise _frame_error(int(-rc), d or L.ErrDetail())

    def write(self, buf):
        b = bytes(buf)
        arr = (C.c_uint8 * max(len(b), 1)).from_buffer_copy(b or b"\0")
        r = L.load().lz4flex_frame_encoder_write(self._h, C.cast(arr, C.c_void_p), len(b))
        self._check(r)
        return int(r)

    write_all = write

    def flush(self):
        self._check(L.load().lz4flex_frame_encoder_flush(self._h))

    def try_finish(self):
        d = L.ErrDetail()
        self._check(L.load().lz4flex_frame_encoder_try_finish(self._h, C.byref(d)), d)

    def finish(self):
        self.try_finish()
        return self._w

    def get_ref(self):
        return self._w

    get_mut = get_ref
    into_inner = get_ref

    def auto_finish(self):
        return AutoFinishEncoder(self)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                L.load().lz4flex_frame_encoder_free(h)
            except Exception:
                pass


class AutoFinishEncoder:
    """frame::AutoFinishEncoder (compress.rs:417-448): finishes the stream when closed / collected."""

    def __init__(self, enc):
        self._e = enc

    def write(self, buf):
        return self._e.write(buf)

    def flush(self):
        self._e.flush()

    def close(self):
        e, self._e = self._e, None
        if e is not None:
            try:
                e.try_finish()
            except Exception:
                pass

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class FrameDecoder:
    """frame::FrameDecoder<R> (decompress.rs:48-107, io::Read :352-408)."""

    def __init__(self, rdr):
        lib = L.load()
        self._r = rdr
        self._exc = None

        def _cb(_user, buf, n):
            try:
                data = self._r.read(n)
                k = len(data)
                if k:
                    C.memmove(buf, data, k)
                return k
            except Exception as e:
                self._exc = e
                return -1
        self._cb = L.READ_FN(_cb)
        self._h = lib.lz4flex_frame_decoder_new(self._cb, None)
        if not self._h:
            raise MemoryError("lz4flex_frame_decoder_new")

    @classmethod
    def new(cls, rdr):
        return cls(rdr)

    def set_batch_bytes(self, n):
        if L.load().lz4flex_frame_decoder_set_batch_bytes(self._h, n):
            raise ValueError("set_batch_bytes")

    def read(self, n=-1):
        """io::Read::read: up to n bytes; b'' at the end of a frame / EOF.  n<0 = read_to_end."""
        if n is None or n < 0:
            return self.read_to_end()
        buf = (C.c_uint8 * max(n, 1))()
        d = L.ErrDetail()
        r = L.load().lz4flex_frame_decoder_read(self._h, C.cast(buf, C.c_void_p), n, C.byref(d))
        if r < 0:
            if self._exc is not None:
                e, self._exc = self._exc, None
                raise IoError(str(e)) from e
            raise _frame_error(int(-r), d)
        return bytes(buf[:r])

    def fill_buf(self):
        """io::BufRead::fill_buf (frame/decompress.rs:410-416): the decoded bytes not consumed yet (b"" at the end)"""
        p = C.c_void_p()
        d = L.ErrDetail()
        r = L.load().lz4flex_frame_decoder_fill_buf(self._h, C.byref(p), C.byref(d))
        if r < 0:
            raise _frame_error(int(-r), d)
        return C.string_at(p, r) if r else b""

    def consume(self, amt):
        """io::BufRead::consume (:418-421)"""
        if L.load().lz4flex_frame_decoder_consume(self._h, int(amt)) != 0:
            raise ValueError("consume(%d): more than fill_buf returned" % amt)

    def read_to_end(self):
        """io::Read::read_to_end (decompress.rs:385-399): until a read returns 0 (one frame)."""
        out = []
        while True:
            b = self.read(1 << 20)
            if not b:
                break
            out.append(b)
        return b"".join(out)

    def get_ref(self):
        return self._r

    get_mut = get_ref
    into_inner = get_ref

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                L.load().lz4flex_frame_decoder_free(h)
            except Exception:
                pass


_new_bytes = C.pythonapi.PyBytes_FromStringAndSize      # an uninitialised bytes object the library writes into: no zero-fill,
_new_bytes.restype = C.py_object                         # no copy out (a 4 MiB ctypes array costs 0.35 ms to create and as much to
_new_bytes.argtypes = [C.c_char_p, C.c_ssize_t]          # turn into bytes -- more than the GPU work on it)


def _as_bytes(data):
    return data if isinstance(data, bytes) else bytes(data)


def compress_frame(data, frame_info=None):
    """one-shot: FrameEncoder::with_frame_info + write_all + finish over flat buffers"""
    lib = L.load()
    fi = (frame_info or FrameInfo())._c()
    b = _as_bytes(data)
    cap = int(lib.lz4flex_frame_compress_bound(len(b), C.byref(fi)))
    out = _new_bytes(None, cap)
    d = L.ErrDetail()
    r = lib.lz4flex_frame_compress(b, len(b), C.byref(fi), out, cap, C.byref(d))
    if r < 0:
        raise _frame_error(int(-r), d)
    return out[:r]


def decompress_frame(data, max_size):
    """one-shot: FrameDecoder::new + read_to_end (first frame). Returns (bytes, consumed)."""
    lib = L.load()
    b = _as_bytes(data)
    out = _new_bytes(None, max(max_size, 1))
    d = L.ErrDetail()
    consumed = C.c_size_t(0)
    r = lib.lz4flex_frame_decompress(b, len(b), out, max_size, C.byref(consumed), C.byref(d))
    if r < 0:
        raise _frame_error(int(-r), d)
    return out[:r], int(consumed.value)         # (the whole object when the frame filled it: no copy)


# ---- many frames at once (include/lz4flex_amd.h "many frames at once": N streams, a frame each, one batch) ---------------------
def _u64(values):
    return (C.c_uint64 * max(len(values), 1))(*values)


def compress_frames(streams, frame_info=None):
    """N host buffers -> N frames (what a FrameEncoder per stream would write), all blocks of all streams in one batch.
    Returns a list of bytes; raises the first stream's error if one failed."""
    lib = L.load()
    fi = (frame_info or FrameInfo())._c()
    bufs = [_as_bytes(s) for s in streams]
    n = len(bufs)
    if n == 0:
        return []
    src = b"".join(bufs)
    in_len = [len(b) for b in bufs]
    in_off, at = [], 0
    for v in in_len:
        in_off.append(at); at += v
    caps = [int(lib.lz4flex_frame_compress_bound(v, C.byref(fi))) for v in in_len]
    out_off, at = [], 0
    for v in caps:
        out_off.append(at); at += v
    out = _new_bytes(None, max(at, 1))
    out_len = (C.c_uint64 * n)()
    status = (C.c_int32 * n)()
    r = lib.lz4flex_frame_compress_many(None, src, _u64(in_off), _u64(in_len), n, C.byref(fi), out, _u64(out_off), _u64(caps), out_len, status,
                                        L.MEM_HOST, None)
    if r != 0:
        raise DeviceError("lz4flex error %d: %s" % (-r, L.last_error()))
    for i in range(n):
        if status[i] != 0:
            raise _frame_error(int(-status[i]), L.ErrDetail())
    return [out[out_off[i]:out_off[i] + int(out_len[i])] for i in range(n)]


def decompress_frames(frames, max_sizes, return_errors=False):
    """N frames (host bytes) -> N byte strings, all blocks of all frames in one batch (Linked frames: N chains side by side).
    max_sizes: an int or one per frame.  return_errors: failed streams come back as exception objects instead of raising."""
    lib = L.load()
    bufs = [_as_bytes(f) for f in frames]
    n = len(bufs)
    if n == 0:
        return []
    caps = [int(max_sizes)] * n if isinstance(max_sizes, int) else [int(v) for v in max_sizes]
    src = b"".join(bufs) or b"\0"
    in_len = [len(b) for b in bufs]
    in_off, at = [], 0
    for v in in_len:
        in_off.append(at); at += v
    out_off, at = [], 0
    for v in caps:
        out_off.append(at); at += v
    out = _new_bytes(None, max(at, 1))
    out_len = (C.c_uint64 * n)()
    status = (C.c_int32 * n)()
    detail = (L.ErrDetail * n)()
    r = lib.lz4flex_frame_decompress_many(None, src, _u64(in_off), _u64(in_len), n, out, _u64(out_off), _u64(caps), out_len, status, detail,
                                          L.MEM_HOST, None)
    if r != 0:
        raise DeviceError("lz4flex error %d: %s" % (-r, L.last_error()))
    res = []
    for i in range(n):
        if status[i] != 0:
            err = _frame_error(int(-status[i]), detail[i])
            if not return_errors:
                raise err
            res.append(err)
        else:
            res.append(out[out_off[i]:out_off[i] + int(out_len[i])])
    return res


def compress_frames_device(src, in_off, in_len, frame_info, dst, out_off, out_cap, stream=None):
    """device-resident streams: src / dst are torch uint8 tensors on the GPU, offsets and lengths plain Python sequences.
    Returns (out_len list, status list); the call has completed when it returns."""
    lib = L.load()
    fi = (frame_info or FrameInfo())._c()
    n = len(in_off)
    out_len = (C.c_uint64 * max(n, 1))()
    status = (C.c_int32 * max(n, 1))()
    r = lib.lz4flex_frame_compress_many(None, C.c_void_p(src.data_ptr()), _u64(in_off), _u64(in_len), n, C.byref(fi), C.c_void_p(dst.data_ptr()),
                                        _u64(out_off), _u64(out_cap), out_len, status, L.MEM_DEVICE, C.c_void_p(stream) if stream else None)
    if r != 0:
        raise DeviceError("lz4flex error %d: %s" % (-r, L.last_error()))
    return list(out_len[:n]), list(status[:n])


def decompress_frames_device(src, in_off, in_len, dst, out_off, out_cap, stream=None):
    """device-resident frames -> device-resident streams; returns (out_len list, status list)"""
    lib = L.load()
    n = len(in_off)
    out_len = (C.c_uint64 * max(n, 1))()
    status = (C.c_int32 * max(n, 1))()
    r = lib.lz4flex_frame_decompress_many(None, C.c_void_p(src.data_ptr()), _u64(in_off), _u64(in_len), n, C.c_void_p(dst.data_ptr()),
                                          _u64(out_off), _u64(out_cap), out_len, status, None, L.MEM_DEVICE, C.c_void_p(stream) if stream else None)
    if r != 0:
        raise DeviceError("lz4flex error %d: %s" % (-r, L.last_error()))
    return list(out_len[:n]), list(status[:n])
