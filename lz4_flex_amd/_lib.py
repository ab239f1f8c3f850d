"""ctypes binding of the C ABI (include/lz4flex_amd.h).  The HIP shared library is the product:
if it is missing or no GPU is usable, calls fail loudly -- there is no CPU path in this package."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblz4flex_amd.so")

# codes (include/lz4flex_amd.h)
E_OUTPUT_TOO_SMALL, E_LITERAL_OUT_OF_BOUNDS, E_EXPECTED_ANOTHER_BYTE, E_OFFSET_ZERO, E_OFFSET_OUT_OF_BOUNDS = 1, 2, 3, 4, 5
FE_COMPRESSION, FE_DECOMPRESSION, FE_IO, FE_UNSUPPORTED_BLOCKSIZE, FE_UNSUPPORTED_VERSION = 16, 17, 18, 19, 20
FE_WRONG_MAGIC, FE_RESERVED_BITS, FE_INVALID_BLOCK_INFO, FE_BLOCK_TOO_BIG, FE_HEADER_CHECKSUM = 21, 22, 23, 24, 25
FE_BLOCK_CHECKSUM, FE_CONTENT_CHECKSUM, FE_SKIPPABLE_FRAME, FE_DICTIONARY_NOT_SUPPORTED = 26, 27, 28, 29
FE_CONTENT_LENGTH, FE_OUTPUT_FULL = 30, 31
E_INVALID_ARG, E_NO_DEVICE, E_HIP, E_NOMEM, E_UNSUPPORTED = 64, 65, 66, 67, 68
MEM_HOST, MEM_DEVICE, MEM_BIG_BLOCKS, MEM_CHAINED = 0, 1, 0x100, 0x200
BLOCK_DEFAULT, BLOCK_FRAME_FIRST, BLOCK_FRAME_CONTINUATION = 0, 2, 3


class ErrDetail(C.Structure):
    _fields_ = [("expected", C.c_uint64), ("actual", C.c_uint64), ("inner", C.c_int32), ("hip_error", C.c_int32)]


class FrameInfoC(C.Structure):
    _fields_ = [("has_content_size", C.c_int32), ("content_size", C.c_uint64), ("block_size", C.c_int32),
                ("block_mode", C.c_int32), ("block_checksums", C.c_int32), ("content_checksum", C.c_int32),
                ("legacy_frame", C.c_int32)]


class ChainBlock(C.Structure):
    _fields_ = [("in_off", C.c_uint64), ("dict_off", C.c_uint64), ("in_len", C.c_uint32), ("in_pos", C.c_uint32),
                ("dict_len", C.c_uint32), ("so", C.c_uint32), ("repos", C.c_uint32), ("flags", C.c_uint32)]


class DecompressExt(C.Structure):
    _fields_ = [("dict_base", C.c_void_p), ("dict_off", C.c_void_p), ("dict_len", C.c_void_p), ("out_pos", C.c_void_p),
                ("chain_prev", C.c_void_p), ("n_chains", C.c_uint32)]


WRITE_FN = C.CFUNCTYPE(C.c_int64, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)
READ_FN = C.CFUNCTYPE(C.c_int64, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)

# every symbol include/lz4flex_amd.h declares: name -> (restype, argtypes)
_VP, _SZ, _I64, _I32, _U32, _U64 = C.c_void_p, C.c_size_t, C.c_int64, C.c_int, C.c_uint32, C.c_uint64
SIGNATURES = {
    "lz4flex_ctx_create": (_I32, [C.POINTER(_VP), _I32]),
    "lz4flex_ctx_destroy": (None, [_VP]),
    "lz4flex_device_count": (_I32, []),
    "lz4flex_version": (C.c_char_p, []),
    "lz4flex_last_error": (C.c_char_p, []),
    "lz4flex_build_id": (C.c_char_p, []),
    "lz4flex_abi_version": (C.c_int, []),
    "lz4flex_get_maximum_output_size": (_SZ, [_SZ]),
    "lz4flex_compress_into": (_I64, [_VP, _SZ, _VP, _SZ]),
    "lz4flex_compress_into_with_dict": (_I64, [_VP, _SZ, _VP, _SZ, _VP, _SZ]),
    "lz4flex_compress_prepend_size": (_I64, [_VP, _SZ, _VP, _SZ]),
    "lz4flex_compress_prepend_size_with_dict": (_I64, [_VP, _SZ, _VP, _SZ, _VP, _SZ]),
    "lz4flex_compress_table_new": (_VP, [_I32]),
    "lz4flex_compress_table_free": (None, [_VP]),
    "lz4flex_compress_table_is_large": (_I32, [_VP]),
    "lz4flex_compress_into_with_table": (_I64, [_VP, _SZ, _VP, _SZ, _VP]),
    "lz4flex_decompress_size_prepended_with_dict": (_I64, [_VP, _SZ, _VP, _SZ, _VP, _SZ, C.POINTER(ErrDetail)]),
    "lz4flex_decompress_into": (_I64, [_VP, _SZ, _VP, _SZ, C.POINTER(ErrDetail)]),
    "lz4flex_decompress_into_with_dict": (_I64, [_VP, _SZ, _VP, _SZ, _VP, _SZ, C.POINTER(ErrDetail)]),
    "lz4flex_uncompressed_size": (_I64, [_VP, _SZ]),
    "lz4flex_decompress_size_prepended": (_I64, [_VP, _SZ, _VP, _SZ, C.POINTER(ErrDetail)]),
    "lz4flex_compress_batch": (_I32, [_VP, _VP, _VP, _VP, _VP, _U32, _VP, _VP, _VP, _VP, _VP, _I32, _VP]),
    "lz4flex_compress_chains": (_I32, [_VP, _VP, C.POINTER(ChainBlock), _U32, _VP, _VP, _U32, _VP, _VP, _VP, _VP, _VP, _VP,
                                        _I32, _VP]),
    "lz4flex_decompress_batch": (_I32, [_VP, _VP, _VP, _VP, _U32, _VP, _VP, _VP, _VP, _VP, _VP, _I32, _VP]),
    "lz4flex_decompress_batch_ex": (_I32, [_VP, _VP, _VP, _VP, _U32, _VP, _VP, _VP, _VP, _VP, _VP,
                                            C.POINTER(DecompressExt), _I32, _VP]),
    "lz4flex_set_tuning": (_I32, [_VP, C.c_char_p, _I32]),
    "lz4flex_get_tuning": (_I32, [_VP, C.c_char_p]),
    "lz4flex_frame_encoder_new": (_VP, [C.POINTER(FrameInfoC), WRITE_FN, _VP]),
    "lz4flex_frame_encoder_write": (_I64, [_VP, _VP, _SZ]),
    "lz4flex_frame_encoder_flush": (_I32, [_VP]),
    "lz4flex_frame_encoder_try_finish": (_I32, [_VP, C.POINTER(ErrDetail)]),
    "lz4flex_frame_encoder_frame_info": (None, [_VP, C.POINTER(FrameInfoC)]),
    "lz4flex_frame_encoder_set_batch_bytes": (_I32, [_VP, _SZ]),
    "lz4flex_frame_encoder_free": (None, [_VP]),
    "lz4flex_frame_decoder_new": (_VP, [READ_FN, _VP]),
    "lz4flex_frame_decoder_read": (_I64, [_VP, _VP, _SZ, C.POINTER(ErrDetail)]),
    "lz4flex_frame_decoder_fill_buf": (_I64, [_VP, C.POINTER(_VP), C.POINTER(ErrDetail)]),
    "lz4flex_frame_decoder_consume": (_I32, [_VP, _SZ]),
    "lz4flex_frame_decoder_set_batch_bytes": (_I32, [_VP, _SZ]),
    "lz4flex_frame_decoder_free": (None, [_VP]),
    "lz4flex_frame_compress": (_I64, [_VP, _SZ, C.POINTER(FrameInfoC), _VP, _SZ, C.POINTER(ErrDetail)]),
    "lz4flex_frame_decompress": (_I64, [_VP, _SZ, _VP, _SZ, C.POINTER(_SZ), C.POINTER(ErrDetail)]),
    "lz4flex_frame_compress_bound": (_SZ, [_SZ, C.POINTER(FrameInfoC)]),
    "lz4flex_frame_info_write": (_I64, [C.POINTER(FrameInfoC), _VP, _SZ]),
    "lz4flex_frame_info_read": (_I64, [_VP, _SZ, C.POINTER(FrameInfoC), C.POINTER(ErrDetail)]),
    "lz4flex_xxh32": (_U32, [_VP, _SZ, _U32]),
    "lz4flex_xxh32_batch_device": (_I32, [_VP, _VP, _VP, _U32, _U32, _VP, _VP]),
    "lz4flex_frame_assemble_device": (_I32, [_VP, _VP, _VP, _VP, _VP, _VP, _U32, _I32, _VP, _VP, _VP, _VP]),
    "lz4flex_copy_batch_device": (_I32, [_VP, _VP, _VP, _VP, _VP, _U32, _VP]),
    "lz4flex_frame_walk_device": (_I32, [_VP, _U64, _U32, _I32, _U32, _U32, _VP, _VP, _VP, _VP]),
    "lz4flex_frame_compress_many": (_I32, [_VP, _VP, _VP, _VP, _U32, C.POINTER(FrameInfoC), _VP, _VP, _VP, _VP, _VP, _I32, _VP]),
    "lz4flex_frame_decompress_many": (_I32, [_VP, _VP, _VP, _VP, _U32, _VP, _VP, _VP, _VP, _VP, _VP, _I32, _VP]),
    "lz4flex_frame_segment_bound": (_U64, [_U64, C.POINTER(FrameInfoC)]),
    "lz4flex_frame_compress_sharded": (_I32, [_VP, _VP, _I32, _I32, _I32, _VP, _U64, _U64, C.POINTER(FrameInfoC), _VP, _U64,
                                               C.POINTER(_U64), _VP]),
    "lz4flex_frame_decompress_sharded": (_I32, [_VP, _VP, _I32, _I32, _I32, _VP, _U64, _VP, _U64, C.POINTER(_U64), C.POINTER(_U64),
                                                 C.POINTER(_U64), C.POINTER(FrameInfoC), C.POINTER(ErrDetail), _VP]),
}

_lib = None


def load():
    """Load liblz4flex_amd.so (raises if it was not built).  torch is imported first so that the
    process uses ONE HIP runtime: torch's bundled libamdhip64 has the same SONAME the library needs."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("LZ4FLEX_LIB") or LIB_PATH     # LZ4FLEX_LIB: tools only (kernel experiments, build.build_variant)
    if not os.path.exists(path):
        raise ImportError(
            "lz4_flex_amd: %s is missing. Build it with `python -m lz4_flex_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    if not os.environ.get("LZ4FLEX_NO_TORCH"):   # torch-free tools load libamdhip64 themselves
        try:
            import torch  # noqa: F401  (HIP runtime unification; plumbing only)
        except Exception:
            pass
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().lz4flex_last_error().decode(errors="replace")
