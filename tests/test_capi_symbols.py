"""CPU test: the C-ABI shared library loads and exports every symbol include/lz4flex_amd.h declares,
and fails loudly (no CPU fallback) when there is no GPU.  No compute calls."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lz4flex_amd.h")


@pytest.fixture(scope="module")
def lib_path():
    from lz4_flex_amd import build
    return build.build()   # hipcc cross-compiles gfx950 without a GPU


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lz4flex_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    names = declared_functions()
    for must in ("lz4flex_compress_into", "lz4flex_decompress_into", "lz4flex_compress_batch", "lz4flex_decompress_batch",
                 "lz4flex_frame_encoder_new", "lz4flex_frame_decoder_new", "lz4flex_get_maximum_output_size"):
        assert must in names


def test_library_exports_every_declared_symbol(lib_path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib_path]).decode()
    exported = set(re.findall(r" T (lz4flex_[a-z0-9_]+)", out))
    missing = [n for n in declared_functions() if n not in exported]
    assert not missing, missing


def test_ctypes_binding_covers_header(lib_path):
    from lz4_flex_amd import _lib
    lib = _lib.load()
    for n in declared_functions():
        assert n in _lib.SIGNATURES, n
        assert getattr(lib, n) is not None


def test_library_contains_gfx950_code_object(lib_path):
    data = open(lib_path, "rb").read()
    assert b"gfx950" in data
    assert b"lz4_decompress_blocks_kernel" in data and b"lz4_compress_blocks_kernel" in data


def test_host_logic_without_gpu(lib_path):
    """pure host entry points work anywhere; compute entry points refuse to run without a device"""
    import ctypes as C
    from lz4_flex_amd import _lib, block, frame
    lib = _lib.load()
    assert block.get_maximum_output_size(65536) == 72109
    assert frame.FrameInfo(block_size=frame.BlockSize.Max64KB).write() == bytes([0x04, 0x22, 0x4D, 0x18, 0x60, 0x40, 0x82])
    assert frame.FrameInfo(block_size=frame.BlockSize.Max64KB, block_mode=frame.BlockMode.Linked).write()[-1] == 0xC0
    fi = frame.FrameInfo.read(bytes([0x04, 0x22, 0x4D, 0x18, 0x40, 0x40, 0xC0]))
    assert fi.block_mode == frame.BlockMode.Linked and fi.block_size == frame.BlockSize.Max64KB
    with pytest.raises(frame.HeaderChecksumError):
        frame.FrameInfo.read(bytes([0x04, 0x22, 0x4D, 0x18, 0x40, 0x40, 0xC1]))
    assert lib.lz4flex_xxh32(b"", 0, 0) == 0x02CC5D05
    if lib.lz4flex_device_count() == 0:
        with pytest.raises(block.DeviceError):
            block.compress(b"no gpu, no codec")
        with pytest.raises(block.DeviceError):
            block.decompress(bytes([0x10, 0x61]), 1)
        assert lib.lz4flex_get_tuning(None, b"compress_mode") == -_lib.E_NO_DEVICE      # the default context needs a device too
        assert lib.lz4flex_set_tuning(None, b"compress_mode", 1) == -_lib.E_NO_DEVICE


def test_stale_library_is_detected(lib_path):
    """the build is keyed on a hash of EVERY file under csrc/ and include/ (round 1 shipped a library that
    predated a header edit because a hand-kept dependency list missed it)"""
    from lz4_flex_amd import _lib, build
    assert not build.needs_build()
    assert _lib.load().lz4flex_build_id().decode() == build.source_hash()
    hdrs = [p for p in build.dep_files() if p.endswith(".h")]
    assert len(hdrs) >= 3
    for victim in hdrs:
        with open(victim, "rb") as f:
            orig = f.read()
        try:
            with open(victim, "ab") as f:
                f.write(b"\n// touched by test_stale_library_is_detected\n")
            assert build.needs_build(), victim
        finally:
            with open(victim, "wb") as f:
                f.write(orig)
    assert not build.needs_build()


def test_dispatch_thresholds_are_readable_without_a_device():
    """the decoder dispatch's batch-size table (lz4_device.h) through lz4flex_get_tuning: ascending, ends where the key is refused --
    tests/test_gpu_dispatch_matrix.py builds its size matrix from it"""
    from lz4_flex_amd import _lib
    lib = _lib.load()
    ts = []
    for i in range(32):
        v = lib.lz4flex_get_tuning(None, b"dispatch_threshold_%d" % i)
        if v < 0:
            break
        ts.append(v)
    assert len(ts) >= 5 and ts == sorted(ts) and ts[-1] == 16384
    assert lib.lz4flex_get_tuning(None, b"dispatch_threshold_x") < 0


def test_decoder_configurations_are_listed_without_a_device():
    """lz4flex_get_tuning "decoder_config_<i>": every decoder configuration the build can be pinned to (variant * 1000 + parameter); the
    GPU decoder matrix (tests/test_gpu_block.py DECODERS) and tools/gpu_fuzz.py are generated from it"""
    from lz4_flex_amd import _lib
    lib = _lib.load()
    cs = []
    for i in range(64):
        v = lib.lz4flex_get_tuning(None, b"decoder_config_%d" % i)
        if v < 0:
            break
        cs.append(v)
    assert {1016, 4064, 7000, 8000, 10000, 11000, 13000} <= set(cs) and len(cs) == len(set(cs))
    assert not ({9000, 12000} & set(cs))     # the plan / replay and the fused decoder left the product library (tools builds only)
    assert not ({5000, 6000} & set(cs))      # the wave decoder and its two-wavefront form: replaced by the sequence decoder (13) in round 6
    for gone in (5, 6, 9, 12):
        assert lib.lz4flex_set_tuning(None, b"decompress_variant", gone) < 0
    assert lib.lz4flex_get_tuning(None, b"decoder_config_") < 0


def test_many_frames_entry_points_check_their_arguments_without_a_device():
    """lz4flex_frame_{compress,decompress}_many: nothing to do is success, missing arrays are refused before any device is touched;
    with no device the calls fail loudly (there is no CPU path)"""
    import ctypes as C
    from lz4_flex_amd import _lib, frame
    lib = _lib.load()
    z = (C.c_uint64 * 1)(0)
    st = (C.c_int32 * 1)(0)
    buf = C.create_string_buffer(64)
    assert lib.lz4flex_frame_compress_many(None, buf, z, z, 0, None, buf, z, z, z, st, _lib.MEM_HOST, None) == 0
    assert lib.lz4flex_frame_decompress_many(None, buf, z, z, 0, buf, z, z, z, st, None, _lib.MEM_HOST, None) == 0
    assert lib.lz4flex_frame_compress_many(None, buf, None, z, 1, None, buf, z, z, z, st, _lib.MEM_HOST, None) == -_lib.E_INVALID_ARG
    assert lib.lz4flex_frame_decompress_many(None, buf, z, z, 1, buf, z, z, z, None, None, _lib.MEM_HOST, None) == -_lib.E_INVALID_ARG
    assert frame.compress_frames([]) == [] and frame.decompress_frames([], 10) == []
    if lib.lz4flex_device_count() == 0:
        cap = (C.c_uint64 * 1)(64)
        assert lib.lz4flex_frame_compress_many(None, buf, z, z, 1, None, buf, z, cap, z, st, _lib.MEM_HOST, None) == -_lib.E_NO_DEVICE
        from lz4_flex_amd import block
        with pytest.raises(block.DeviceError):
            frame.compress_frames([b"no gpu, no codec"])
