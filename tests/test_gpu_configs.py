"""GPU parity tests (-m gpu) shaped like the BASELINE.json configs (at sizes the oracle finishes in seconds,
plus size-independent properties at full block counts)."""
import hashlib
import io

import numpy as np
import pytest
import torch

import corpus
import oracle_api as O

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("exact_encoder")]   # encoder bytes are compared with lz4_flex's: reference-exact mode


@pytest.fixture(scope="module")
def mods():
    from lz4_flex_amd import _lib, block, frame, sharded, workloads
    assert _lib.load().lz4flex_device_count() >= 1
    return block, frame, sharded, workloads


def _device_roundtrip(sharded, src, bs):
    flags = np.zeros((src.numel() + bs - 1) // bs, dtype=np.uint32)
    comp, comp_off, comp_len, in_len = sharded.compress_blocks_device(src, bs, flags)
    out, out_len, st = sharded.decompress_blocks_device(comp, comp_off, comp_len, None, bs)
    torch.cuda.synchronize()
    assert int((st != 0).sum().item()) == 0
    assert torch.equal(out_len.to(torch.int32), in_len.to(torch.int32))
    assert torch.equal(out[:src.numel()], src)
    return comp, comp_off, comp_len


def test_config2_json_blocks_sample_vs_oracle_and_full_roundtrip(mods):
    """configs[1]: 64 KiB JSON tiles. 256 blocks: GPU bytes == oracle bytes; 4096 blocks (256 MiB): device round
    trip bit-exact + ratio (the bench verifies all 16 384)."""
    block, frame, sharded, W = mods
    plain = O.fixture_plain("compression_66k_JSON")
    src = W.json_tiles(plain, 4096 * 65536, device="cuda")
    comp, comp_off, comp_len = _device_roundtrip(sharded, src, 65536)
    ratio = float(comp_len.sum().item()) / src.numel()
    assert 0.225 < ratio < 0.24            # 0.2321 on JSON tiles
    h_src = src[:256 * 65536].cpu().numpy().tobytes()
    h_comp, h_off, h_len = comp.cpu().numpy(), comp_off.cpu().tolist(), comp_len.cpu().tolist()
    for i in range(0, 256, 5):
        exp = O.compress(h_src[i * 65536:(i + 1) * 65536])
        assert bytes(h_comp[h_off[i]:h_off[i] + h_len[i]]) == exp, i


def test_config3_text_tiles(mods):
    """configs[2] substitute: dickens.txt is absent from the reference mount (.MISSING_LARGE_BLOBS), so
    compression_65k.txt (English text) is tiled to 10 MiB = 160 blocks, as SURVEY 8(d) prescribes."""
    block, frame, sharded, W = mods
    text = O.fixture_plain("compression_65k")
    src = W.json_tiles(text, 160 * 65536, device="cuda")
    comp, comp_off, comp_len = _device_roundtrip(sharded, src, 65536)
    ratio = float(comp_len.sum().item()) / src.numel()
    assert 0.55 < ratio < 0.59             # 0.5708
    h_src = src.cpu().numpy().tobytes()
    h_comp, h_off, h_len = comp.cpu().numpy(), comp_off.cpu().tolist(), comp_len.cpu().tolist()
    for i in (0, 1, 77, 159):
        assert bytes(h_comp[h_off[i]:h_off[i] + h_len[i]]) == O.compress(h_src[i * 65536:(i + 1) * 65536])


def test_config4_log_stream_4mb_frame_sharded_world1(mods):
    """configs[3] at 48 MiB: BlockIndependent + Max4MB frame over the synthetic log stream through the sharded
    path (world 1 here; ranks 2/4 are covered with gloo in test_sharded_cpu.py): bytes == oracle FrameEncoder."""
    block, frame, sharded, W = mods
    n = 12 * (4 << 20) + 128 * 1000          # 12 full blocks + a partial one
    src = W.log_stream(0, n, device="cuda")
    fi = frame.FrameInfo(block_size=frame.BlockSize.Max4MB)
    fr = sharded.compress_frame_sharded(src, 0, fi)
    torch.cuda.synchronize()
    host = src.cpu().numpy().tobytes()
    rc, exp = O.frame_compress(host, block_size=7)
    got = fr.cpu().numpy().tobytes()
    assert rc == 0 and got == exp
    assert 0.25 < len(got) / n < 0.33
    out, (lo, hi), _ = sharded.decompress_frame_sharded(fr)
    assert (lo, hi) == (0, 13) and torch.equal(out, src)
    assert O.c_frame_decompress(got, n) == host
    # the io::Write-shaped encoder produces the same frame
    buf = io.BytesIO()
    e = frame.FrameEncoder.with_frame_info(fi, buf)
    e.write_all(host); e.finish()
    assert buf.getvalue() == exp


def test_config5_linked_64k_frames(mods):
    """configs[4] at 2 MiB: the config-2 bytes through a BlockMode::Linked, Max64KB frame (one dependency chain)"""
    block, frame, sharded, W = mods
    plain = O.fixture_plain("compression_66k_JSON")
    data = W.json_tiles(plain, 32 * 65536).numpy().tobytes()
    fi = frame.FrameInfo(block_mode=frame.BlockMode.Linked, block_size=frame.BlockSize.Max64KB)
    buf = io.BytesIO()
    e = frame.FrameEncoder.with_frame_info(fi, buf)
    e.write_all(data); e.finish()
    rc, exp = O.frame_compress(data, block_mode=1, block_size=4)
    assert rc == 0 and buf.getvalue() == exp
    assert frame.FrameDecoder.new(io.BytesIO(exp)).read_to_end() == data
    # linked blocks can reach into the previous block: never worse than the independent frame on this input
    # (the tile period, 66 675 B, exceeds the 65 535 B window, so the gain is small)
    assert len(exp) <= len(O.frame_compress(data, block_mode=0, block_size=4)[1])


def test_big_blocks_device_batch(mods):
    """4 MiB blocks take the u32-table encoder; decode is block-size agnostic"""
    block, frame, sharded, W = mods
    src = W.log_stream(128 * 5000, 3 * (4 << 20), device="cuda")
    comp, comp_off, comp_len = _device_roundtrip(sharded, src, 4 << 20)
    h = src[:4 << 20].cpu().numpy().tobytes()
    got = comp[:int(comp_len[0].item())].cpu().numpy().tobytes()
    assert got == O.compress(h)


def test_xxh32_batch_device_and_checksummed_sharded_frame(mods):
    """device XXH32 (block checksums) == oracle XXH32; a block-checksummed frame built on device == oracle's bytes"""
    block, frame, sharded, W = mods
    data = W.log_stream(0, 128 * 20000, device="cuda")
    lens = [0, 1, 3, 4, 15, 16, 17, 31, 32, 33, 100, 1000, 65536, 70001]
    offs, o = [], 5
    for n in lens:
        offs.append(o); o += n + 3
    got = sharded.xxh32_blocks_device(data, torch.tensor(offs, device="cuda"), torch.tensor(lens, device="cuda")).cpu().tolist()
    host = data.cpu().numpy().tobytes()
    assert got == [O.xxh32(host[a:a + n]) for a, n in zip(offs, lens)]
    src = torch.cat([W.log_stream(0, 128 * 3000, device="cuda"),
                     torch.frombuffer(bytearray(corpus.lcg_bytes(70000, 5, 256, 1)), dtype=torch.uint8).cuda()])
    fi = frame.FrameInfo(block_size=frame.BlockSize.Max64KB, block_checksums=True)
    fr = sharded.compress_frame_sharded(src, 0, fi)
    h = src.cpu().numpy().tobytes()
    rc, exp = O.frame_compress(h, block_size=4, block_checksums=True)
    assert rc == 0 and fr.cpu().numpy().tobytes() == exp
    out, _, _ = sharded.decompress_frame_sharded(fr)
    assert torch.equal(out, src)
    bad = fr.clone(); bad[-9] ^= 0x55          # corrupt the last block's checksum
    with pytest.raises(RuntimeError, match="BlockChecksumError"):
        sharded.decompress_frame_sharded(bad)


def test_device_block_header_walk_equals_host_walk(mods):
    """lz4flex_frame_walk_device (the sharded decoder's header walk, frame/decompress.rs:231-247) against the host walk: offsets,
    lengths, stored-raw bits, with and without block checksums; truncated frames and BlockTooBig are reported, not walked."""
    block, frame, sharded, W = mods
    rng = np.random.default_rng(11)
    data = bytes(W.log_stream(0, 300 * W.LINE * 16).numpy()) + rng.integers(0, 256, 200000, dtype=np.uint8).tobytes()   # compressible + stored blocks
    for bc in (False, True):
        fi = frame.FrameInfo(block_size=frame.BlockSize.Max64KB, block_checksums=bc)
        fr = frame.compress_frame(data, fi)
        hdr = len(fi.write())
        want, end = sharded.walk_blocks(np.frombuffer(fr, dtype=np.uint8), hdr, bc, 65536)
        dev = torch.frombuffer(bytearray(fr), dtype=torch.uint8).cuda()
        got = sharded.walk_blocks_device(dev, hdr, bc, 65536)
        assert got == want and any(r for _, _, r in got) and not all(r for _, _, r in got)
        with pytest.raises(ValueError):
            sharded.walk_blocks_device(dev[:len(fr) - 9].contiguous(), hdr, bc, 65536)
        with pytest.raises(frame.BlockTooBig):
            sharded.walk_blocks_device(dev, hdr, bc, 1000)


def test_native_sharded_entry_points_world1(mods):
    """lz4flex_frame_compress_sharded / lz4flex_frame_decompress_sharded (the C ABI's one-shot multi-GPU entry points, here with a
    world of one rank: no RCCL call is made): the frame == the oracle's FrameEncoder bytes == what lz4_flex_amd/sharded.py
    builds over torch.distributed, decoding returns the stream; block checksums and stored (incompressible) blocks included."""
    import ctypes as C
    from lz4_flex_amd import _lib as L
    block, frame, sharded, W = mods
    lib = L.load()
    rng = np.random.default_rng(5)
    cases = [(W.log_stream(0, 5 * (4 << 20) + 128 * 777, device="cuda"), 7, False),
             (torch.cat([W.log_stream(0, 128 * 3000, device="cuda"), torch.from_numpy(rng.integers(0, 256, 200000, dtype=np.uint8)).cuda(),
                         W.log_stream(128 * 50, 128 * 2000, device="cuda")]), 4, True)]
    for src, bs_code, bc in cases:
        fi = frame.FrameInfo(block_size=frame.BlockSize(bs_code), block_checksums=bc)
        fic = L.FrameInfoC(0, 0, bs_code, 0, 1 if bc else 0, 0, 0)
        n = int(src.numel())
        cap = int(lib.lz4flex_frame_segment_bound(n, C.byref(fic))) + 32
        out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        flen = C.c_uint64(0)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = lib.lz4flex_frame_compress_sharded(None, None, 0, 1, 0, C.c_void_p(src.data_ptr()), n, 0, C.byref(fic), C.c_void_p(out.data_ptr()),
                                                cap, C.byref(flen), stream)
        assert rc == 0, (rc, L.last_error())
        got = out[:flen.value].cpu().numpy().tobytes()
        host = src.cpu().numpy().tobytes()
        rc_o, exp = O.frame_compress(host, block_size=bs_code, block_checksums=bc)
        assert rc_o == 0 and got == exp
        assert got == sharded.compress_frame_sharded(src, 0, fi).cpu().numpy().tobytes()
        bs = fi.block_size.get_size()
        nblk = (n + bs - 1) // bs
        back = torch.zeros(nblk * bs, dtype=torch.uint8, device="cuda")
        olen, first, nb = C.c_uint64(0), C.c_uint64(99), C.c_uint64(0)
        info = L.FrameInfoC()
        det = L.ErrDetail()
        rc = lib.lz4flex_frame_decompress_sharded(None, None, 0, 1, 0, C.c_void_p(out.data_ptr()), flen.value, C.c_void_p(back.data_ptr()),
                                                  nblk * bs, C.byref(olen), C.byref(first), C.byref(nb), C.byref(info), C.byref(det), stream)
        assert rc == 0, (rc, L.last_error())
        assert (olen.value, first.value, nb.value, info.block_size) == (n, 0, nblk, bs_code)
        assert torch.equal(back[:n], src)
        if bc:                                                  # a corrupted block checksum is reported, not decoded
            bad = out.clone(); bad[flen.value - 9] ^= 0x55
            rc = lib.lz4flex_frame_decompress_sharded(None, None, 0, 1, 0, C.c_void_p(bad.data_ptr()), flen.value, C.c_void_p(back.data_ptr()),
                                                      nblk * bs, C.byref(olen), C.byref(first), C.byref(nb), None, None, stream)
            assert rc == -L.FE_BLOCK_CHECKSUM
    # what does not shard is refused
    fic = L.FrameInfoC(0, 0, 4, 1, 0, 0, 0)                     # Linked
    assert lib.lz4flex_frame_compress_sharded(None, None, 0, 1, 0, C.c_void_p(src.data_ptr()), 100, 0, C.byref(fic), C.c_void_p(out.data_ptr()),
                                              cap, C.byref(flen), stream) == -L.E_UNSUPPORTED
