"""CPU tests (gloo, world_size 2 and 4) of the multi-rank frame path: block partition, table-mode flags,
size all-gather + prefix sum + variable-size gather, header walk + scatter.  The per-rank block codec is the
ORACLE here (no GPU in this container); the assembled frame must be byte-identical to the single-process
FrameEncoder restatement, and to the 1-rank result."""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def oracle_compress_blocks(src, block_size, flags):
    import oracle_api as O
    data = bytes(src.numpy().tobytes())
    n = (len(data) + block_size - 1) // block_size
    stride = (O.max_out(block_size) + 63) // 64 * 64
    comp = torch.zeros(n * stride, dtype=torch.uint8)
    lens, ilens = [], []
    for i in range(n):
        blk = data[i * block_size:(i + 1) * block_size]
        c = O.compress_frame_block(blk, first_block=(int(flags[i]) == 2))
        comp[i * stride:i * stride + len(c)] = torch.frombuffer(bytearray(c), dtype=torch.uint8)
        lens.append(len(c)); ilens.append(len(blk))
    return (comp, torch.arange(n, dtype=torch.int64) * stride, torch.tensor(lens, dtype=torch.int32),
            torch.tensor(ilens, dtype=torch.int32))


def oracle_decompress_blocks(comp, comp_off, comp_len, _unused, block_size):
    import oracle_api as O
    n = comp_off.numel()
    out = torch.zeros(n * block_size, dtype=torch.uint8)
    lens, st = [], []
    raw = bytes(comp.numpy().tobytes())
    for i in range(n):
        o, l = int(comp_off[i]), int(comp_len[i])
        s, data = O.decompress(raw[o:o + l], block_size)
        if s == "ok":
            out[i * block_size:i * block_size + len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8)
            lens.append(len(data)); st.append(0)
        else:
            lens.append(0); st.append(1)
    return out, torch.tensor(lens, dtype=torch.int32), torch.tensor(st, dtype=torch.int32)


def oracle_xxh32_blocks(base, off, length, seed=0):
    import oracle_api as O
    raw = bytes(base.numpy().tobytes())
    return torch.tensor([O.xxh32(raw[int(o):int(o) + int(l)], seed) for o, l in zip(off.tolist(), length.tolist())],
                        dtype=torch.int64)


def _stream(total):
    from lz4_flex_amd import workloads as W
    import oracle_api as O
    log = W.log_stream(0, total // 2 // 128 * 128)
    js = W.json_tiles(O.fixture_plain("compression_66k_JSON"), total - log.numel())
    import corpus
    rnd = torch.frombuffer(bytearray(corpus.lcg_bytes(70000, 99, 256, 1)), dtype=torch.uint8)
    return torch.cat([log, rnd, js])      # the random part exercises the store-raw rule


def _worker(rank, world, port, bs_code, total, q, block_checksums=False):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lz4_flex_amd import sharded as S
        from lz4_flex_amd.frame import BlockSize, FrameInfo
        import oracle_api as O
        torch.manual_seed(0)
        data = _stream(total) if rank == 0 else None
        meta = [None]
        if rank == 0:
            meta = [bytes(data.numpy().tobytes())]
        dist.broadcast_object_list(meta, src=0)
        stream = meta[0]
        bs = BlockSize(bs_code).get_size()
        n_blocks = (len(stream) + bs - 1) // bs
        lo, hi = S.partition(n_blocks, world)[rank]
        local = torch.frombuffer(bytearray(stream[lo * bs:hi * bs]), dtype=torch.uint8) if hi > lo else torch.empty(0, dtype=torch.uint8)
        fi = FrameInfo(block_size=BlockSize(bs_code), block_checksums=block_checksums)
        frame = S.compress_frame_sharded(local, lo, fi, compress_blocks=oracle_compress_blocks, xxh32_blocks=oracle_xxh32_blocks)
        if rank == 0:
            fb = bytes(frame.numpy().tobytes())
            rc, exp = O.frame_compress(stream, block_size=bs_code, block_checksums=block_checksums)
            assert rc == 0 and fb == exp, "sharded frame differs from the single-encoder frame"
            assert O.c_frame_decompress(fb, len(stream)) == stream
        out, (l2, h2), _ = S.decompress_frame_sharded(frame if rank == 0 else None, decompress_blocks=oracle_decompress_blocks,
                                                     xxh32_blocks=oracle_xxh32_blocks)
        assert (l2, h2) == (lo, hi)
        assert bytes(out.numpy().tobytes()) == stream[lo * bs:hi * bs]
        q.put((rank, "ok", hashlib.md5(bytes(frame.numpy().tobytes())).hexdigest() if rank == 0 else ""))
    except Exception as e:   # surface the failure in the parent
        import traceback
        q.put((rank, "fail", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _run(world, bs_code, total, block_checksums=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bs_code, total, q, block_checksums)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
    bad = [r for r in res if r[1] != "ok"]
    assert not bad, bad[0][2]
    return [r for r in res if r[0] == 0][0][2]


def test_partition_and_flags():
    sys.path.insert(0, ROOT)
    from lz4_flex_amd import sharded as S
    assert S.partition(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert S.partition(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    f = S.block_flags(0, 4, 65536)
    assert list(f) == [2, 3, 3, 3]
    assert list(S.block_flags(5, 3, 65536)) == [3, 3, 3]
    # 4 MiB blocks: the table is repositioned (next block behaves like a first block) when
    # src_stream_offset + block_size + 64 KiB >= 2^31 - 1, i.e. before block 511 (frame/compress.rs:266-271)
    f = S.block_flags(0, 1100, 4 << 20)
    firsts = [i for i, v in enumerate(f) if v == 2]
    assert firsts == [0, 511, 1022]
    assert list(S.block_flags(510, 3, 4 << 20)) == [3, 2, 3]


def test_log_stream_is_pinned():
    sys.path.insert(0, ROOT)
    from lz4_flex_amd import workloads as W
    x = W.log_stream(0, 1 << 20)
    assert hashlib.md5(bytes(x.numpy().tobytes())).hexdigest() == "d2c4057d850c2857b4c41c4fe60f23ed"
    y = W.log_stream(128 * 1000, 128 * 50)
    assert torch.equal(y, x[128 * 1000:128 * 1050])


@pytest.mark.parametrize("world,bs_code,bc", [(2, 4, False), (4, 4, False), (2, 5, False), (2, 4, True)])
def test_sharded_frame_matches_single_encoder(world, bs_code, bc):
    total = 9 * 65536 + 12345 if bs_code == 4 else 5 * 262144 + 777
    md5s = {w: _run(w, bs_code, total, bc) for w in (1, world)}
    assert md5s[1] == md5s[world]
