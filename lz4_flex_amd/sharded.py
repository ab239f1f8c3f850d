"""Frame layer across ranks (BASELINE configs[3]): BlockMode::Independent frames shard
naturally -- every rank owns a contiguous range of blocks, compresses them with the batched kernels, and the
frame is reassembled with ONE exchange step: an all-gather of the per-rank segment sizes, an exclusive
prefix sum, and a variable-size gather of the segments to the root (point-to-point sends over RCCL/xGMI;
`ncclGather` needs equal counts).  The bytes are identical to what a single FrameEncoder produces
(reference src/frame/compress.rs:261-371): block k>0 is compressed in the "continuation" table mode and the
2 GiB reposition rule (:266-271) is a pure function of the global block index.

Linked frames and `content_checksum` do not shard (each block / the running XXH32 depends on everything
before it): replicas only.  torch.distributed is plumbing here: barrier-free, three collectives per frame.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib as L
from .frame import BlockMode, BlockSize, FrameInfo

WINDOW_SIZE = 65536
UNCOMPRESSED_BIT = 0x80000000


def partition(n_blocks, world):
    """contiguous block ranges [lo, hi) per rank, sizes differ by at most one"""
    base, extra = divmod(n_blocks, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def block_flags(first_block, n_local, block_size):
    """per-block table mode of a FrameEncoder that has already written `first_block` blocks of `block_size`:
    FRAME_FIRST when src_stream_offset == 0 (block 0, and the first block after each reposition), else
    FRAME_CONTINUATION (src/frame/compress.rs:266-271, :357-367)."""
    flags = np.empty(n_local, dtype=np.uint32)
    so = 0
    # replay the offsets up to first_block (O(n) integer work, no data)
    limit = 0xFFFFFFFF // 2
    for k in range(first_block + n_local):
        if so + block_size + WINDOW_SIZE >= limit:
            so = 0
        if k >= first_block:
            flags[k - first_block] = L.BLOCK_FRAME_FIRST if so == 0 else L.BLOCK_FRAME_CONTINUATION
        so += block_size
    return flags


# ---- batched block codec on device tensors (C ABI, LZ4FLEX_MEM_DEVICE) -------------------------------------
def _p(t):
    return C.c_void_p(t.data_ptr())


def _packed(dev, **arrays):
    """the small per-block arrays of one launch as ONE device buffer: every array (numpy, or (dtype, count) for an output
    the kernel fills) is laid out 16-byte aligned in a host buffer, uploaded in one transfer, and handed back as typed
    views -- a dozen arange / full / zeros launches and their allocations per call were a tenth of a 1 GiB frame's time"""
    at, lay = 0, {}
    for k, v in arrays.items():
        if isinstance(v, tuple):
            dt, cnt = np.dtype(v[0]), int(v[1])
            lay[k] = (at, dt, cnt, None)
        else:
            v = np.ascontiguousarray(v)
            dt, cnt = v.dtype, int(v.size)
            lay[k] = (at, dt, cnt, v)
        at = (at + dt.itemsize * cnt + 15) // 16 * 16
    host = np.zeros(max(at, 16), dtype=np.uint8)
    for k, (o, dt, cnt, v) in lay.items():
        if v is not None and cnt:
            host[o:o + dt.itemsize * cnt] = v.view(np.uint8).reshape(-1)
    buf = torch.from_numpy(host).to(dev)
    tdt = {"int32": torch.int32, "uint32": torch.int32, "int64": torch.int64, "uint64": torch.int64}
    out = {k: buf[o:o + dt.itemsize * cnt].view(tdt[dt.name]) for k, (o, dt, cnt, v) in lay.items()}
    out["_buf"] = buf
    return out


def _compress_blocks_device(src, block_size, flags, extra=None):
    """compress_blocks_device + the packed descriptor buffer (with `extra` arrays for the caller's next launch) and the
    compressed lengths on the host: ONE read-back brings lengths and status"""
    lib = L.load()
    dev = src.device
    total = src.numel()
    n = (total + block_size - 1) // block_size
    stride = (int(lib.lz4flex_get_maximum_output_size(block_size)) + 63) // 64 * 64
    ar = np.arange(n, dtype=np.int64)
    in_len = np.full(n, block_size, dtype=np.int32)
    if total % block_size:
        in_len[-1] = total % block_size
    d = _packed(dev, in_off=ar * block_size, comp_off=ar * stride, in_len=in_len, cap=np.full(n, stride, dtype=np.int32),
                flags=np.ascontiguousarray(flags, dtype=np.uint32), lens_status=("int32", 2 * n), **(extra or {}))
    d["comp_len"], d["status"] = d["lens_status"][:n], d["lens_status"][n:]
    comp = torch.empty(n * stride, dtype=torch.uint8, device=dev)
    kind = L.MEM_DEVICE | (L.MEM_BIG_BLOCKS if block_size > 65536 else 0)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    with torch.cuda.device(dev):
        rc = lib.lz4flex_compress_batch(None, _p(src), _p(d["in_off"]), _p(d["in_len"]), _p(d["flags"]), n, _p(comp), _p(d["comp_off"]),
                                        _p(d["cap"]), _p(d["comp_len"]), _p(d["status"]), kind, stream)
    if rc:
        raise RuntimeError("lz4flex_compress_batch: %d %s" % (rc, L.last_error()))
    h = d["lens_status"].cpu().numpy()
    if h[n:].any():
        raise RuntimeError("compress status != 0")
    return comp, d, h[:n].astype(np.int64), in_len.astype(np.int64)


def compress_blocks_device(src, block_size, flags):
    """src: uint8 CUDA tensor; returns (comp, comp_off[i64], comp_len[i32], in_len[i32]) on the same device"""
    comp, d, _hl, _il = _compress_blocks_device(src, block_size, flags)
    return comp, d["comp_off"], d["comp_len"], d["in_len"]


def decompress_blocks_device(comp, comp_off, comp_len, out_len_expected, block_size):
    """returns (out, out_len[i32], status[i32]); block i decodes into out[i*block_size : ...]"""
    lib = L.load()
    dev = comp.device
    n = comp_off.numel()
    out = torch.empty(n * block_size, dtype=torch.uint8, device=dev)
    d = _packed(dev, out_off=np.arange(n, dtype=np.int64) * block_size, cap=np.full(n, block_size, dtype=np.int32),
                out_len=("int32", n), status=("int32", n))
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    with torch.cuda.device(dev):
        rc = lib.lz4flex_decompress_batch(None, _p(comp), _p(comp_off), _p(comp_len), n, _p(out), _p(d["out_off"]), _p(d["cap"]),
                                          _p(d["out_len"]), _p(d["status"]), None, L.MEM_DEVICE | (L.MEM_BIG_BLOCKS if block_size > 65536 else 0), stream)
    if rc:
        raise RuntimeError("lz4flex_decompress_batch: %d %s" % (rc, L.last_error()))
    return out, d["out_len"], d["status"]


def xxh32_blocks_device(base, off, length, seed=0):
    """XXH32 of base[off[i] : off[i]+length[i]] for every i, on the device (block checksums)"""
    lib = L.load()
    dev = base.device
    n = off.numel()
    out = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    o64 = off.to(torch.int64).contiguous()
    l32 = length.to(torch.int32).contiguous()
    with torch.cuda.device(dev):
        rc = lib.lz4flex_xxh32_batch_device(_p(base), _p(o64), _p(l32), n, seed, _p(out), stream)
    if rc:
        raise RuntimeError("lz4flex_xxh32_batch_device: %d %s" % (rc, L.last_error()))
    return out.to(torch.int64) & 0xFFFFFFFF


def _le32(word):
    return torch.stack([(word >> s) & 0xFF for s in (0, 8, 16, 24)], dim=1).to(torch.uint8)


# ---- segment assembly ------------------------------------------------------------------------------------
def build_segment(src, comp, comp_off, comp_len, in_len, block_size, block_checksums=False, xxh32_blocks=None):
    """[4-byte block header | payload | (XXH32 of the payload)]* for this rank's blocks; store-raw rule of
    frame/compress.rs:301-306, block checksum :313-316.  CUDA tensors: three kernel launches for any number of blocks
    (lz4flex_frame_assemble_device, csrc/frame_kernels.hip); CPU tensors (the gloo tests, whose codec is the oracle):
    the same layout built with tensor slices."""
    dev = src.device
    n = comp_len.numel()
    if dev.type == "cuda" and xxh32_blocks is None:
        lib = L.load()
        src_off = torch.arange(n, dtype=torch.int64, device=dev) * block_size
        seg = torch.empty(int(src.numel()) + 8 * n + 16, dtype=torch.uint8, device=dev)
        seg_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        scratch = torch.empty(16 * max(n, 1), dtype=torch.uint8, device=dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        il = in_len.to(torch.int32).contiguous()
        cl = comp_len.to(torch.int32).contiguous()
        co = comp_off.to(torch.int64).contiguous()
        with torch.cuda.device(dev):
            rc = lib.lz4flex_frame_assemble_device(_p(src), _p(src_off), _p(il), _p(comp), _p(co), _p(cl), n, int(bool(block_checksums)),
                                                   _p(seg), _p(seg_off), _p(scratch), stream)
        if rc:
            raise RuntimeError("lz4flex_frame_assemble_device: %d %s" % (rc, L.last_error()))
        return seg[:int(seg_off[n].item())]                       # one scalar read-back sizes the exchange
    clen = comp_len.to(torch.int64)
    ilen = in_len.to(torch.int64)
    raw = clen >= ilen
    size = torch.where(raw, ilen, clen)
    per = size + 4 + (4 if block_checksums else 0)
    seg_off = torch.cumsum(per, 0) - per
    total = int(per.sum().item())
    seg = torch.empty(total, dtype=torch.uint8, device=dev)
    hdr = _le32(torch.where(raw, ilen | UNCOMPRESSED_BIT, clen))
    h_seg, h_size, h_raw, h_coff = seg_off.tolist(), size.tolist(), raw.tolist(), comp_off.tolist()
    for i in range(n):
        o = h_seg[i]
        seg[o:o + 4] = hdr[i]
        if h_raw[i]:
            seg[o + 4:o + 4 + h_size[i]] = src[i * block_size:i * block_size + h_size[i]]
        else:
            seg[o + 4:o + 4 + h_size[i]] = comp[h_coff[i]:h_coff[i] + h_size[i]]
    if block_checksums and n:
        sums = (xxh32_blocks or xxh32_blocks_device)(seg, seg_off + 4, size)
        cs = _le32(sums)
        for i in range(n):
            o = h_seg[i] + 4 + h_size[i]
            seg[o:o + 4] = cs[i]
    return seg


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


# Collectives on device tensors go straight to RCCL (backend "nccl").  Under gloo -- the CPU tests, and bench.py's
# oversubscribed functional run with several ranks on one GPU, which RCCL refuses -- device tensors are staged through the host.
def _staged(t, group):
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _bcast(t, src, group):
    if _staged(t, group):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)


def _send(t, dst, group):
    dist.send(t.cpu() if _staged(t, group) else t, dst=dst, group=group)


def _isend(t, dst, group):
    return dist.isend(t.cpu() if _staged(t, group) else t, dst=dst, group=group)


def _recv_into(view, src, group):
    """returns a completion callable"""
    if _staged(view, group):
        h = torch.empty(view.shape, dtype=view.dtype)
        q = dist.irecv(h, src=src, group=group)

        def done():
            q.wait()
            view.copy_(h)
        return done
    q = dist.irecv(view, src=src, group=group)
    return q.wait


def compress_frame_sharded(local, first_block, frame_info, group=None, root=0, compress_blocks=compress_blocks_device,
                           xxh32_blocks=None):
    """Every rank passes the bytes of its contiguous block range (`local`, uint8 tensor) and the global index
    of its first block.  Returns the complete frame (uint8 tensor on the root's device) on `root`, None elsewhere.
    `compress_blocks(src, block_size, flags) -> (comp, comp_off, comp_len, in_len)`."""
    fi = frame_info
    if fi.block_mode != BlockMode.Independent:
        raise ValueError("Linked frames do not shard: every block depends on the previous 64 KiB (replicas only)")
    if fi.content_checksum:
        raise ValueError("content_checksum: one XXH32 over the whole stream is serial (SURVEY H6); not sharded")
    if fi.block_size == BlockSize.Auto or fi.content_size is not None:
        raise ValueError("sharded frames need an explicit block_size and no content_size")
    rank, world = _world(group)
    bs = fi.block_size.get_size()
    dev = local.device
    n_local = (local.numel() + bs - 1) // bs
    tail = 4 if fi.block_checksums else 0
    # Device tensors with the library's own codec: the segment's size is known on the host as soon as the compressed lengths are
    # (store-raw rule, frame/compress.rs:301-306), so the sizes are exchanged BEFORE the segment is assembled and the root
    # assembles its own segment straight into the frame (a 1 GiB rank saved a 300 MB copy and a dozen small launches).
    direct = local.is_cuda and compress_blocks is compress_blocks_device and xxh32_blocks is None
    seg = asm = None
    if n_local and direct:
        flags = block_flags(first_block, n_local, bs)
        comp, d, h_clen, h_ilen = _compress_blocks_device(local, bs, flags, extra={"seg_off": ("int64", n_local + 1), "scratch": ("int64", 2 * n_local)})
        my_size = int(np.where(h_clen >= h_ilen, h_ilen, h_clen).sum()) + n_local * (4 + tail)

        def asm(dst_tensor, at):        # [header | payload | checksum]* of my blocks -> dst_tensor[at : at + my_size]
            lib = L.load()
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            with torch.cuda.device(dev):
                rc = lib.lz4flex_frame_assemble_device(_p(local), _p(d["in_off"]), _p(d["in_len"]), _p(comp), _p(d["comp_off"]), _p(d["comp_len"]),
                                                       n_local, int(bool(fi.block_checksums)), C.c_void_p(dst_tensor.data_ptr() + at),
                                                       _p(d["seg_off"]), _p(d["scratch"]), stream)
            if rc:
                raise RuntimeError("lz4flex_frame_assemble_device: %d %s" % (rc, L.last_error()))
    elif n_local:
        flags = block_flags(first_block, n_local, bs)
        comp, comp_off, comp_len, in_len = compress_blocks(local, bs, flags)
        seg = build_segment(local, comp, comp_off, comp_len, in_len, bs, fi.block_checksums, xxh32_blocks)
        my_size = int(seg.numel())
    else:
        my_size = 0
    # 1) all-gather of the segment sizes, 2) exclusive prefix sum
    if world > 1:
        my = torch.tensor([my_size], dtype=torch.int64, device=dev)
        gdev = torch.device("cpu") if _staged(my, group) else dev
        sizes = [torch.zeros(1, dtype=torch.int64, device=gdev) for _ in range(world)]
        dist.all_gather(sizes, my.to(gdev), group=group)
        sizes = [int(x) for x in torch.cat(sizes).cpu().tolist()]
    else:
        sizes = [my_size]
    header = np.frombuffer(fi.write(), dtype=np.uint8)
    offs = [int(header.size)]
    for sz in sizes[:-1]:
        offs.append(offs[-1] + sz)
    total = offs[-1] + sizes[-1] + 4
    # 3) variable-size gather to the root
    if rank == root:
        frame = torch.empty(total, dtype=torch.uint8, device=dev)
        ends = np.zeros(header.size + 4, dtype=np.uint8)        # header and EndMark (frame/compress.rs:222-224): one small upload
        ends[:header.size] = header
        e = torch.from_numpy(ends).to(dev)
        frame[:header.size] = e[:header.size]
        frame[total - 4:] = e[header.size:]
        reqs = []
        for r in range(world):
            view = frame[offs[r]:offs[r] + sizes[r]]
            if r == rank:
                if asm is not None:
                    asm(frame, offs[r])
                elif seg is not None:
                    view.copy_(seg)
            elif sizes[r]:
                reqs.append(_recv_into(view, r, group))
        for done in reqs:
            done()
        return frame
    if my_size:
        if asm is not None:
            seg = torch.empty(my_size, dtype=torch.uint8, device=dev)
            asm(seg, 0)
        _send(seg, root, group)
    return None


def walk_blocks(frame_host, header_len, block_checksums=False, block_size=None):
    """host-side block-header walk (frame/decompress.rs:231-241): returns [(payload_off, len, raw)], end offset.
    A block longer than the frame's block size is BlockTooBig (:242-247), as in the reference."""
    from .frame import BlockTooBig
    out, p = [], header_len
    n = len(frame_host)
    while True:
        if p + 4 > n:
            raise ValueError("truncated frame")
        w = int.from_bytes(bytes(frame_host[p:p + 4]), "little")
        p += 4
        if w == 0:
            return out, p
        raw = bool(w & UNCOMPRESSED_BIT)
        ln = w & ~UNCOMPRESSED_BIT
        if block_size is not None and ln > block_size:
            raise BlockTooBig()
        if p + ln + (4 if block_checksums else 0) > n:
            raise ValueError("truncated frame")
        out.append((p, ln, raw))
        p += ln + (4 if block_checksums else 0)


def _walk_blocks_device_host(frame, header_len, block_checksums, block_size):
    """walk_blocks for a frame in device memory: the chain of block headers is followed by a kernel
    (lz4flex_frame_walk_device); the table comes back in ONE transfer.  Returns numpy arrays (payload_off[i64],
    len_word[i64]: bit 31 = stored raw) -- 12 bytes per block, not the frame."""
    from .frame import BlockTooBig
    lib = L.load()
    dev = frame.device
    max_blocks = max(1024, 4 * (frame.numel() // block_size) + 16)      # grown below if the frame holds more (tiny blocks)
    while True:
        buf = torch.empty(16 + 12 * max_blocks, dtype=torch.uint8, device=dev)     # info (4 x i32) | off (i64) | len (i32)
        buf[:16] = 0
        base = buf.data_ptr()
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib.lz4flex_frame_walk_device(frame.data_ptr(), frame.numel(), header_len, 1 if block_checksums else 0, block_size, max_blocks,
                                           base + 16, base + 16 + 8 * max_blocks, base, stream)
        if rc != 0:
            raise RuntimeError("lz4flex_frame_walk_device: %d %s" % (rc, L.last_error()))
        h = buf.cpu().numpy()
        n, st = (int(x) for x in h[:8].view(np.int32))
        if st == 3 and max_blocks < (1 << 26):
            max_blocks *= 8
            continue
        break
    if st == 2:
        raise BlockTooBig()
    if st != 0:
        raise ValueError("truncated frame")
    off = h[16:16 + 8 * max_blocks].view(np.int64)[:n].copy()
    ln = h[16 + 8 * max_blocks:].view(np.int32)[:n].astype(np.int64) & 0xFFFFFFFF
    return off, ln


def walk_blocks_device_tensors(frame, header_len, block_checksums, block_size):
    """the same as DEVICE tensors (payload_off[i64], len_word[i64])"""
    off, ln = _walk_blocks_device_host(frame, header_len, block_checksums, block_size)
    return torch.from_numpy(off).to(frame.device), torch.from_numpy(ln).to(frame.device)


def walk_blocks_device(frame, header_len, block_checksums, block_size):
    """the same as a host list [(payload_off, len, raw)] (tests)"""
    off, words = _walk_blocks_device_host(frame, header_len, block_checksums, block_size)
    return [(int(o), int(w) & ~UNCOMPRESSED_BIT, bool(int(w) & UNCOMPRESSED_BIT)) for o, w in zip(off.tolist(), words.tolist())]


def decompress_frame_sharded(frame, group=None, root=0, decompress_blocks=decompress_blocks_device, device=None,
                             xxh32_blocks=None):
    """`frame` (uint8 tensor) is needed on the root only.  The root walks the block headers, every rank
    receives and decodes a contiguous block range.  Returns (local_out tensor, (lo, hi) block range, FrameInfo).
    Only tensors move between the ranks: a 3-word description of the frame, the block table (payload offset and length
    word per block: two broadcasts) and every rank's contiguous byte range (point-to-point); no Python objects.  Each rank
    turns the table into the descriptors of its own launch on the host (numpy, 16 bytes per block) and uploads them once."""
    rank, world = _world(group)
    dev = frame.device if frame is not None else torch.device(device or "cpu")
    off = words = None
    meta = np.zeros(3, dtype=np.int64)                            # block count, BlockSize code, block checksums
    if rank == root:
        fi = FrameInfo.read(bytes(frame[:19].cpu().numpy()))      # validates magic, version, flags, header checksum (header.rs:277-373)
        hdr_len = len(fi.write())
        if fi.legacy_frame or fi.block_mode != BlockMode.Independent or fi.content_checksum:
            raise ValueError("only Independent frames without a content checksum shard")
        if frame.is_cuda:
            off, words = _walk_blocks_device_host(frame, hdr_len, fi.block_checksums, fi.block_size.get_size())
        else:
            blocks, _end = walk_blocks(frame.numpy(), hdr_len, fi.block_checksums, fi.block_size.get_size())
            off = np.array([b[0] for b in blocks], dtype=np.int64)
            words = np.array([b[1] | (UNCOMPRESSED_BIT if b[2] else 0) for b in blocks], dtype=np.int64)
        meta[:] = (off.size, int(fi.block_size), int(bool(fi.block_checksums)))
    if world > 1:
        m = torch.from_numpy(meta).to(dev)
        _bcast(m, root, group)
        meta = m.cpu().numpy()
    nb, bs_code, has_bc = (int(x) for x in meta)
    if world > 1:
        table = torch.empty(2 * nb, dtype=torch.int64, device=dev)
        if rank == root:
            table.copy_(torch.from_numpy(np.concatenate([off, words])))
        if nb:
            _bcast(table, root, group)
        t = table.cpu().numpy()
        off, words = t[:nb], t[nb:]
    bs = BlockSize(bs_code).get_size()
    tail = 4 if has_bc else 0
    ranges = partition(nb, world)
    lo, hi = ranges[rank]
    n = hi - lo
    length = words & ~UNCOMPRESSED_BIT
    israw = (words & UNCOMPRESSED_BIT) != 0

    def span(l2, h2):                                             # the bytes of a block range are contiguous in the frame
        return (int(off[l2]), int(off[h2 - 1] + length[h2 - 1] + tail)) if h2 > l2 else (0, 0)
    a, b = span(lo, hi)
    if rank == root:
        reqs = []
        for r, (l2, h2) in enumerate(ranges):
            if r == rank or l2 == h2:
                continue
            ra, rb = span(l2, h2)
            reqs.append(_isend(frame[ra:rb].contiguous(), r, group))
        local = frame[a:b]
        for q in reqs:
            q.wait()
    else:
        local = torch.empty(b - a, dtype=torch.uint8, device=dev)
        if b > a:
            _recv_into(local, root, group)()
    poff = off[lo:hi] - a                                         # my blocks: payload offset in `local`, length, stored-raw bit
    plen = length[lo:hi]
    praw = israw[lo:hi]
    if has_bc and n:   # verify the block checksums (frame/decompress.rs:255-261,275-278) before decoding
        t_off, t_len = torch.from_numpy(np.ascontiguousarray(poff)).to(dev), torch.from_numpy(np.ascontiguousarray(plen)).to(dev)
        got = (xxh32_blocks or xxh32_blocks_device)(local, t_off, t_len)
        idx = (t_off + t_len).unsqueeze(1) + torch.arange(4, device=dev).unsqueeze(0)
        stored = (local[idx].to(torch.int64) << torch.tensor([0, 8, 16, 24], device=dev)).sum(dim=1)
        if not torch.equal(got.cpu(), stored.cpu()):
            raise RuntimeError("BlockChecksumError")
    out = torch.empty(n * bs, dtype=torch.uint8, device=dev)
    produced = np.zeros(n, dtype=np.int64)
    comp_idx = np.nonzero(~praw)[0]
    raw_idx = np.nonzero(praw)[0]
    nc, nr = int(comp_idx.size), int(raw_idx.size)
    if dev.type == "cuda" and decompress_blocks is decompress_blocks_device:
        # every block goes straight to its place: compressed ones through the batched decoder (out_off = i * bs), stored ones
        # through one batched copy (csrc/frame_kernels.hip); the descriptors of both launches travel in one upload
        lib = L.load()
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        d = _packed(dev, coff=poff[comp_idx], ooff=comp_idx * bs, roff=poff[raw_idx], doff=raw_idx * bs,
                    clen=plen[comp_idx].astype(np.int32), ocap=np.full(nc, bs, dtype=np.int32), rlen=plen[raw_idx].astype(np.int32),
                    dlen_st=("int32", 2 * nc))
        if nc:
            with torch.cuda.device(dev):
                rc = lib.lz4flex_decompress_batch(None, _p(local), _p(d["coff"]), _p(d["clen"]), nc, _p(out), _p(d["ooff"]), _p(d["ocap"]),
                                                  _p(d["dlen_st"][:nc]), _p(d["dlen_st"][nc:]), None,
                                                  L.MEM_DEVICE | (L.MEM_BIG_BLOCKS if bs > 65536 else 0), stream)
            if rc:
                raise RuntimeError("lz4flex_decompress_batch: %d %s" % (rc, L.last_error()))
        if nr:
            with torch.cuda.device(dev):
                rc = lib.lz4flex_copy_batch_device(_p(local), _p(d["roff"]), _p(d["rlen"]), _p(out), _p(d["doff"]), nr, stream)
            if rc:
                raise RuntimeError("lz4flex_copy_batch_device: %d %s" % (rc, L.last_error()))
            produced[raw_idx] = plen[raw_idx]
        if nc:
            h = d["dlen_st"].cpu().numpy()                        # lengths and status: one read-back
            if h[nc:].any():
                raise RuntimeError("DecompressionError in a sharded block")
            produced[comp_idx] = h[:nc]
    else:
        # CPU tensors (the gloo tests; the codec is injected): per-block placement in Python is test plumbing, not the product path
        if nc:
            coff = torch.from_numpy(np.ascontiguousarray(poff[comp_idx]))
            clen = torch.from_numpy(plen[comp_idx].astype(np.int32))
            dec, dlen, st = decompress_blocks(local, coff, clen, None, bs)
            if int((st != 0).sum().item()):
                raise RuntimeError("DecompressionError in a sharded block")
            dl = dlen.tolist()
            for k, i in enumerate(comp_idx.tolist()):
                out[i * bs:i * bs + dl[k]] = dec[k * bs:k * bs + dl[k]]
                produced[i] = dl[k]
        h_off, h_len = poff.tolist(), plen.tolist()
        for i in raw_idx.tolist():
            out[i * bs:i * bs + h_len[i]] = local[h_off[i]:h_off[i] + h_len[i]]
            produced[i] = h_len[i]
    # blocks are full except possibly the frame's last one: compact view
    total = int(produced.sum())
    if n > 1 and bool((produced[:-1] != bs).any()):
        pr = produced.tolist()
        out = torch.cat([out[i * bs:i * bs + pr[i]] for i in range(n)])
    else:
        out = out[:total]
    return out, (lo, hi), FrameInfo(block_size=BlockSize(bs_code))
