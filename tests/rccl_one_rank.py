"""The C ABI's multi-rank frame entry points (csrc/sharded.cpp) through the REAL librccl with a communicator of one rank
(LZ4FLEX_FORCE_COLLECTIVES=1: ncclAllGather, ncclBroadcast and a grouped ncclSend + ncclRecv to itself are all really called).
Run as a process of its own by tests/test_gpu_sharded_native.py (the library resolves its RCCL once per process, and the other tests
of that file load a mock).  Prints 'RCCL-ONE-RANK OK ...' or raises."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["LZ4FLEX_FORCE_COLLECTIVES"] = "1"
os.environ.pop("LZ4FLEX_RCCL_LIB", None)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def main():
    import torch
    import oracle_api as O
    from lz4_flex_amd import _lib as L, workloads
    rccl = None
    for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"):
        try:
            rccl = C.CDLL(name, mode=C.RTLD_GLOBAL)
            break
        except OSError:
            continue
    if rccl is None:
        print("RCCL-ONE-RANK SKIP: librccl not found")
        return
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    uid = UniqueId()
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rc = rccl.ncclGetUniqueId(C.byref(uid))
    assert rc == 0, "ncclGetUniqueId %d" % rc
    comm = C.c_void_p()
    rc = rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0)
    assert rc == 0 and comm.value, "ncclCommInitRank %d" % rc
    lib = L.load()
    checked = 0
    for mode in ("exact", "fast"):
        for bs_code, n_blocks, tail, bc in ((4, 11, 12288, False), (4, 3, 0, True), (7, 3, 128 * 999, False)):
            bs = {4: 65536, 7: 4 << 20}[bs_code]
            total = n_blocks * bs + tail
            src = workloads.log_stream(0, total, device="cuda")
            host = src.cpu().numpy().tobytes()
            fic = L.FrameInfoC(0, 0, bs_code, 0, 1 if bc else 0, 0, 0)
            ctx = C.c_void_p()
            assert lib.lz4flex_ctx_create(C.byref(ctx), 0) == 0
            assert lib.lz4flex_set_tuning(ctx, b"compress_mode", 1 if mode == "exact" else 0) == 0
            cap = int(lib.lz4flex_frame_segment_bound(total, C.byref(fic))) + 64
            frame = torch.zeros(cap, dtype=torch.uint8, device="cuda")
            flen = C.c_uint64(0)
            rc = lib.lz4flex_frame_compress_sharded(ctx, comm, 0, 1, 0, C.c_void_p(src.data_ptr()), total, 0, C.byref(fic), C.c_void_p(frame.data_ptr()),
                                                    cap, C.byref(flen), None)
            assert rc == 0, (rc, L.last_error())
            got = frame[:flen.value].cpu().numpy().tobytes()
            rc_o, back, used = O.frame_decompress(got, total)
            assert rc_o == 0 and back == host and used == len(got), "the oracle's FrameDecoder does not return the stream"
            if mode == "exact":
                rc_e, exp = O.frame_compress(host, block_size=bs_code, block_checksums=bc)
                assert rc_e == 0 and got == exp, "frame != the oracle's FrameEncoder bytes"
            out = torch.zeros(((total + bs - 1) // bs) * bs, dtype=torch.uint8, device="cuda")
            olen, first, nblk = C.c_uint64(0), C.c_uint64(99), C.c_uint64(99)
            rc = lib.lz4flex_frame_decompress_sharded(ctx, comm, 0, 1, 0, C.c_void_p(frame.data_ptr()), flen.value, C.c_void_p(out.data_ptr()),
                                                      int(out.numel()), C.byref(olen), C.byref(first), C.byref(nblk), None, None, None)
            assert rc == 0, (rc, L.last_error())
            assert (first.value, nblk.value, olen.value) == (0, (total + bs - 1) // bs, total)
            assert out[:total].cpu().numpy().tobytes() == host
            lib.lz4flex_ctx_destroy(ctx)
            checked += 1
    rccl.ncclCommDestroy(comm)
    print("RCCL-ONE-RANK OK: %d frames through ncclAllGather / ncclBroadcast / grouped ncclSend + ncclRecv of the real librccl" % checked)


if __name__ == "__main__":
    main()
