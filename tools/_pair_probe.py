import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, oracle_api as O
from lz4_flex_amd import _lib as L, workloads
lib = L.load(); dev = torch.device("cuda", 0)
p = lambda t: C.c_void_p(t.data_ptr()); stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for data in ("text", "json"):
    for n in (64, 96, 128):
        B = 65536
        src = workloads.json_tiles(O.fixture_plain("compression_65k" if data == "text" else "compression_66k_JSON"), n * B, device=dev)
        stride = 72128
        comp = torch.empty(n * stride, dtype=torch.uint8, device=dev); back = torch.empty(n * B, dtype=torch.uint8, device=dev)
        ar = torch.arange(n, dtype=torch.int64, device=dev); in_off, comp_off = ar * B, ar * stride
        in_len = torch.full((n,), B, dtype=torch.int32, device=dev); cap = torch.full((n,), stride, dtype=torch.int32, device=dev)
        clen = torch.zeros(n, dtype=torch.int32, device=dev); st = torch.full((n,), -1, dtype=torch.int32, device=dev)
        blen = torch.zeros(n, dtype=torch.int32, device=dev); bst = torch.full((n,), -1, dtype=torch.int32, device=dev)
        ctx = C.c_void_p(); assert lib.lz4flex_ctx_create(C.byref(ctx), 0) == 0
        assert lib.lz4flex_compress_batch(ctx, p(src), p(in_off), p(in_len), None, n, p(comp), p(comp_off), p(cap), p(clen), p(st), L.MEM_DEVICE, stream) == 0
        row = []
        for pair in (0, 2):
            assert lib.lz4flex_set_tuning(ctx, b"decompress_variant", 7) == 0
            assert lib.lz4flex_set_tuning(ctx, b"decompress_pcd_pair", pair) == 0
            ts = []
            for r in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                assert lib.lz4flex_decompress_batch(ctx, p(comp), p(comp_off), p(clen), n, p(back), p(in_off), p(in_len), p(blen), p(bst), None, L.MEM_DEVICE, stream) == 0
                e1.record(); torch.cuda.synchronize()
                if r: ts.append(e0.elapsed_time(e1))
            ok = int((bst != 0).sum().item()) == 0 and torch.equal(back, src)
            ts.sort(); row.append("pair %d: %.3f ms %s" % (pair, ts[len(ts) // 2], "" if ok else "WRONG"))
        print(data, n, " | ".join(row), flush=True)
        lib.lz4flex_ctx_destroy(ctx)
