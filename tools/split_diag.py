#!/usr/bin/env python3
"""diagnostic: where does a decoder variant differ from the input on a JSON batch?  (tools)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle_api as O
from lz4_flex_amd import _lib as L, workloads
lib = L.load()
n = int(sys.argv[1]); v = int(sys.argv[2]); bpw = int(sys.argv[3]) if len(sys.argv) > 3 else 0
B = 65536
dev = torch.device("cuda", 0)
src = workloads.json_tiles(O.fixture_plain("compression_66k_JSON"), n * B, device=dev)
stride = 72128
comp = torch.empty(n * stride, dtype=torch.uint8, device=dev); back = torch.zeros(n * B, dtype=torch.uint8, device=dev)
ar = torch.arange(n, dtype=torch.int64, device=dev); in_off, comp_off = ar * B, ar * stride
in_len = torch.full((n,), B, dtype=torch.int32, device=dev); cap = torch.full((n,), stride, dtype=torch.int32, device=dev)
clen = torch.zeros(n, dtype=torch.int32, device=dev); st = torch.full((n,), -1, dtype=torch.int32, device=dev)
blen = torch.zeros(n, dtype=torch.int32, device=dev); bst = torch.full((n,), -1, dtype=torch.int32, device=dev)
ctx = C.c_void_p(); assert lib.lz4flex_ctx_create(C.byref(ctx), 0) == 0
p = lambda t: C.c_void_p(t.data_ptr()); stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
assert lib.lz4flex_compress_batch(ctx, p(src), p(in_off), p(in_len), None, n, p(comp), p(comp_off), p(cap), p(clen), p(st), L.MEM_DEVICE, stream) == 0
assert lib.lz4flex_set_tuning(ctx, b"decompress_variant", v) == 0
if bpw: assert lib.lz4flex_set_tuning(ctx, b"decompress_blocks_per_wg", bpw) == 0
for rep in range(3):
    back.zero_(); bst.fill_(-1); blen.zero_()
    assert lib.lz4flex_decompress_batch(ctx, p(comp), p(comp_off), p(clen), n, p(back), p(in_off), p(in_len), p(blen), p(bst), None, L.MEM_DEVICE, stream) == 0
    torch.cuda.synchronize()
    neq = (back.view(n, B) != src.view(n, B))
    badb = torch.nonzero(neq.any(dim=1)).flatten().tolist()
    sts = torch.nonzero(bst != 0).flatten().tolist()
    print("rep", rep, "n", n, "variant", v, "bpw", bpw, "blocks differing:", len(badb), badb[:20], "status != 0:", len(sts), [(i, int(bst[i])) for i in sts[:10]], "len != B:", int((blen != B).sum().item()))
    for i in badb[:3]:
        pos = torch.nonzero(neq[i]).flatten()
        print("   block", i, "first bad byte", int(pos[0]), "last", int(pos[-1]), "count", int(pos.numel()), "out_len", int(blen[i]))
