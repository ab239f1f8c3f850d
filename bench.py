#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config[1]: 1 GiB = 16 384 independent 64 KiB JSON blocks,
block format, compress + decompress on MI355X.

A "step" is one pass of the hot path over the batch: one batched compress launch (1 GiB -> LZ4 blocks)
followed by one batched decompress launch (those blocks -> 1 GiB), inputs resident in HBM.
`value` = uncompressed MiB carried through that round trip per second, whole job (all ranks).
Per-kernel MiB/s, ratio and roofline numbers ride along in the same JSON line.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Multi-GPU: blocks are independent, so ranks own disjoint block ranges (weak scaling: 1 GiB per GPU) and
the data path has no collective; torch.distributed (RCCL) only provides the barrier and the max-reduce
of the timings.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BLOCK = 65536
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec


def load_json_fixture(block_mod):
    """The reference's benches/compression_66k_JSON.txt, recovered by decoding its golden LZ4 block with
    the GPU codec itself and checked against the md5 recorded from the reference file."""
    g = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(g, "manifest.json")) as f:
        m = json.load(f)["compression_66k_JSON"]
    with open(os.path.join(g, "compression_66k_JSON.lz4blk"), "rb") as f:
        blk = f.read()
    plain = block_mod.decompress(blk, m["plain_len"])
    assert hashlib.md5(plain).hexdigest() == m["plain_md5"], "fixture md5 mismatch"
    return plain


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=16384, help="64 KiB blocks per GPU (16384 = 1 GiB)")
    ap.add_argument("--decompress-lanes", type=int, default=0)
    ap.add_argument("--compress-lanes", type=int, default=0)
    ap.add_argument("--in-pad", type=int, default=0, help="diagnostic: bytes of padding between input blocks (needs --no-verify)")
    ap.add_argument("--compress-variant", type=int, default=0)
    ap.add_argument("--decompress-variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="kernel experiments only: skip the bit-exact check (never for reported numbers)")
    ap.add_argument("--ablate", type=int, default=0, help="kernel timing ablations (wrong output; implies --no-verify)")
    ap.add_argument("--only", choices=["both", "compress", "decompress"], default="both",
                    help="profiling aid: run only one kernel in the timed steps (value then covers that kernel only)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from lz4_flex_amd import _lib, block

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs a GPU: the HIP kernels are the product"
    ndev = torch.cuda.device_count()
    oversub = world > ndev          # functional test of the multi-process path on a box with fewer GPUs than ranks
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if oversub:
            dist.init_process_group("gloo")        # RCCL refuses two ranks on one device; timing collectives are tiny
        else:
            dist.init_process_group("nccl", device_id=dev)
    lib = _lib.load()
    ctx = C.c_void_p()
    rc = lib.lz4flex_ctx_create(C.byref(ctx), dev_index)
    assert rc == 0, _lib.last_error()
    if args.decompress_lanes:
        assert lib.lz4flex_set_tuning(ctx, b"decompress_lanes", args.decompress_lanes) == 0
    if args.compress_lanes:
        assert lib.lz4flex_set_tuning(ctx, b"compress_lanes", args.compress_lanes) == 0
    if args.compress_variant:
        assert lib.lz4flex_set_tuning(ctx, b"compress_variant", args.compress_variant) == 0
    if args.decompress_variant:
        assert lib.lz4flex_set_tuning(ctx, b"decompress_variant", args.decompress_variant) == 0
    if args.ablate:
        args.no_verify = True

    # ---- workload: buf[i] = json[(i + phase) mod 66 675], cut into 64 KiB blocks
    n = args.blocks
    total = n * BLOCK
    plain = load_json_fixture(block)
    jt = torch.frombuffer(bytearray(plain), dtype=torch.uint8).to(dev)
    phase = (rank * 7919) % len(plain)            # every rank gets different bytes
    reps = (total + phase) // len(plain) + 2
    src = jt.repeat(reps)[phase:phase + total].contiguous()
    del jt
    stride = 72128                                # >= get_maximum_output_size(65536) = 72109, 64 B aligned
    comp = torch.empty(n * stride, dtype=torch.uint8, device=dev)
    back = torch.empty(total, dtype=torch.uint8, device=dev)
    ar = torch.arange(n, dtype=torch.int64, device=dev)
    in_off = (ar * BLOCK).contiguous()            # u64 view of non-negative i64
    if args.in_pad:                               # diagnostic layout: blocks no longer at 64 KiB multiples
        istride = BLOCK + args.in_pad
        src2 = torch.zeros(n * istride, dtype=torch.uint8, device=dev)
        src2.view(n, istride)[:, :BLOCK] = src.view(n, BLOCK)
        src = src2
        back = torch.empty(n * istride, dtype=torch.uint8, device=dev)
        in_off = (ar * istride).contiguous()
    comp_off = (ar * stride).contiguous()
    in_len = torch.full((n,), BLOCK, dtype=torch.int32, device=dev)
    comp_cap = torch.full((n,), stride, dtype=torch.int32, device=dev)
    comp_len = torch.zeros(n, dtype=torch.int32, device=dev)
    c_status = torch.full((n,), -1, dtype=torch.int32, device=dev)
    back_cap = torch.full((n,), BLOCK, dtype=torch.int32, device=dev)
    back_len = torch.zeros(n, dtype=torch.int32, device=dev)
    d_status = torch.full((n,), -1, dtype=torch.int32, device=dev)

    def p(t):
        return C.c_void_p(t.data_ptr())

    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def do_compress():
        r = lib.lz4flex_compress_batch(ctx, p(src), p(in_off), p(in_len), None, n, p(comp), p(comp_off), p(comp_cap),
                                       p(comp_len), p(c_status), _lib.MEM_DEVICE, stream)
        assert r == 0, _lib.last_error()

    def do_decompress():
        r = lib.lz4flex_decompress_batch(ctx, p(comp), p(comp_off), p(comp_len), n, p(back), p(in_off), p(back_cap),
                                         p(back_len), p(d_status), None, _lib.MEM_DEVICE, stream)
        assert r == 0, _lib.last_error()

    def barrier():
        if world > 1:
            dist.barrier()

    # one untimed pass to produce the compressed side (also needed when --only decompress)
    do_compress()
    do_decompress()
    torch.cuda.synchronize()
    if args.ablate:
        assert lib.lz4flex_set_tuning(ctx, b"ablate", args.ablate) == 0
    for _ in range(args.warmup):
        if args.only in ("both", "compress"):
            do_compress()
        if args.only in ("both", "decompress"):
            do_decompress()
    torch.cuda.synchronize()

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        ev[s][0].record()
        if args.only in ("both", "compress"):
            do_compress()
        ev[s][1].record()
        if args.only in ("both", "decompress"):
            do_decompress()
        ev[s][2].record()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if oversub else dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    # ---- verify after the timed loop
    if not args.no_verify:
        assert int((c_status != 0).sum().item()) == 0 and int((d_status != 0).sum().item()) == 0, "per-block status != 0"
        assert int((back_len != BLOCK).sum().item()) == 0
        assert torch.equal(back, src), "round trip mismatch"
    comp_bytes = int(comp_len.to(torch.int64).sum().item())
    ratio = comp_bytes / total

    t_c = sum(ev[s][0].elapsed_time(ev[s][1]) for s in range(args.steps)) / args.steps * 1e-3   # s per launch
    t_d = sum(ev[s][1].elapsed_time(ev[s][2]) for s in range(args.steps)) / args.steps * 1e-3
    alg_bytes = total + comp_bytes    # compress moves u (read) + c (write); decompress c (read) + u (write)

    def roof(t):
        if t <= 0:
            return None
        a = alg_bytes / t / 1e9
        return {"bound": "hbm", "achieved": round(a, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(a / HBM_PEAK_GBS, 5), "traffic": None}

    kernels = {}
    if args.only in ("both", "compress"):
        kernels["compress"] = {"kernel": "lz4_compress_blocks_kernel", "ms_per_launch": round(t_c * 1e3, 4),
                               "MiB_per_s": round(total / 1048576 / t_c, 1), "roofline": roof(t_c)}
    if args.only in ("both", "decompress"):
        dv = args.decompress_variant or (3 if n > 20480 else 4)   # capi.cpp launch_decompress_fast's choice by batch size
        kernels["decompress"] = {"kernel": {4: "lz4_decompress_split_kernel", 3: "lz4_decompress_pipe_kernel", 2: "lz4_decompress_lds_kernel",
                                            1: "lz4_decompress_blocks_kernel"}[dv], "ms_per_launch": round(t_d * 1e3, 4),
                                 "MiB_per_s": round(total / 1048576 / t_d, 1), "roofline": roof(t_d)}
    # measured HBM traffic per launch (rocprofv3 --pmc passes, corrected per MI355X_MICROARCH.md) if recorded
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                tr = json.load(f)
            for k in kernels:
                if k in tr and kernels[k]["roofline"] and tr[k].get("blocks") == n:
                    kernels[k]["roofline"]["traffic"] = tr[k]["hbm_bytes_per_launch"]
        except Exception:
            pass
    dominant = max(kernels, key=lambda k: kernels[k]["ms_per_launch"])

    ms_per_step = elapsed / args.steps * 1e3
    value = world * (total / 1048576) / (elapsed / args.steps)

    out = {
        "metric": "MiB/s compress+decompress round trip, 1 GiB of 64 KiB JSON blocks per GPU (LZ4 block format)",
        "value": round(value, 1),
        "unit": "MiB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic: benches/compression_66k_JSON.txt tiled cyclically (buf[i]=json[(i+phase) mod 66675]), "
                "%d x 64 KiB blocks per GPU" % n,
        "config": {"workload": "BASELINE configs[1]: %d independent 64 KiB JSON blocks (%.3f GiB) per GPU, block format, "
                               "compress then decompress, device-resident" % (n, total / 2**30),
                   "blocks_per_gpu": n, "block_bytes": BLOCK, "parallelism": "blocks sharded across ranks, no data-path collective",
                   "only": args.only},
        "ratio": round(ratio, 5),
        "compress_MiB_per_s_per_gpu": kernels.get("compress", {}).get("MiB_per_s"),
        "decompress_MiB_per_s_per_gpu": kernels.get("decompress", {}).get("MiB_per_s"),
        "roofline": dict(kernels[dominant]["roofline"], kernel=kernels[dominant]["kernel"]),
        "kernels": kernels,
        "oversubscribed_ranks_per_gpu": (world + ndev - 1) // ndev if oversub else 1,
        "verified": "NOT VERIFIED (--no-verify)" if args.no_verify else "round trip bit-exact on device; all per-block status 0",
    }

    # ---- CPU baseline: the oracle (a C port of lz4_flex's block codec) on the host cores, bounded sample
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(src, comp, comp_off, comp_len, n)
        except Exception as e:   # the baseline is a report, never a reason to lose the GPU line
            out["cpu_baseline"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(out))
    lib.lz4flex_ctx_destroy(ctx)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(src, comp, comp_off, comp_len, n):
    """Times oracle/ (kind 'port': lz4_flex is Rust, no toolchain here) on the same bytes: a bounded sample
    of the workload, all host cores, best of 3 passes per direction."""
    import subprocess
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    if not os.path.exists(os.path.join(ROOT, "oracle", "liblz4flex_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    import oracle_api as O
    o = O.lib()
    cores = os.cpu_count() or 1
    ns = min(n, 4096)                              # 256 MiB sample
    h_src = src[:ns * BLOCK].cpu().numpy()
    stride = int(comp_off[1].item()) if n > 1 else 72128
    h_in_off = (np.arange(ns, dtype=np.uint64) * BLOCK)
    h_in_len = np.full(ns, BLOCK, dtype=np.uint32)
    h_out = np.zeros(ns * stride, dtype=np.uint8)
    h_out_off = (np.arange(ns, dtype=np.uint64) * stride)
    h_out_cap = np.full(ns, stride, dtype=np.uint32)
    h_out_len = np.zeros(ns, dtype=np.uint32)

    def vp(a):
        return C.c_void_p(a.ctypes.data)

    res = {}
    for threads in (cores, 1):
        tc = o.lz4o_bench_batch(0, vp(h_src), vp(h_in_off), vp(h_in_len), vp(h_out), vp(h_out_off), vp(h_out_cap),
                                vp(h_out_len), ns, threads, 3)
        # the oracle's blocks must equal the GPU's
        g_len = comp_len[:ns].cpu().numpy().astype(np.uint32)
        assert (h_out_len == g_len).all(), "GPU encoder output size differs from the oracle"
        h_back = np.zeros(ns * BLOCK, dtype=np.uint8)
        h_back_len = np.zeros(ns, dtype=np.uint32)
        td = o.lz4o_bench_batch(1, vp(h_out), vp(h_out_off), vp(h_out_len), vp(h_back), vp(h_in_off), vp(h_in_len),
                                vp(h_back_len), ns, threads, 3)
        assert (h_back == h_src).all()
        mib = ns * BLOCK / 1048576
        res[threads] = (mib / tc, mib / td, mib / (tc + td))
    g = comp[:ns * stride].cpu().numpy()
    for i in (0, 1, ns // 2, ns - 1):
        a = g[i * stride:i * stride + int(h_out_len[i])]
        b = h_out[i * stride:i * stride + int(h_out_len[i])]
        assert (a == b).all(), "GPU encoder bytes differ from the oracle"
    return {"value": round(res[cores][2], 1), "unit": "MiB/s", "cores": cores, "kind": "port",
            "sample": "first %d of the %d blocks (%d MiB), compress+decompress round trip, best of 3, %d threads; "
                      "oracle = C restatement of lz4_flex block codec, gcc -O3 -march=native" % (ns, n, ns * BLOCK >> 20, cores),
            "compress_MiB_per_s": round(res[cores][0], 1), "decompress_MiB_per_s": round(res[cores][1], 1),
            "single_thread": {"compress_MiB_per_s": round(res[1][0], 1), "decompress_MiB_per_s": round(res[1][1], 1),
                              "round_trip_MiB_per_s": round(res[1][2], 1)}}


if __name__ == "__main__":
    main()
