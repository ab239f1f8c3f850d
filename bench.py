#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: MiB/s of LZ4 block compress + decompress on MI355X, inputs resident in HBM.

Default (= `--config 2`, BASELINE configs[1], the configuration the metric is quoted on): 1 GiB = 16 384 independent
64 KiB JSON blocks per GPU, block format.  A "step" is one pass of the hot path over the batch: one batched compress
launch (1 GiB -> LZ4 blocks) followed by one batched decompress launch (those blocks -> 1 GiB).
`value` = uncompressed MiB carried through that round trip per second, whole job (all ranks).

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Other BASELINE configs (parity cases; each prints one JSON line of the same shape):
  --config 3   text tiles (dickens.txt is absent from the reference mount: compression_65k.txt tiled to 10 MiB)
  --config 4   frame format, BlockIndependent + Max4MB over the synthetic log stream, 1 GiB per rank, sharded across the
               ranks with the frame reassembled on rank 0 (the only place a collective is on the data path)
  --config 5   frame format, BlockLinked 64 KiB blocks: one dependency chain (a stress, not a throughput path)

Encoder: `--compress-mode fast` (default) is the throughput encoder (its own parse: a valid LZ4 block that lz4_flex's
decoder returns to the input -- north_star's compress contract; ratio reported), `exact` reproduces lz4_flex's bytes.
The decoder is bit-exact in either case.  Multi-GPU: blocks are independent, so ranks own disjoint block ranges (weak
scaling: 1 GiB per GPU); torch.distributed (RCCL) provides the barrier and the max-reduce of the timings, and in
--config 4 the size all-gather + segment gather of the frame.
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
ROUND = 5               # stamped into the traffic figure's provenance


def median(xs):
    xs = sorted(xs)
    n = len(xs)
    return xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])


def relaunch_ranks(gpus):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks here (one process per GPU over
    torch.distributed / RCCL, rendezvous on 127.0.0.1) and become the launcher.  Under torch.distributed.run (WORLD_SIZE is
    set) this is never reached."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


def load_fixture(block_mod, stem):
    """A reference bench file (benches/*.txt), recovered by decoding its golden LZ4 block with the GPU codec itself and
    checked against the md5 recorded from the reference file."""
    g = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(g, "manifest.json")) as f:
        m = json.load(f)[stem]
    with open(os.path.join(g, stem + ".lz4blk"), "rb") as f:
        blk = f.read()
    plain = block_mod.decompress(blk, m["plain_len"])
    assert hashlib.md5(plain).hexdigest() == m["plain_md5"], "fixture md5 mismatch"
    return plain


def roof(alg_bytes, t):
    if t <= 0:
        return None
    a = alg_bytes / t / 1e9
    return {"bound": "hbm", "achieved": round(a, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a / HBM_PEAK_GBS, 5),
            "traffic": None}


def attach_traffic(kernels, n, mode, config=2):
    """measured HBM bytes per launch (separate rocprofv3 --pmc passes, corrected per MI355X_MICROARCH.md), recorded in
    profiles/traffic.json together with the round and the counter files they came from"""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return
    try:
        with open(tpath) as f:
            tr = json.load(f)
        for k in kernels:
            key = k + ("_" + mode if k == "compress" else "")
            e = tr.get("config%d_%s" % (config, k)) if config != 2 else (tr.get(key) or tr.get(k))
            # kernel_key = the kernel's plain name (the "kernel" field may carry template arguments and a description)
            if e and kernels[k]["roofline"] and e.get("blocks") == n and e.get("kernel_key", e.get("kernel")) == kernels[k]["kernel"]:
                kernels[k]["roofline"]["traffic"] = e["hbm_bytes_per_launch"]
                kernels[k]["roofline"]["traffic_source"] = "profiles/traffic.json (round %s, %s)" % (e.get("round", "?"), e.get("source", "?"))
    except Exception:
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5], help="BASELINE.json configs[config-1]")
    ap.add_argument("--blocks", type=int, default=0, help="blocks per GPU (default: the config's size)")
    ap.add_argument("--compress-mode", choices=["fast", "exact"], default="fast")
    ap.add_argument("--decompress-variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="kernel experiments only: skip the bit-exact check (never for reported numbers)")
    ap.add_argument("--only", choices=["both", "compress", "decompress"], default="both",
                    help="profiling aid: run only one kernel in the timed steps (value then covers that kernel only)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="the default line measures roofline.traffic of the dominant kernel in this run (two rocprofv3 --pmc passes of a "
                         "short compress-only child run, ~30 s); this switches that off (the figure then comes from profiles/traffic.json)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="the default line (config 2, one GPU) also runs configs 3, 4 and 5 for a few steps behind its timed loop and "
                         "carries their figures under `other_configs`; this switches that off")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        relaunch_ranks(args.gpus)                  # does not return

    import torch
    import torch.distributed as dist
    from lz4_flex_amd import _lib, block

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but the launcher started %d rank(s): n_gpus must be what was asked for" % (args.gpus, world)
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
            dist.init_process_group("gloo")        # RCCL refuses two ranks on one device
        else:
            dist.init_process_group("nccl", device_id=dev)
    lib = _lib.load()
    block.set_compress_mode(args.compress_mode)    # this thread's default context (frame layer, scalar calls)
    ctx = C.c_void_p()
    rc = lib.lz4flex_ctx_create(C.byref(ctx), dev_index)
    assert rc == 0, _lib.last_error()
    assert lib.lz4flex_set_tuning(ctx, b"compress_mode", 1 if args.compress_mode == "exact" else 0) == 0
    if args.decompress_variant:
        assert lib.lz4flex_set_tuning(ctx, b"decompress_variant", args.decompress_variant) == 0

    env = {"torch": torch, "dist": dist, "lib": lib, "ctx": ctx, "dev": dev, "world": world, "rank": rank, "oversub": oversub,
           "block": block, "_lib": _lib}
    if args.config in (2, 3):
        out = run_blocks(args, env)
    elif args.config == 4:
        out = run_sharded_frame(args, env)
    else:
        out = run_linked_frame(args, env)
    # The driver only runs the default line: the other BASELINE configs ride on it (one GPU, behind the headline's timed loop and its
    # verification; a few steps each, the same code as `--config K`), so that their figures are driver-observed too
    if (args.config == 2 and world == 1 and args.only == "both" and not args.blocks and not args.no_cpu_baseline and not args.no_other_configs
            and not args.no_verify and not args.decompress_variant):
        out["other_configs"] = other_configs(args, env)
        if not args.no_live_traffic:
            live_traffic(out, args)
    if rank == 0:
        print(json.dumps(out))
    lib.lz4flex_ctx_destroy(ctx)
    if world > 1:
        dist.destroy_process_group()


def live_traffic(out, args):
    """roofline.traffic of the dominant kernel (the encoder) measured in THIS run, as MI355X_MICROARCH.md prescribes: rocprofv3 --pmc
    FETCH_SIZE and --pmc WRITE_SIZE in separate passes (only --kernel-trace beside the counter), each over a short compress-only child
    run of this script on the same workload; median over the kernel's launches; FETCH_SIZE x 2 (gfx950 tallies 128-byte requests at
    64 B), units KiB.  Any failure leaves the figure from profiles/traffic.json in place (traffic_source says which it is)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return
    tmp = tempfile.mkdtemp(prefix="lz4flex_pmc_", dir="/tmp")
    try:
        for which in ("compress", "decompress"):
            kname = out.get("kernels", {}).get(which, {}).get("kernel")
            if not kname:
                continue
            med = {}
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                d = os.path.join(tmp, which + "_" + counter)
                cmd = [rp, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                       os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--only", which, "--no-cpu-baseline", "--no-other-configs",
                       "--no-live-traffic", "--compress-mode", args.compress_mode]
                subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=150, check=True)
                vals = []
                for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                    with open(f) as fh:
                        for r in csv.DictReader(fh):
                            if kname in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                                vals.append(float(r["Counter_Value"]))
                if len(vals) < 3:          # (fewer: the other kernel's single untimed pass, or nothing)
                    med = None
                    break
                vals.sort()
                med[counter] = (vals[len(vals) // 2], len(vals))
            if not med:
                continue
            hbm = int(med["FETCH_SIZE"][0] * 2048 + med["WRITE_SIZE"][0] * 1024)
            src = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (--kernel-trace only beside the counter) over a "
                   "%s-only child run (1 + 3 launches each); medians %.0f / %.0f KiB over %d / %d launches; FETCH_SIZE x 2 (gfx950), KiB -> bytes"
                   % (which, med["FETCH_SIZE"][0], med["WRITE_SIZE"][0], med["FETCH_SIZE"][1], med["WRITE_SIZE"][1]))
            targets = [out.get("kernels", {}).get(which, {}).get("roofline")]
            if out["roofline"].get("kernel") == kname:
                targets.append(out["roofline"])
            for r in targets:
                if r:
                    if r.get("traffic") is not None:
                        r["traffic_profiles_file"] = r["traffic"]
                    r["traffic"] = hbm
                    r["traffic_source"] = src
    except Exception as e:
        out["roofline"]["live_traffic_error"] = repr(e)[:200]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def other_configs(args, env):
    """configs 3, 4 (one GPU's 1 GiB share) and 5 through their own run_* functions: 5 timed steps behind 2 warm-up steps each"""
    import copy
    res = {}
    for k, fn in ((3, run_blocks), (4, run_sharded_frame), (5, run_linked_frame)):
        a2 = copy.copy(args)
        a2.config, a2.steps, a2.warmup, a2.blocks = k, 5, 2, 0
        t0 = time.perf_counter()
        try:
            env["torch"].cuda.empty_cache()
            o = fn(a2, env)
            cb = o.get("cpu_baseline") or {}
            res["config%d" % k] = {"metric": o["metric"], "value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": a2.steps,
                                   "ratio": o.get("ratio"), "verified": o.get("verified"), "workload": o["config"]["workload"],
                                   "cpu_baseline": {kk: cb.get(kk) for kk in ("value", "unit", "cores", "kind", "sample", "error", "by_threads") if kk in cb},
                                   "roofline_frac": (o.get("roofline") or {}).get("frac"), "wall_s": None}
            for extra in ("parts_ms", "windows_64k", "windows_32k", "many_streams", "kernels"):
                if o.get(extra) is not None:
                    res["config%d" % k][extra] = o[extra]
        except Exception as e:     # a failing side configuration must not cost the headline line
            res["config%d" % k] = {"error": repr(e)}
        res["config%d" % k]["wall_s"] = round(time.perf_counter() - t0, 2)
    # medium batches (round 6): the block codec on fewer blocks than the headline's 16 384 -- the sizes at which the default dispatch takes the
    # sequence decoder (a wavefront per block, a lane per sequence).  Same code as --config 2 / 3 --blocks N: round trip verified on the device
    res["medium_batches"] = {}
    for cfg, nb in ((2, 4096), (2, 8192), (3, 4096)):
        a2 = copy.copy(args)
        a2.config, a2.steps, a2.warmup, a2.blocks, a2.no_cpu_baseline = cfg, 5, 2, nb, True
        key = "%s_%d_blocks" % ("json" if cfg == 2 else "text", nb)
        try:
            env["torch"].cuda.empty_cache()
            o = run_blocks(a2, env)
            res["medium_batches"][key] = {"value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "ratio": o.get("ratio"),
                                          "verified": o.get("verified"), "kernels": o.get("kernels")}
        except Exception as e:
            res["medium_batches"][key] = {"error": repr(e)}
    # zero pages (round 6, run windows): 16 blocks of 4 MiB of zeros through the block codec -- the reference encodes such a block as ONE
    # sequence (src/block/compress.rs:156-216: count_same_bytes is unbounded), this encoder as one per 48 KiB window since round 6
    try:
        env["torch"].cuda.empty_cache()
        res["zero_pages"] = run_zero_pages(args, env)
    except Exception as e:
        res["zero_pages"] = {"error": repr(e)}
    return res


def run_zero_pages(args, env, n=16, size=4 << 20):
    torch, lib, ctx, dev, _lib = (env[k] for k in ("torch", "lib", "ctx", "dev", "_lib"))
    src = torch.zeros(n * size, dtype=torch.uint8, device=dev)
    stride = (int(lib.lz4flex_get_maximum_output_size(size)) + 63) // 64 * 64
    comp = torch.empty(n * stride, dtype=torch.uint8, device=dev)
    back = torch.full((n * size,), 0xEE, dtype=torch.uint8, device=dev)
    ar = torch.arange(n, dtype=torch.int64, device=dev)
    in_off, comp_off = (ar * size).contiguous(), (ar * stride).contiguous()
    in_len = torch.full((n,), size, dtype=torch.int32, device=dev)
    cap = torch.full((n,), stride, dtype=torch.int32, device=dev)
    clen = torch.zeros(n, dtype=torch.int32, device=dev)
    st = torch.full((n,), -1, dtype=torch.int32, device=dev)
    blen = torch.zeros(n, dtype=torch.int32, device=dev)
    bst = torch.full((n,), -1, dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    flags = _lib.MEM_DEVICE | _lib.MEM_BIG_BLOCKS

    def comp_once():
        assert lib.lz4flex_compress_batch(ctx, p(src), p(in_off), p(in_len), None, n, p(comp), p(comp_off), p(cap), p(clen), p(st), flags, stream) == 0, _lib.last_error()

    def dec_once():
        assert lib.lz4flex_decompress_batch(ctx, p(comp), p(comp_off), p(clen), n, p(back), p(in_off), p(in_len), p(blen), p(bst), None, flags, stream) == 0, _lib.last_error()

    def ms(fn, reps=5):
        ts = []
        for _ in range(reps + 2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return median(ts[2:])

    comp_once(); dec_once(); torch.cuda.synchronize()
    ok = int((st != 0).sum().item()) == 0 and int((bst != 0).sum().item()) == 0 and bool(torch.equal(back, src))
    tc, td = ms(comp_once), ms(dec_once)
    total_c = int(clen.to(torch.int64).sum().item())
    verified = "round trip bit-exact on the device" if ok else "ROUND TRIP FAILED"
    try:                                               # the oracle's decoder on the first block (test infrastructure: the checker)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
        import oracle_api as O
        c0 = bytes(comp[:int(clen[0].item())].cpu().numpy())
        verified += "; block 0 (%d bytes) decoded by the oracle == 4 MiB of zeros: %s" % (len(c0), O.decompress(c0, size) == ("ok", bytes(size)))
    except Exception as e:
        verified += "; oracle check skipped (%r)" % (e,)
    return {"what": "%d blocks of %d MiB of zeros, device-resident, compress_mode fast" % (n, size >> 20), "compress_ms": round(tc, 4), "decompress_ms": round(td, 4),
            "ratio": round(total_c / (n * size), 6), "decompress_GB_per_s": round(n * size / td / 1e6, 1), "compress_GB_per_s": round(n * size / tc / 1e6, 1),
            "verified": verified}


def timed(args, env, step):
    """W untimed warmup steps, K timed steps between barrier + synchronize, max over ranks"""
    torch, dist, world = env["torch"], env["dist"], env["world"]
    for _ in range(args.warmup):
        step(None)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(s)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if env["oversub"] else env["dev"])
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    return elapsed


# --------------------------------------------------------------------------------------- configs 2 and 3: block batches
def run_blocks(args, env):
    torch, lib, ctx, dev, world, rank, _lib, block = (env[k] for k in ("torch", "lib", "ctx", "dev", "world", "rank", "_lib", "block"))
    from lz4_flex_amd import workloads
    if args.config == 2:
        n = args.blocks or 16384
        plain = load_fixture(block, "compression_66k_JSON")
        what = "benches/compression_66k_JSON.txt tiled cyclically (buf[i]=json[(i+phase) mod 66675])"
        cfg = "BASELINE configs[1]: %d independent 64 KiB JSON blocks (%.3f GiB) per GPU" % (n, n * BLOCK / 2**30)
    else:
        n = args.blocks or 160
        plain = load_fixture(block, "compression_65k")
        what = ("dickens.txt is absent from the reference mount (.MISSING_LARGE_BLOBS): benches/compression_65k.txt (English text) "
                "tiled cyclically to 10 MiB, as SURVEY 8(d) prescribes")
        cfg = "BASELINE configs[2] (substitute text): %d independent 64 KiB text blocks (%.1f MiB) per GPU" % (n, n * BLOCK / 2**20)
    total = n * BLOCK
    src = workloads.json_tiles(plain, total, phase=(rank * 7919) % len(plain), device=dev)   # every rank gets different bytes
    stride = 72128                                # >= get_maximum_output_size(65536) = 72109, 64 B aligned
    comp = torch.empty(n * stride, dtype=torch.uint8, device=dev)
    back = torch.empty(total, dtype=torch.uint8, device=dev)
    ar = torch.arange(n, dtype=torch.int64, device=dev)
    in_off = (ar * BLOCK).contiguous()
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

    do_compress()       # one untimed pass produces the compressed side (also needed when --only decompress)
    do_decompress()
    torch.cuda.synchronize()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]

    def step(s):
        if s is not None:
            ev[s][0].record()
        if args.only in ("both", "compress"):
            do_compress()
        if s is not None:
            ev[s][1].record()
        if args.only in ("both", "decompress"):
            do_decompress()
        if s is not None:
            ev[s][2].record()

    elapsed = timed(args, env, step)

    # ---- verify after the timed loop: every block decodes to its input on the device
    if not args.no_verify:
        assert int((c_status != 0).sum().item()) == 0 and int((d_status != 0).sum().item()) == 0, "per-block status != 0"
        assert int((back_len != BLOCK).sum().item()) == 0
        assert torch.equal(back, src), "round trip mismatch"
    comp_bytes = int(comp_len.to(torch.int64).sum().item())
    ratio = comp_bytes / total
    # per-kernel durations: MEDIAN over the timed steps, events on the stream the kernels are launched on
    t_c = median([ev[s][0].elapsed_time(ev[s][1]) for s in range(args.steps)]) * 1e-3   # s per launch
    t_d = median([ev[s][1].elapsed_time(ev[s][2]) for s in range(args.steps)]) * 1e-3
    alg_bytes = total + comp_bytes    # compress moves u (read) + c (write); decompress c (read) + u (write)
    kernels = {}
    if args.only in ("both", "compress"):
        kname = "lz4_compress_wave_kernel" if args.compress_mode == "fast" else "lz4_compress_blocks_kernel"
        kernels["compress"] = {"kernel": kname, "ms_per_launch": round(t_c * 1e3, 4), "MiB_per_s": round(total / 1048576 / t_c, 1),
                               "roofline": roof(alg_bytes, t_c)}
    if args.only in ("both", "decompress"):
        # capi.cpp launch_decompress_fast's choice by batch size: the thresholds are the library's (lz4_device.h DISPATCH_*)
        th = [lib.lz4flex_get_tuning(ctx, b"dispatch_threshold_%d" % i) for i in range(6)]     # pair of workgroups, pcd 1024 / 512 / 256 lanes, a wavefront per block, full chip
        assert min(th) > 0 and th == sorted(th), th
        dv = args.decompress_variant or (7 if n <= th[3] else (13 if n <= th[4] else 4))
        kernels["decompress"] = {"kernel": {7: "lz4_decompress_pcd_kernel", 8: "lz4_decompress_pcd_kernel", 13: "lz4_decompress_seq_kernel",
                                            4: "lz4_decompress_split_kernel", 1: "lz4_decompress_blocks_kernel",
                                            9: "lz4_replay_kernel", 10: "lz4_decompress_pcd_kernel", 11: "lz4_decompress_pcd_kernel", 12: "lz4_decompress_fused_kernel"}[dv],
                                 "ms_per_launch": round(t_d * 1e3, 4), "MiB_per_s": round(total / 1048576 / t_d, 1),
                                 "roofline": roof(alg_bytes, t_d)}
    attach_traffic(kernels, n, args.compress_mode, args.config)
    dominant = max(kernels, key=lambda k: kernels[k]["ms_per_launch"])
    out = {
        "metric": "MiB/s compress+decompress round trip, 1 GiB of 64 KiB JSON blocks per GPU (LZ4 block format)" if args.config == 2
                  else "MiB/s compress+decompress round trip, 64 KiB text blocks (LZ4 block format)",
        "value": round(world * (total / 1048576) / (elapsed / args.steps), 1),
        "unit": "MiB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic: %s, %d x 64 KiB blocks per GPU" % (what, n),
        "config": {"workload": cfg + ", block format, compress then decompress, device-resident",
                   "blocks_per_gpu": n, "block_bytes": BLOCK, "compress_mode": args.compress_mode,
                   "compress_contract": ("valid LZ4 blocks that the reference decoder returns to the input (own parse)" if args.compress_mode == "fast"
                                         else "bytes identical to the oracle restatement of lz4_flex's encoder"),
                   "parallelism": "blocks sharded across ranks, no data-path collective", "only": args.only,
                   "oversubscribed": ("%d ranks on %d device(s): functional run of the multi-process path, not a scaling figure" % (world, torch.cuda.device_count()))
                                     if env["oversub"] else None},
        "ratio": round(ratio, 5),
        "compress_MiB_per_s_per_gpu": kernels.get("compress", {}).get("MiB_per_s"),
        "decompress_MiB_per_s_per_gpu": kernels.get("decompress", {}).get("MiB_per_s"),
        "roofline": dict(kernels[dominant]["roofline"], kernel=kernels[dominant]["kernel"]),
        "kernels": kernels,
        "verified": "NOT VERIFIED (--no-verify)" if args.no_verify else "round trip bit-exact on device; all per-block status 0",
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            stash = {}
            out["cpu_baseline"] = cpu_baseline(src, comp, comp_off, comp_len, n, args.compress_mode, stash)
            # north_star's decompress contract is "bit-exact on the reference's own block bytes": the timed loop above decodes the
            # GPU encoder's blocks, so the decoder is timed once more on the ORACLE-encoded workload (= lz4_flex's bytes), after
            # the timed region, events on the launch stream, and checked against the source
            h_out, h_out_len = stash["oracle_blocks"]
            comp.copy_(torch.from_numpy(h_out).to(dev))
            comp_len.copy_(torch.from_numpy(h_out_len.astype("int32")).to(dev))
            back.zero_()
            do_decompress()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); do_decompress(); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            assert int((d_status != 0).sum().item()) == 0 and torch.equal(back, src), "reference-encoded blocks: decode mismatch"
            out["decompress_ref_blocks_ms"] = round(median(ts), 4)
            out["decompress_ref_blocks"] = ("all %d blocks encoded by the oracle (lz4_flex's encoder restated: %d compressed bytes), decoded on the "
                                            "GPU, median of 5 launches, output == source" % (n, int(h_out_len.sum())))
        except Exception as e:   # the baseline is a report, never a reason to lose the GPU line
            out.setdefault("cpu_baseline", {"error": repr(e)})
            out["decompress_ref_blocks_ms"] = None
            out["decompress_ref_blocks"] = "failed: %r" % (e,)
    return out


# --------------------------------------------------------------------------------------- config 4: sharded frames
def run_sharded_frame(args, env):
    torch, dist, dev, world, rank = (env[k] for k in ("torch", "dist", "dev", "world", "rank"))
    from lz4_flex_amd import sharded, workloads
    from lz4_flex_amd.frame import BlockMode, BlockSize, FrameInfo
    bs = 4 << 20
    n = args.blocks or 256                             # 4 MiB blocks per rank: 1 GiB per GPU (8 GiB over 8 GPUs)
    per_rank = n * bs
    local = workloads.log_stream(rank * per_rank, per_rank, device=dev)
    fi = FrameInfo(block_size=BlockSize.Max4MB, block_mode=BlockMode.Independent)
    state = {}
    t_parts = {"compress+assemble+gather": 0.0, "scatter+decompress": 0.0}

    def step(s):
        t0 = time.perf_counter()
        frame = sharded.compress_frame_sharded(local, rank * n, fi)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        out, (lo, hi), _ = sharded.decompress_frame_sharded(frame, device=dev)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if s is not None:
            t_parts["compress+assemble+gather"] += t1 - t0
            t_parts["scatter+decompress"] += t2 - t1
        state["frame"], state["out"], state["range"] = frame, out, (lo, hi)

    elapsed = timed(args, env, step)
    # verify: every rank's decoded range equals the bytes it owns of the stream
    lo, hi = state["range"]
    if not args.no_verify:
        exp = workloads.log_stream(lo * bs, (hi - lo) * bs, device=dev)
        assert torch.equal(state["out"], exp), "sharded frame round trip mismatch"
    total = per_rank * world
    oracle_checked = None
    if rank == 0 and not args.no_verify and total <= (2 << 30):
        # the gathered frame through the REFERENCE's frame decoder (oracle restatement), once, after the timed loop
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_api as O
        h_frame = state["frame"].cpu().numpy().tobytes()
        rc, back, _used = O.frame_decompress(h_frame, total)
        assert rc == 0, "the reference's FrameDecoder (oracle) rejects the gathered frame: %r" % (rc,)
        want = b"".join(workloads.log_stream(r * per_rank, per_rank, device="cpu").numpy().tobytes() for r in range(world))
        assert back == want, "the reference's FrameDecoder (oracle) does not return the stream"
        oracle_checked = "gathered frame (%d bytes) decoded by the oracle's FrameDecoder == the stream" % len(h_frame)
        del h_frame, back, want
    frame_bytes = int(state["frame"].numel()) if rank == 0 else 0
    alg = total + frame_bytes
    # The timed figure is the library's default: the 64 KiB windows of a 4 MiB block advance by 48 KiB, so every window start has 16 KiB of
    # history (the reference's window slides continuously; ratio just below the reference's).  The same steps with windows that advance
    # by 64 KiB ("compress_sliding_window" 0: round 3's bytes, every byte indexed once) and by 32 KiB (1: round 4's default, every byte
    # indexed twice) are timed beside it, outside the reported value.
    windows_64k = windows_32k = None
    if world == 1 and not args.no_verify and args.compress_mode != "exact":
        from lz4_flex_amd import _lib as L
        lib = L.load()
        default = lib.lz4flex_get_tuning(None, b"compress_sliding_window")

        def with_setting(value, what):
            try:
                assert lib.lz4flex_set_tuning(None, b"compress_sliding_window", value) == 0
                for _ in range(2):
                    step(None)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                k = max(3, min(args.steps, 5))
                for _ in range(k):
                    step(None)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / k
                assert torch.equal(state["out"], workloads.log_stream(lo * bs, (hi - lo) * bs, device=dev))
                return {"what": what, "value": round((total / 1048576) / dt, 1), "ms_per_step": round(dt * 1e3, 3),
                        "ratio": round(int(state["frame"].numel()) / total, 5)}
            except Exception as e:
                return {"error": repr(e)}
            finally:
                lib.lz4flex_set_tuning(None, b"compress_sliding_window", default)
        windows_64k = with_setting(0, "the same step with compress_sliding_window = 0 (windows advance by 64 KiB)")
        windows_32k = with_setting(1, "the same step with compress_sliding_window = 1 (windows advance by 32 KiB: round 4's default)")
    out = {
        "metric": "MiB/s frame compress + decompress, BlockIndependent Max4MB, synthetic log stream, 1 GiB per GPU, frame gathered on rank 0",
        "value": round((total / 1048576) / (elapsed / args.steps), 1), "unit": "MiB/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8",
        "data": "synthetic: lz4_flex_amd.workloads.log_stream (128-byte log lines from a counter-based generator), %d x 4 MiB blocks per GPU" % n,
        "config": {"workload": "BASELINE configs[3]: frame format, BlockIndependent + Max4MB, %d blocks (%.2f GiB) per GPU, sharded across %d rank(s), "
                               "segments gathered to rank 0 (size all-gather + point-to-point gather), then scattered and decoded" % (n, per_rank / 2**30, world),
                   "compress_mode": args.compress_mode, "timing": "end to end on device tensors: kernels + frame assembly + exchange"},
        "ratio": round(frame_bytes / total, 5) if frame_bytes else None,
        "parts_ms": {k: round(v / args.steps * 1e3, 3) for k, v in t_parts.items()},
        "windows_64k": windows_64k,
        "windows_32k": windows_32k,
        "roofline": dict(roof(alg, elapsed / args.steps), kernel="whole step (lz4_compress_wave_kernel + frame assembly + lz4_decompress_pcd_kernel)"),
        "verified": "NOT VERIFIED" if args.no_verify else "every rank's decoded block range equals the stream bytes it owns" +
                    ("; " + oracle_checked if oracle_checked else ""),
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline_buffer(local[:64 * bs], bs)
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    return out


# --------------------------------------------------------------------------------------- config 5: Linked frame
def run_linked_frame(args, env):
    torch, dev, world, rank, block = (env[k] for k in ("torch", "dev", "world", "rank", "block"))
    from lz4_flex_amd import frame as F, workloads
    # fast (default): every block's matches reach into the 32 KiB of the stream in front of it (LZ4FLEX_BLOCK_HISTORY: the history is
    # input, so the blocks still encode in one launch); exact: the reference's bytes, one dependency chain on the device.
    # Decoding: one chained launch per batch of blocks either way -- the frame carries dependencies either way.
    block.set_compress_mode(args.compress_mode)
    exact = args.compress_mode == "exact"
    n = args.blocks or 64
    plain = load_fixture(block, "compression_66k_JSON")
    data = workloads.json_tiles(plain, n * BLOCK, phase=rank * 7919, device="cpu").numpy().tobytes()
    fi = F.FrameInfo(block_size=F.BlockSize.Max64KB, block_mode=F.BlockMode.Linked)
    state = {}

    def step(s):
        fr = F.compress_frame(data, fi)
        state["frame"] = fr
        state["back"] = F.decompress_frame(fr, len(data))[0]

    elapsed = timed(args, env, step)
    oracle_checked = None
    if not args.no_verify:
        assert state["back"] == data
        # the reference's FrameDecoder (oracle restatement) over the frame the GPU wrote, once, after the timed loop
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_api as O
        rc, back, used = O.frame_decompress(state["frame"], len(data))
        assert rc == 0 and back == data and used == len(state["frame"]), "the reference's FrameDecoder (oracle) does not return the data"
        oracle_checked = "frame (%d bytes) decoded by the oracle's FrameDecoder == the data" % len(state["frame"])
    total = len(data) * world
    alg = len(data) + len(state["frame"])
    # BASELINE calls this configuration a sequential-dependency stress.  Since round 4 the timed frame IS dependency-carrying in
    # compress_mode fast too (every block but the first refers to its predecessor's bytes; round 3's blocks were parsed on their
    # own).  The same data also goes through the reference-exact chain encoder (lz4_flex's bytes) and back, outside the timed loop:
    # that is what the reference's own byte stream costs on this design.
    dep = None
    if rank == 0 and not exact and not args.no_verify:
        try:
            block.set_compress_mode("exact")
            t0 = time.perf_counter(); fr_x = F.compress_frame(data, fi); torch.cuda.synchronize(); t1 = time.perf_counter()
            back_x = F.decompress_frame(fr_x, len(data))[0]; torch.cuda.synchronize(); t2 = time.perf_counter()
            assert back_x == data
            rc_o, fr_o = O.frame_compress(data, block_mode=1, block_size=4)
            dep = {"what": "the same data through the reference-exact chain encoder (lz4_flex's own bytes) and the chained decoder, one pass, untimed warm caches",
                   "compress_ms": round((t1 - t0) * 1e3, 2), "decompress_ms": round((t2 - t1) * 1e3, 2),
                   "round_trip_MiB_per_s": round(len(data) / 1048576 / (t2 - t0), 2), "ratio": round(len(fr_x) / len(data), 5),
                   "frame_equals_oracle_FrameEncoder": bool(rc_o == 0 and fr_o == fr_x)}
        except Exception as e:
            dep = {"error": repr(e)}
        finally:
            block.set_compress_mode(args.compress_mode)
    many = None
    if rank == 0:
        try:
            many = many_linked_streams(args, env, plain)
        except Exception as e:
            many = {"error": repr(e)}
    base = {"note": "--no-cpu-baseline"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            base = cpu_baseline_linked(data)
        except Exception as e:
            base = {"error": repr(e)}
    return {
        "metric": "MiB/s frame compress + decompress, BlockLinked 64 KiB blocks, host buffers",
        "value": round((total / 1048576) / (elapsed / args.steps), 2), "unit": "MiB/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8",
        "data": "synthetic: JSON tiles, %d x 64 KiB blocks in ONE Linked frame per GPU" % n,
        "config": {"workload": "BASELINE configs[4]: frame format, BlockLinked 64 KiB blocks -- sequential-dependency stress: the blocks of a frame are "
                               "decoded by ONE chained launch (lz4_decompress_pcd_kernel: every block's token chain at once; a block waits for its "
                               "predecessors only where a match reaches behind its start)" +
                               ("; compress_mode exact: the chain encoder does too (the reference's bytes)" if exact else
                                "; compress_mode fast: one encoder launch, every block's matches reach up to 64 KiB back into the blocks before it "
                                "(32 KiB of history in front of each block, windows advancing by 32 KiB): a dependency-carrying frame, ratio below the reference's") +
                               "; host buffers (PCIe included)",
                   "blocks": n, "compress_mode": args.compress_mode},
        "ratio": round(len(state["frame"]) / len(data), 5),
        "roofline": dict(roof(alg, elapsed / args.steps), kernel=("lz4_compress_chain_kernel" if exact else "lz4_compress_wave_kernel") + " + lz4_decompress_pcd_kernel (chained batch)"),
        "verified": "NOT VERIFIED" if args.no_verify else "frame round trip bit-exact; " + oracle_checked,
        "dependency_carrying_frame": dep,
        "many_streams": many,
        "cpu_baseline": base,
    }


def many_linked_streams(args, env, plain, n=256, size=4 << 20):
    """config 5 in the shape that has parallelism in it: n independent streams, a Linked frame of 64 KiB blocks each, device-resident
    (lz4flex_frame_compress_many / lz4flex_frame_decompress_many: all blocks of all streams in one encoder launch; one chained decode
    launch, n chains side by side).  Wall clock around the calls (they return when the work is done), median of 5 after 2 warm-ups;
    every frame is then decoded by the oracle's FrameDecoder."""
    torch, dev = env["torch"], env["dev"]
    from lz4_flex_amd import _lib, frame as F, workloads
    lib = _lib.load()
    fi = F.FrameInfo(block_size=F.BlockSize.Max64KB, block_mode=F.BlockMode.Linked)
    src = workloads.json_tiles(plain, n * size, device=dev)
    cap = int(lib.lz4flex_frame_compress_bound(size, fi._c()))
    frames = torch.zeros(n * cap, dtype=torch.uint8, device=dev)
    back = torch.zeros(n * size, dtype=torch.uint8, device=dev)
    in_off, f_off = [i * size for i in range(n)], [i * cap for i in range(n)]
    tc, td, flen = [], [], None
    for it in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        flen, st = F.compress_frames_device(src, in_off, [size] * n, fi, frames, f_off, [cap] * n)
        t1 = time.perf_counter()
        assert st == [0] * n
        olen, st = F.decompress_frames_device(frames, f_off, flen, back, in_off, [size] * n)
        t2 = time.perf_counter()
        assert st == [0] * n and olen == [size] * n
        if it >= 2:
            tc.append(t1 - t0); td.append(t2 - t1)
    tc.sort(); td.sort()
    c_ms, d_ms = tc[len(tc) // 2] * 1e3, td[len(td) // 2] * 1e3
    checked = None
    if not args.no_verify:
        assert torch.equal(back, src)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_api as O
        host = src.cpu().numpy()
        fh = frames.cpu().numpy()
        for i in range(n):
            f = fh[f_off[i]:f_off[i] + flen[i]].tobytes()
            rc, b, used = O.frame_decompress(f, size)
            assert rc == 0 and used == len(f) and b == host[i * size:(i + 1) * size].tobytes(), "stream %d: the oracle's FrameDecoder does not return it" % i
        checked = "round trip bit-exact on the device; each of the %d frames decoded by the oracle's FrameDecoder == its stream" % n
    out = {"what": "%d streams x %d MiB, one Linked frame of 64 KiB blocks each, device-resident, compress_mode %s" % (n, size >> 20, args.compress_mode),
           "compress_ms": round(c_ms, 3), "decompress_ms": round(d_ms, 3),
           "round_trip_MiB_per_s": round(n * size / 1048576 / ((c_ms + d_ms) / 1e3), 1),
           "round_trip_GiB_per_s": round(n * size / (1 << 30) / ((c_ms + d_ms) / 1e3), 2),
           "ratio": round(sum(flen) / (n * size), 5), "verified": checked or "NOT VERIFIED"}
    if not args.no_cpu_baseline and not args.no_verify:
        out["cpu_every_thread_a_frame"] = cpu_many_frames(host[:size].tobytes())
    return out


def cpu_many_frames(data):
    """the reference's answer to many Linked streams: a thread per stream (oracle FrameEncoder + FrameDecoder, every hardware thread)"""
    import threading
    O = _oracle_fresh()
    hw, _phys = host_topology()
    o = O.lib()
    fi = O.frame_info(block_mode=1, block_size=4)
    cap = len(data) + len(data) // 100 + (len(data) // 65536 + 2) * 16 + 64

    def one_pass(reps):
        out = C.create_string_buffer(cap)
        back = C.create_string_buffer(len(data))
        used = C.c_size_t(0)
        d = O.ErrDetail()
        for _ in range(reps):
            n = o.lz4o_frame_compress(data, len(data), None, 0, C.byref(fi), out, cap, C.byref(d))
            m = o.lz4o_frame_decompress(out, n, back, len(data), C.byref(used), C.byref(d))
            assert n > 0 and m == len(data)
    best, reps = 0.0, 1
    for attempt in range(6):
        th = [threading.Thread(target=one_pass, args=(reps,)) for _ in range(hw)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        if dt >= 0.3:
            best = max(best, hw * reps * len(data) / 1048576 / dt)
            if attempt >= 3:
                break
        else:
            reps = max(reps + 1, int(reps * 0.4 / max(dt, 1e-3)))
    return {"threads": hw, "round_trip_MiB_per_s": round(best, 1), "sample": "one 4 MiB stream per thread, oracle FrameEncoder + FrameDecoder, passes of >= 0.3 s"}


# --------------------------------------------------------------------------------------- CPU baseline
def _oracle_fresh():
    """the oracle (test infrastructure) rebuilt ON THIS NODE: -march=native must mean the node that times it"""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    subprocess.check_call(["make", "-B", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    import oracle_api as O
    return O


def host_topology():
    """(hardware threads this process may run on, physical cores among them): the thread counts the CPU baseline is timed at"""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    cores = set()
    try:
        cur = {}
        with open("/proc/cpuinfo") as f:
            for line in f.read().split("\n") + [""]:
                if ":" in line:
                    k, v = line.split(":", 1)
                    cur[k.strip()] = v.strip()
                elif cur:
                    if int(cur.get("processor", -1)) in cpus:
                        cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                    cur = {}
    except OSError:
        pass
    return len(cpus), (len(cores) or len(cpus))


def _time_codecs(O, h_src, ns, bs, stride):
    """oracle port and system liblz4 1.9.3 over ns blocks of bs bytes.  A persistent thread pool per measurement (threads are
    created outside the timed passes), passes of >= 0.3 s (the sweep is repeated inside a pass), best of 3; at three thread
    counts: every hardware thread, one thread per physical core, one thread (over an eighth of the sample)."""
    import numpy as np
    o = O.lib()
    hw, phys = host_topology()
    h_in_off = (np.arange(ns, dtype=np.uint64) * bs)
    h_in_len = np.full(ns, bs, dtype=np.uint32)
    h_out = np.zeros(ns * stride, dtype=np.uint8)
    h_out_off = (np.arange(ns, dtype=np.uint64) * stride)
    h_out_cap = np.full(ns, stride, dtype=np.uint32)
    h_out_len = np.zeros(ns, dtype=np.uint32)
    h_back = np.zeros(ns * bs, dtype=np.uint8)
    h_back_len = np.zeros(ns, dtype=np.uint32)

    def vp(a):
        return C.c_void_p(a.ctypes.data)

    res = {}
    try:
        l4 = O.clz4()
        fns = {"liblz4_1.9.3": (C.cast(l4.LZ4_compress_default, C.c_void_p), C.cast(l4.LZ4_decompress_safe, C.c_void_p))}
    except Exception:
        fns = {}
    fns["oracle_port"] = (None, None)
    mib = ns * bs / 1048576
    counts = [hw] + ([phys] if phys != hw else []) + [1]
    for name, (fc, fd) in fns.items():
        for threads in counts:
            sub = ns if threads > 1 else max(ns // 8, 1)                 # one thread: an eighth of the sample
            used = C.c_int(0)
            tc = o.lz4o_bench_pool(0, fc, vp(h_src), vp(h_in_off), vp(h_in_len), vp(h_out), vp(h_out_off), vp(h_out_cap),
                                   vp(h_out_len), sub, threads, 3, 0.3, C.byref(used))
            td = o.lz4o_bench_pool(1, fd, vp(h_out), vp(h_out_off), vp(h_out_len), vp(h_back), vp(h_in_off), vp(h_in_len),
                                   vp(h_back_len), sub, threads, 3, 0.3, C.byref(used))
            assert tc > 0 and td > 0, "thread pool could not be created"
            assert (h_back[:sub * bs] == h_src[:sub * bs]).all(), name
            m = mib * sub / ns
            res[(name, threads)] = (m / tc, m / td, m / (tc + td), float(h_out_len[:sub].sum()) / (sub * bs), used.value)
    return res, (hw, phys), h_out, h_out_len


def _baseline_dict(res, topo, sample):
    hw, phys = topo
    best_t = max((t for (nm, t) in res if nm == "oracle_port" and t > 1), key=lambda t: res[("oracle_port", t)][2], default=1)
    port = res[("oracle_port", best_t)]

    def leg(r):
        return {"compress_MiB_per_s": round(r[0], 1), "decompress_MiB_per_s": round(r[1], 1), "round_trip_MiB_per_s": round(r[2], 1),
                "threads_run": r[4]}

    out = {"value": round(port[2], 1), "unit": "MiB/s", "cores": best_t, "kind": "port", "sample": sample,
           "compress_MiB_per_s": round(port[0], 1), "decompress_MiB_per_s": round(port[1], 1),
           "host": {"hardware_threads": hw, "physical_cores": phys},
           "by_threads": {str(t): leg(res[("oracle_port", t)]) for (nm, t) in sorted(res) if nm == "oracle_port"},
           "single_thread": leg(res[("oracle_port", 1)])}
    if ("liblz4_1.9.3", hw) in res:
        a = res[("liblz4_1.9.3", max((t for (nm, t) in res if nm == "liblz4_1.9.3" and t > 1), key=lambda t: res[("liblz4_1.9.3", t)][2], default=1))]
        out["liblz4_1.9.3"] = {"note": "system C liblz4 (LZ4_compress_default / LZ4_decompress_safe), the library the reference's tests "
                                       "cross-check against; same sample, same thread counts, same pool",
                               "round_trip_MiB_per_s": round(a[2], 1), "compress_MiB_per_s": round(a[0], 1),
                               "decompress_MiB_per_s": round(a[1], 1), "ratio": round(a[3], 5),
                               "by_threads": {str(t): leg(res[("liblz4_1.9.3", t)]) for (nm, t) in sorted(res) if nm == "liblz4_1.9.3"},
                               "single_thread": leg(res[("liblz4_1.9.3", 1)])}
    return out


def cpu_baseline(src, comp, comp_off, comp_len, n, mode, stash=None):
    """Times oracle/ (kind 'port': lz4_flex is Rust, no toolchain here) and the system liblz4 on the same bytes as the GPU: the
    whole batch on all host threads / all physical cores, an eighth of it on one thread; then the ORACLE's decoder (= lz4_flex's,
    restated) decodes EVERY block the GPU encoder wrote and must return the input."""
    import numpy as np
    O = _oracle_fresh()
    ns = n
    h_src = src[:ns * BLOCK].cpu().numpy()
    stride = int(comp_off[1].item()) if n > 1 else 72128
    res, topo, h_out, h_out_len = _time_codecs(O, h_src, ns, BLOCK, stride)
    if stash is not None:
        stash["oracle_blocks"] = (h_out, h_out_len)      # the oracle port's output of its last compress pass: every block, at the GPU's stride
    g_len = comp_len[:ns].cpu().numpy().astype(np.uint32)
    g = comp[:ns * stride].cpu().numpy()
    if mode == "exact":                            # the reference-exact encoder: same sizes and bytes as the oracle
        assert (h_out_len == g_len).all(), "GPU encoder output size differs from the oracle"
        for i in range(ns):
            assert (g[i * stride:i * stride + int(g_len[i])] == h_out[i * stride:i * stride + int(g_len[i])]).all(), i
    # either encoder: the reference's decoder (oracle) over every GPU-written block, in one batch call
    g_off = (np.arange(ns, dtype=np.uint64) * stride)
    b_off = (np.arange(ns, dtype=np.uint64) * BLOCK)
    b_cap = np.full(ns, BLOCK, dtype=np.uint32)
    b_len = np.zeros(ns, dtype=np.uint32)
    back = np.zeros(ns * BLOCK, dtype=np.uint8)

    def vp(a):
        return C.c_void_p(a.ctypes.data)
    O.lib().lz4o_bench_pool(1, None, vp(g), vp(g_off), vp(g_len), vp(back), vp(b_off), vp(b_cap), vp(b_len), ns, topo[1], 1, 0.0, None)
    assert (b_len == BLOCK).all(), "the reference decoder (oracle) rejects %d GPU-encoded block(s)" % int((b_len != BLOCK).sum())
    assert (back == h_src).all(), "the reference decoder (oracle) does not return the input for a GPU-encoded block"
    d = _baseline_dict(res, topo, "all %d blocks of the workload (%d MiB), compress + decompress round trip; thread pool created outside the "
                       "timed passes, passes of >= 0.3 s, best of 3; one thread: an eighth of the blocks; oracle = C restatement of "
                       "lz4_flex's block codec, rebuilt on this node with gcc -O3 -march=native" % (ns, ns * BLOCK >> 20))
    d["oracle_ratio"] = round(res[("oracle_port", topo[0])][3], 5)
    d["gpu_blocks_decoded_by_oracle"] = int(ns)
    return d


def cpu_baseline_linked(data):
    """config 5: the oracle's FrameEncoder + FrameDecoder (lz4_flex's, restated) on the same bytes, Linked 64 KiB blocks: one thread
    on one frame (a Linked frame is one dependency chain: that is its CPU speed), and every hardware thread on a frame each
    (what the host does with many such streams).  Passes of >= 0.3 s, best of 3."""
    import threading
    O = _oracle_fresh()
    hw, phys = host_topology()
    rc, fr = O.frame_compress(data, block_mode=1, block_size=4)
    assert rc == 0
    assert O.frame_decompress(fr, len(data))[1] == data
    o = O.lib()
    fi = O.frame_info(block_mode=1, block_size=4)
    cap = len(data) + len(data) // 100 + (len(data) // 65536 + 2) * 16 + 64

    def one_pass(reps):          # the oracle's C entry points on preallocated buffers: no Python copies inside the timed passes
        out = C.create_string_buffer(cap)
        back = C.create_string_buffer(len(data))
        used = C.c_size_t(0)
        d = O.ErrDetail()
        for _ in range(reps):
            n = o.lz4o_frame_compress(data, len(data), None, 0, C.byref(fi), out, cap, C.byref(d))
            m = o.lz4o_frame_decompress(out, n, back, len(data), C.byref(used), C.byref(d))
            assert n == len(fr) and m == len(data)

    def rate(threads):
        reps, best = 1, 0.0
        for attempt in range(8):
            th = [threading.Thread(target=one_pass, args=(reps,)) for _ in range(threads)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            dt = time.perf_counter() - t0
            if dt >= 0.3:
                best = max(best, threads * reps * len(data) / 1048576 / dt)
                if attempt >= 4:
                    break
            else:
                reps = max(reps + 1, int(reps * 0.4 / max(dt, 1e-3)))
        return best
    one = rate(1)
    allc = rate(hw)
    return {"value": round(one, 1), "unit": "MiB/s", "cores": 1, "kind": "port",
            "sample": "the same %d bytes, one Linked frame of 64 KiB blocks: oracle FrameEncoder + FrameDecoder round trip on ONE thread (a Linked frame is one "
                      "dependency chain); passes of >= 0.3 s, best of 3; oracle rebuilt on this node with gcc -O3 -march=native" % len(data),
            "all_threads": {"threads": hw, "round_trip_MiB_per_s": round(allc, 1),
                            "note": "every hardware thread encodes + decodes its own copy of the frame (ctypes calls release the GIL)"},
            "host": {"hardware_threads": hw, "physical_cores": phys}, "oracle_ratio": round(len(fr) / len(data), 5)}


def cpu_baseline_buffer(buf, bs):
    O = _oracle_fresh()
    h = buf.cpu().numpy()
    ns = h.size // bs
    stride = int(O.max_out(bs)) + 64
    res, topo, _o, _l = _time_codecs(O, h, ns, bs, stride)
    d = _baseline_dict(res, topo, "first %d blocks of %d bytes of rank 0's stream, block codec only (no frame bytes); thread pool created "
                       "outside the timed passes, passes of >= 0.3 s, best of 3" % (ns, bs))
    d["oracle_ratio"] = round(res[("oracle_port", topo[0])][3], 5)
    return d


if __name__ == "__main__":
    main()
