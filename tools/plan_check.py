#!/usr/bin/env python3
"""The PLAN kernel alone on the GPU, its plans replayed by the HOST MODEL (tests/sim/plan_model.cpp: the replay kernel's lanes byte
for byte, with guards): says for every block whether the kernel made a plan, and if the plan is wrong, which guard fires or where the
bytes differ.  A tool (kernel development)."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=256)
    ap.add_argument("--data", default="json")
    ap.add_argument("--adversarial", action="store_true")
    args = ap.parse_args()
    import numpy as np
    import torch
    import corpus
    import oracle_api as O
    import plan_model as M
    import wave_model as W
    from lz4_flex_amd import _lib as L, workloads
    lib = L.load()
    lib.lz4flex_debug_plan.restype = C.c_int
    lib.lz4flex_debug_plan.argtypes = [C.c_void_p] * 5 + [C.c_uint] + [C.c_void_p] * 5
    lib.lz4flex_debug_plan_slot_words.restype = C.c_uint
    slot = lib.lz4flex_debug_plan_slot_words()
    dev = torch.device("cuda", 0)
    if args.adversarial:
        cases = [(c, k, O.decompress(c, k)) for c, k in corpus.adversarial_blocks()]
    else:
        B = 65536
        plain = O.fixture_plain("compression_66k_JSON" if args.data == "json" else "compression_65k")
        src = bytes(workloads.json_tiles(plain, args.blocks * B, device="cpu").numpy())
        cases = []
        for i in range(args.blocks):
            d = src[i * B:(i + 1) * B]
            c = W.compress(d) if i % 2 == 0 else O.compress(d)
            cases.append((c, B, ("ok", d)))
    n = len(cases)
    in_len = np.array([len(c) for c, _, _ in cases], dtype=np.uint32)
    in_off = np.concatenate([[0], np.cumsum(in_len[:-1].astype(np.uint64) + 0)]).astype(np.uint64)
    h_in = np.frombuffer(b"".join(c for c, _, _ in cases) + b"", dtype=np.uint8).copy()
    caps = np.array([k for _, k, _ in cases], dtype=np.uint32)
    out_off = np.concatenate([[0], np.cumsum(caps[:-1].astype(np.uint64) + 64)]).astype(np.uint64)
    t = lambda a: torch.from_numpy(a).to(dev)
    d_in, d_ioff, d_ilen, d_ooff, d_cap = t(h_in if h_in.size else np.zeros(1, np.uint8)), t(in_off.astype(np.int64)), t(in_len.astype(np.int32)), t(out_off.astype(np.int64)), t(caps.astype(np.int32))
    d_plans = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
    d_words = torch.zeros(n * slot, dtype=torch.int32, device=dev)
    d_olen = torch.zeros(n, dtype=torch.int32, device=dev)
    d_st = torch.full((n,), -1, dtype=torch.int32, device=dev)
    p = lambda x: C.c_void_p(x.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(2):
        e0.record()
        rc = lib.lz4flex_debug_plan(p(d_in), p(d_ioff), p(d_ilen), p(d_ooff), p(d_cap), n, p(d_plans), p(d_words), p(d_olen), p(d_st),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
        e1.record()
        torch.cuda.synchronize()
        assert rc == 0
    print("plan kernel: %d blocks in %.3f ms" % (n, e0.elapsed_time(e1)))
    plans = d_plans.cpu().numpy().view(np.uint32).reshape(n, 8)
    words = d_words.cpu().numpy().view(np.uint32)
    olen = d_olen.cpu().numpy()
    st = d_st.cpu().numpy()
    u32p = C.POINTER(C.c_uint32)
    kinds = {}
    shown = 0
    for i, (c, k, exp) in enumerate(cases):
        first_word, tail_word, tail_op, nt_fl = int(plans[i][4]), int(plans[i][5]), int(plans[i][6]), int(plans[i][7])
        n_tail, flags = nt_fl & 0xFFFF, nt_fl >> 16
        if flags != 0 or st[i] != 0:
            verdict = "no plan (status %#x, reason %d)" % (int(st[i]) & 0xFFFFFFFF, tail_op)
            key = "no-plan/" + ("valid reason %d" % tail_op if exp[0] == "ok" else "invalid")
        elif exp[0] != "ok":
            verdict, key = "PLAN FOR AN INVALID BLOCK (%s)" % exp[0], "plan-for-invalid"
        else:
            E = int(olen[i])
            out = C.create_string_buffer(max(k, 1))
            w = words[first_word:first_word + slot].copy()
            code = M.lib().plan_replay(c, len(c), w.ctypes.data_as(u32p), tail_word - first_word, n_tail, E, out, k)
            if code != 0:
                verdict, key = "replay guard %d (E %d, want %d, tail_op %d, n_tail %d)" % (code, E, len(exp[1]), tail_op, n_tail), "guard %d" % code
            elif out.raw[:E] != exp[1]:
                verdict, key = "bytes differ (E %d, want %d)" % (E, len(exp[1])), "bytes"
            else:
                verdict, key = "ok", "ok"
        kinds[key] = kinds.get(key, 0) + 1
        if key not in ("ok", "no-plan/invalid") and shown < 12:
            print("  block %d (%d compressed): %s" % (i, len(c), verdict))
            shown += 1
    print(sorted(kinds.items()))


if __name__ == "__main__":
    main()
