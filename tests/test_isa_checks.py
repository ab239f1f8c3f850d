"""CPU: properties of the generated gfx950 ISA that the source relies on but the compiler does not guarantee.

lz4_compress_wave.hip's indexer issues its window loads as inline assembly, IDX_DEPTH chunks ahead, and waits for
them with hand-counted `s_waitcnt vmcnt(N)`.  The compiler does not know these registers are in flight: if it copied
or spilled one between the load and its wait it would copy garbage, silently.  This test compiles the file to assembly
and checks that no instruction touches a destination register between its "lz4w-load" and the "lz4w-wait" naming it."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "lz4_flex_amd", "csrc", "lz4_compress_wave.hip")


def _regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def _mentioned(line):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|\bv\d+\b", line.split(";")[0]):
        out |= _regs(tok)
    return out


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = str(tmp_path_factory.mktemp("isa") / "wave.s")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only", "-S", SRC,
                           "-o", out], stderr=subprocess.DEVNULL)
    return open(out).read().splitlines()


def test_async_loads_are_not_touched_before_their_wait(isa):
    in_flight = {}          # register -> line number of the load
    n_loads = n_waits = 0
    for ln, line in enumerate(isa, 1):
        code = line.split(";")[0].strip()
        if "lz4w-load" in line:
            n_loads += 1
            dst = re.search(r"global_load_dword(?:x4)?\s+(v\[\d+:\d+\]|v\d+),\s*(v\[\d+:\d+\])", code)
            assert dst, line
            # the address registers may alias the destination (read at issue): only the destination is in flight afterwards
            for r in _regs(dst.group(1)):
                in_flight[r] = ln
            continue
        if "lz4w-wait" in line:
            n_waits += 1
            named = set()
            for tok in re.findall(r"v\[\d+:\d+\]|\bv\d+\b", line.split("lz4w-wait")[1]):
                named |= _regs(tok)
            assert named, line
            if re.search(r"vmcnt\(0\)", code):
                in_flight.clear()                       # everything has landed
            else:
                for r in named:
                    in_flight.pop(r, None)
            continue
        if not code or code.endswith(":") or code.startswith("."):
            continue
        if code.startswith("s_waitcnt") and "vmcnt(0)" in code:
            in_flight.clear()
            continue
        bad = _mentioned(line) & set(in_flight)
        assert not bad, "line %d touches v%s, requested at line %d and not waited for yet: %s" % (
            ln, sorted(bad), min(in_flight[r] for r in bad), line.strip())
    assert n_loads >= 16 and n_waits >= 8          # the pipeline is really there (prologue + unrolled steady state)


def test_no_spills_inside_the_hot_functions(isa):
    """scratch traffic is allowed only as the callee-saved save / restore at function entry / exit"""
    text = "\n".join(isa)
    for fn in ("index_window", "match_segment"):
        m = re.search(r"^(_ZN11lz4flex_dev4wave\d+%s\w*):.*?^\.Lfunc_end\d+:" % fn, text, re.S | re.M)
        assert m, fn
        body = m.group(0).splitlines()
        spill_lines = [i for i, l in enumerate(body) if "scratch_" in l]
        loops = [i for i, l in enumerate(body) if "Loop Header" in l]
        assert loops, fn
        first_loop, last_branch = loops[0], max(i for i, l in enumerate(body) if "s_cbranch" in l)
        inside = [body[i].strip() for i in spill_lines if first_loop < i < last_branch and "Folded" in body[i]]
        # tolerate reloads of loop-invariant pointers, never stores (a store inside a loop is a live value being spilled)
        assert not [l for l in inside if "scratch_store" in l], (fn, inside[:5])
