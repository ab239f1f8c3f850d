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


@pytest.fixture(scope="module")
def isa():
    from lz4_flex_amd import build
    return build.wave_isa()


def test_async_loads_are_not_touched_before_their_wait(isa):
    """the check lz4_flex_amd/build.py runs at build time (a failure there switches the file to plain loads): with today's
    toolchain it passes, and the pipeline is really there (prologue + unrolled steady state)"""
    from lz4_flex_amd import build
    ok, msg, n_loads, n_waits = build.check_async_loads(isa)
    assert ok, msg
    assert n_loads >= 16 and n_waits >= 8
    assert build.wave_extra_flags() == []


def test_the_check_sees_a_touched_register(isa):
    """negative control: the same listing with one instruction inserted between a marked load and its wait"""
    from lz4_flex_amd import build
    import re
    for i, line in enumerate(isa):
        if "lz4w-load" in line and "dwordx4" in line:
            dst = re.search(r"global_load_dwordx4\s+v\[(\d+):(\d+)\]", line)
            bad = isa[:i + 1] + ["\tv_mov_b32_e32 v0, v%s" % dst.group(1)] + isa[i + 1:]
            ok, msg, _l, _w = build.check_async_loads(bad)
            assert not ok and "not waited for yet" in msg
            return
    raise AssertionError("no marked load in the listing")


def test_plain_load_fallback_compiles_without_hand_counted_waits():
    """what the build switches to when the check fails: no marked loads, no hand-counted waits"""
    from lz4_flex_amd import build
    lines = build.wave_isa([build.PLAIN_LOADS])
    assert not any("lz4w-load" in l or "lz4w-wait" in l for l in lines)
    assert any("global_load_dwordx4" in l for l in lines)


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
        # tolerate reloads of loop-invariant pointers; stores (a live value being spilled inside a loop): none in the indexer, at most ONE in
        # match_segment -- since encode_seqs is inlined (round 6) half of the cand[] prefetch is parked in scratch around the lane-parallel
        # stores of a call, once per ~4 supersteps (measured with it: profiles/r06_encoder.txt); a second one means a hot value is spilled
        stores = [l for l in inside if "scratch_store" in l]
        assert len(stores) <= (1 if fn == "match_segment" else 0), (fn, inside[:5])


# ---- the replay decoder (lz4_decompress_replay.hip): same discipline, tag "lz4r"
@pytest.fixture(scope="module")
def replay_isa():
    from lz4_flex_amd import build
    return build.replay_isa()


def test_replay_async_loads_are_not_touched_before_their_wait(replay_isa):
    from lz4_flex_amd import build
    ok, msg, n_loads, n_waits = build.check_async_loads(replay_isa, "lz4r")
    assert ok, msg
    assert n_loads >= 48 and n_waits >= 24
    assert build.replay_extra_flags() == []


def test_the_check_sees_a_touched_replay_register(replay_isa):
    from lz4_flex_amd import build
    for i, line in enumerate(replay_isa):
        if "lz4r-load" in line and "dwordx4" in line:
            dst = re.search(r"global_load_dwordx4\s+v\[(\d+):(\d+)\]", line)
            bad = replay_isa[:i + 2] + ["\tv_mov_b32_e32 v0, v%s" % dst.group(1)] + replay_isa[i + 2:]   # (behind the line that restores exec)
            ok, msg, _l, _w = build.check_async_loads(bad, "lz4r")
            assert not ok and "not waited for yet" in msg
            return
    raise AssertionError("no marked load in the listing")


def test_replay_plain_load_fallback_compiles_without_hand_counted_waits():
    from lz4_flex_amd import build
    lines = build.replay_isa([build.REPLAY_PLAIN_LOADS])
    assert not any("lz4r-load" in l or "lz4r-wait" in l for l in lines)
    assert any("global_load_dwordx4" in l for l in lines)
    build.replay_isa()      # (leave the default listing behind)


# ---- the fused decoder (lz4_decompress_fused.hip): the replay decoder's loads behind an LDS step queue, tag "lz4f"
@pytest.fixture(scope="module")
def fused_isa():
    from lz4_flex_amd import build
    return build.fused_isa()


def test_fused_async_loads_are_not_touched_before_their_wait(fused_isa):
    from lz4_flex_amd import build
    ok, msg, n_loads, n_waits = build.check_async_loads(fused_isa, "lz4f")
    assert ok, msg
    assert n_loads >= 8 and n_waits >= 8
    assert build.fused_extra_flags() == []
    # no scratch memory at all: the quads keep LOOKAHEAD slots of five registers each alive around the turn
    assert not any("scratch_" in l for l in fused_isa)


def test_the_check_sees_a_touched_fused_register(fused_isa):
    from lz4_flex_amd import build
    for i, line in enumerate(fused_isa):
        if "lz4f-load" in line and "dwordx4" in line:
            dst = re.search(r"global_load_dwordx4\s+v\[(\d+):(\d+)\]", line)
            bad = fused_isa[:i + 2] + ["\tv_mov_b32_e32 v0, v%s" % dst.group(1)] + fused_isa[i + 2:]   # (behind the line that restores exec)
            ok, msg, _l, _w = build.check_async_loads(bad, "lz4f")
            assert not ok and "not waited for yet" in msg
            return
    raise AssertionError("no marked load in the listing")


def test_fused_plain_load_fallback_compiles_without_hand_counted_waits():
    from lz4_flex_amd import build
    lines = build.fused_isa([build.FUSED_PLAIN_LOADS])
    assert not any("lz4f-load" in l or "lz4f-wait" in l for l in lines)
    assert any("global_load_dwordx4" in l for l in lines)
    build.fused_isa()      # (leave the default listing behind)
