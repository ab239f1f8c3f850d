"""Build the in-tree HIP shared library (gfx950) with hipcc.  No JIT cache, no pip install:
the .so lands next to this file so it travels with the repository snapshot.

Staleness is decided by CONTENT, not by a hand-kept dependency list: every file under csrc/ and include/ plus
the compiler flags are hashed; the hash is stored next to the library (build/source_hash.txt) and compiled into
it (`lz4flex_build_id()`), so a test can prove that the loaded binary was built from the sources in the tree."""
import glob
import hashlib
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(HERE, "..", "include")
LIB = os.path.join(HERE, "liblz4flex_amd.so")
BDIR = os.path.join(HERE, "build")
STAMP = os.path.join(BDIR, "source_hash.txt")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def sources():
    """every translation unit under csrc/ (kernels: *.hip, host: *.cpp)"""
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def dep_files():
    """everything a translation unit can include: all of csrc/ and include/"""
    out = []
    for d in (CSRC, INCLUDE):
        for p in sorted(glob.glob(os.path.join(d, "*"))):
            if os.path.isfile(p) and p.rsplit(".", 1)[-1] in ("hip", "cpp", "h", "hpp", "inc"):
                out.append(p)
    return out


def source_hash():
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for p in dep_files():
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP kernels are the product; there is no fallback build")


def built_hash():
    try:
        with open(STAMP) as f:
            return f.read().strip()
    except OSError:
        return None


def needs_build():
    return not os.path.exists(LIB) or built_hash() != source_hash()


# ---- a property of the generated ISA that lz4_compress_wave.hip relies on and the compiler does not guarantee ----------------
def _vregs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check_async_loads(isa_lines, tag="lz4w"):
    """The encoder's indexer (tag "lz4w") and the replay decoder (tag "lz4r") issue loads as inline assembly, several steps
    ahead, and wait for them with hand-counted `s_waitcnt vmcnt(N)` (marked "<tag>-load" / "<tag>-wait <registers>").  The
    compiler does not know these registers are in flight: an instruction that reads, copies or spills one between the load and
    its wait would move garbage, silently.  Returns (ok, message, loads, waits) for a `hipcc -S` listing of that file."""
    load_mark, wait_mark = tag + "-load", tag + "-wait"
    in_flight = {}          # register -> line number of the load
    n_loads = n_waits = 0
    for ln, line in enumerate(isa_lines, 1):
        code = line.split(";")[0].strip()
        if load_mark in line:
            n_loads += 1
            dst = re.search(r"global_load_dword(?:x[24])?\s+(v\[\d+:\d+\]|v\d+),\s*(v\[\d+:\d+\])", code)
            if not dst:
                return False, "line %d: unexpected form of a marked load: %s" % (ln, line.strip()), n_loads, n_waits
            for r in _vregs(dst.group(1)):      # (the address registers may alias the destination: read at issue)
                in_flight[r] = ln
            continue
        if wait_mark in line:
            n_waits += 1
            named = set()
            for tok in re.findall(r"v\[\d+:\d+\]|\bv\d+\b", line.split(wait_mark)[1]):
                named |= _vregs(tok)
            if not named:
                return False, "line %d: a wait that names no register: %s" % (ln, line.strip()), n_loads, n_waits
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
        touched = set()
        for tok in re.findall(r"v\[\d+:\d+\]|\bv\d+\b", code):
            touched |= _vregs(tok)
        bad = touched & set(in_flight)
        if bad:
            return False, "line %d touches v%s, requested at line %d and not waited for yet: %s" % (
                ln, sorted(bad), min(in_flight[r] for r in bad), line.strip()), n_loads, n_waits
    return True, "", n_loads, n_waits


WAVE_SRC = "lz4_compress_wave.hip"
PLAIN_LOADS = "-DLZ4W_PLAIN_LOADS"     # the indexer's loads as ordinary C++ loads: slower (the compiler sinks them to their use), always right


REPLAY_SRC = "lz4_decompress_replay.hip"
REPLAY_PLAIN_LOADS = "-DLZ4R_PLAIN_LOADS"   # the replay decoder's loads in every lane, waited for by the compiler: slower, always right


FUSED_SRC = "lz4_decompress_fused.hip"
FUSED_PLAIN_LOADS = "-DLZ4F_PLAIN_LOADS"     # the fused decoder's loads under a plain branch, waited for by the compiler: slower, always right


def file_isa(src, extra_flags=()):
    """the `hipcc -S` listing (device code) of one source file with the build's flags"""
    out = os.path.join(BDIR, src.rsplit(".", 1)[0] + "_check.s")
    os.makedirs(BDIR, exist_ok=True)
    cmd = [_hipcc()] + [f for f in FLAGS if f != "-fPIC"] + list(extra_flags) + ["-x", "hip", "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("hipcc -S failed on %s:\n%s" % (src, r.stdout.decode(errors="replace")))
    with open(out) as f:
        return f.read().splitlines()


def wave_isa(extra_flags=()):
    """the listing of the throughput encoder"""
    return file_isa(WAVE_SRC, extra_flags)


def replay_isa(extra_flags=()):
    """the listing of the replay decoder"""
    return file_isa(REPLAY_SRC, extra_flags)


def replay_extra_flags():
    """[] when this toolchain leaves the replay kernel's in-flight registers alone, else [REPLAY_PLAIN_LOADS]"""
    ok, msg, loads, waits = check_async_loads(replay_isa(), "lz4r")
    if ok and loads >= 48 and waits >= 24:
        return []
    print("lz4_flex_amd.build: %s: the hand-scheduled loads of the replay decoder are not safe with this compiler (%s); "
          "building it with %s" % (REPLAY_SRC, msg or "markers missing: %d loads, %d waits" % (loads, waits), REPLAY_PLAIN_LOADS), file=sys.stderr)
    return [REPLAY_PLAIN_LOADS]


def fused_isa(extra_flags=()):
    """the listing of the fused decoder (a -DLZ4FLEX_TOOLS kernel since round 6: the product library does not hold it)"""
    return file_isa(FUSED_SRC, ["-DLZ4FLEX_TOOLS"] + list(extra_flags))


def fused_extra_flags():
    """[] when this toolchain leaves the fused decoder's in-flight registers alone, else [FUSED_PLAIN_LOADS]"""
    ok, msg, loads, waits = check_async_loads(fused_isa(), "lz4f")
    if ok and loads >= 8 and waits >= 8:
        return []
    print("lz4_flex_amd.build: %s: the hand-scheduled loads of the fused decoder are not safe with this compiler (%s); "
          "building it with %s" % (FUSED_SRC, msg or "markers missing: %d loads, %d waits" % (loads, waits), FUSED_PLAIN_LOADS), file=sys.stderr)
    return [FUSED_PLAIN_LOADS]


def wave_extra_flags():
    """[] when this toolchain leaves the in-flight registers alone (today's does), else [PLAIN_LOADS] -- decided at build time on
    the ISA that is about to be shipped, so that a compiler update degrades the encoder's speed and not its output"""
    ok, msg, loads, waits = check_async_loads(wave_isa())
    if ok and loads >= 16 and waits >= 8:
        return []
    print("lz4_flex_amd.build: %s: the hand-scheduled loads of the encoder's indexer are not safe with this compiler (%s); "
          "building it with %s" % (WAVE_SRC, msg or "markers missing: %d loads, %d waits" % (loads, waits), PLAIN_LOADS), file=sys.stderr)
    return [PLAIN_LOADS]


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    per_file = {WAVE_SRC: wave_extra_flags(), REPLAY_SRC: replay_extra_flags()}      # (the fused decoder is compiled in tools builds only: build_variant(..., ["-DLZ4FLEX_TOOLS"] + fused_extra_flags()))
    hipcc = _hipcc()
    os.makedirs(BDIR, exist_ok=True)
    sh = source_hash()
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(BDIR, src + ".o")
        cmd = [hipcc] + FLAGS + per_file.get(src, []) + ['-DLZ4FLEX_BUILD_ID="%s"' % sh, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        if verbose and out.strip():
            print(out.decode(errors="replace"), file=sys.stderr)
    tmp = LIB + ".tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode(errors="replace"))
    os.replace(tmp, LIB)
    with open(STAMP, "w") as f:
        f.write(sh + "\n")
    return LIB


def build_variant(name, extra_flags):
    """tools only: a second copy of the library compiled with extra -D flags (kernel experiments), next to the
    objects in build/; selected by LZ4FLEX_LIB=<path> (see _lib.load)"""
    hipcc = _hipcc()
    vdir = os.path.join(BDIR, "variant_" + name)
    os.makedirs(vdir, exist_ok=True)
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(vdir, src + ".o")
        cmd = [hipcc] + FLAGS + list(extra_flags) + ['-DLZ4FLEX_BUILD_ID="variant-%s"' % name, "-x", "hip", "-c",
                                                     os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
    lib = os.path.join(vdir, "liblz4flex_amd.so")
    # -Bsymbolic: the variant calls its OWN launchers even when another build of the library is loaded in the same process
    # (tools/dec_variants.py times several variants in one run)
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", lib] + objs, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode(errors="replace"))
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:      # python -m lz4_flex_amd.build --variant w11 -DLZ4W_WORKERS=11
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
