"""Build libmpe_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build() and by hand:

    python -m multiagent_particle_envs_amd._build [--force]

The shared object lands in multiagent_particle_envs_amd/lib/ (git-ignored, but it travels to the
GPU box with the repo snapshot).  hipcc cross-compiles for gfx950 without a GPU.
Flags that matter for parity: -ffp-contract=off (no silent fma: NumPy rounding sequence for
dx*dx+dy*dy, see csrc/mpe_device.h) and the hipcc defaults for correctly rounded fp32 sqrt/div.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmpe_hip.so")
SOURCES = ["mpe_abi.hip", "mpe_narrow.hip", "mpe_split.hip", "mpe_wide.hip", "mpe_rng.hip", "mpe_rows.hip"]
HEADERS = ["mpe_device.h", "mpe_internal.h", os.path.join("..", "..", "include", "mpe_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function",
         # kernarg preload (gfx94x/gfx950): the CP writes the first dwords of the kernarg segment into user SGPRs at wave
         # launch; kernels whose leading arguments are scalars / pointers (k_split, the bookkeeping kernels) start their
         # global loads without a scalar round trip.  Kernels that begin with a by-value struct are unaffected.
         "-mllvm", "-amdgpu-kernarg-preload-count=14",
         # keep the device assembly next to the objects (build/*-gfx950.s): tests/test_host_cpu.py reads the kernels' resource
         # metadata from it -- a kernel that starts using scratch memory (private_segment_fixed_size > 0) fails the CPU suite
         "-save-temps=obj"]


STRESS_SOURCES = ("split", "wide", "rows")
STRESS_VARIANTS = {"stress": ["-DMPE_STRESS_DELAY_WAVE=1"],
                   "stress_racy": ["-DMPE_STRESS_DELAY_WAVE=1", "-DMPE_STRESS_STORE_BEFORE_BARRIER"],
                   # measurement build (tools/device_span.py): every wave of k_split stamps the device wall clock at its
                   # start and end -> per-launch spans and periods from the device's own clock
                   "span": ["-DMPE_DEVICE_SPAN"]}
VARIANT_SOURCES = {"span": ("split",), "teamgrid": ("split", "narrow"), "rowsclock": ("rows",)}     # which kernel files a variant recompiles (default: STRESS_SOURCES)
# A/B builds made on request only (`python -m multiagent_particle_envs_amd._build --ab teamgrid`), never by build():
#   teamgrid   round 3's 19 extra k_split entries (simple_adversary / simple_world_comm team sizes) put back, to hold the
#              row-program path that replaced them against (profiles/r4_team_sizes_ab.txt)
#   rowsclock  mpe_rows.hip with phase stamps (tools/rows_clock.py: where a row-program step's cycles go)
AB_VARIANTS = {"teamgrid": ["-DMPE_SPLIT_TEAM_GRID"], "rowsclock": ["-DMPE_ROWS_CLOCK"]}


ROWS_CACHE = os.path.join(LIBDIR, "rows_cache")      # compiled row programs: <sha256 of header + sources + flags>.hsaco


_rows_sources_digest = None


def _rows_image_path(header_text):
    """lib/rows_cache/<sha256(header, mpe_rows.hip + its headers, flags)>.hsaco, and the flags"""
    global _rows_sources_digest
    import hashlib
    _rows_sources_digest = None if os.environ.get("MPE_ROWS_IMAGE_FLAGS") else _rows_sources_digest
    # (-pragma-unroll-threshold: the op loops of a compiled program MUST unroll -- every op a constant is the whole point -- and
    #  LLVM refuses `#pragma unroll` past 16 K instructions of pre-folding body, which a 15-op reward program already exceeds)
    flags = [f for f in FLAGS if f not in ("-fPIC", "-save-temps=obj", "-Wall")] + ["-mllvm", "-pragma-unroll-threshold=16777216"]
    flags += os.environ.get("MPE_ROWS_IMAGE_FLAGS", "").split()      # (A/B builds of the images: part of the cache key)
    if _rows_sources_digest is None:
        h = hashlib.sha256()
        for f in [os.path.join(CSRC, "mpe_rows.hip")] + [os.path.join(CSRC, x) for x in HEADERS]:
            with open(f, "rb") as fh:
                h.update(fh.read())
        h.update(" ".join(flags).encode())
        _rows_sources_digest = h.digest()
    h = hashlib.sha256(_rows_sources_digest)
    h.update(header_text.encode())
    return os.path.join(ROWS_CACHE, h.hexdigest()[:32] + ".hsaco"), flags


def cached_rows_image(header_text):
    """The code object of a compiled row program if lib/rows_cache/ holds it (never runs hipcc), else None."""
    out, _ = _rows_image_path(header_text)
    if not os.path.exists(out):
        return None
    with open(out, "rb") as fh:
        return fh.read()


def compile_rows_image(header_text, verbose=False):
    """A row program compiled in: csrc/mpe_rows.hip built with the generated header (mpe_rows_static_source) as a gfx950 code
    object (`hipcc --genco`, the library's own flags: same arithmetic, same kernarg preload) -> its bytes, for
    mpe_rows_load_image.  Cached by content under lib/rows_cache/ (travels with the tree; delete at will)."""
    out, flags = _rows_image_path(header_text)
    if not os.path.exists(out):
        os.makedirs(ROWS_CACHE, exist_ok=True)
        hdr = out[:-6] + ".h"
        with open(hdr, "w") as fh:
            fh.write(header_text)
        tmp = out + ".tmp%d" % os.getpid()
        cmd = [_hipcc(), "--genco"] + flags + ["-include", hdr, "-I", CSRC, "-I", os.path.join(HERE, "..", "include"),
                                               os.path.join(CSRC, "mpe_rows.hip"), "-o", tmp]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on a compiled row program:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        os.replace(tmp, out)
    with open(out, "rb") as fh:
        return fh.read()


def variant_lib(tag):
    return os.path.join(LIBDIR, "libmpe_hip_%s.so" % tag)


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _drop_temps(d, keep_asm=True):
    """-save-temps leaves ~25 MB of intermediates per source; only the product build's device assembly (*-gfx950.s) is wanted."""
    import glob
    for pat in ("*.bc", "*.hipi", "*.out", "*.resolution.txt", "*.hipfb", "*-host-*.s", "*-gfx950.o") + (() if keep_asm else ("*-gfx950.s",)):
        for f in glob.glob(os.path.join(d, pat)):
            try:
                os.remove(f)
            except OSError:
                pass


def build(force=False, verbose=True, variants=True, ab=()):
    """Compile and link libmpe_hip.so; with `variants` also the two test-only stress builds (tests/test_gpu_race.py).
    The product library is linked FIRST and is what decides staleness on its own: a box that holds only libmpe_hip.so
    newer than the sources (the GPU box) never invokes hipcc for it, and a compile failure in a variant cannot keep the
    product from being linked."""
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    want = [LIB] + ([variant_lib(t) for t in list(STRESS_VARIANTS) + list(ab)] if variants else [])
    if not force and not any(_stale(l, srcs + hdrs) for l in want):
        return LIB  # the shipped .so files are newer than every source: nothing to do (GPU box)
    hipcc = _hipcc()

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-8000:]))
        if verbose and r.stderr.strip():
            print(r.stderr[-4000:])

    # ---- every compile job (product + variants) in ONE pool, the long ones (mpe_split: ~29 s) first; the product library is
    # linked as soon as its own objects exist, the variants after theirs
    long_first = {"mpe_split.hip": 0, "mpe_narrow.hip": 1}
    jobs, objs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append((long_first.get(src, 2), "product", [hipcc] + FLAGS + ["-c", s, "-o", o]))
    # test-only variants of the role-split kernels (tests/test_gpu_race.py): one wave of every workgroup is held back
    # (MPE_STRESS_DELAY_WAVE: ~30 us before its first load in k_split / k_duo, a few us at every step of k_duo_roll);
    # "_racy" additionally restores the store-before-barrier ordering the round-1 k_split had, as the negative control
    # that shows the test can fail
    todo = []
    wanted = dict(STRESS_VARIANTS) if variants else {}
    wanted.update({t: AB_VARIANTS[t] for t in ab})
    vflags = [f for f in FLAGS if f != "-save-temps=obj"]      # (no assembly kept for the variants)
    for tag, defs in wanted.items():
        vo = {}
        for stem in VARIANT_SOURCES.get(tag, STRESS_SOURCES):      # the kernel files that carry the variant's hooks
            src = os.path.join(CSRC, "mpe_%s.hip" % stem)
            os.makedirs(os.path.join(OBJ, tag), exist_ok=True)
            o = os.path.join(OBJ, tag, "mpe_%s.o" % stem)
            vo["mpe_%s.o" % stem] = o
            if force or _stale(o, [src] + hdrs):
                jobs.append((long_first.get("mpe_%s.hip" % stem, 2), tag, [hipcc] + vflags + defs + ["-c", src, "-o", o]))
        todo.append((tag, vo))
    jobs.sort(key=lambda j: j[0])
    with ThreadPoolExecutor(max_workers=max(4, os.cpu_count() or 4)) as ex:
        futs = [(tag, ex.submit(run, cmd)) for _, tag, cmd in jobs]
        for tag, f in futs:
            if tag == "product":
                f.result()       # a product compile error surfaces here, before anything is linked
        if force or any(t == "product" for t, _ in futs) or _stale(LIB, objs):
            run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
        _drop_temps(OBJ)
        for tag, f in futs:      # (a compile failure in a variant cannot keep the product from being linked: it is, by now)
            f.result()
        links = []
        for tag, vo in todo:
            vlib = variant_lib(tag)
            vobjs = [vo.get(os.path.basename(x), x) for x in objs]
            if force or _stale(vlib, vobjs):
                links.append(ex.submit(run, [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", vlib] + vobjs))
        for f in links:
            f.result()
    return LIB


if __name__ == "__main__":
    ab = [sys.argv[i + 1] for i, a in enumerate(sys.argv[:-1]) if a == "--ab"]
    print(build(force="--force" in sys.argv, variants="--no-variants" not in sys.argv, ab=ab))
