"""Build libsttm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m sttm_amd.build [--force] [--verbose] [--dev]

The library lands in sttm_amd/lib/libsttm_hip.so; it is git-ignored but travels with the tree.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsttm_hip.so")
SOURCES = ["quadtree_spatial.hip", "spatial_f32.hip", "spatial_bf16.hip", "spatial_f16.hip", "spatial_f32_head.hip",
           "spatial_bf16_head.hip", "spatial_f16_head.hip", "spatial_pooled_f32.hip", "spatial_pooled_bf16.hip", "spatial_pooled_f16.hip",
           "spatial_col_f32.hip", "spatial_col_bf16.hip", "spatial_col_f16.hip", "temporal_merge.hip", "tome.hip", "pool2d.hip", "dycoke.hip", "octree.hip", "api.hip"]
HEADERS = ["sttm_common.h", "sttm_kernels.h", "sttm_pairs.inc", "quadtree_spatial.inc", "spatial_pooled.inc", "spatial_col.inc", os.path.join("..", "..", "include", "sttm_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
if os.environ.get("STTM_NT_STREAM") == "1":
    FLAGS.append("-DSTTM_NT_STREAM")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    mt = os.path.getmtime(target)
    return any(os.path.getmtime(d) > mt for d in deps)


def source_tag(extra_flags=()):
    """Short hash of every source the library is built from AND of the compile flags: baked into the library (sttm_build_tag)
    so that measurements committed under profiles/ can say which build they were taken on (two builds of the same sources
    with different flags -- STTM_NT_STREAM, extra_flags -- report different tags and do not share object files)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES) + sorted(HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    h.update(("\0".join(list(FLAGS) + sorted(extra_flags))).encode())
    return h.hexdigest()[:12]


def flags_tag(extra_flags=()):
    """Hash of the compile flags alone: when IT changes every object is rebuilt; a changed source only rebuilds the objects that
    are older than it (and api.o, which carries the source + flags tag)."""
    import hashlib
    return hashlib.sha256(("\0".join(list(FLAGS) + sorted(extra_flags))).encode()).hexdigest()[:12]


def build(force=False, verbose=False, extra_flags=(), dev=False):
    """dev=True builds libsttm_hip_dev.so with -DSTTM_DEV (the measurement hooks of tools/*_ticks.py, tools/k1_ablate.py);
    the product library never contains them."""
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "dev") if dev else LIBDIR
    os.makedirs(objdir, exist_ok=True)
    lib = os.path.join(LIBDIR, "libsttm_hip_dev.so") if dev else LIB
    if dev:
        extra_flags = tuple(extra_flags) + ("-DSTTM_DEV",)
    common = [os.path.join(CSRC, h) for h in HEADERS if not h.endswith(".inc")]
    # the two big template files are included by some translation units only
    inc_users = {"quadtree_spatial.inc": lambda src: src.startswith("spatial_"), "spatial_pooled.inc": lambda src: src.startswith("spatial_pooled_"),
                 "spatial_col.inc": lambda src: src.startswith("spatial_col_"),
                 "sttm_pairs.inc": lambda src: src == "temporal_merge.hip"}

    def deps(src):
        return common + [os.path.join(CSRC, inc) for inc, uses in inc_users.items() if uses(src)]
    tag = source_tag(extra_flags) + ("-dev" if dev else "")
    tag_file = os.path.join(objdir, ".build_tag")
    old = open(tag_file).read().split() if os.path.exists(tag_file) else []
    old_tag = old[0] if old else ""
    ftag = flags_tag(extra_flags)
    flags_changed = len(old) < 2 or old[1] != ftag        # (a tag file of an older layout: rebuild once)
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        extra = [f'-DSTTM_BUILD_TAG="{tag}"'] if src == "api.hip" else []
        if src == "tome.hip":
            # the four-wave match kernel names its accumulators (all 256 AGPRs) literally in inline assembly: the compiler must not park
            # spilled VGPRs there
            extra += ["-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"]
        # changed FLAGS rebuild every object (they apply to all of them); a changed source rebuilds the objects older than it, plus
        # api.o, which has the source + flags tag baked in (sttm_build_tag)
        if force or flags_changed or _stale(o, [s] + deps(src)) or (src == "api.hip" and old_tag != tag):
            jobs.append([_hipcc(), *FLAGS, *extra_flags, *extra, "-c", s, "-o", o])
    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr:
            print(r.stderr)
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(lib, objs):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    with open(tag_file, "w") as fh:
        fh.write(tag + "\n" + ftag + "\n")
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, dev="--dev" in sys.argv))
