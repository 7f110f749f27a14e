"""Register-allocation guard (CPU): the kernel descriptors of the BUILT library, read out of its embedded gfx950 code objects.

The instantiations that the BASELINE.json configurations and the production shapes launch must not use scratch memory
(`private_segment_fixed_size == 0`): a spilled value sits on the critical path of these latency-bound kernels and enabling the
private segment costs every wave of the launch ("any scratch use is poison").  The list is by mangled-name
fragment; a listed kernel that is missing from the library fails too, so a renamed template cannot silently drop out.
"""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"

# (what it is, regex on the mangled kernel name)
NO_SCRATCH = [
    ("K1 headline / C2 / C3 / C5: fp32 C=1024, 3-level tree", r"k_spatialIfLi4ELi3ELi0ELi256ELb0E"),
    ("K1 production: bf16 C=3584, 32-byte packs", r"k_spatialINS_6bf16_tELi16ELi3ELi0ELi256ELb0E"),
    ("K1 72B width: bf16 C=8192, 16-byte packs, 1024 threads", r"k_spatialINS_6bf16_tELi8ELi3ELi0ELi1024ELb0E"),
    ("K1 bf16 C<=2048", r"k_spatialINS_6bf16_tELi8ELi3ELi0ELi256ELb0E"),
    ("K1 fp16 C<=2048", r"k_spatialINS_5f16_tELi8ELi3ELi0ELi256ELb0E"),
    ("K1 C4 grids (20x36, 18x26): fp32 4-level tree", r"k_spatialIfLi4ELi3ELi1ELi256ELb0E"),
    ("K1 split form, pass A (trees of 4+ levels: one workgroup per 3-level block), fp32", r"k_spatial_blocksIfLi4ELi256E"),
    ("K1 split form, pass A, bf16 32-byte packs", r"k_spatial_blocksINS_6bf16_tELi16ELi256E"),
    ("K1 split form, pass B over one upper level (C4 grids)", r"k_spatial_upperIfLi4ELi1ELi256E"),
    ("K1 split form, pass B over three upper levels (6-level trees)", r"k_spatial_upperIfLi4ELi3ELi256E"),
    ("K2 fp32 (8-wide row packs)", r"k_pairs256IfLi8ELi5ELi32E"),
    ("K2 fp32, the column walk's 16-byte packs (round 6: the headline's pair kernel)", r"k_pairs256IfLi4ELi5ELi32E"),
    ("K2 bf16", r"k_pairs256INS_6bf16_tELi8ELi5ELi64E"),
    ("K3 fused label stage", r"k_col_labelsILi2E"),
    ("K5 fp32", r"k_group_meanIfLi8ELi6E"),
    ("K5 bf16", r"k_group_meanINS_6bf16_tELi8ELi6E"),
    ("K5 on root cells of 17-64 leaves (C4 grids): two member rows in flight", r"k_group_meanIfLi8ELi5ELb1ELb0E"),
    ("K5 on root cells of more than 64 leaves: long groups by the whole workgroup", r"k_group_meanIfLi8ELi4ELb1ELb1E"),
    ("K1 on the unpooled token map (get_2dPool fused into the leaf load), fp32 C<=1024", r"k_spatial_pooledIfLi4ELi256E"),
    ("ToMe 256-tile match, fp32 two-plane", r"k_tome_match_gldsILi2ELi4E"),
    ("ToMe 256-tile match, one plane", r"k_tome_match_gldsILi1ELi1E"),
    ("ToMe 256-tile match, four-wave form (accumulators = the whole AGPR file, named in inline assembly), fp32 two-plane", r"k_tome_match_gldsILi2ELi3ENS_5f16_tELi0ELi4E"),
    ("ToMe 256-tile match, four-wave form, bf16", r"k_tome_match_gldsILi1ELi1ENS_6bf16_tELi0ELi4E"),
]


def _kernels_of(lib):
    """{mangled name: (private_segment_fixed_size, vgpr_count)} of every kernel in the library's gfx950 code objects."""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(tmp, "copy.so")], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for n, s in enumerate(starts):
            e = starts[n + 1] if n + 1 < len(starts) else len(blob)
            part = os.path.join(tmp, f"b{n}.bin")
            with open(part, "wb") as fh:
                fh.write(blob[s:e])
            co = os.path.join(tmp, f"b{n}.co")
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                f"--input={part}", f"--output={co}", "--unbundle"], capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk)
                priv = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
                vg = re.search(r"\.vgpr_count:\s+(\d+)", blk)
                if name and priv:
                    out[name.group(1)] = (int(priv.group(1)), int(vg.group(1)) if vg else -1)
    return out


@pytest.fixture(scope="module")
def kernels():
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sttm_amd", "lib", "libsttm_hip.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    for tool in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"):
        if not os.path.exists(f"{LLVM}/{tool}") and not shutil.which(tool):
            pytest.skip(f"{tool} not available")
    k = _kernels_of(lib)
    assert len(k) > 50, "could not read the kernel descriptors of the library"
    return k


@pytest.mark.parametrize("what,pattern", NO_SCRATCH, ids=[w for w, _ in NO_SCRATCH])
def test_headline_and_production_kernels_use_no_scratch(kernels, what, pattern):
    hits = {n: v for n, v in kernels.items() if re.search(pattern, n)}
    assert hits, f"no kernel matches {pattern!r} ({what}): the guard list is stale"
    spilled = {n: v for n, v in hits.items() if v[0] != 0}
    assert not spilled, f"{what}: scratch in use (bytes per lane, VGPRs): {spilled}"


def test_four_wave_tome_kernels_keep_the_compiler_out_of_the_agprs():
    """csrc/tome.hip, TomeAcc: the four-wave match kernels name their 256 accumulator registers literally in inline assembly; a compiler
    spill into an AGPR (v_accvgpr_write) or a compiler copy out of one would corrupt / duplicate them silently.  The disassembly of the
    built kernels must hold exactly the 512 v_accvgpr_read of TomeAcc::read (16 per block, one statement pair per block, once in each of
    the two passes of the running max) and no v_accvgpr_write / v_accvgpr_mov at all."""
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sttm_amd", "lib", "libsttm_hip.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    for tool in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump"):
        if not os.path.exists(f"{LLVM}/{tool}"):
            pytest.skip(f"{tool} not available")
    found = 0
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(tmp, "copy.so")], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for n, s0 in enumerate(starts):
            e = starts[n + 1] if n + 1 < len(starts) else len(blob)
            if b"k_tome_match_glds" not in blob[s0:e]:
                continue
            part, co = os.path.join(tmp, f"b{n}.bin"), os.path.join(tmp, f"b{n}.co")
            with open(part, "wb") as fh:
                fh.write(blob[s0:e])
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                f"--input={part}", f"--output={co}", "--unbundle"], capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", co], capture_output=True, text=True, check=True).stdout
            for blk in re.split(r"\n(?=[0-9a-f]+ <)", dis):
                head = blk.split("\n", 1)[0]
                if "k_tome_match_glds" not in head or not re.search(r"ELi0ELi4EE", head):
                    continue
                found += 1
                assert len(re.findall(r"v_accvgpr_read_b32", blk)) == 512, head
                assert not re.search(r"v_accvgpr_write|v_accvgpr_mov", blk), head
    assert found >= 3, "the four-wave ToMe match kernels (fp32 two-plane, bf16, fp16) are missing from the library"
