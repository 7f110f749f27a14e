"""Kernel resource table of one .hip translation unit (VGPR / SGPR / scratch / occupancy / LDS), from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.   usage: python tools/kres.py sttm_amd/csrc/temporal_merge.hip [filter]"""
import re, subprocess, sys, shutil
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:], capture_output=True, text=True)
filt = shutil.which("c++filt")
for b in re.split(r"remark: Function Name: ", r.stderr)[1:]:
    name = b.split()[0]
    dem = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip() if filt else name
    if flt and flt not in dem: continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    print(dem[:90].ljust(90), "SGPR", g("TotalSGPRs"), "VGPR", g(" VGPRs"), "AGPR", g("AGPRs"), "scratch", g(r"ScratchSize \[bytes/lane\]"),
          "occ", g(r"Occupancy \[waves/SIMD\]"), "lds", g(r"LDS Size \[bytes/block\]"))
