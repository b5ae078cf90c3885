#!/usr/bin/env python3
"""Static resources of every search-kernel instantiation, read from the gfx950 code objects inside the built
objects (no GPU needed): VGPRs, SGPRs, scratch, and the waves per CU the VGPR count allows (512 VGPRs per SIMD
lane, 8-register granules, 4 SIMDs, at most 8 waves per SIMD).  LDS is dynamic (sized per launch from L and the
visited table) -- see DESIGN.md.   usage: python profiles/kernel_resources.py [object ...] > profiles/<name>.txt"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def code_object(obj, tmp):
    fat = os.path.join(tmp, "fat.bin")
    co = os.path.join(tmp, "dev.co")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj, os.path.join(tmp, "x.o")])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--input={fat}", f"--output={co}", "--unbundle"])
    return co


def kernels(co):
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    out = []
    for blk in notes.split("- .agpr_count:")[1:]:
        g = lambda k: re.search(rf"\.{k}:\s+(\S+)", blk)
        name = g("name").group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        out.append((dem, int(g("vgpr_count").group(1)), int(g("sgpr_count").group(1)),
                    int(g("private_segment_fixed_size").group(1)), int(g("group_segment_fixed_size").group(1))))
    return out


def main():
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "diskann_amd", "build", "*.o")))
    print(f"{'kernel':100s} {'VGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'LDS(static)':>11s} {'waves/CU by VGPR':>17s}")
    for obj in objs:
        with tempfile.TemporaryDirectory() as tmp:
            try:
                ks = kernels(code_object(obj, tmp))
            except subprocess.CalledProcessError:
                continue
        for dem, vg, sg, scr, lds in sorted(ks):
            dem = re.sub(r"^void dann::\(anonymous namespace\)::", "", dem)
            dem = re.sub(r"\(.*$", "", dem)
            granule = (vg + 7) // 8 * 8
            waves = 4 * min(8, 512 // max(granule, 8))
            print(f"{dem[:100]:100s} {vg:5d} {sg:5d} {scr:8d} {lds:11d} {waves:17d}")


if __name__ == "__main__":
    main()
