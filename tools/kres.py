#!/usr/bin/env python3
"""Per-kernel resources (VGPRs, AGPRs, SGPRs, scratch, LDS) read from the built objects: tools/kres.py <file.o stem> [name-substring ...]
e.g. tools/kres.py conv_wino4 wino4_kernel"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    stem, pats = sys.argv[1], sys.argv[2:]
    obj = os.path.join(REPO, "eemflow_amd", "csrc", "build", stem + ".o")
    with tempfile.TemporaryDirectory() as d:
        fat, co = f"{d}/fat.bin", f"{d}/k.co"
        subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, f"{d}/host.o"], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name:
            continue
        nm = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
        nm = re.sub(r"^void \(anonymous namespace\)::", "", nm).replace("(EncConvArgs)", "")
        if pats and not any(p in nm for p in pats):
            continue
        g = lambda k: (re.search(rf"\.{k}:\s+(\d+)", blk) or [None, "?"])[1]   # noqa: E731
        agpr = re.match(r"\s*(\d+)", blk).group(1)
        print(f"{nm[:100]:100s} vgpr {g('vgpr_count'):>3s} agpr {agpr:>3s} sgpr {g('sgpr_count'):>3s} scratch {g('private_segment_fixed_size'):>4s} lds {g('group_segment_fixed_size'):>6s}")


if __name__ == "__main__":
    main()
