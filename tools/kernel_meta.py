"""Register / spill / LDS figures of the kernels in a built library (no GPU needed):
    python tools/kernel_meta.py [path/to/lib.so] [regex]
Reads the code-object metadata the way tests/test_kernel_budget.py does."""
import os
import re
import subprocess
import sys
import tempfile

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib):
    found = {}
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", lib,
                        os.path.join(tmp, "ignored.so")], check=True)
        data = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
        for i, p in enumerate(starts):
            piece = os.path.join(tmp, f"b{i}.bin")
            with open(piece, "wb") as f:
                f.write(data[p:starts[i + 1] if i + 1 < len(starts) else len(data)])
            co = os.path.join(tmp, f"b{i}.co")
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={piece}",
                            f"--output={co}"], check=True)
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True,
                                   capture_output=True, text=True).stdout
            body = notes.split("---", 1)[1].rsplit("...", 1)[0]
            for k in yaml.safe_load(body)["amdhsa.kernels"]:
                found[k[".name"]] = k
    names = list(found)
    plain = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True,
                           check=True).stdout.split("\n")
    return {p: found[n] for n, p in zip(names, plain)}


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(
        ROOT, "aind_exaspim_neuron_segmentation_amd", "csrc", "libexaspim_affinity.so")
    pat = sys.argv[2] if len(sys.argv) > 2 else "."
    for name, k in sorted(kernels(lib).items()):
        if re.search(pat, name):
            short = name.replace("exaspim::", "").replace("(ConvArgs, int, int, int)", "").replace("void ", "")
            print(f"{short[:90]:90s} vgpr {k['.vgpr_count']:4d} spills {k['.vgpr_spill_count']:3d} "
                  f"sgpr-spills {k['.sgpr_spill_count']:3d} scratch {k['.private_segment_fixed_size']:4d} "
                  f"lds {k['.group_segment_fixed_size']}")
