"""
Register / scratch / LDS budget of the compiled gfx950 kernels, read from the code-object
metadata of the in-tree library (no GPU needed): the MFMA convolutions are built for a fixed
number of workgroups per CU, and a change that pushes one of them over its register or LDS
budget would silently halve its occupancy or send accumulators to scratch memory.
"""

import os
import re
import shutil
import subprocess

import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "aind_exaspim_neuron_segmentation_amd", "csrc", "libexaspim_affinity.so")
LLVM = "/opt/rocm/lib/llvm/bin"
LDS_PER_CU = 160 * 1024


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not os.path.exists(LIB) or not all(os.path.exists(t) for t in tools):
        pytest.skip("library or LLVM binary utilities not available")
    tmp = tmp_path_factory.mktemp("codeobj")
    fat = str(tmp / "fat.bin")
    subprocess.run([tools[0], f"--dump-section=.hip_fatbin={fat}", LIB, str(tmp / "ignored.so")], check=True)
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
    found = {}
    for i, p in enumerate(starts):
        piece = str(tmp / f"bundle{i}.bin")
        with open(piece, "wb") as f:
            f.write(data[p:starts[i + 1] if i + 1 < len(starts) else len(data)])
        co = str(tmp / f"bundle{i}.co")
        subprocess.run([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={piece}", f"--output={co}"], check=True)
        notes = subprocess.run([tools[2], "--notes", co], check=True, capture_output=True, text=True).stdout
        body = notes.split("---", 1)[1].rsplit("...", 1)[0]
        for k in yaml.safe_load(body)["amdhsa.kernels"]:
            found[k[".name"]] = k
    filt = shutil.which("c++filt") or shutil.which(os.path.join(LLVM, "llvm-cxxfilt"))
    if filt:     # readable names; the patterns below match the demangled template arguments
        names = list(found)
        plain = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        found = {p: found[n] for n, p in zip(names, plain)}
    else:
        pytest.skip("no C++ demangler")
    assert found
    return found


def _sel(kernels, pattern):
    out = {n: k for n, k in kernels.items() if re.search(pattern, n)}
    assert out, pattern
    return out


def test_every_kernel_is_built_for_gfx950_wave64(kernels):
    for name, k in kernels.items():
        assert k[".wavefront_size"] == 64, name
        assert k[".vgpr_count"] <= 512, name      # (.vgpr_count is the unified allocation, AGPRs included)


def test_memory_bound_kernels_use_no_scratch(kernels):
    for name, k in _sel(kernels, r"gather|stitch|finalize|histogram|maxpool2|upsample2|pad_|head_kernel|export_f16|"
                                 r"splitk_reduce|conv_first|convt2|synth").items():
        assert k[".private_segment_fixed_size"] == 0 and k[".vgpr_spill_count"] == 0, name


def test_t14_kernels_fit_their_occupancy_without_spills(kernels):
    """Two waves per SIMD (256 registers) at least, two workgroups' LDS per CU, nothing spilled."""
    for name, k in _sel(kernels, r"conv3x3x3_t14<").items():
        assert k[".vgpr_spill_count"] == 0 and k[".private_segment_fixed_size"] == 0, name
        assert k[".vgpr_count"] <= 256, name
        if "F32Tag" not in name:      # (float32 records: the 64-cout shapes never take the pooled epilogue)
            assert 2 * k[".group_segment_fixed_size"] <= LDS_PER_CU, name


def test_z_column_kernels_stay_inside_their_budget(kernels):
    """conv3x3x3_zpipe runs two four-wave workgroups per CU: at most 256 registers and half the LDS.
    Since round 3 nothing may go to scratch memory in the instantiations without the fused head: the
    lane-derived constants that hipcc used to hoist out of the tile loop and spill (each reload a
    scratch_load whose s_waitcnt vmcnt(0) also waited for the previous tile's stores) are recomputed
    per tile (fresh_lane, conv3d.hip). The fused-head instantiations of the six-plane tile still spill
    a few staging offsets around their epilogue (partial sums of six planes x up to four outputs on top
    of the next tile's staged pieces); that is bounded so that it cannot grow unnoticed."""
    for name, k in _sel(kernels, r"conv3x3x3_zpipe<").items():
        assert k[".vgpr_count"] <= 256, name
        assert 2 * k[".group_segment_fixed_size"] <= LDS_PER_CU, name
        m = re.search(r"Tag, (\d+), 8, 16, 2, 4, (\d+), (true|false)>", name)
        assert m, name
        planes, head = int(m.group(1)), int(m.group(2))
        if head == 0 or planes == 4:
            limit = 0
        else:
            limit = {1: 16, 2: 16, 3: 64, 4: 160}[head]      # bytes of scratch per lane
        assert k[".private_segment_fixed_size"] <= limit, (name, k[".private_segment_fixed_size"])


def test_no_mfma_kernel_waits_on_a_spill_inside_its_tap_loop(kernels):
    """(metadata only) every conv3x3x3_t14 instantiation and the z-column kernels above: the count of
    spilled registers is what the two tests above bound; this one pins the totals so that a compiler
    or source change that moves them shows up in review."""
    total = sum(k[".vgpr_spill_count"] for n, k in kernels.items() if "conv3x3x3_t14<" in n)
    assert total == 0
