"""The device code of libcpc2_hip.so carries no packed-f32 VALU instruction (cpc2_amd/build.py, DEVICE_FLAGS): with them,
kernels returned slightly different sums from launch to launch whenever another process shared the MI355X (round 3,
DESIGN.md section 5).  CPU only: the gfx950 code objects are taken out of the library and disassembled."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")), reason="needs the ROCm LLVM tools")
def test_no_packed_f32_valu_in_device_code(tmp_path):
    lib = os.path.join(ROOT, "cpc2_amd", "libcpc2_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()
    fat = tmp_path / "fat.bin"
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", lib], check=True)
    blob = fat.read_bytes()
    starts = []
    pos = blob.find(MAGIC)
    while pos >= 0:
        starts.append(pos)
        pos = blob.find(MAGIC, pos + 1)
    assert len(starts) >= 8, "one bundle per .hip source expected"
    packed, fmas, kernels = 0, 0, 0
    for i, a in enumerate(starts):
        piece = tmp_path / f"bundle{i}.bin"
        piece.write_bytes(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = tmp_path / f"dev{i}.co"
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle", f"--input={piece}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
        asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", str(co)], check=True, capture_output=True, text=True).stdout
        packed += sum(asm.count(op) for op in ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32"))
        fmas += asm.count("v_fma_f32") + asm.count("v_fmac_f32")
        kernels += asm.count("_kernel")
    assert fmas > 1000 and kernels > 50, (fmas, kernels)        # (the disassembly did see the kernels)
    assert packed == 0, f"{packed} packed-f32 VALU instructions in the device code: build.py's DEVICE_FLAGS were not applied"


def test_no_library_stream_is_created_with_a_priority():
    """A lowest-priority side stream cost nothing in a single-process run and ~45 % of every kernel once the process had
    initialised RCCL (profiles/r03_dist_priority_bisect.txt): the library's streams are created with the default priority."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cpc2_amd", "csrc")
    hits = []
    for name in sorted(os.listdir(root)):
        if name.endswith((".hip", ".cpp", ".h")):
            for no, line in enumerate(open(os.path.join(root, name)), 1):
                code = line.split("//")[0]
                if "hipStreamCreateWithPriority" in code:
                    hits.append(f"{name}:{no}")
    assert not hits, hits
