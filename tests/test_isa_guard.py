"""Guard against the register-indexing miscompile of hipcc 7.2 (scripts/microbench/gpr_idx_guard.hip: guarded stores into
a register-resident array through a RUN-TIME index are lowered to s_set_gpr_idx writes executed ahead of the guard, out of
range included - a memory fault in r03's first depth-4 speculative tree) and against scratch spills: the gfx950 code
objects of the BUILT library are extracted (every clang offload bundle of its .hip_fatbin section), disassembled with
llvm-objdump, and no kernel may contain s_set_gpr_idx* / v_movrel* or use scratch memory.  CPU only: no GPU is touched."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pysurfinv_amd", "lib", "libsurfdisp_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TOOLS = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump", "llvm-readelf")]

pytestmark = pytest.mark.skipif(not (os.path.exists(LIB) and all(os.path.exists(t) for t in TOOLS)),
                                reason="built library or the LLVM binary tools are absent")


@pytest.fixture(scope="module")
def code_objects(tmp_path_factory):
    """[(path of a gfx950 code object, its disassembly, its note section text)] - one per translation unit with kernels."""
    d = tmp_path_factory.mktemp("isa")
    fat = str(d / "fat.bin")
    subprocess.check_call([TOOLS[0], "-O", "binary", "--only-section=.hip_fatbin", LIB, fat])
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    assert starts, "no offload bundle in .hip_fatbin"
    out = []
    for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(blob)])):
        part, co = str(d / f"bundle{n}.bin"), str(d / f"k{n}.hsaco")
        open(part, "wb").write(blob[a:b])
        subprocess.check_call([TOOLS[1], "--unbundle", "--type=o", f"--input={part}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        if os.path.getsize(co) == 0:
            continue
        asm = subprocess.run([TOOLS[2], "-d", co], capture_output=True, text=True, check=True).stdout
        notes = subprocess.run([TOOLS[3], "--notes", co], capture_output=True, text=True, check=True).stdout
        out.append((co, asm, notes))
    shutil.rmtree(d, ignore_errors=True)
    return out


def _kernels(asm):
    """{symbol: text} of the functions of one disassembly"""
    parts = re.split(r"^[0-9a-f]+ <([^>]+)>:\n", asm, flags=re.M)
    return dict(zip(parts[1::2], parts[2::2]))


def test_every_translation_unit_with_kernels_is_seen(code_objects):
    names = set()
    for _, asm, _ in code_objects:
        names |= set(_kernels(asm))
    dem = subprocess.run(["c++filt"], input="\n".join(sorted(names)), capture_output=True, text=True).stdout
    for k in ("surfdisp_phase_kernel", "surfdisp_group_kernel", "surfdisp_ellip_kernel", "surfdisp_layers_kernel",
              "surfdisp_thermal_kernel", "surfdisp_mcmc_propose_kernel", "surfdisp_mcmc_accept_kernel"):
        assert k in dem, f"{k} not found in the library's gfx950 code objects"


def test_no_runtime_register_indexing(code_objects):
    bad = []
    for _, asm, _ in code_objects:
        for name, text in _kernels(asm).items():
            hits = re.findall(r"\b(s_set_gpr_idx\w*|v_movrel\w*)", text)
            if hits:
                bad.append((name, sorted(set(hits)), len(hits)))
    assert not bad, f"run-time indexed register moves (see scripts/microbench/gpr_idx_guard.hip): {bad}"


def test_no_kernel_uses_scratch(code_objects):
    nk = 0
    for _, asm, notes in code_objects:
        sizes = re.findall(r"\.private_segment_fixed_size:\s*(\d+)", notes)
        names = re.findall(r"\.name:\s*(\S+)", notes)
        nk += len(sizes)
        spilled = [(n, int(s)) for n, s in zip(names, sizes) if int(s) != 0]
        assert not spilled, f"kernels with scratch: {spilled}"
        assert not re.search(r"\bscratch_(load|store)\w*", asm), "scratch_load / scratch_store in the ISA"
    assert nk >= 70
