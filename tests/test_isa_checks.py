"""Build-time checks on generated ISA (no GPU): kernels that issue operand requests through volatile asm and wait for them by
hand (csrc/zk_sep_strip.hip; the LDS operand reads of zk_patch_direct_kernel in csrc/zk_direct_patches.hip) must keep every in-flight destination register untouched until the covering s_waitcnt -- the
compiler does not know those registers are still to be written.  tools/check_async_requests.py walks the control-flow graph
of the assembly; a violation was a GPU memory fault in round 3 (a kernel-argument reload racing a late table row)."""
import os
import subprocess
import sys

from conftest import ROOT

CSRC = os.path.join(ROOT, "motif-learn_amd", "csrc")
CHECK = os.path.join(ROOT, "motif-learn_amd", "tools", "check_async_requests.py")


def test_hand_issued_requests_have_no_register_hazards(tmp_path):
    asm = tmp_path / "strip.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I../../include", "-I.", "-S",
                           "--cuda-device-only", "-o", str(asm), "zk_sep_strip.hip"], cwd=CSRC, stderr=subprocess.DEVNULL)
    out = subprocess.run([sys.executable, CHECK, str(asm)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-3000:]
    assert "26 kernel(s) with hand-issued requests checked, 0 hazard(s)" in out.stdout


def test_direct_batch_kernel_operand_reads_have_no_register_hazards(tmp_path):
    """zk_patch_direct_kernel reads its MFMA operands (pixels and table rows) from LDS a step ahead, by asm reads into a second set
    of registers: 2 input types x 3 widths of the wider chunks (each kernel also holds the body one block narrower)."""
    asm = tmp_path / "direct.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I../../include", "-I.", "-S",
                           "--cuda-device-only", "-o", str(asm), "zk_direct_patches.hip"], cwd=CSRC, stderr=subprocess.DEVNULL)
    out = subprocess.run([sys.executable, CHECK, str(asm)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-3000:]
    assert "6 kernel(s) with hand-issued requests checked, 0 hazard(s)" in out.stdout
