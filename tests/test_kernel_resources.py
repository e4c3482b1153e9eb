"""Build-time guard: the CG SpMV kernel owes ~20 % of its speed to running 8 wavefronts per SIMD
(measured: a change that raised it from 64 to 78 VGPRs took the 10 M-dof SpMV from 0.396 to 0.486 ms), and
it sits exactly at the 64-VGPR limit of that occupancy.  hipcc cross-compiles without a GPU, so the register
count of the shipped variants is checked here, where a regression is cheap to see."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_spmv_kernel_keeps_full_occupancy(tmp_path):
    src = os.path.join(ROOT, "performance-test_amd", "csrc", "zzz_spmv.hip")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-I" + os.path.dirname(src),
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "performance-test_amd", "host"), "-c", src, "-o",
           str(tmp_path / "spmv.o"), "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    # remarks come in blocks: "Function Name: <mangled>" followed by the resource lines of that kernel
    blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        # spmv_tile_kernel<DOT, NT, PIPE=false, TILE=2048>: the variants the solver launches by default
        m = re.match(r"_ZN3zzz16spmv_tile_kernelILb([01])ELb([01])ELb0ELi2048EEE", name)
        if not m:
            continue
        vgprs = int(re.search(r"VGPRs: (\d+)", b).group(1))
        occ = int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", b).group(1))
        spill = int(re.search(r"VGPRs Spill: (\d+)", b).group(1))
        lds = int(re.search(r"LDS Size \[bytes/block\]: (\d+)", b).group(1))
        assert vgprs <= 64 and occ == 8 and spill == 0, (name, vgprs, occ, spill)
        assert lds * 8 <= 160 * 1024, (name, lds)  # eight workgroups per CU must fit the 160 KB of LDS
        seen += 1
    assert seen == 4


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_operator_stream_kernel_keeps_full_occupancy(tmp_path):
    """The product on the operator stream (zzz_sellp.hip) has no LDS and must stay at 8 wavefronts per SIMD (<= 64 VGPRs:
    its launch bound); without spills it would need 74 registers = 6 wavefronts, measured exactly as fast, so the few
    spilled dwords are tolerated -- but not growth beyond that."""
    src = os.path.join(ROOT, "performance-test_amd", "csrc", "zzz_sellp.hip")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-I" + os.path.dirname(src),
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "performance-test_amd", "host"), "-c", src, "-o",
           str(tmp_path / "sellp.o"), "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        if not re.match(r"_ZN3zzz17spmv_sellp_kernelILb[01]ELb[01]ELb[01]ELb[01]ELb[01]ELi[0123]EEE", name):
            continue
        vgprs = int(re.search(r"VGPRs: (\d+)", b).group(1))
        occ = int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", b).group(1))
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        lds = int(re.search(r"LDS Size \[bytes/block\]: (\d+)", b).group(1))
        sgprs = int(re.search(r"TotalSGPRs: (\d+)", b).group(1))
        # LDS: the cross-wavefront sums and the folded all-reduce's tail (zzz_tail.h), a few hundred bytes per workgroup;
        # SGPRs <= 80 keeps eight 256-thread workgroups per CU admissible (MI355X_MICROARCH.md, Residency)
        assert vgprs <= 64 and occ == 8 and scratch <= 64 and lds <= 512 and sgprs <= 80, (name, vgprs, occ, scratch, lds, sgprs)
        seen += 1
    # (dot products, load policy, row permutation, Chebyshev-term epilogue) + x windows (natural order), each with the
    # values as doubles, as codes into a dictionary in memory, as codes into a dictionary in LDS, as 8-bit codes into
    # per-slice dictionaries
    assert seen == 88  # (slice dictionaries and x windows exclude each other)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src_name,kernel_re,lds_budget,variants", [
    ("zzz_sellp_blk.hip", r"_ZN3zzz16spmv_blk3_kernelILb[01]ELb[01]ELb[01]ELi[12]ELb[01]EEE", 2200 * 72, 20),
    ("zzz_sellp_win.hip", r"_ZN3zzz15spmv_win_kernelILb[01]ELb[01]ELb[01]ELb[01]EEE", (11264 + 8192) * 8, 10)])
def test_special_product_kernels_fit_one_workgroup_of_1024_lanes_per_cu(tmp_path, src_name, kernel_re, lds_budget, variants):
    """The block-row and block-window products (round 6) run ONE workgroup of 1 024 lanes per CU: sixteen wavefronts, four per SIMD,
    i.e. at most 128 VGPRs per lane, with (nearly) no scratch; their dynamic LDS (block table / window + dictionary) plus the static
    part must fit the CU's 160 KB."""
    src = os.path.join(ROOT, "performance-test_amd", "csrc", src_name)
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-I" + os.path.dirname(src),
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "performance-test_amd", "host"), "-c", src, "-o",
           str(tmp_path / "k.o"), "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        if not re.match(kernel_re, name):
            continue
        vgprs = int(re.search(r"VGPRs: (\d+)", b).group(1))
        occ = int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", b).group(1))
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        lds = int(re.search(r"LDS Size \[bytes/block\]: (\d+)", b).group(1))
        assert vgprs <= 128 and occ >= 4 and scratch <= 64, (name, vgprs, occ, scratch)
        assert lds + lds_budget <= 160 * 1024, (name, lds)
        seen += 1
    assert seen == variants, seen  # (dot / single reduction / load policy [/ table form], and the Chebyshev-epilogue variants)
