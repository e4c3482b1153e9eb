"""Timing probe (tools build, wrong results by construction): the block-row product at C4 with x read as three component planes
(three dense 8-B loads per block) instead of 24-B node records (a straddling 16-B + 8-B load).  ZZZ_HIP_LIB must point at
libzzz_hip_exp.so.  Prints ms per product for ZZZ_BK_XPROBE = 0 / 1."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "performance-test_amd"))
import zzz

n = int(sys.argv[1]) if len(sys.argv) > 1 else 110
with zzz.Context(0) as c:
    c.cube_generate("elasticity", 1, n, n, n, 1, 0)
    c.pattern_build()
    c.assemble_matrix(zzz.FORM_ELASTICITY)
    c.assemble_vector(zzz.FORM_ELASTICITY)
    c.cg_solve(pc=zzz.PC_JACOBI, rtol=1e-30, max_it=3)
    print(c.spmv_values_info()["special_form"])
    for probe in ("0", "1", "0", "1"):
        os.environ["ZZZ_BK_XPROBE"] = probe
        print("xprobe", probe, "ms per product", min(c.spmv_time(200) for _ in range(3)))
