"""Assembly kernels alone (no solve), for rocprofv3 --kernel-trace --stats: python3 asm_probe.py [problem order n reps].
With ZZZ_HIP_LIB = the tools build and ZZZ_ASM_PROBE = a mask (1 no column search, 2 no coordinate gathers, 4 no flag
gathers, 8 no geometry) the P1 matrix kernel runs a timing ablation (wrong values)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import zzz

problem = sys.argv[1] if len(sys.argv) > 1 else "poisson"
order = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = int(sys.argv[3]) if len(sys.argv) > 3 else 215
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
with zzz.Context(0) as c:
    info = c.cube_generate(problem, order, n, n, n, 1, 0)
    c.pattern_build()
    form = zzz.FORM_ELASTICITY if problem == "elasticity" else zzz.FORM_POISSON
    for _ in range(reps):
        c.assemble_matrix(form)
        c.assemble_vector(form)
    c.vec_download(zzz.VEC_B)
    print("dofs", int(info[0]))
