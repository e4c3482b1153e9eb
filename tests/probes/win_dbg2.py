import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "performance-test_amd"))
import numpy as np, zzz
os.environ["ZZZ_DEBUG_SYNC"] = "1"
dims = tuple(int(v) for v in sys.argv[1:4])
with zzz.Context(0) as c:
    c.cube_generate("poisson", 3, *dims, 1, 0)
    c.pattern_build(); c.assemble_matrix(zzz.FORM_POISSON); c.assemble_vector(zzz.FORM_POISSON)
    vi = c.spmv_values_info()
    print(dims, vi["special_form"], vi["block_chunks"], flush=True)
    print("time", c.spmv_time(10))
