import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "performance-test_amd"))
import numpy as np, zzz
os.environ["ZZZ_SELLP_BWIN"] = "2"
for sellp in (None, "2"):
    for feed in ("device", "host"):
        if sellp: os.environ["ZZZ_SELLP"] = sellp
        else: os.environ.pop("ZZZ_SELLP", None)
        with zzz.Context(0) as c:
            if feed == "device":
                c.cube_generate("poisson", 3, 12, 12, 12, 1, 0)
            else:
                c.upload_part(zzz.Part("poisson", 3, 12, 12, 12))
            c.pattern_build(); c.assemble_matrix(zzz.FORM_POISSON)
            vi = c.spmv_values_info()
            print("ZZZ_SELLP", sellp, feed, vi["special_form"], vi["block_chunks"], c.spmv_info_raw(), flush=True)
