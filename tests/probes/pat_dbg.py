import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "performance-test_amd"))
import numpy as np, zzz
prob = sys.argv[1]; dims = tuple(int(v) for v in sys.argv[2:5]); order = int(sys.argv[5]) if len(sys.argv) > 5 else 1
P = zzz.Part(prob, order, *dims)
with zzz.Context(0) as ctx:
    ctx.upload_part(P); print("upload", flush=True)
    ctx.pattern_build(); ctx.sync(); print("pattern ok", prob, dims, ctx.csr_sizes(), flush=True)
