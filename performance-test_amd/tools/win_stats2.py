"""For a PARTIAL x window: per group of 256 rows (internal order), segments (gaps <= 8 filled) ranked by references per
slot; which share of the group's gathers would a window of W doubles serve?"""
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import zzz  # noqa: E402

for problem, order, dims in (("poisson", 3, (28, 28, 28)), ("poisson", 2, (40, 40, 40))):
    bs = 1
    with zzz.Context(0) as c:
        c.cube_generate(problem, order, *dims, 1, 0)
        c.pattern_build()
        rp, cl, _ = c.csr_download(values=False)
        perm, kind = c.internal_order()
    n = rp.shape[0] - 1
    A = sp.csr_matrix((np.ones(cl.shape[0], np.int8), cl, rp.astype(np.int64)), shape=(n, n))
    Ai = A[perm][:, perm].tocsr()
    for W in (2048, 3072, 4096):
        served, total = 0, 0
        for g0 in range(0, n, 256):
            cols = Ai.indices[Ai.indptr[g0]:Ai.indptr[min(g0 + 256, n)]]
            u, cnt = np.unique(cols, return_counts=True)
            brk = np.flatnonzero(np.diff(u) > 8)
            starts = np.concatenate(([0], brk + 1))
            ends = np.concatenate((brk, [u.size - 1]))
            seglen = u[ends] - u[starts] + 1
            segref = np.add.reduceat(cnt, starts)
            order_ = np.argsort(-(segref / seglen))
            room = W
            for i in order_:
                if seglen[i] <= room:
                    room -= seglen[i]
                    served += segref[i]
            total += cols.size
        print(problem, order, "window", W, "doubles: share of gathers served from LDS", round(served / total, 3), flush=True)
