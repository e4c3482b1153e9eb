"""Column-code statistics of the operator stream for long rows: per chunk (64 rows x 8 slots, internal order), does the
spread of every slot's columns fit 8 bits -- over the whole wavefront, or over its two halves / four quarters separately?"""
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import zzz  # noqa: E402

for problem, order, dims in (("poisson", 3, (28, 28, 28)), ("poisson", 2, (40, 40, 40))):
    with zzz.Context(0) as c:
        c.cube_generate(problem, order, *dims, 1, 0)
        c.pattern_build()
        c.assemble_matrix(zzz.FORM_POISSON)
        rp, cl, v = c.csr_download()
        perm, kind = c.internal_order()
    n = rp.shape[0] - 1
    A = sp.csr_matrix((v, cl, rp.astype(np.int64)), shape=(n, n))
    A.eliminate_zeros()
    Ai = A[perm][:, perm].tocsr()
    Ai.sort_indices()
    tot = fit1 = fit2 = fit4 = 0
    for s0 in range(0, n - 63, 64 * 7):     # a sample of the slices
        rows = [Ai.indices[Ai.indptr[r]:Ai.indptr[r + 1]] for r in range(s0, s0 + 64)]
        mlen = max(len(r) for r in rows)
        M = np.full((64, (mlen + 7) // 8 * 8), -1, np.int64)
        for i, r in enumerate(rows):
            M[i, :len(r)] = r
        for j in range(M.shape[1] // 8):
            blk = M[:, 8 * j:8 * j + 8]
            def fits(parts):
                for e in range(8):
                    col = blk[:, e]
                    for q in np.array_split(np.arange(64), parts):
                        cq = col[q][col[q] >= 0]
                        if cq.size and cq.max() - cq.min() > 255:
                            return False
                return True
            tot += 1
            f1 = fits(1)
            fit1 += f1
            f2 = f1 or fits(2)
            fit2 += f2
            fit4 += f2 or fits(4)
    print(problem, order, "chunks sampled", tot, "8-bit codes fit: one base per slot", round(fit1 / tot, 3), "| two", round(fit2 / tot, 3),
          "| four", round(fit4 / tot, 3), flush=True)
