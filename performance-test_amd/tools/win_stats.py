"""Window statistics for an LDS-staged x: per group of 4 slices (256 rows, the library's internal order), how many
distinct columns the group's rows reference, in how many contiguous segments (gaps <= 8 merged)."""
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import zzz  # noqa: E402

for problem, order, dims in (("poisson", 3, (28, 28, 28)), ("poisson", 2, (40, 40, 40)), ("elasticity", 1, (48, 48, 48)),
                             ("elasticity", 2, (22, 22, 22))):
    bs = 3 if problem == "elasticity" else 1
    with zzz.Context(0) as c:
        c.cube_generate(problem, order, *dims, 1, 0)
        c.pattern_build()
        rp, cl, _ = c.csr_download(values=False)
        perm, kind = c.internal_order()          # perm[internal block] = caller block
    n = rp.shape[0] - 1
    A = sp.csr_matrix((np.ones(cl.shape[0], np.int8), cl, rp.astype(np.int64)), shape=(n, n))
    rows = (perm[:, None].astype(np.int64) * bs + np.arange(bs)[None, :]).ravel()   # internal scalar row -> caller row
    inv = np.empty(n, np.int64)
    inv[rows] = np.arange(n)
    Ai = A[rows][:, rows].tocsr()               # the matrix in the library's internal order
    uniq, segs, span = [], [], []
    for g0 in range(0, n, 256):
        cols = np.unique(Ai.indices[Ai.indptr[g0]:Ai.indptr[min(g0 + 256, n)]])
        uniq.append(cols.size)
        gaps = np.diff(cols)
        segs.append(1 + int(np.count_nonzero(gaps > 8)))
        span.append(int(cols.size + gaps[gaps <= 8].sum() - np.count_nonzero(gaps <= 8)))  # window incl. small gaps
    uniq, segs, span = np.array(uniq), np.array(segs), np.array(span)
    q = lambda a: [int(np.percentile(a, p)) for p in (50, 90, 99, 100)]  # noqa: E731
    print(problem, order, "rows", n, "nnz/row", round(cl.shape[0] / n, 1), "| distinct columns per 256 rows p50/90/99/max", q(uniq),
          "| window with gaps <= 8 filled", q(span), "| segments", q(segs), "| groups over 3072:", float((span > 3072).mean()),
          "over 4096:", float((span > 4096).mean()), flush=True)
