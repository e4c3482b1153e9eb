"""One rank's share of BASELINE configs[2] at N GPUs on one GPU (bench.run_rank_size), for A/B runs under knobs:
python3 rank_probe.py N [N ...]  ->  one line per N with the per-iteration time of both CG forms' better one."""
import os, sys, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "performance-test_amd"))
import bench  # noqa: E402
import zzz  # noqa: E402

nx, ny, nz, r = zzz.mesh_size(10000000, True, 1, 1, 1)
nx, ny, nz = nx << r, ny << r, nz << r
for n in [int(v) for v in sys.argv[1:]] or [8]:
    rec = bench.run_rank_size("poisson", 1, nx, ny, -(-nz // n), f"c3 at {n} GPUs")
    print(json.dumps({k: rec[k] for k in ("rows", "krylov_iterations", "solve_ms", "us_per_iteration", "cg_form")}), flush=True)
