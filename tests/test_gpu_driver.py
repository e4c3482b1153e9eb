"""GPU parity tests, part 5: the drop-in driver binary (CLI, timers, stdout) and bench.py's contract."""
from _gpu_helpers import *  # noqa: F401,F403 -- helpers, fixtures (ctx), np / os / zzz / zo

pytestmark = pytest.mark.gpu  # noqa: F405


def test_driver_binary_surface():
    """dolfinx-scaling-test keeps the reference's CLI, timer names and stdout lines
    (src/main.cpp:57-74,186-205,232-233; src/mesh.cpp:192-193; README.md:148-161)."""
    import subprocess

    exe = os.path.join(zzz.PKG, "dolfinx-scaling-test")
    assert os.path.exists(exe)
    # the reference's CI configuration: weak, 50 000 dofs, P1 (ccpp.yml:56-70) with CG + Jacobi
    cmd = [exe, "--problem_type", "poisson", "--scaling_type", "weak", "--ndofs", "50000", "-ksp_type", "cg",
           "-pc_type", "jacobi", "-ksp_rtol", "1.0e-8", "-log_view", "-options_left", "--some_unknown_flag", "7"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    s = out.stdout
    assert "UnitCube (36x32x40) to be refined 0 times" in s
    for line in ("Test problem summary", "  Problem type:    poisson", "  Scaling type:    weak", "  Num processes:   1",
                 "  Num cells:       276480 (276 thousand)", "  Total degrees of freedom:               50061 (50.1 thousand)",
                 "  Average degrees of freedom per process: 50061", "Summary of timings", "ZZZ Create Mesh",
                 "ZZZ FunctionSpace", "ZZZ Assemble ", "ZZZ Create boundary conditions", "ZZZ Create RHS function",
                 "ZZZ Assemble matrix", "ZZZ Assemble vector", "ZZZ Solve", "*** Number of Krylov iterations: ",
                 "*** Solution norm:  "):
        assert line in s, line
    its = int(s.split("*** Number of Krylov iterations: ")[1].split()[0])
    nrm = float(s.split("*** Solution norm:  ")[1].split()[0])
    # SURVEY.md 8c provisional sanity values for this config: 194 iterations, |u| = 47.56358
    assert abs(its - 194) <= 2 and abs(nrm - 47.56358) < 1e-3
    # elasticity + P2, cgpoisson, bad options
    out = subprocess.run([exe, "--problem_type", "elasticity", "--order", "2", "--ndofs", "20000", "-pc_type", "jacobi",
                          "-ksp_rtol", "1e-8"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ZZZ Create near-nullspace" in out.stdout and "ZZZ Create forms" in out.stdout
    out = subprocess.run([exe, "--problem_type", "cgpoisson", "--ndofs", "30000"], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and "CG matrix-free action processed: " in out.stdout and "Gdof/s" in out.stdout
    assert "ZZZ Assemble matrix" not in out.stdout
    for bad in (["--scaling_type", "sideways"], ["--problem_type", "stokes"], ["--order", "4"]):
        out = subprocess.run([exe] + bad, capture_output=True, text=True, timeout=60)
        assert out.returncode != 0
    # --mesh_type unstructured (the reference's CI runs it, ccpp.yml:102-117): the ring-with-spurs mesh through the host
    # feed; iteration count and norm against the oracle on the same feed
    out = subprocess.run([exe, "--problem_type", "poisson", "--mesh_type", "unstructured", "--scaling_type", "weak", "--ndofs",
                          "50000", "-ksp_type", "cg", "-pc_type", "jacobi", "-ksp_rtol", "1.0e-8"], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-1000:]
    su = out.stdout
    m = zzz.host().zzzh_spoke_size(50000, 1)
    assert f"each cut {m}x{m}x{m}" in su and "ZZZ Assemble matrix" in su and "ZZZ Solve" in su
    Pu = zzz.Part.spoke("poisson", 1, m)
    assert f"  Total degrees of freedom:               {Pu.n_owned}" in su and f"  Num cells:       {Pu.ncells}" in su
    orp, ocl = zo.pattern(Pu.n_owned, Pu.cell_dofs, 1)
    ovu = zo.assemble_matrix(0, 1, Pu.x, Pu.cells, Pu.cell_dofs, Pu.bc_marker(), orp, ocl)
    obu = zo.assemble_vector(0, 1, Pu.x, Pu.cells, Pu.cell_dofs, Pu.f, Pu.g, Pu.facets, Pu.bc_marker())
    oitu, ouu, _, _ = zo.pcg(orp, ocl, ovu, obu, rtol=1e-8)
    assert abs(int(su.split("*** Number of Krylov iterations: ")[1].split()[0]) - oitu) <= 2
    assert abs(float(su.split("*** Solution norm:  ")[1].split()[0]) - np.linalg.norm(ouu)) <= 1e-5 * np.linalg.norm(ouu)
    # ... and cut into sectors over three ranks (mpirun -np 2 in the reference's CI; host-mediated communicator, one GPU):
    # the same problem (strong scaling), so the same iteration count and norm
    un = [exe, "--problem_type", "poisson", "--mesh_type", "unstructured", "--scaling_type", "strong", "--ndofs", "50000", "-ksp_type",
          "cg", "-pc_type", "jacobi", "-ksp_rtol", "1.0e-8"]
    o1 = subprocess.run(un, capture_output=True, text=True, timeout=300)
    o3 = subprocess.run(un + ["--ngpus", "3", "--comm", "local"], capture_output=True, text=True, timeout=300)
    assert o1.returncode == 0 and o3.returncode == 0, o3.stderr[-1000:]
    it1, it3 = (int(o.stdout.split("*** Number of Krylov iterations: ")[1].split()[0]) for o in (o1, o3))
    n1, n3 = (float(o.stdout.split("*** Solution norm:  ")[1].split()[0]) for o in (o1, o3))
    assert abs(it1 - it3) <= 1 and abs(n1 - n3) <= 1e-8 * n1 and "  Num processes:   3" in o3.stdout
    # the polynomial preconditioner through the options database: same solution norm as Jacobi's run above, fewer
    # iterations; on 2 ranks (host-mediated communicator, both on this GPU) the same again; options checked
    base = [exe, "--problem_type", "poisson", "--scaling_type", "weak", "--ndofs", "50000", "-ksp_type", "cg", "-ksp_rtol",
            "1.0e-8", "-pc_type"]

    def its_norm(args):
        o = subprocess.run(args, capture_output=True, text=True, timeout=300)
        assert o.returncode == 0, o.stderr[-1000:]
        return (int(o.stdout.split("*** Number of Krylov iterations: ")[1].split()[0]),
                float(o.stdout.split("*** Solution norm:  ")[1].split()[0]), o.stdout)

    seen = []
    for extra, opts in (([], []),
                        ([], ["-pc_chebyshev_jacobi_degree", "4", "-pc_chebyshev_jacobi_ratio", "40", "-pc_chebyshev_jacobi_esteig",
                              "-1", "-ksp_view"]),
                        (["--ngpus", "2", "--comm", "local"], [])):
        its_j, nrm_j, _ = its_norm(base + ["jacobi"] + extra)
        its_c, nrm_c, text = its_norm(base + ["chebyshev_jacobi"] + extra + opts)
        assert 20 < its_c < 0.45 * its_j and abs(nrm_c - nrm_j) < 1e-6 * nrm_j, (extra, opts, its_c, its_j, nrm_c, nrm_j)
        seen.append(its_c)
        if "-ksp_view" in opts:
            assert "PC Object: type: chebyshev_jacobi" in text
    assert seen[1] < seen[0]   # degree 4 against 3
    base = base + ["chebyshev_jacobi"]
    its_s, nrm_s, _ = its_norm(base + ["-ksp_cg_single_reduction"])   # one reduction point per three products
    assert abs(its_s - seen[0]) <= 2 and abs(nrm_s - 47.56358) < 1e-3
    # --operator matfree (an extension): KSPCG + Jacobi with no matrix, one rank and two; the assembled run's numbers
    jac = [exe, "--problem_type", "poisson", "--scaling_type", "weak", "--ndofs", "50000", "--order", "2", "-ksp_type", "cg",
           "-ksp_rtol", "1.0e-8", "-pc_type", "jacobi"]
    its_a, nrm_a, _ = its_norm(jac)
    its_m, nrm_m, text = its_norm(jac + ["--operator", "matfree", "-ksp_view"])
    assert abs(its_m - its_a) <= 2 and abs(nrm_m - nrm_a) < 1e-6 * nrm_a and "type=shell" in text and "ZZZ Assemble matrix" in text
    two = ["--ngpus", "2", "--comm", "local"]  # (weak scaling: twice the problem)
    its_a2, nrm_a2, _ = its_norm(jac + two)
    its_m2, nrm_m2, _ = its_norm(jac + ["--operator", "matfree"] + two)
    assert abs(its_m2 - its_a2) <= 2 and abs(nrm_m2 - nrm_a2) < 1e-6 * nrm_a2
    for bad in (["--problem_type", "elasticity"], ["-pc_type", "chebyshev_jacobi"], ["--operator", "sparse"]):
        o = subprocess.run(jac + ["--operator", "matfree"] + bad, capture_output=True, text=True, timeout=60)
        assert o.returncode != 0
    # --memory_profiling: the logging thread of src/mem.cpp (VSIZE / RSS in kB every 100 ms, here plus used HBM)
    out = subprocess.run([exe, "--problem_type", "poisson", "--ndofs", "2000000", "--memory_profiling", "-pc_type", "jacobi",
                          "-ksp_rtol", "1e-8"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1000:]
    mem = [ln for ln in out.stderr.splitlines() if "[MEM] [warning] VSIZE=" in ln]
    assert mem and all("RSS=" in ln and "HBM=" in ln for ln in mem)
    assert int(mem[-1].split("HBM=")[1]) > 100000  # kB: the 2 M-dof problem is resident on the device


def test_driver_multi_rank_threads_on_one_gpu():
    """The driver's multi-rank machinery (one thread per rank, barriers, max-over-ranks timers, summary)
    with the host-mediated communicator: 3 ranks on GPU 0 must print the single-rank iteration count
    and solution norm."""
    import subprocess

    exe = os.path.join(zzz.PKG, "dolfinx-scaling-test")
    base = [exe, "--problem_type", "poisson", "--scaling_type", "strong", "--ndofs", "60000", "-ksp_type", "cg",
            "-pc_type", "jacobi", "-ksp_rtol", "1e-8"]
    one = subprocess.run(base, capture_output=True, text=True, timeout=300)
    three = subprocess.run(base + ["--ngpus", "3", "--comm", "local"], capture_output=True, text=True, timeout=300)
    assert one.returncode == 0 and three.returncode == 0, three.stderr

    def parse(s):
        return (int(s.split("*** Number of Krylov iterations: ")[1].split()[0]),
                float(s.split("*** Solution norm:  ")[1].split()[0]))

    i1, n1 = parse(one.stdout)
    i3, n3 = parse(three.stdout)
    assert "Num processes:   3" in three.stdout
    assert abs(i3 - i1) <= 1 and abs(n3 - n1) <= 1e-6 * n1
    # weak scaling: ndofs is per process (src/mesh.cpp:87-90)
    w = subprocess.run([exe, "--problem_type", "elasticity", "--scaling_type", "weak", "--ndofs", "9000", "--ngpus", "2",
                        "--comm", "local", "-pc_type", "jacobi", "-ksp_rtol", "1e-8"], capture_output=True, text=True, timeout=300)
    assert w.returncode == 0 and "Num processes:   2" in w.stdout, w.stderr
    tot = int(w.stdout.split("Total degrees of freedom:")[1].split()[0])
    assert 15000 < tot < 21000
    # cgpoisson (matrix-free action + src/cg.h, 100 iterations) partitioned: same norm as on one rank
    mf = [exe, "--problem_type", "cgpoisson", "--scaling_type", "strong", "--ndofs", "40000", "--order", "2"]
    m1 = subprocess.run(mf, capture_output=True, text=True, timeout=300)
    m2 = subprocess.run(mf + ["--ngpus", "2", "--comm", "local"], capture_output=True, text=True, timeout=300)
    assert m1.returncode == 0 and m2.returncode == 0, m2.stderr
    (j1, q1), (j2, q2) = parse(m1.stdout), parse(m2.stdout)
    assert j1 == j2 == 100 and abs(q2 - q1) <= 1e-9 * q1
    # two ranks on ONE GPU, peer-memory all-reduce, a stream whose slice dictionaries are built and DECLINED at the first
    # product of the solve (elasticity P2: the 60 % rule): nothing on that path may free device memory -- hipFree waits for
    # the whole device, i.e. for the other rank's kernel that polls its mailbox for this rank (round 5: a 3-s time-out)
    e2 = subprocess.run([exe, "--problem_type", "elasticity", "--order", "2", "--scaling_type", "strong", "--ndofs", "150000",
                         "--ngpus", "2", "--comm", "local", "-ksp_type", "cg", "-pc_type", "jacobi", "-ksp_rtol", "1e-8"],
                        capture_output=True, text=True, timeout=300)
    assert e2.returncode == 0 and "timed out" not in e2.stderr and "Num processes:   2" in e2.stdout, e2.stderr[-1500:]


def test_bench_contract_line():
    """bench.py prints ONE JSON line (last line of stdout) with the contract's keys, also when the RCCL
    code path is attached (RCCL's init banner must not reach stdout)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(zzz.PKG)
    for extra in ([], ["--no_cpu_baseline", "--force_comm"]):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--ndofs", "40000", "--steps", "1",
                              "--warmup", "0"] + extra, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
        d = json.loads(lines[-1])
        assert len([ln for ln in lines if ln.lstrip().startswith("{")]) == 1
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline"):
            assert k in d, k
        assert d["unit"] == "DoF/s" and d["dtype"] == "f64" and d["n_gpus"] == 1 and d["vs_baseline"] is None
        assert "workload" in d["config"] and "model" not in d["config"]
        r = d["roofline"]
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
        it = r["iteration"]
        assert it["bytes"] > r["bytes_per_launch"] and abs(it["frac"] - it["achieved"] / r["peak"]) < 1e-12
        # the scalar copies the driver's parser keeps (round 5): the same numbers as the nested records
        assert r["iteration_frac"] == it["frac"] and r["iteration_us"] > 0
        assert ("full_pattern" in r) == ("full_pattern_frac" in r)
        if not extra:
            c = d["cpu_baseline"]
            assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
            # the CPU leg solves the whole problem (nothing extrapolated) and reports its own iteration count
            assert abs(c["krylov_iterations"] - d["config"]["krylov_iterations"]) <= 2 and c["solve_s"] > 0
            # BASELINE configs[0] on one host thread, and what one rank of the multi-GPU configurations does per iteration
            assert c["c1_1_thread_dofs_per_s"] > 0 and c["c1_1_thread"]["threads"] == 1
            # (other_configs.rank_sizes exists for the default 10 M-dof run only: its record is checked below at a small size)
    # one record of other_configs.rank_sizes (a rank's share as a problem of its own, communication path attached), in a child
    # process as bench.py runs it
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r); import bench; "
            "print(json.dumps(bench.run_rank_size('poisson', 1, 20, 18, 7, 'test')))" % (root, zzz.PKG))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["rows"] == 21 * 19 * 8 and rec["us_per_iteration"] > 0 and rec["krylov_iterations"] > 0 and rec["solve_ms"] > 0
    assert rec["cg_form"] in ("classical", "single_reduction") and rec["scalar_allreduce"] in ("peer-memory mailboxes", "ncclAllReduce")
    # the N > 1 machinery on one GPU (1-rank communicator): mailbox attach, warm-up probe, CG-form tuning
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--ndofs", "60000", "--steps", "2", "--warmup", "1",
                          "--no_cpu_baseline", "--force_comm"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    cfg = d["config"]
    assert set(cfg["cg_form_tuning_s"]) == {"single_reduction+peer_memory", "classical+peer_memory",
                                            "single_reduction+ncclAllReduce", "classical+ncclAllReduce"}
    assert all(v > 0 for v in cfg["cg_form_tuning_s"].values())
    assert cfg["scalar_allreduce"] in ("ncclAllReduce", "peer-memory mailboxes over xGMI (one kernel: reduce + exchange)")
    assert ("-ksp_cg_single_reduction" in cfg["workload"]) == min(
        cfg["cg_form_tuning_s"], key=cfg["cg_form_tuning_s"].get).startswith("single_reduction")


def test_bench_multi_gpu_process_layout_on_one_gpu():
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one process per GPU,
    env:// rendezvous on 127.0.0.1) with --force_dist: libzzz_hip and /opt/rocm's RCCL bound before torch's own
    copies, gloo process group, unique-id broadcast, ncclCommInitRank + ncclCommSplit, mailbox handle all_gather and
    attach, device-generated slab feed, warm-up vote, CG-form / transport tuning, max-over-ranks timing.  One rank is
    what a 1-GPU box can run of it; the partition logic itself is covered by the *_partitioned_* tests."""
    import json
    import socket
    import subprocess
    import sys

    root = os.path.dirname(zzz.PKG)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--force_dist", "--ndofs", "200000",
           "--steps", "2", "--warmup", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    cfg = d["config"]
    assert d["n_gpus"] == 1 and d["value"] > 0 and "cpu_baseline" not in d
    assert cfg["feed"].startswith("generated on the device") and len(cfg["cg_form_tuning_s"]) == 4
    assert cfg["scalar_allreduce"] in ("ncclAllReduce", "peer-memory mailboxes over xGMI (one kernel: reduce + exchange)")
    assert abs(cfg["krylov_iterations"] - 306) <= 40 and cfg["relative_residual"] <= 1e-8


@pytest.mark.parametrize("cg", ["classical", "single_reduction"])
@pytest.mark.parametrize("p2p", ["0", "1"])
def test_bench_multi_gpu_branches_on_one_gpu(cg, p2p):
    """Every branch the N > 1 tuning of bench.py can select -- {classical, single_reduction} x {ncclAllReduce, peer-memory
    mailboxes} -- runs under test on one GPU, launched the way the driver launches N > 1 (torch.distributed.run, env://
    on 127.0.0.1, gloo group, unique-id broadcast, 1-rank RCCL communicator): `--cg` given explicitly means NO tuning
    solves, the JSON line says which combination ran and why, carries the per-rank halo wait, and the solve is the
    single-GPU one (975-iteration problem scaled down: same count in every branch)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(zzz.PKG)
    env = dict(os.environ, ZZZ_P2P=p2p, MASTER_ADDR="127.0.0.1")
    port = 29600 + (0 if cg == "classical" else 2) + int(p2p)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--force_dist", "--ndofs", "200000",
           "--steps", "1", "--warmup", "1", "--cg", cg]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")][-1])
    cfg = d["config"]
    assert cfg["cg_form"]["form"] == cg and cfg["cg_form"]["chosen_by"] == "--cg " + cg
    assert cfg["cg_form"]["scalar_allreduce"] == ("peer_memory" if p2p == "1" else "ncclAllReduce")
    assert "cg_form_tuning_s" not in cfg  # explicit --cg: no warm-up solves of other combinations
    assert ("-ksp_cg_single_reduction" in cfg["workload"]) == (cg == "single_reduction")
    rk = cfg["ranks"][0]
    assert rk["ranks"] == 1 and rk["peer_memory_allreduce"] == int(p2p) and "halo_wait_ns_per_product" in rk
    assert 250 <= cfg["krylov_iterations"] <= 330 and cfg["relative_residual"] <= 1e-8
