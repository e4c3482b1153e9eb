cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu -k "partitioned or rccl or peer_memory or driver" 2>&1 | tail -8
for cg in classical single_reduction; do for p in 0 1; do
  ZZZ_P2P=$p timeout 300 python bench.py --no_cpu_baseline --cg $cg --force_comm --ndofs 1250000 --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/p2p_${cg}_$p.json
done; done
timeout 300 python bench.py --no_cpu_baseline --ndofs 1250000 --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/p2p_nocomm.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/p2p_*.json')):
    try:
        d=json.load(open(f)); print(f, d['config']['krylov_iterations'], round(d['phases_ms']['ZZZ Solve'],2), round(d['phases_ms']['ZZZ Solve']/d['config']['krylov_iterations']*1e3,1),'us/it', d['config'].get('scalar_allreduce'))
    except Exception as e: print(f, 'ERR', e)
PY
