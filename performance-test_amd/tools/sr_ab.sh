set -x
cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
for cg in classical single_reduction; do
  timeout 300 python bench.py --no_cpu_baseline --cg $cg --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/sr_${cg}.json
  timeout 300 python bench.py --no_cpu_baseline --cg $cg --force_comm --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/sr_${cg}_comm.json
done
for n in 1250000 2500000; do for cg in classical single_reduction; do
  timeout 300 python bench.py --no_cpu_baseline --cg $cg --ndofs $n --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/sr_${cg}_$n.json
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/sr_*.json')):
    try:
        d=json.load(open(f)); print(f, d['config']['krylov_iterations'], round(d['phases_ms']['ZZZ Solve'],2), round(d['phases_ms']['ZZZ Solve']/d['config']['krylov_iterations']*1e3,1),'us/it', round(d['roofline']['avg_launch_ms']*1e3,1))
    except Exception as e: print(f, 'ERR', e)
PY
