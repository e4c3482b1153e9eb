cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lpr
run() { name=$1; lpr=$2; shift; shift; ZZZ_SPMV_LPR=$lpr timeout 600 python bench.py --no_cpu_baseline --steps 1 --warmup 1 "$@" 2>/dev/null | tail -1 > gpurun_out/lpr/${name}_$lpr.json; }
for lpr in 1 2 4 8 16; do
  run c5_p3_6m $lpr --order 3 --ndofs 6250000
  run c4_el_p1_4m $lpr --problem_type elasticity --ndofs 4000000
  run el_p3_1m $lpr --problem_type elasticity --order 3 --ndofs 1000000
  run p2_5m $lpr --order 2 --ndofs 5000000
done
run c2_p1_10m 1 --ndofs 10000000
run c2_p1_10m 2 --ndofs 10000000
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/lpr/*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']; c=d['config']
        print(f.split('/')[-1][:-5], 'its',c['krylov_iterations'],'solve %.1f ms spmv %.1f us %.0f GB/s'%(d['phases_ms']['ZZZ Solve'],r['avg_launch_ms']*1e3,r['achieved']), 'norm', c['solution_norm'])
    except Exception as e: print(f,'ERR',e)
PY
