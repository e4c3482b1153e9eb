# single-GPU sweep over the BASELINE configs' per-GPU shapes; one summary line each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sweep
run() { name=$1; shift; timeout 600 python bench.py --no_cpu_baseline --steps 2 --warmup 1 "$@" 2>gpurun_out/sweep/$name.err | tail -1 > gpurun_out/sweep/$name.json; }
run c1_p1_500k --ndofs 500000
run c2_p1_10m --ndofs 10000000
run c4_el_p1_500k --problem_type elasticity --scaling_type weak --ndofs 500000
run c4_el_p1_4m --problem_type elasticity --scaling_type strong --ndofs 4000000
run c5_p3_6m --order 3 --ndofs 6250000
run p2_5m --order 2 --ndofs 5000000
run el_p2_2m --problem_type elasticity --order 2 --ndofs 2000000
run el_p3_1m --problem_type elasticity --order 3 --ndofs 1000000
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/sweep/*.json')):
    try:
        d=json.load(open(f)); c=d['config']; ph=d['phases_ms']; r=d['roofline']
        print(f.split('/')[-1][:-5], 'dofs',c['dofs'],'nnz',c['nnz_rank0'],'its',c['krylov_iterations'],'| pattern %.1f asmA %.2f asmb %.2f solve %.1f ms | spmv %.1f us %.0f GB/s | value %.3g'%(
          ph['create_matrix (sparsity pattern, adjacency, tiles)'],ph['ZZZ Assemble matrix'],ph['ZZZ Assemble vector'],ph['ZZZ Solve'],r['avg_launch_ms']*1e3,r['achieved'],d['value']), '|', c.get('spmv_column_stream'))
    except Exception as e:
        print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
PY
