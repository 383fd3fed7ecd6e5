#!/bin/bash
# Turns what tools/collect_profiles_r02.sh left under gpurun_out/r02/ into the files kept in profiles/.
set -e
O=gpurun_out/r02
python tools/kernel_stats_csv.py $O/stats/stats_kernel_stats.csv profiles/r02_kernel_stats.csv
python tools/kernel_stats_csv.py $O/statsh/statsh_kernel_stats.csv profiles/r02_hessian_kernel_stats.csv
python tools/kernel_stats_csv.py $O/cfg3/cfg3_kernel_stats.csv profiles/r02_cfg3_kernel_stats.csv
python tools/kernel_stats_csv.py $O/hess3/hess3_kernel_stats.csv profiles/r02_hess_cfg3_kernel_stats.csv
python tools/pmc_summary_csv.py $O/fetch/fetch_counter_collection.csv $O/write/write_counter_collection.csv profiles/r02_pmc_hbm.json > /dev/null
cp $O/bench.json profiles/r02_bench.json
cp $O/bench_single.json profiles/r02_bench_single_stream.json
python tools/kernel_stats_csv.py $O/stats2/stats2_kernel_stats.csv profiles/r02_kernel_stats_two_streams.csv
F='amdgpu.ids\|rocprofv3\|output_stream\|tool.cpp'
grep -v "$F" $O/cfg3.log > profiles/r02_cfg3.log
grep -v "$F" $O/pipeline.log > profiles/r02_pipeline.log
grep -v "$F" $O/hess3.log > profiles/r02_hess_cfg3.log
grep -v "$F" $O/model.log > profiles/r02_model_create.log
grep -v "$F" $O/hess_small.log > profiles/r02_hessian_small.log
grep -v "$F" $O/lockstep.log > profiles/r02_lockstep.log
grep -v "$F" $O/fit.log > profiles/r02_fit.log
grep -v "$F" $O/phase_cycles.log > profiles/r02_phase_cycles.log
(grep "^pass" $O/bo_loop.log | sed 's/^/--hessian=1 (default): /'; grep "^pass" $O/bo_loop_lbfgs.log | sed 's/^/--hessian=0: /') > profiles/r02_bo_loop.log
python - <<'PY'
import json
h = json.load(open('profiles/r02_pmc_hbm.json'))
p = 'profiles/pmc_traffic.json'
d = json.load(open(p))
step = {k: h[k]['hbm_bytes_per_launch'] for k in ('k_gram<true>', 'k_gemm_nt_tall', 'k_stage2<64, 20>', 'k_gemm_grad<8>', 'k_grad_fin', 'k_best_record')}
d['moments_sdp_hbm_bytes_per_launch'] = step['k_stage2<64, 20>']
d['step_hbm_bytes_by_kernel'] = step
d['step_hbm_bytes'] = sum(step.values())
json.dump(d, open(p, 'w'), indent=1)
print(step, d['step_hbm_bytes'])
PY
