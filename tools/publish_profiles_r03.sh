#!/bin/bash
# Turns what tools/collect_profiles_r03.sh left under gpurun_out/r03/ into the files kept in profiles/.
set -e
O=gpurun_out/r03
python tools/kernel_stats_csv.py $O/stats/stats_kernel_stats.csv profiles/r03_kernel_stats.csv
python tools/kernel_stats_csv.py $O/stats2/stats2_kernel_stats.csv profiles/r03_kernel_stats_two_streams.csv
python tools/kernel_stats_csv.py $O/statsh/statsh_kernel_stats.csv profiles/r03_hessian_kernel_stats.csv
python tools/kernel_stats_csv.py $O/cfg3/cfg3_kernel_stats.csv profiles/r03_cfg3_kernel_stats.csv
python tools/kernel_stats_csv.py $O/hess3/hess3_kernel_stats.csv profiles/r03_hess_cfg3_kernel_stats.csv
python tools/kernel_stats_csv.py $O/round_lbfgs/round_lbfgs_kernel_stats.csv profiles/r03_round_lbfgs_kernel_stats.csv
python tools/kernel_stats_csv.py $O/round_newton/round_newton_kernel_stats.csv profiles/r03_round_newton_kernel_stats.csv
python tools/pmc_summary_csv.py $O/fetch/fetch_counter_collection.csv $O/write/write_counter_collection.csv profiles/r03_pmc_hbm.json > /dev/null
cp $O/bench.json profiles/r03_bench.json
cp $O/bench_single.json profiles/r03_bench_single_stream.json
cp $O/bench_collective.json profiles/r03_bench_collective.json
python tools/pmc_summary_csv.py $O/cfg3_fetch/fetch_counter_collection.csv $O/cfg3_write/write_counter_collection.csv profiles/r03_cfg3_pmc_hbm.json > /dev/null
python - <<'PY'
import json
r = json.load(open('gpurun_out/r03/cfg3_roofline.json'))
h = json.load(open('profiles/r03_cfg3_pmc_hbm.json'))['k_stage2<256, 0>']
r['traffic'] = h['hbm_bytes_per_launch']
r['traffic_note'] = ('HBM bytes per launch of k_stage2<256, 0> at B = 2048 from two counter passes (FETCH_SIZE doubled as on gfx950, '
                     'WRITE_SIZE; profiles/r03_cfg3_pmc_hbm.json); algorithmic bytes per launch of this kernel: the A rows of 2048 candidates '
                     '(2048 * 50 * 1024 * 8 = 839 MB) read twice, for the moments and for Z = M11 A after the solve, and Z written once: 2.52 GB')
json.dump(r, open('profiles/r03_cfg3_roofline.json', 'w'), indent=1)
PY
F='amdgpu.ids\|rocprofv3\|output_stream\|tool.cpp'
grep -v "$F" $O/cfg3.log > profiles/r03_cfg3.log
grep -v "$F" $O/hess3.log > profiles/r03_hess_cfg3.log
grep -v "$F" $O/model.log > profiles/r03_model_create.log
grep -v "$F" $O/hess_small.log > profiles/r03_hessian_small.log
grep -v "$F" $O/lockstep.log > profiles/r03_lockstep.log
grep -v "$F" $O/fit.log > profiles/r03_fit.log
grep -v "$F" $O/phase_cycles.log > profiles/r03_phase_cycles.log
(grep -v "$F" $O/phase_cycles_q32.log; grep -v "$F" $O/phase_cycles_q50.log) > profiles/r03_phase_cycles_q32_q50.log
grep -v "$F" $O/hess_wide_phase_cycles.log > profiles/r03_hess_wide_phase_cycles.log
cp $O/micro_mm_lds.log profiles/r03_micro_mm_lds.log
python tools/pmc_summary_csv.py $O/zfree_fetch/fetch_counter_collection.csv $O/zfree_write/write_counter_collection.csv profiles/r03_zfree_pmc_hbm.json > /dev/null
python tools/kernel_stats_csv.py $O/zfree_stats/stats_kernel_stats.csv profiles/r03_zfree_kernel_stats.csv
cp $O/bench_zfree.json profiles/r03_bench_zfree.json
if [ -f $O/micro_mixed_pipes.log ]; then cp $O/micro_mixed_pipes.log profiles/r03_micro_mixed_pipes.log; grep -v "$F" $O/warm_probe_first_order.log > profiles/r03_warm_probe_first_order.log; fi
(grep -v "$F" $O/round_lbfgs.log; grep -v "$F" $O/round_newton.log) > profiles/r03_round.log
(grep "^pass" $O/bo_loop_lbfgs.log | sed 's/^/--nl_solver=lockstep (default) --hessian=0: /'; grep "^pass" $O/bo_loop_newton.log | sed 's/^/--nl_solver=lockstep-newton: /') > profiles/r03_bo_loop.log
python - <<'PY'
import json
h = json.load(open('profiles/r03_pmc_hbm.json'))
p = 'profiles/pmc_traffic.json'
d = json.load(open(p))
step = {k: h[k]['hbm_bytes_per_launch'] for k in ('k_gram<true, 8>', 'k_gemm_nt_tall', 'k_stage2<64, 20>', 'k_gemm_grad<8, 1>', 'k_grad_fin', 'k_best_record')}
d['moments_sdp_hbm_bytes_per_launch'] = step['k_stage2<64, 20>']
d['step_hbm_bytes_by_kernel'] = step
d['step_hbm_bytes'] = sum(step.values())
d['source'] = 'profiles/r03_pmc_hbm.json'
json.dump(d, open(p, 'w'), indent=1)
print(step, d['step_hbm_bytes'])
PY
