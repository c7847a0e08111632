exec < /dev/null
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r2_pytest_final3.log 2>&1; tail -3 gpurun_out/r2_pytest_final3.log
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2_bench_final3.json 2> gpurun_out/r2_bench_final3.err
python -c "
import json; d=json.load(open('gpurun_out/r2_bench_final3.json')); print({k:d[k] for k in ('value','ms_per_step','tta_only_images_per_s','dice','kept_masks','speedup_vs_cpu_baseline')}); print(d['dice_parity']); print(d['roofline']['avg_iterations_per_stage'], d['roofline']['avg_launch_ms'], d['roofline']['us_per_iteration'], d['roofline']['frac']); print({k:(round(v['value'],2),v['dice']['Dice Coefficient'],v['gagm_avg_launch_ms']) for k,v in d['ab'].items()}); print(d['cpu_baseline']['seconds_all'], d['cpu_baseline']['value']); print([(r['kernel'], round(r['avg_launch_ms'],4), r['frac']) for r in d['roofline_other_kernels']])"
bash tools/collect_profiles.sh r02 2>&1 | tail -4
timeout -k 10 300 python bench.py --images 2048 --warmup 5 --no-ab --no-cpu-baseline > gpurun_out/r02/bench_cfg4_n1.json 2> /dev/null; python -c "
import json; d=json.load(open('gpurun_out/r02/bench_cfg4_n1.json')); print('cfg4 n1', round(d['value'],2), d['steps'], d['dice'], d['kept_masks'], d['roofline']['avg_iterations_per_stage'])"
timeout -k 10 400 python bench.py --workload cfg5 --steps 20 --warmup 5 --no-ab > gpurun_out/r02/bench_cfg5.json 2> /dev/null; python -c "
import json; d=json.load(open('gpurun_out/r02/bench_cfg5.json')); print('cfg5', round(d['value'],2), d['dice'], d['kept_masks'], d['roofline']['avg_iterations_per_stage'], d['dtype'])"
