exec < /dev/null
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r2_pytest_final2.log 2>&1; tail -4 gpurun_out/r2_pytest_final2.log
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2_bench_final2.json 2> gpurun_out/r2_bench_final2.err
python -c "
import json; d=json.load(open('gpurun_out/r2_bench_final2.json')); print({k:d[k] for k in ('value','ms_per_step','tta_only_images_per_s','dice','kept_masks','speedup_vs_cpu_baseline')}); print(d['dice_parity']); print(d['roofline']['avg_iterations_per_stage'], d['roofline']['avg_launch_ms'], d['roofline']['us_per_iteration']); print({k:(round(v['value'],2),v['dice']['Dice Coefficient'],v['gagm_avg_launch_ms']) for k,v in d['ab'].items()}); print(d['cpu_baseline']['seconds_all'], d['cpu_baseline']['value'])"
bash tools/collect_profiles.sh r02 2>&1 | tail -8
