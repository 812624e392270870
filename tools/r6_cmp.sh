for rep in 1 2; do
python bench_extra.py --workload dcpt --dtype bf16 --size 256 --steps 5 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench_extra 256', d['ms_per_step'])"
python bench_extra.py --workload dcpt --dtype bf16 --size 128 --steps 5 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench_extra 128', d['ms_per_step'])"
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['secondary']; print('bench.py headline', d['ms_per_step'], '256', s['dcpt_all_bf16_256']['ms_per_step'], '128', s['dcpt_all_bf16_128']['ms_per_step'], 'naf_bf16', s['naf_bf16_256']['ms_per_step'])"
done
