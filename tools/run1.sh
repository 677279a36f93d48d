(cd /tmp && time python3 $OLDPWD/bench.py --gpus 1 --steps 20 --warmup 5 > $OLDPWD/gpurun_out/bench_c.json 2> $OLDPWD/gpurun_out/bench_c.err); tail -3 gpurun_out/bench_c.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/bench_c.json'))
print(d['value'], d['ms_per_step'], d['roofline']['read_basis']['overlapped_calls'])
print(json.dumps(d.get('consumer_loop'))[:900])
PY
