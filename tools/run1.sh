timeout 400 python3 tools/fuzz_pcm.py 150 2 2>&1 | tail -2
for c in mixed:25 mixed:10; do
(cd /tmp && python3 $OLDPWD/bench.py --steps 20 --warmup 5 --iq-class $c --no-cpu-baseline --no-bit-exact-leg > $OLDPWD/gpurun_out/bench_$c.json 2>/dev/null)
python3 - $c <<'PY'
import json,sys
d=json.load(open('gpurun_out/bench_%s.json'%sys.argv[1]))
c=d.get('consumer_loop') or {}
print(sys.argv[1], d['roofline'].get('kernel'), {k:c.get(k) for k in ('kernel','ms_per_call_pcm_and_audio','ms_per_call_pcm_only','ms_per_call_demodulator_alone','sink_chain_ok')})
PY
done
