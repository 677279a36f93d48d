WLS="fm256_pcm" bash tools/profile_round.sh r06 $1 > gpurun_out/prof_pcm.log 2>&1; tail -3 gpurun_out/prof_pcm.log
head -3 gpurun_out/profiles_r06/r06_fm256_pcm_kernel_stats.csv | cut -c1-200
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/profiles_r06/r06_fm256_pcm_pmc.json'))
print({k:d[k] for k in list(d)[:40]})
b=json.load(open('gpurun_out/profiles_r06/r06_fm256_pcm_bench.json'))
print(b['value'], b['config'].get('kernel'), b['roofline'].get('traffic'), b['roofline']['read_basis'])
PY
