timeout 900 python3 -m pytest tests/test_pcm_sink_gpu.py tests/test_c_frontend_gpu.py -x -q -m gpu 2>&1 | tail -5
for i in 1 2 3; do timeout 60 ./examples/consumer_loop_main 256 10 300 6 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print({k:(v['steady_us_per_call'] if isinstance(v,dict) else v) for k,v in d.items() if k!='kernel'})"; done
