timeout 900 python3 -m pytest tests/test_pcm_sink_gpu.py tests/test_c_frontend_gpu.py tests/test_overlap_gpu.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python3 tools/consumer_loop.py 2>&1 | tail -12 | cut -c1-250
