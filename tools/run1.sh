timeout 900 python3 -m pytest tests/test_c_frontend_gpu.py -x -q -m gpu 2>&1 | tail -6
