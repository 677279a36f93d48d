mkdir -p gpurun_out/soak
(timeout 1500 python3 tools/fuzz_parity.py 200 7 2>&1 | tail -2) > gpurun_out/soak/fuzz_parity.txt; tail -n 2 gpurun_out/soak/fuzz_parity.txt
