python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -7
timeout 600 python3 tools/fuzz_pcm.py 240 1 2>&1 | tail -5
