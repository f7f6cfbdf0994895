#!/bin/bash
cd "$(dirname "$0")/../.."
python3 tools/many_bench.py 16000000000 8 4096
python3 tools/many_bench.py 16000000000 2 4096
python3 tools/many_bench.py 16000000000 4 4096
python3 tools/many_bench.py 16000000000 16 4096
python3 tools/many_bench.py 16000000000 8 4000
