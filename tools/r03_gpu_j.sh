#!/bin/bash
cd "$(dirname "$0")/.."
python3 -m pytest tests/test_gpu_features.py -x -q -k "sorting_changes_nothing" 2>&1 | tail -2
bash tools/r03_final.sh b
