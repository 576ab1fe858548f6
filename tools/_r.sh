timeout 900 python -m pytest tests/test_lap_gpu.py -m gpu -x -q 2>&1 | tail -2
TAGS="p1 base p1 base" tools/ab_tags.sh
