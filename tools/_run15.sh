python -m pytest tests/test_parity_gpu.py -m gpu -x -q --timeout=600 -k "small_frame or drain_work" 2>&1 | tail -15
