python tools/latency_probe.py 2>&1 | tail -4
python -m pytest tests/test_gpu_extract.py tests/test_shim_ref.py tests/test_shim_cpp.py tests/test_stereo.py -q -x 2>&1 | tail -3
