"""Larger randomised parity sweep on the GPU box: runs tests/test_gpu_fuzz.py with the case counts scaled.
usage: python tools/fuzz_parity.py [scale=5]     (scale 1 = what the -m gpu suite runs)"""
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, ORBFE_FUZZ_SCALE=sys.argv[1] if len(sys.argv) > 1 else "5")
sys.exit(subprocess.call([sys.executable, "-m", "pytest", "-m", "gpu", "-x", "-q", os.path.join(root, "tests", "test_gpu_fuzz.py")],
                         cwd=root, env=env))
