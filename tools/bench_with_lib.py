"""A/B a kernel variant: run bench.py against another build of the C-ABI library.
    PYTHONPATH=. python tools/bench_with_lib.py path/to/libds2hip_variant.so --steps 20 --warmup 3 --no-cpu-baseline"""
import runpy, sys
from deepspeech.pytorch_amd import _lib
_lib.LIB_PATH = sys.argv[1]
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
