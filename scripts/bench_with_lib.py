"""A/B helper: run bench.py against another build of the library (e.g. one linked with a previous version of one source):
   python scripts/bench_with_lib.py stain2stain_amd/libstain2stain_hip_prev.so --steps 1000 --warmup 20 --no-extras
The product package never loads anything but its own library; this script does so explicitly and says so."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stain2stain_amd import _native  # noqa: E402

lib = os.path.abspath(sys.argv[1])
_native.load_library(lib)
sys.argv = ["bench.py"] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
