"""Build (under the package's build lock) and load the -DS2S_ABLATE library -- the superseded forward loops and the timing
ablations whose results are wrong by construction (S2S_CONV_DBG bits 1-32, S2S_CONV_DMA) -- for the measurement scripts in
this directory: ``import ablate_lib`` as the FIRST import of a script that wants them.  The product package never loads
this library; there is no environment variable that makes it."""
import fcntl
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stain2stain_amd import _native  # noqa: E402

with open(os.path.join(_native.HERE, ".build.lock"), "w") as lock:
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        _native.build(ablate=True, jobs=6)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
_native.load_library(_native.ABLATE_LIB_PATH)
