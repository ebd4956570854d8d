"""One-off A/B helper: the default bench step (main workload only) through a variant library.
    TVAE_LIB_OVERRIDE=<lib.so> [TVAE_DEBUG_NO_SYNC=1] TAG=<label> python tools/bench_override.py"""
import os, sys
sys.path.insert(0, os.getcwd())
from transformer_vae_amd import _lib
if os.environ.get("TVAE_LIB_OVERRIDE"):
    _lib.LIB_PATH = os.path.abspath(os.environ["TVAE_LIB_OVERRIDE"])
import bench, json, io, contextlib
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-side-configs", "--no-cpu-baseline"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(os.environ.get("TAG"), d["ms_per_step"], d["roofline"]["kernel_ms"])
