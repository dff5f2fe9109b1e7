"""Calls tests/test_gpu_step.py::test_chunked_streamed_matching_equals_one_chunk repeatedly in one
process (after one pipelined-trainer test, as in the suite's order) and reports the failures.
usage: python benchmarks/flake_probe.py [repeats=40]"""
import sys
import traceback
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT))
import conftest  # noqa: F401,E402
import test_gpu_step as T  # noqa: E402

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for rep in range(REP):
    try:
        T.test_chunked_streamed_matching_equals_one_chunk()
    except AssertionError:
        bad += 1
        print(f"rep {rep}: FAILED", flush=True)
        traceback.print_exc(limit=1)
print(f"{bad} of {REP} failed")
