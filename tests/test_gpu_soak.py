"""A slice of every differential soak (tests/soak/) in the GPU suite (the long runs are recorded in profiles/r05_fuzz_*.txt):
random shapes / parameters / image statistics through the extractor and the stream matcher, random cases through every
ORBmatcher entry point, a random schedule of calls on one handle, frame sets + stereo + vocabulary, the frame-set searches
with the caller's queries + BoW on the set -- each against the CPU oracle, each with a seed the long runs did not use."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,args,says", [
    ("fuzz_soak.py", ["60", "101"], "fuzz soak: 60 cases"),
    ("fuzz_matchers.py", ["6", "102"], "matcher soak: 6 rounds"),
    ("fuzz_stream.py", ["120", "103"], "stream soak: 120 operations"),
    ("fuzz_frontend.py", ["8", "104"], "front-end soak: 8 rounds"),
    ("fuzz_tracking.py", ["10", "105"], "tracking soak: 10 rounds"),
])
def test_soak_slice(gpu, oracle, tool, args, says):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "soak", tool)] + args, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert says in out.stdout and "equal" in out.stdout
