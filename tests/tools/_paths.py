"""sys.path for the tools: the package, the oracle (test infrastructure) and the test helpers, as tests/conftest.py sets them."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "solstrale-rust_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
