import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests"), os.path.join(REPO, "alphazero-4-player-chess_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Make sure the native pieces exist (fresh checkout: *.so are git-ignored).  Builds with hipcc /
    g++ exactly like `python __graft_entry__.py`; a no-op when everything is up to date."""
    import __graft_entry__
    try:
        __graft_entry__.build()
    except Exception as exc:                      # report, let the tests that need it fail loudly
        print("WARNING: __graft_entry__.build() failed: %r" % (exc,))
