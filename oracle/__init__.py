"""CPU oracle for the Viterbi / ViterbiNet hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package.  The shipped detectors never do (see oracle/mvn_oracle.c header).
"""
from .oracle import *  # noqa: F401,F403
