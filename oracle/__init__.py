"""CPU oracle for the A-NeRF rendering hot path of mgholamikn/PoseGen.

TEST INFRASTRUCTURE ONLY.  Nothing under ``posegen_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker.

The oracle is a from-scratch fp32 restatement (torch CPU + numpy) of the
reference algorithm.  It is pinned by golden vectors captured from the real
reference implementation imported in the build container
(``tools/gen_golden.py`` -> ``tests/golden/*.npz``); see
``tests/test_oracle_golden.py``.
"""
from .anerf_oracle import *  # noqa: F401,F403
