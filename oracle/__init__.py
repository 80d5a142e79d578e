"""CPU oracle for the A*-guided trajectory-generation path of TS-TrajGen.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker (or as the timed CPU baseline),
never as the thing that is shipped.  The product path (``ts-trajgen_amd``) calls
hand-written gfx950 HIP kernels through the C-ABI in ``include/tsg.h`` and fails
loudly when that library is missing.

Every function restates the reference algorithm on the CPU (torch-CPU / numpy
fp32 arithmetic, stdlib ``heapq`` for the priority queue) and cites the reference
file:line it follows.  Parity pins: ``tests/golden/*.npz|json`` were produced by
importing the reference itself in the build container (``tools/gen_golden.py``);
``tests/test_oracle_golden.py`` checks this oracle against every one of them.

One boundary is *parity unpinned*: the reference calls
``geopy.distance.distance`` (WGS-84 geodesic, third-party, no version pinned,
not installed here and not installable).  ``oracle/geodesic.py`` restates the
published Vincenty inverse on WGS-84 and the golden generator injected that same
function into the reference in place of geopy.
"""
