"""CPU oracle: torch/numpy restatements of the reference's hot-path arithmetic.

TEST INFRASTRUCTURE ONLY. Nothing under sgl-kernel-xpu_amd/ imports this
package; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
do, and only as the checker / the timed CPU baseline. The product path is the
HIP library and fails loudly when it is missing.

Pinning status (see DESIGN.md "Oracle"): the reference's SYCL kernels cannot be
built here (icpx, ocloc, Level-Zero and the un-vendored intel/sycl-tla@525faea3
are absent), so there is no oracle/_ref. Every function here is pinned against
golden vectors produced by importing the reference's own torch-eager test
references in the authoring container (tests/golden/make_golden.py, outputs
committed under tests/golden/), and cites the reference file:line it restates.
"""
