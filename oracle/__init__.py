"""CPU oracle for the OEI evaluation path -- TEST INFRASTRUCTURE ONLY.

This package is a plain numpy/scipy fp64 restatement of the reference's
algorithm for the Optimistic-Expected-Improvement path (GP posterior -> Omega
-> SDP -> gradient -> SDP sensitivity -> Hessian).  It exists so that tests,
the smoke check and the `cpu_baseline` leg of the benchmark have something to
compare the HIP path against.  Nothing in the product package imports it, and
the product path never falls back to it.

PARITY UNPINNED: the reference repository stores no golden vectors for this
path and none of its third-party solvers (SCS, pypardiso, gpflow 0.5,
TensorFlow 1.x) can be imported here, so this restatement cannot be checked
against outputs of the reference itself.  It is pinned instead by
solver-independent certificates (KKT conditions of the SDP, the closed form of
the q=1 problem, finite differences) and by the reference's own regenerable
test inputs (tests/test_sdp.py:45-62, legacy numpy seed-0 stream).

Modules
-------
gp        kernels, GP posterior, Omega and its closed-form derivatives
sdp_ref   the SDP as the reference poses it: pack/unpack, C_i, an SCS-style
          first-order solver, a KKT-Newton polish, the dense sensitivity solve,
          and the KKT certificate checker
oei       acquisition / acquisition_hessian / project orchestration
hart6     the Hartmann-6 objective and the [-1/2,1/2]^d rescaling used to build
          synthetic GP models
"""
