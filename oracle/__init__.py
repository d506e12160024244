"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference hot path (TheMTank/GridUniverse
`core/envs/griduniverse_env.py` step/reset/look_step_ahead, the tabular DP
sweeps of `core/algorithms/utils.py`, and the build-defined per-env counter
RNG).  It exists so that the HIP kernels can be checked bit-for-bit.

Rules (enforced by tests/test_layout.py):
  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
    leg may import or execute anything in this directory;
  * the product package `griduniverse_amd/` never imports it and has no CPU
    fallback -- it fails loudly when libgu.so (the HIP library) is missing.

Parity status: PINNED.  Every function here is checked against golden vectors
captured from the real reference imported in the build container
(`tests/golden/make_golden.py` -> `tests/golden/`), including the reference's own ten
known-answer tests (`tests/test_griduniverse.py`).  See tests/test_oracle_*.py.

The reference is pure Python, so there is nothing to compile into
`oracle/_ref/`; the golden fixtures are the link to the real reference.
"""
