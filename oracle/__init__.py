"""oracle/ — CPU restatement of the reference's branch-sequence hot path.  TEST INFRASTRUCTURE ONLY.

This package is the parity checker, never the product:
  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
  * nothing under `branch-gan_amd/` imports it, and the product path raises if the HIP library is
    missing instead of falling back to anything here.

It restates, in plain PyTorch fp32 ops on the CPU, what FreddeFrallan/Branch-GAN computes on the path
SURVEY.md §8 scopes (each function cites the reference file:line it follows).  The reference is pure
Python on PyTorch, so "the algorithm" is a chain of ATen ops; ATen itself (matmul, softmax, sort,
cumsum, binary_cross_entropy, cross_entropy) is the third-party arithmetic and is called, not
re-derived.  torch version here: 2.10.0 (the reference pins none).

Pinning: the reference ships no tests or golden vectors, so every function here is
pinned against vectors captured from the *imported reference itself* in the build container by
`tests/golden/make_goldens.py` (fixtures `tests/golden/*.npz`); `tests/test_oracle_*.py` hold the checks.
"""
