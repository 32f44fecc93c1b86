"""CPU oracle for the DDPG-from-pixels hot path.  TEST INFRASTRUCTURE ONLY.

Nothing in `cartpoleplusplus_amd/` may import this package.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` use it, and
only as the checker / the timed CPU baseline -- never as a product code path.

Pinning status (see DESIGN.md "Oracle"):
  * replay path  -- PINNED by the reference's own known answers
    (replay_memory_test.py:19-30, :32-56, :58-86 and the soak invariant in
    replay_memory.py:166-200), restated in tests/test_oracle_replay.py.
  * network / optimiser path -- PARITY UNPINNED.  The reference holds no
    numerical test of any network output and cannot be run here (Python 2 +
    TensorFlow 0.x, neither installable).  The restatement
    is cross-checked against torch-CPU autograd (an independent implementation,
    not the reference) in tests/test_oracle_vs_torch.py.
"""
