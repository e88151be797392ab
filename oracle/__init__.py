"""CPU oracle — TEST INFRASTRUCTURE ONLY.

Restatements of the reference's algorithms for the hot path, used as the
checker by tests/, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg.  Nothing in
the product package (`lmrl-gym_amd/`) imports this package; the product fails loudly when
its HIP library is missing instead of falling back to anything here.
"""
