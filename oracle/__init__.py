"""TEST INFRASTRUCTURE ONLY.

CPU restatement ("oracle") of the JoHof/lungmask hot path.  Nothing in the
product package (`lungmask_amd/`) may import this; only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do.
"""
