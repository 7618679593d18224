"""TEST INFRASTRUCTURE: the CPU parity oracle.  Import only from tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke(); never from skred_amd/."""
