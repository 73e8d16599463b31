"""CPU oracle for the RNN-T joint + transducer-loss hot path.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (rnnt_amd/) never imports this.
"""
