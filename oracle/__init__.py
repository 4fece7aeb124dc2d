"""TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference algorithms on the hot path used as the
parity checker.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package; the product package `tts_amd` never does.
"""
