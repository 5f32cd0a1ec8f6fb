"""legion1_amd -- MI355X-native implementation of Legion's GPU-initiated mini-batch
pipeline (sampler + frontier compaction, unified feature cache gather, partitioned
CSR store) behind Legion's Operator / IPC-service surface.

The product is the C-ABI library ``csrc/liblegion_amd.so`` (include/legion_amd.h);
this package is the thin Python host layer used by tests, bench.py and the
``ipc_service`` trainer extension.  There is no CPU fallback: importing
``legion1_amd.capi`` fails loudly when the HIP library has not been built.
"""
__version__ = "0.1.0"
