#!/usr/bin/env python3
"""Checker of an experiment (profiles/r05_sampler.md; lives under tests/ because it uses the oracle): a -DLEGION_POS32 variant library wipes its u32 position table every 126 batches.
400 consecutive batches on one pool against the oracle, word for word: three wipes, every epoch value used.
    LEGION_LIB=$PWD/legion-1_amd/csrc/variants/liblegion_amd_pos32.so python3 tests/pos32_wrap_check.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]   # tests/ = this directory (conftest)
import legion1_amd.capi as K  # noqa: E402
import legion1_amd.synth as S  # noqa: E402
import oracle as O  # noqa: E402
from conftest import assert_batch_equal  # noqa: E402

L = K.lib()
L.legion_set_error_mode(K.ERR_RETURN)
L.SetGPUDevice(0)
ds = S.generate(S.spec_for("products", scale=0.01))
B, fan = 200, [10, 5, 3]
lab = ds.labels[ds.train]
orc = O.OracleRunner(ds.indptr, ds.indices, ds.features, ds.spec.V, ds.spec.F, B, fan)
eng = K.Engine(ds.indptr, ds.indices, ds.features, ds.spec.V, ds.spec.F, dict(train=[(ds.train, lab)]), B, fan)
eng.alloc_features()
steps = (len(ds.train) + B - 1) // B
refs = {}
for i in range(400):
    it = i % steps
    if it not in refs:
        refs[it] = orc.run_batch(ds.train, lab, it)
    eng.run_batch(0, it, per_level=bool(i & 1))
    assert_batch_equal(refs[it], eng.result(0))
print("pos32 wrap check: 400 consecutive batches bit-identical to the oracle; library", K.lib_path(), "serial now", L.GPUMemoryPool_GetBatchSerial(eng.pools[0]))
eng.close()
