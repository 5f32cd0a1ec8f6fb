"""bench.py's legs after the headline with real ranks (two processes on the one GPU of the test box, gloo): the extra legs run
and land in the line; a leg that raises on one rank is named in "legs_failed" and the run still exits 0; a leg that hangs on one
rank ends EVERY rank with exit code 3 after rank 0 printed the headline (no rank may be left in a collective)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(extra_env, *flags, timeout=400):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(LEGION_BENCH_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scale", "0.02", "--steps", "4", "--warmup", "1",
                        "--min-time", "0.05", "--extra-min-time", "0.05", "--presc-steps", "2"] + list(flags),
                       env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_two_rank_line_has_every_leg():
    r, line = _bench({})
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["n_gpus"] == 2 and line["legs_failed"] == [] and line["value"] > 0 and line["value_overlap"] > 0
    u = line["unified_cache"]
    assert u["Kg"] == 2 and u["value"] > 0 and min(u["rows_last_batch"][k] for k in ("own_shard", "peer_shards", "backing_table")) > 0
    x = u["exchange_variant"]
    assert x.get("error") is None and x["host_syncs_per_batch"] == 1.0 and x["allocations_in_timed_windows"] == 0
    assert u["xgmi_read_GBps_per_gpu"] is None                       # two ranks on one device: not an xGMI number
    lp, uk = line["extra_legs"]["lp"], line["extra_legs"]["uk_union"]
    assert lp["value"] > 0 and lp["batch"] % 3 == 0 and uk["value"] > 0 and uk["Kg"] == 2 and uk["topo_rows_per_gpu"] > 0 and uk["F"] == 256


def test_a_failed_leg_is_named_and_the_other_legs_survive():
    """A leg that raises on every rank (a shape that does not fit, a refused argument): reported, the run goes on, exit code 0.
    (A failure on SOME ranks only leaves the others inside the leg's collectives: that is the hung-leg case below.)"""
    r, line = _bench({"LEGION_BENCH_INJECT_ERROR": "lp"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert [f["leg"] for f in line["legs_failed"]] == ["lp"] and line["legs_failed"][0]["hung"] is False
    assert "rank 0" in line["legs_failed"][0]["error"] and "rank 1" in line["legs_failed"][0]["error"] and "error" in line["extra_legs"]["lp"]
    assert line["extra_legs"]["uk_union"]["value"] > 0 and line["unified_cache"]["value"] > 0


def test_a_hung_leg_ends_every_rank_with_exit_code_3_and_the_headline_is_printed():
    r, line = _bench({"LEGION_BENCH_INJECT_HANG": "unified_cache:1"}, "--unified-timeout", "25")
    assert r.returncode == 3, (r.returncode, r.stderr[-3000:])
    assert line is not None and line["value"] > 0 and line["n_gpus"] == 2            # the headline survived
    assert line["legs_failed"][0]["leg"] == "unified_cache" and line["legs_failed"][0]["hung"] is True
    assert "did not finish" in line["unified_cache"]["error"]
    assert "the headline line was printed" in r.stderr


def test_a_leg_that_fails_on_one_rank_only_is_reported_within_seconds():
    """Rank 1 raises inside `lp`, rank 0 is left in the leg's collectives: the failure travels through the job's key-value store and
    rank 0 ends the run with the named error after a few seconds -- not after the leg's whole timeout (150 s)."""
    import time
    t0 = time.time()
    r, line = _bench({"LEGION_BENCH_INJECT_ERROR": "lp:1"})
    assert r.returncode == 3 and time.time() - t0 < 130, (r.returncode, time.time() - t0, r.stderr[-2000:])
    f = line["legs_failed"][0]
    assert f["leg"] == "lp" and f["hung"] is True and "rank 1" in f["error"] and "injected failure" in f["error"]
    assert line["value"] > 0 and line["unified_cache"]["value"] > 0           # everything before the failed leg is in the line
