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


def _bench(extra_env, *flags, timeout=400, gpus=2):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    if gpus > 1:
        env.update(LEGION_BENCH_FORCE_DEVICE="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--scale", "0.02", "--steps", "4", "--warmup", "1",
                        "--min-time", "0.05", "--extra-min-time", "0.05", "--presc-steps", "2"] + list(flags),
                       env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    # whatever the ranks, the HIP library and their children print, the command's stdout is the ONE JSON line
    assert r.stdout.count("\n") <= 1 and len(lines) == len(r.stdout.splitlines()), r.stdout[:2000]
    return r, (json.loads(lines[-1]) if lines else None)


def test_two_rank_line_has_every_leg():
    r, line = _bench({})
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["n_gpus"] == 2 and line["legs_failed"] == [] and line["value"] > 0 and line["value_overlap"] > 0
    # next 2c: the CPU baseline (short, sampler only, rank 0) and the roofline are in the N > 1 line too; nothing was skipped
    assert line["legs_skipped"] == [] and line["roofline"]["frac"] > 0 and line["time_budget"]["used_s"] < line["time_budget"]["budget_s"]
    cb = line["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] == 1 and "no feature gather" in cb["sample"] and cb["seconds"] <= 5.0
    assert "xGMI Clique" in r.stderr or "Feature Cache" in r.stderr or "Start solve cost model" in r.stderr      # the library's chatter: stderr
    u = line["unified_cache"]
    assert u["Kg"] == 2 and u["value"] > 0 and min(u["rows_last_batch"][k] for k in ("own_shard", "peer_shards", "backing_table")) > 0
    x = u["exchange_variant"]
    assert x.get("error") is None and x["host_syncs_per_batch"] == 1.0 and x["allocations_in_timed_windows"] == 0
    # two ranks on one device run the all-to-alls over gloo, staged through the host: labelled, and never printed as a result
    assert x["staged_through_host"] is True and "value" not in x and x["rehearsal_ms_per_step_not_a_result"] > 0
    assert u["xgmi_read_GBps_per_gpu"] is None                       # two ranks on one device: not an xGMI number
    lp, uk = line["extra_legs"]["lp"], line["extra_legs"]["uk_union"]
    assert lp["value"] > 0 and lp["batch"] % 3 == 0 and uk["value"] > 0 and uk["Kg"] == 2 and uk["topo_rows_per_gpu"] > 0 and uk["F"] == 256
    # the reference's deployment: ONE server process over both (logical) GPUs + one consumer process per GPU, started by rank 0 (last leg)
    sa = line["extra_legs"]["served_all"]
    assert list(line["extra_legs"]) == ["lp", "uk_union", "served_all"]
    assert sa.get("error") is None and sa["n_gpus"] == 2 and sa["value"] > 0 and len(sa["ms_per_step_per_gpu"]) == 2 and sa["shared_device"] is True, sa
    assert sa["gpu0_batches_equal_rank0s_timed_ones"] is True and "REHEARSAL" in sa["what"] and len(sa["server_gather"]) == 2
    if os.environ.get("LEGION_DEVICE_AUDIT") == "1":
        # the whole two-rank command -- replicated headline, unified-cache leg with in-kernel peer reads and the exchange
        # variant, config 4 / 5 legs, and ONE server over two logical GPUs with two consumers -- under the logical-device audit: rank 0's library
        # and the served_all server checked thousands of launches / copies / events / tables against their logical GPU and found nothing
        da, sda = line["device_audit"], sa["server_device_audit"]
        assert da["violations"] == 0 and da["checks"] > 1000, da
        assert sda["violations"] == 0 and sda["unattributed"] == 0 and sda["checks"] > 1000, sda
        from conftest import note_server_audit
        note_server_audit(sda)


def test_a_failed_leg_is_named_and_the_other_legs_survive():
    """A leg that raises on every rank (a shape that does not fit, a refused argument): reported, the run goes on, exit code 0.
    (A failure on SOME ranks only leaves the others inside the leg's collectives: that is the hung-leg case below.)"""
    r, line = _bench({"LEGION_BENCH_INJECT_ERROR": "lp"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert [f["leg"] for f in line["legs_failed"]] == ["lp"] and line["legs_failed"][0]["hung"] is False
    assert "rank 0" in line["legs_failed"][0]["error"] and "rank 1" in line["legs_failed"][0]["error"] and "error" in line["extra_legs"]["lp"]
    assert line["extra_legs"]["uk_union"]["value"] > 0 and line["unified_cache"]["value"] > 0


def test_a_hung_leg_ends_every_rank_with_exit_code_3_and_the_headline_is_printed():
    r, line = _bench({"LEGION_BENCH_INJECT_HANG": "unified_cache:1"}, "--unified-timeout", "12")
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


def test_one_gpu_line_has_every_single_gpu_baseline_path():
    """the driver's N = 1 command carries every BASELINE path one GPU can measure, each as a guarded leg:
    lp (config 5), cached_gather (config 3's FindFeat + gather), products_2hop (config 1, with its CPU legs), products_3hop (config 2),
    partitioned_csr (config 4's sampler over CSR fragments).  Here at 2 % of the shapes; the driver runs the full ones."""
    r, line = _bench({}, "--cpu-baseline-seconds", "1.5", "--measure-traffic", "off", gpus=1)
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["n_gpus"] == 1 and line["legs_failed"] == [] and line["value"] > 0 and line["cpu_baseline"]["value"] > 0
    legs = line["extra_legs"]
    assert list(legs) == ["served", "lp", "cached_gather", "products_2hop", "products_3hop", "partitioned_csr", "partitioned_csr_host_spill"]
    sv = legs.pop("served")
    # the server binary (dataset source synth:papers100M:0.02) + a consumer process, as fresh children of the bench
    assert sv.get("error") is None and sv["value"] > 0 and sv["ms_per_step"] > 0 and sv["windows"] >= 1, sv
    assert sv["schedule"]["train_steps"] == (int(11105995 * 0.02) - 1) // 8000 and sv["schedule"]["epochs"] >= 2
    assert sv["served_batches_equal_the_timed_ones"] is True and sv["server_tables"] == "generated in HBM"
    assert sv["ratio_to_alt_schedule_levels"] > 0 and sv["fanout"] == [25, 10, 5] and sv["F"] == 128
    assert line["value_served"] == sv["value"] and line["ms_per_step_served"] == sv["ms_per_step"]        # also at the top level, next to value / value_overlap
    # value_served is the figure of a consumer that READS what it is served -- every feature row and both COO arrays, on its own
    # stream, before the pipe goes back -- with one batch's sums checked (features against the generator's closed form of the rows the batch names);
    # the null consumer (the hand-off alone) sits beside it
    rc = sv["reading_consumer"]
    assert sv["consumer"] == "reading" and rc["read_GB_per_batch"] > 0 and rc["consumer_read_GB_per_s_of_wall_clock"] > 0
    ck = rc["checksum"]
    assert ck["features_equal_the_generators_rows"] is True and ck["coo_src_equal_host_copy"] is True and ck["coo_dst_equal_host_copy"] is True and ck["rows"] > 0, ck
    nul = sv["null_consumer"]
    assert nul.get("error") is None and nul["value"] > 0 and nul["served_batches_equal_the_timed_ones"] is True and sv["ratio_to_null_consumer"] > 0
    assert line["value_served_null"] == nul["value"] and line["ms_per_step_served_null"] == nul["ms_per_step"]
    for name, leg in legs.items():
        assert leg.get("error") is None and leg["value"] > 0 and leg["ms_per_step"] > 0, (name, leg)
        assert 0 < leg["gather_frac_of_hbm_peak"] < 1 and leg["sampler_us_per_batch"] > 0 and 0 < leg["pipeline_frac"] < 1, (name, leg)
    assert legs["lp"]["batch"] % 3 == 0 and legs["lp"]["F"] == 128
    lps = legs["lp"]["served"]        # config 5 through the server: it generates the [src | pos | neg] lists itself (synth: source + meta flag 2)
    assert lps.get("error") is None and lps["served_batches_equal_the_timed_ones"] is True and lps["value"] > 0 and "link-prediction" in lps["schedule"]["seeds"], lps
    cg = legs["cached_gather"]
    assert cg["Kg"] == 1 and 0.2 < cg["cached_fraction_of_V"] < 0.3 and cg["F"] == 128
    assert min(cg["rows_last_batch"][k] for k in ("own_shard", "backing_table")) > 0 and cg["rows_last_batch"]["peer_shards"] == 0
    cgs = cg["served"]     # the cached path through the server: cost model + FillUp + cached gather on a synth: source ($LEGION_SYNTH_CACHE=1)
    assert cgs.get("error") is None and cgs["served_batches_equal_the_timed_ones"] is True and cgs["server_cache"].startswith("Feat capacity"), cgs
    p2, p3 = legs["products_2hop"], legs["products_3hop"]
    assert p2["fanout"] == [25, 10] and p3["fanout"] == [25, 10, 5] and p2["F"] == p3["F"] == 100
    assert p2["cpu_baseline"]["value"] > 0 and p2["cpu_baseline"]["dgl_semantics"]["value"] > 0 and "cpu_baseline" not in p3
    assert p3["value_overlap"] > 0 and p3["pipeline_frac_overlap"] > 0
    assert "training batches per epoch" in p2["served"]["error"]          # 2 % of products has no full training batch: reported, not fatal
    pc = legs["partitioned_csr"]
    assert pc["F"] == 256 and pc["fanout"] == [25, 10] and pc["topo_rows_per_gpu"] > 0 and pc["Kg"] == 1
    sr = pc["served_replicated"]     # config 4's shape through the server (replicated: the uk-union tables fit one GPU's HBM)
    assert sr.get("error") is None and sr["value"] > 0 and sr["F"] == 256 and sr["served_batches_equal_the_timed_ones"] is None and sr["schedule"]["train_steps"] > 0
    # config 4 as BASELINE states it -- the feature table in pinned host memory behind the 10 % HBM cache, misses over PCIe
    hs = legs["partitioned_csr_host_spill"]
    assert hs["F"] == 256 and hs["fanout"] == [25, 10] and hs["topo_rows_per_gpu"] > 0 and "PINNED HOST" in hs["what"]
    sp = hs["host_spill"]
    assert 0 < sp["hit_rate"] < 1 and sp["miss_rows_last_batch"] > 0 and 0 < sp["pcie_read_GBps"] < 2 * sp["pcie_peak_GBps"], sp


def test_a_symmetric_exchange_failure_does_not_cost_the_rest_of_the_line():
    """the exchange variant failing on EVERY rank is a caught, reported part of the unified leg: the ranks agree
    right behind it and go on together -- the in-kernel numbers and the extra legs survive, exit code 0 (round 3 ended every rank
    with exit code 3 five seconds after the announcement)."""
    r, line = _bench({"LEGION_BENCH_INJECT_ERROR": "exchange"})
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    u = line["unified_cache"]
    assert "injected failure of the exchange variant" in u["exchange_variant"]["error"] and u["value"] > 0 and u["Kg"] == 2
    assert line["legs_failed"] == [] and line["extra_legs"]["lp"]["value"] > 0 and line["extra_legs"]["uk_union"]["value"] > 0


def test_served_all_failing_on_rank_0_costs_nothing_else():
    """The one leg only rank 0 works in (it starts the server and the consumers; the other ranks wait in the agreement collective): a failure there --
    a server that dies or is not ready inside the leg's own deadline -- is a FAILED leg, named in legs_failed, with every other leg intact and exit code 0."""
    r, line = _bench({"LEGION_BENCH_INJECT_ERROR": "served_all:0"})
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    assert [f["leg"] for f in line["legs_failed"]] == ["served_all"] and line["legs_failed"][0]["hung"] is False
    assert "error" in line["extra_legs"]["served_all"] and line["extra_legs"]["lp"]["value"] > 0 and line["extra_legs"]["uk_union"]["value"] > 0
    assert line["unified_cache"]["value"] > 0 and line["value"] > 0
