"""bench.py's launch contract (CPU only): `--gpus N` without torchrun spawns N ranks from a parent that never
touches the GPU; a rank count that differs from what was asked for is an error, never a silent smaller run."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launch_plan_matrix():
    one, two, eight = bench.parse([]), bench.parse(["--gpus", "2"]), bench.parse(["--gpus", "8"])
    assert bench.launch_plan(one, {}) == "worker"
    assert bench.launch_plan(two, {}) == "spawn"
    assert bench.launch_plan(eight, {}) == "spawn"
    assert bench.launch_plan(eight, {"WORLD_SIZE": "8"}) == "worker"        # torchrun / our own children
    assert bench.launch_plan(one, {"WORLD_SIZE": "1"}) == "worker"
    # the round-1 hole: `--gpus 8` under WORLD_SIZE=1 (or any other count) ran one rank and printed n_gpus: 1
    assert bench.launch_plan(eight, {"WORLD_SIZE": "1"}).startswith("error")
    assert bench.launch_plan(two, {"WORLD_SIZE": "4"}).startswith("error")
    assert bench.launch_plan(one, {"WORLD_SIZE": "2"}).startswith("error")
    assert bench.launch_plan(bench.parse(["--gpus", "0"]), {}).startswith("error")


def test_child_env_is_a_torchrun_style_rank():
    env = bench.child_env({"PATH": "/bin", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}, 3, 8, 29511)
    assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"]) == ("3", "3", "8")
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29511"
    assert env["PATH"] == "/bin" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert bench.child_env({}, 0, 2, 1)["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


class FakeProc:
    def __init__(self, polls, rc):
        self.polls, self.rc, self.returncode, self.killed = list(polls), rc, None, False

    def poll(self):
        if self.returncode is None and self.polls and self.polls.pop(0):
            self.returncode = self.rc
        return self.returncode

    def kill(self):
        self.killed = True
        self.returncode = -9

    def wait(self):
        return self.returncode


def _launch(procs, **kw):
    started = []

    def popen(cmd, env, stdout=None):
        started.append((cmd, env, stdout))
        return procs[len(started) - 1]
    args = bench.parse(["--gpus", str(len(procs))])
    rc = bench.launch_children(args, ["--gpus", str(len(procs)), "--steps", "3"], popen=popen, poll_s=0.0, **kw)
    return rc, started


def test_parent_starts_n_children_and_passes_the_flags_through():
    procs = [FakeProc([False, True], 0) for _ in range(4)]
    rc, started = _launch(procs)
    assert rc == 0 and len(started) == 4
    for r, (cmd, env, stdout) in enumerate(started):
        assert cmd[0] == sys.executable and cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "4", "--steps", "3"]
        assert env["RANK"] == str(r) and env["WORLD_SIZE"] == "4"
        assert (stdout is None) == (r == 0) and (r == 0 or stdout is sys.stderr)      # only rank 0 keeps the command's stdout
        assert float(env["LEGION_BENCH_T0"]) > 0                                      # the time budget counts from the parent's start
    assert len({env["MASTER_PORT"] for _, env, _ in started}) == 1
    assert not any(p.killed for p in procs)


def test_parent_fails_when_a_rank_fails_and_stops_the_stragglers():
    procs = [FakeProc([True], 7), FakeProc([False] * 10 ** 6, 0)]   # rank 1 would hang in a collective forever
    rc, _ = _launch(procs, grace_s=0.0)
    assert rc == 7
    assert procs[1].killed and not procs[0].killed


def test_gpus_2_without_gpus_is_an_error_not_a_silent_run():
    """Real processes: this container has no GPU, so both spawned ranks must refuse and the parent must fail."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LEGION_BENCH_FORCE_DEVICE")}
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "refusing" in r.stderr and "n_gpus" not in r.stdout
    # and a rank count that contradicts the flag is refused before anything is imported
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and r.stdout == ""


# ---- legs after the headline: LegGuard (a hung leg must not look like success, and no rank may be left in a collective) ----
class _Ctx:
    rank, world = 0, 1

    def __init__(self):
        import legion1_amd.dist as D
        self.D = D


def test_a_leg_that_raises_is_reported_and_the_run_goes_on():
    line = {"legs_failed": [], "extra_legs": {}}
    g = bench.LegGuard(_Ctx(), line)

    def boom():
        raise RuntimeError("shard import failed")
    res = g.run("uk_union", 5.0, boom)
    assert "shard import failed" in res["error"]
    assert line["legs_failed"] == [{"leg": "uk_union", "error": "rank 0: " + res["error"], "hung": False}]
    assert g.run("lp", 5.0, lambda: {"value": 1.0}) == {"value": 1.0} and len(line["legs_failed"]) == 1


def test_a_leg_that_hangs_prints_the_headline_names_the_leg_and_exits_3():
    code = ("import sys, json, time; sys.path.insert(0, %r)\n"
            "import bench, legion1_amd.dist as D\n"
            "class _Ctx:\n"
            "    rank, world, D = 0, %d, D\n"
            "line = {'metric': 'm', 'value': 42.0, 'legs_failed': [], 'extra_legs': {}}\n"
            "g = bench.LegGuard(_Ctx(), line)\n"
            "def stuck():\n"
            "    g.partial = {'value': 7.0}\n"
            "    time.sleep(60)\n"
            "g.run('unified_cache', 0.5, stuck)\n"
            "print('never')\n")
    # N = 1 (the driver's BENCH run): the line is printed (headline and finished legs are valid), but a watchdog that fired on a
    # process that has touched the GPU never exits 0: the same documented code as at N > 1
    r1 = subprocess.run([sys.executable, "-c", code % (ROOT, 1)], capture_output=True, text=True, timeout=60, cwd=ROOT)
    assert r1.returncode == bench.LEG_HUNG_EXIT and "never" not in r1.stdout and '"hung": true' in r1.stdout
    r = subprocess.run([sys.executable, "-c", code % (ROOT, 2)], capture_output=True, text=True, timeout=60, cwd=ROOT)
    assert r.returncode == bench.LEG_HUNG_EXIT == 3 and "never" not in r.stdout
    import json
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["value"] == 42.0                                          # the headline survives
    assert line["legs_failed"][0]["leg"] == "unified_cache" and line["legs_failed"][0]["hung"] is True
    assert "time.sleep" in line["legs_failed"][0]["stuck_at"] or "stuck" in line["legs_failed"][0]["stuck_at"]
    assert line["unified_cache"] == {"value": 7.0, "error": "did not finish within 0 s"} or line["unified_cache"]["value"] == 7.0
    assert "Thread" in r.stderr or "File" in r.stderr                     # faulthandler dump of every thread


def test_parent_passes_the_hung_leg_exit_code_on():
    procs = [FakeProc([True], bench.LEG_HUNG_EXIT) for _ in range(4)]
    rc, _ = _launch(procs)
    assert rc == bench.LEG_HUNG_EXIT
    # a rank that died for another reason still wins over the "leg hung" code
    procs = [FakeProc([True], bench.LEG_HUNG_EXIT), FakeProc([True], 1)]
    rc, _ = _launch(procs, grace_s=0.0)
    assert rc == 1


# ---- announcements of a caught PART of a leg are settled by that part's agreement collective ----------------------
_GUARD_PRELUDE = ("import sys, json, time; sys.path.insert(0, %r)\n"
                  "import bench\n"
                  "class FakeD:\n"
                  "    @staticmethod\n"
                  "    def allgather_object(o, world): return [o] * world\n"
                  "    @staticmethod\n"
                  "    def barrier(world): pass\n"
                  "class FakeStore:\n"
                  "    def __init__(self): self.d = {}\n"
                  "    def set(self, k, v): self.d[k] = v.encode() if isinstance(v, str) else v\n"
                  "    def get(self, k): return self.d[k]\n"
                  "    def check(self, ks): return all(k in self.d for k in ks)\n"
                  "class _Ctx:\n"
                  "    rank, world, D = 0, 2, FakeD\n"
                  "bench.LegGuard.PEER_GRACE_S = 1.0\n"
                  "line = {'metric': 'm', 'value': 42.0, 'legs_failed': [], 'extra_legs': {}}\n"
                  "g = bench.LegGuard(_Ctx(), line)\n"
                  "g.store = FakeStore()\n") % ROOT


def _guard_script(body):
    return subprocess.run([sys.executable, "-c", _GUARD_PRELUDE + body], capture_output=True, text=True, timeout=60, cwd=ROOT)


def test_a_part_failure_every_rank_agreed_on_does_not_end_the_leg():
    r = _guard_script("def leg():\n"
                      "    g.store.set('legion_leg_failed/unified_cache/part', 'rank 1: exchange variant: boom')   # a peer's announcement\n"
                      "    g.announce('exchange variant: boom')                                                     # and this rank's own\n"
                      "    g.part_agreed()              # the agreement all-gather behind the part came back: every rank left it\n"
                      "    time.sleep(3.0)              # the rest of the leg takes longer than the grace period\n"
                      "    return {'value': 5.0}\n"
                      "print('RES', g.run('unified_cache', 20.0, leg), line['legs_failed'])\n")
    assert r.returncode == 0 and "RES {'value': 5.0} []" in r.stdout, (r.returncode, r.stdout, r.stderr[-1500:])


def test_a_part_failure_a_rank_never_leaves_ends_the_run_once():
    r = _guard_script("def leg():\n"
                      "    g.store.set('legion_leg_failed/unified_cache/part', 'rank 1: exchange variant: boom')\n"
                      "    time.sleep(30.0)             # this rank sits in the part's collectives: no agreement\n"
                      "print('RES', g.run('unified_cache', 1.2, leg))\n")      # the plain timeout fires at about the same time
    assert r.returncode == bench.LEG_HUNG_EXIT and "RES" not in r.stdout
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                          # timer and watcher race: printed ONCE
    import json
    f = json.loads(lines[0])["legs_failed"]
    assert len(f) == 1 and f[0]["leg"] == "unified_cache" and f[0]["hung"] is True


def test_a_rank_ignores_its_own_leg_level_announcement():
    r = _guard_script("g.store.set('legion_leg_failed/lp', 'rank 0: something this rank said itself')\n"
                      "def leg():\n"
                      "    time.sleep(2.5)\n"
                      "    return {'value': 1.0}\n"
                      "print('RES', g.run('lp', 20.0, leg))\n")
    assert r.returncode == 0 and "RES {'value': 1.0}" in r.stdout, (r.returncode, r.stdout, r.stderr[-1500:])
    r = _guard_script("g.store.set('legion_leg_failed/lp', 'rank 1: a peer raised')\n"
                      "def leg():\n"
                      "    time.sleep(30.0)\n"
                      "print('RES', g.run('lp', 20.0, leg))\n")
    assert r.returncode == bench.LEG_HUNG_EXIT and "a peer raised" in r.stdout and "RES" not in r.stdout


# ---- which legs the one command runs (every single-GPU-measurable BASELINE path in the N = 1 line) ----------------
def test_extra_leg_plan():
    class C:
        pass

    def names(world, *flags):
        c = C()
        c.world, c.args = world, bench.parse(list(flags))
        return bench.extra_leg_names(c)
    assert names(1) == ["served", "lp", "cached_gather", "products_2hop", "products_3hop", "partitioned_csr",
                        "partitioned_csr_host_spill"]     # legs that re-use the headline graph first; the 137 GB pinned-host table last
    assert names(8) == ["lp", "uk_union", "served_all"] and names(2) == ["lp", "uk_union", "served_all"]    # r03 keys + (round 5, last) the one-server-process deployment
    assert names(1, "--extra-legs", "none") == [] and names(8, "--extra-legs", "none") == []
    assert names(1, "--workload", "products") == [] and names(1, "--task", "lp") == [] and names(1, "--headline-only") == []   # auto: default workload only
    assert names(1, "--extra-legs", "products_3hop,lp") == ["lp", "products_3hop"]                      # explicit lists run in the canonical order
    assert names(4, "--extra-legs", "uk_union") == ["uk_union"]
    import pytest
    with pytest.raises(SystemExit):
        names(1, "--extra-legs", "nonsense")


# ---- the first multi-GPU run must not be lost to plumbing --------------------------------------------------------------
_FAKE_WORKER = ("import os, sys, subprocess; sys.path.insert(0, %r)\n"
                "import bench\n"
                "bench.claim_stdout()                       # what worker() does first\n"
                "rank = int(os.environ['RANK'])\n"
                "print('python noise of rank %%d' %% rank, flush=True)\n"
                "os.write(1, b'raw fd-1 noise (a C++ std::cout / printf of the library)\\n')\n"
                "subprocess.run(['echo', 'noise of a child process of rank %%d' %% rank])\n"
                "assert os.environ.get('LEGION_LOG') == 'stderr'\n"
                "if rank == 0:\n"
                "    bench.emit_line({'metric': 'm', 'value': 1.5, 'n_gpus': int(os.environ['WORLD_SIZE'])})\n"
                "print('more noise after the line', flush=True)\n") % ROOT


def test_stdout_of_a_4_rank_launch_is_one_json_line_whatever_the_ranks_print(tmp_path):
    """Real processes: the self-launching parent starts 4 ranks that all print to stdout -- from Python, from file descriptor 1 directly
    (the HIP library's std::cout chatter: "xGMI Clique ...", "Feature Cache Hit ...") and from their own children.  The command's stdout
    must hold exactly the one JSON line of rank 0; everything else arrives on stderr."""
    fake = tmp_path / "fake_worker.py"
    fake.write_text(_FAKE_WORKER)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import bench\n"
            "args = bench.parse(['--gpus', '4'])\n"
            "sys.exit(bench.launch_children(args, ['--gpus', '4'], script=%r))\n") % (ROOT, str(fake))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LEGION_LOG")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout
    assert json.loads(lines[0]) == {"metric": "m", "value": 1.5, "n_gpus": 4}
    for rank in range(4):
        assert "python noise of rank %d" % rank in r.stderr and "noise of a child process of rank %d" % rank in r.stderr
    assert r.stderr.count("raw fd-1 noise") == 4 and r.stderr.count("more noise after the line") == 4


def test_time_budget_arithmetic():
    clock = [1000.0]
    b = bench.Budget(480.0, t0=1000.0, now=lambda: clock[0])
    assert b.left() == 480.0
    assert b.grant(240.0, 60.0) == 240.0                      # plenty left: the leg's own timeout
    clock[0] += 300.0                                         # 180 s left, 15 s reserve
    assert b.grant(240.0, 60.0) == 165.0                      # clamped to what is left minus the reserve
    assert b.grant(150.0, 30.0) == 150.0
    clock[0] += 120.0                                         # 60 s left
    assert b.grant(240.0, 60.0) == 0.0                        # 45 s < the 60 s the leg needs: skipped
    assert b.grant(150.0, 30.0) == 45.0
    clock[0] += 100.0                                         # over budget
    assert b.grant(150.0, 30.0) == 0.0 and b.left() < 0
    assert bench.Budget(0.0, t0=0.0, now=lambda: 1e9).grant(150.0, 30.0) == 150.0       # --time-budget 0: no budget
    # the parent's start time reaches the ranks through LEGION_BENCH_T0
    assert bench.Budget(480.0, "123.5").t0 == 123.5
    # the worst case of the default command fits the driver's 600 s limit: every grant ends before budget - reserve
    a = bench.parse([])
    assert a.time_budget <= 480.0 and a.time_budget + 60.0 <= 600.0


def test_a_hung_leg_and_the_budget_the_line_is_printed_in_time():
    """Budget 6 s: leg A hangs -> it gets what the budget grants (not its own 240 s timeout), the watchdog prints the line with the headline,
    names the leg and exits non-zero, all well inside the budget.  And a leg that the budget cannot admit any more is skipped, named in
    legs_skipped, and the line is printed normally."""
    code = ("import sys, json, time; sys.path.insert(0, %r)\n"
            "import bench, legion1_amd.dist as D\n"
            "bench.Budget.RESERVE_S = 1.0\n"
            "bench.LEG_LEAST_S.update(unified_cache=2.0, lp=2.0, quick=1.0)\n"
            "class _Ctx:\n"
            "    rank, world, D, children = 0, 1, D, []\n"
            "c = _Ctx(); c.budget = bench.Budget(%%s)\n"
            "line = {'metric': 'm', 'value': 42.0, 'legs_failed': [], 'legs_skipped': [], 'extra_legs': {}}\n"
            "g = bench.LegGuard(c, line)\n"
            "t0 = time.time()\n"
            "%%s\n"
            "bench.emit_line(line)\n") % ROOT
    hang = ("line['extra_legs']['quick'] = bench.run_budgeted(c, line, g, 'quick', 240.0, lambda: {'value': 1.0})\n"
            "line['unified_cache'] = bench.run_budgeted(c, line, g, 'unified_cache', 240.0, lambda: time.sleep(600))\n")
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code % ("6.0", hang)], capture_output=True, text=True, timeout=60, cwd=ROOT)
    took = time.time() - t0
    import json
    assert r.returncode == bench.LEG_HUNG_EXIT and took < 12.0, (r.returncode, took, r.stderr[-1500:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["value"] == 42.0 and line["extra_legs"]["quick"] == {"value": 1.0}
    assert line["legs_failed"][0]["leg"] == "unified_cache" and line["legs_failed"][0]["hung"] is True
    assert "did not finish within 5 s" in line["legs_failed"][0]["error"]           # 6 s budget - 1 s reserve, not the leg's 240 s
    skip = ("time.sleep(4.5)\n"
            "line['extra_legs']['lp'] = bench.run_budgeted(c, line, g, 'lp', 150.0, lambda: {'value': 2.0})\n")
    r = subprocess.run([sys.executable, "-c", code % ("6.0", skip)], capture_output=True, text=True, timeout=60, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-1500:]
    line = json.loads(r.stdout.splitlines()[-1])
    assert line["extra_legs"]["lp"] == {"skipped": "time budget"} and line["legs_skipped"][0]["leg"] == "lp" and line["legs_failed"] == []


def test_served_schedule_windows_stay_inside_the_training_part_of_an_epoch():
    """The served leg's timing windows (bench.served_schedule_windows): K consecutive TRAINING batches of one epoch, the first `warm` of every
    epoch left out, never a validation / test batch inside, measured arrival to arrival."""
    ts, vs, epochs, K, warm = 24, 1, 3, 9, 5
    total = (ts + vs) * epochs + 1
    t = [0.001 * i for i in range(total)]
    t[ts] += 0.5                                   # the validation batch of epoch 0 arrives late: no window may see it
    wins = bench.served_schedule_windows(t, ts, vs, epochs, K, warm)
    per = ts + vs
    assert [w[1] for w in wins] == [e * per + warm + k * K for e in range(epochs) for k in range((ts - warm) // K)]
    for secs, first in wins:
        e = first // per
        assert first - e * per >= warm and first + K <= e * per + ts          # inside the training part, behind the warm-up
        assert abs(secs - K * 0.001) < 1e-9                                    # K inter-arrival gaps; the late validation batch is outside
    assert bench.served_schedule_windows(t, ts, vs, epochs, 30, warm) == []    # a window longer than an epoch's training part: none


def test_a_served_leg_stops_its_children_inside_the_watchdogs_grant():
    """bench.served_deadline: a served leg's own deadline for its child processes lies 25 s inside min(--extra-timeout, what the budget leaves), never
    less than 20 s from now: a server or consumer that never comes back becomes a FAILED leg (reported, exit code 0), not a hung one (exit code 3)."""
    import time

    class C:
        pass
    c = C()
    c.args = bench.parse(["--extra-timeout", "150"])
    clock = [1000.0]
    c.budget = bench.Budget(480.0, t0=1000.0, now=lambda: clock[0])
    assert abs((bench.served_deadline(c) - time.time()) - 125.0) < 1.0          # 150 - 25
    clock[0] += 400.0                                                             # 80 s of the budget left: 80 - 15 - 25 = 40
    assert abs((bench.served_deadline(c) - time.time()) - 40.0) < 1.0
    clock[0] += 70.0                                                              # 10 s left: the floor
    assert abs((bench.served_deadline(c) - time.time()) - 20.0) < 1.0
