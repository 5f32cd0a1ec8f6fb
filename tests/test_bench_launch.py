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

    def popen(cmd, env):
        started.append((cmd, env))
        return procs[len(started) - 1]
    args = bench.parse(["--gpus", str(len(procs))])
    rc = bench.launch_children(args, ["--gpus", str(len(procs)), "--steps", "3"], popen=popen, poll_s=0.0, **kw)
    return rc, started


def test_parent_starts_n_children_and_passes_the_flags_through():
    procs = [FakeProc([False, True], 0) for _ in range(4)]
    rc, started = _launch(procs)
    assert rc == 0 and len(started) == 4
    for r, (cmd, env) in enumerate(started):
        assert cmd[0] == sys.executable and cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "4", "--steps", "3"]
        assert env["RANK"] == str(r) and env["WORLD_SIZE"] == "4"
    assert len({env["MASTER_PORT"] for _, env in started}) == 1
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
            "    rank, world, D = 0, 1, D\n"
            "line = {'metric': 'm', 'value': 42.0, 'legs_failed': [], 'extra_legs': {}}\n"
            "g = bench.LegGuard(_Ctx(), line)\n"
            "def stuck():\n"
            "    g.partial = {'value': 7.0}\n"
            "    time.sleep(60)\n"
            "g.run('unified_cache', 0.5, stuck)\n"
            "print('never')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, cwd=ROOT)
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
