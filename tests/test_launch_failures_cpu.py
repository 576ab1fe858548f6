"""CPU: what can go wrong the first time the multi-GPU path runs (reference: one run_robot.py process per
instance, energies run_robot.py:306-321) -- the launcher of `bench.py --gpus N` / `python -m reart_amd.sweep --gpus N`
(reart_amd/launch.py) must not leave ranks behind: (1) a signal sent to the launching process (an outer `timeout`, a harness
kill) takes the ranks' process group down; (2) a rank that DIES mid-sweep ends the job with a non-zero status within the
timeout instead of leaving rank 0 in the gather."""
import os
import signal
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VICTIM = os.path.join(ROOT, "tests", "launch_victim.py")
DRIVER = ("import sys; sys.path.insert(0, {root!r}); from reart_amd.launch import self_launch; "
          "sys.exit(self_launch({victim!r}, {argv!r}, 2, timeout={timeout}))")


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    try:                                     # a zombie waiting for its reaper is not a rank holding a GPU
        with open(f"/proc/{pid}/stat") as f:
            return f.read().rsplit(")", 1)[1].split()[0] != "Z"
    except OSError:
        return False


def _wait_pids(d, n, limit):
    t0 = time.time()
    while time.time() - t0 < limit:
        pids = [os.path.join(d, f"pid.{r}") for r in range(n)]
        if all(os.path.exists(p) and os.path.getsize(p) > 0 for p in pids):
            return [int(open(p).read()) for p in pids]
        time.sleep(0.2)
    raise AssertionError("the ranks never started")


def _gone(pids, limit):
    t0 = time.time()
    while time.time() - t0 < limit:
        if not any(_alive(p) for p in pids):
            return True
        time.sleep(0.2)
    return False


def test_sigterm_to_the_launching_process_kills_the_ranks(tmp_path):
    code = DRIVER.format(root=ROOT, victim=VICTIM, argv=["hang", str(tmp_path)], timeout=300)
    drv = subprocess.Popen([sys.executable, "-c", code], stderr=subprocess.PIPE)
    try:
        pids = _wait_pids(str(tmp_path), 2, 120)
        assert all(_alive(p) for p in pids)
        drv.send_signal(signal.SIGTERM)
        rc = drv.wait(timeout=60)
        err = drv.stderr.read().decode(errors="replace")
        assert rc == 128 + signal.SIGTERM, (rc, err[-400:])
        assert "process group killed" in err
        assert _gone(pids, 30), "ranks survived the launcher's parent"
    finally:
        if drv.poll() is None:
            drv.kill()
        for f in os.listdir(tmp_path):
            if f.startswith("pid."):
                try:
                    os.kill(int(open(tmp_path / f).read()), signal.SIGKILL)
                except (ProcessLookupError, ValueError):
                    pass


def test_a_rank_that_dies_mid_sweep_ends_the_job(tmp_path):
    save = tmp_path / "out"
    code = DRIVER.format(root=ROOT, victim=VICTIM, argv=["crash", str(tmp_path), str(save)], timeout=240)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=400)
    took = time.time() - t0
    assert r.returncode not in (0, 124), (r.returncode, r.stderr.decode(errors="replace")[-600:])     # failed, and not by the timeout
    assert took < 240
    assert b"n_gpus" not in r.stdout                                      # no line that says the job ran on 2 GPUs
    pids = [int(open(tmp_path / f).read()) for f in os.listdir(tmp_path) if f.startswith("pid.")]
    assert len(pids) == 2 and _gone(pids, 30), "rank 0 was left in the gather"
    assert not (save / "sweep.json").exists()


def _cleanup(tmp_path, drv=None):
    if drv is not None and drv.poll() is None:
        drv.kill()
    for f in os.listdir(tmp_path):
        if f.startswith("pid."):
            try:
                os.kill(int(open(tmp_path / f).read()), signal.SIGKILL)
            except (ProcessLookupError, ValueError):
                pass


def test_a_second_signal_does_not_abort_the_clean_up(tmp_path):
    """`timeout -k`, a harness that sends TERM and then INT, a double Ctrl-C.  The ranks ignore SIGTERM, so the
    clean-up is still waiting for the launcher when the second signal arrives; it must go on to its SIGKILL stage."""
    code = DRIVER.format(root=ROOT, victim=VICTIM, argv=["stubborn", str(tmp_path)], timeout=300)
    drv = subprocess.Popen([sys.executable, "-c", code], stderr=subprocess.PIPE)
    try:
        pids = _wait_pids(str(tmp_path), 2, 120)
        t0 = time.time()
        while not all((tmp_path / f"armed.{r}").exists() for r in range(2)):
            assert time.time() - t0 < 30
            time.sleep(0.1)
        drv.send_signal(signal.SIGTERM)
        time.sleep(1.5)
        drv.send_signal(signal.SIGINT)
        time.sleep(0.5)
        drv.send_signal(signal.SIGTERM)
        rc = drv.wait(timeout=90)
        err = drv.stderr.read().decode(errors="replace")
        assert rc == 128 + signal.SIGTERM, (rc, err[-400:])
        assert _gone(pids, 30), "a second signal left ranks behind"
    finally:
        _cleanup(tmp_path, drv)


def test_ranks_that_outlive_a_dead_launcher_are_killed(tmp_path):
    """once the launcher is reaped its ranks are no longer its descendants and its pid is free.  The launching
    process records the ranks while the launcher lives (and is a child subreaper), and never signals the launcher's pid
    afterwards."""
    code = DRIVER.format(root=ROOT, victim=VICTIM, argv=["orphan", str(tmp_path)], timeout=300)
    drv = subprocess.Popen([sys.executable, "-c", code], stderr=subprocess.PIPE)
    try:
        pids = _wait_pids(str(tmp_path), 2, 120)
        rc = drv.wait(timeout=120)
        err = drv.stderr.read().decode(errors="replace")
        assert rc not in (0, 124), (rc, err[-400:])
        assert "failed" in err
        assert _gone(pids, 30), "orphaned ranks survived"
    finally:
        _cleanup(tmp_path, drv)


def test_descendants_are_stamped_and_checked():
    from reart_amd.launch import descendants, still_same
    child = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(30)"])
    try:
        mine = dict(descendants(os.getpid(), stamped=True))
        assert child.pid in mine and child.pid in descendants(os.getpid())
        assert still_same(child.pid, mine[child.pid])
        assert not still_same(child.pid, mine[child.pid] + 1)          # the same pid, another process
    finally:
        child.kill()
        child.wait()
    assert not still_same(child.pid, mine[child.pid])
