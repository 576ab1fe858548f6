"""One process per GPU: the launch half of the multi-GPU path.

A command that is started WITHOUT ``RANK`` in its environment and asked for N > 1 GPUs starts N ranks of itself under
``python -m torch.distributed.run`` (one rank per GPU, rendezvous on 127.0.0.1) and relays rank 0's standard output.
The parent never touches the GPU: it must decide and spawn before any HIP call (a process that has initialised the
GPU must not be replaced or forked into ranks), so this module imports nothing from torch."""
import json
import os
import socket
import subprocess
import sys


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _proc_table():
    """pid -> (ppid, start time in clock ticks) for every process in /proc."""
    table = {}
    for name in os.listdir("/proc"):
        if not name.isdigit():
            continue
        try:
            with open(f"/proc/{name}/stat") as f:
                rest = f.read().rsplit(")", 1)[1].split()     # state ppid ... (field 22 = start time: index 19 here)
            table[int(name)] = (int(rest[1]), int(rest[19]))
        except (OSError, IndexError, ValueError):
            continue
    return table


def descendants(pid, stamped=False):
    """Every live descendant of ``pid`` (children first), from /proc: the ranks of a torch.distributed.run launcher are its
    children but neither in its process group nor in its session.  ``stamped``: (pid, start time) pairs -- a pid alone may
    be recycled by the time somebody signals it, the pair may not."""
    table = _proc_table()
    kids = {}
    for child, (parent, _) in table.items():
        kids.setdefault(parent, []).append(child)
    out, todo = [], [int(pid)]
    while todo:
        for k in kids.get(todo.pop(), []):
            out.append((k, table[k][1]) if stamped else k)
            todo.append(k)
    return out


def still_same(pid, start):
    """Is ``pid`` still the process that was started at ``start`` (clock ticks since boot)?"""
    try:
        with open(f"/proc/{pid}/stat") as f:
            rest = f.read().rsplit(")", 1)[1].split()
        return rest[0] != "Z" and int(rest[19]) == int(start)
    except (OSError, IndexError, ValueError):
        return False


def under_launcher(env=None):
    """True when this process is one rank of a torch.distributed.run job."""
    env = os.environ if env is None else env
    return "RANK" in env and "WORLD_SIZE" in env


def torchrun_command(script, argv, nproc, port=None, module=False):
    """The exact command the round driver uses for N > 1 (and the one this module spawns).  ``module=True``: ``script`` is
    a module name (``-m``)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(nproc)}",
            "--master-addr", "127.0.0.1", "--master-port", str(port if port is not None else free_port()),
            *(["-m"] if module else []), script, *argv]


def check_world(requested, env=None):
    """A rank's view: the job must have as many ranks as ``--gpus`` asked for.  Raises SystemExit otherwise -- a line that
    says n_gpus = 1 for a request of 8 is worse than no line."""
    env = os.environ if env is None else env
    world = int(env.get("WORLD_SIZE", "1"))
    if int(requested) != world:
        raise SystemExit(f"--gpus {requested} but the job has WORLD_SIZE={world} rank(s): start it with "
                         f"`python -m torch.distributed.run --nproc-per-node {requested} ...` or let the command launch "
                         f"its own ranks (no RANK in the environment)")
    return world


def self_launch(script, argv, nproc, expect_json_key="n_gpus", timeout=None, module=False):
    """Start ``nproc`` ranks of ``script argv`` and wait.  The ranks' stdout is read line by line AS IT ARRIVES: a line that
    parses as a JSON object is kept (rank 0 prints exactly one; the last one is printed at the end), every other line goes
    to stderr at once -- progress is visible while the job runs and a hung rendezvous shows what was printed before it.
    The launcher runs in its own session: on ``timeout`` (seconds), an interrupt or a SIGTERM / SIGHUP / SIGINT sent to THIS
    process (an outer ``timeout``, a harness kill: they signal the parent's group only, and the ranks are no longer in it)
    the whole process group is killed, so no rank is left holding a GPU; should this process die without running a handler
    (SIGKILL), the kernel sends the launcher SIGTERM (PR_SET_PDEATHSIG) and torch.distributed.run takes its ranks down.
    Exit status: the launcher's, 124 after a timeout, 128 + signal after a signal, 3 when no JSON line came back, 4 when the
    line's ``expect_json_key`` differs from ``nproc``."""
    import signal
    import threading
    import time

    cmd = torchrun_command(script, argv, nproc, module=module)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // max(1, int(nproc)))))
    def die_with_parent():                                # in the child, before exec: SIGTERM when the parent is gone
        try:
            import ctypes
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, int(signal.SIGTERM), 0, 0, 0)      # PR_SET_PDEATHSIG = 1
        except Exception:                                 # (no libc by that name: the handlers below still cover signals)
            pass

    found = []
    known = {}                                            # pid -> start time of every descendant the launcher ever had

    class _Signalled(BaseException):
        def __init__(self, signum):
            self.signum = signum

    state = {"cleaning": False, "pending": None}

    def on_signal(signum, frame):
        if state["cleaning"]:                             # a second TERM / INT while the ranks are being taken down
            state["pending"] = signum                     # (timeout -k, a double Ctrl-C) must not abort the clean-up
            return
        raise _Signalled(signum)                          # unwinds the wait below into the clean-up

    def pump():
        for raw in iter(proc.stdout.readline, b""):
            text = raw.decode(errors="replace").rstrip("\n")
            s = text.strip()
            if s.startswith("{") and s.endswith("}"):
                try:
                    json.loads(s)
                    found.append(s)
                    continue
                except ValueError:
                    pass
            if s:
                print(text, file=sys.stderr, flush=True)

    def snapshot():
        """Record the launcher's descendants WHILE IT LIVES: once it is reaped its ranks are somebody else's children and
        its pid may belong to anyone."""
        if proc.returncode is None:
            for pid, start in descendants(proc.pid, stamped=True):
                known[pid] = start

    def kill_ranks(sig):
        for pid, start in list(known.items()):
            if not still_same(pid, start):
                known.pop(pid, None)
                continue
            for kill in (os.killpg, os.kill):            # a rank leads its own group (its children with it)
                try:
                    kill(pid, sig)
                except (ProcessLookupError, PermissionError):
                    pass

    def kill_group():
        """torch.distributed.run starts every rank in a session of its OWN (subprocess_handler: start_new_session), so the
        launcher's process group holds the launcher alone: SIGKILL to it would orphan the ranks with their GPUs.  The ranks
        are the descendants recorded while the launcher was alive (pid + start time: a recycled pid is never signalled);
        a live launcher gets SIGTERM (its handler closes its workers) and a few seconds, then whatever is left -- launcher,
        ranks, the ranks' own process groups -- SIGKILL.  A launcher that has been reaped is never signalled: its pid is
        free."""
        state["cleaning"] = True
        snapshot()
        if proc.poll() is None:
            try:
                os.killpg(proc.pid, signal.SIGTERM)      # start_new_session: the launcher's pid is its group's id
            except ProcessLookupError:
                pass
            try:
                proc.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass
        else:
            kill_ranks(signal.SIGTERM)
            deadline = time.monotonic() + 5
            while time.monotonic() < deadline and any(still_same(p, s) for p, s in known.items()):
                time.sleep(0.1)
        if proc.poll() is None:
            for kill in (os.killpg, os.kill):
                try:
                    kill(proc.pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    pass
            proc.wait()
        kill_ranks(signal.SIGKILL)

    def wait_recording(limit):
        """proc.wait(timeout=limit) in short slices, recording the launcher's descendants between them."""
        deadline = None if limit is None else time.monotonic() + float(limit)
        while True:
            snapshot()
            left = None if deadline is None else deadline - time.monotonic()
            if left is not None and left <= 0:
                raise subprocess.TimeoutExpired(cmd, limit)
            try:
                return proc.wait(timeout=0.5 if left is None else min(0.5, left))
            except subprocess.TimeoutExpired:
                continue

    old_handlers = {}
    proc = reader = None
    try:
        if threading.current_thread() is threading.main_thread():   # (signal.signal is the main thread's privilege)
            for sg in (signal.SIGTERM, signal.SIGHUP, signal.SIGINT):
                old_handlers[sg] = signal.signal(sg, on_signal)
        try:                                              # orphaned ranks become OUR children, not init's: still findable
            import ctypes
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(36, 1, 0, 0, 0)                       # PR_SET_CHILD_SUBREAPER
        except Exception:
            pass
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True, preexec_fn=die_with_parent)
        reader = threading.Thread(target=pump, daemon=True)
        reader.start()
        wait_recording(timeout)
    except subprocess.TimeoutExpired:
        kill_group()
        reader.join(timeout=5)
        print(f"launch of {nproc} ranks timed out after {timeout} s (process group killed): {' '.join(cmd)}", file=sys.stderr)
        return 124
    except _Signalled as sg:
        if proc is not None:
            kill_group()
        print(f"launch of {nproc} ranks ended by signal {sg.signum} (process group killed)", file=sys.stderr)
        return 128 + int(sg.signum)
    except BaseException:
        if proc is not None:
            kill_group()
        raise
    finally:
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
    reader.join(timeout=30)
    line = found[-1] if found else None
    if proc.returncode != 0:
        state["cleaning"] = True
        kill_ranks(signal.SIGTERM)                        # a rank that outlived a failed launcher: recorded while it lived
        time.sleep(0.5 if known else 0)
        kill_ranks(signal.SIGKILL)
        print(f"launch of {nproc} ranks failed (exit {proc.returncode}): {' '.join(cmd)}", file=sys.stderr)
        return proc.returncode
    if line is None:
        print("the ranks printed no JSON line", file=sys.stderr)
        return 3
    if expect_json_key is not None and json.loads(line).get(expect_json_key) != int(nproc):
        print(f"asked for {nproc} GPUs, the line reports {expect_json_key}={json.loads(line).get(expect_json_key)}",
              file=sys.stderr)
        return 4
    print(line)
    return 0
