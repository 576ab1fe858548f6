"""Helper of tests/test_launch_failures_cpu.py: one RANK of a job started by reart_amd.launch.self_launch.
  hang  DIR          every rank writes DIR/pid.<rank> and sleeps (the test signals the launcher's parent)
  stubborn DIR      the same, but the ranks ignore SIGTERM: taking them down needs the SIGKILL stage of the clean-up
  orphan DIR        rank 1 SIGKILLs the LAUNCHER (its parent) after a while; both ranks sleep on as orphans
  crash DIR ROOT     the sweep's command line under gloo with the oracle as runner; rank 1 dies hard (os._exit) in the
                     middle of its share while rank 0 goes on to the gather of the energies"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    mode, d = sys.argv[1], sys.argv[2]
    rank = int(os.environ["RANK"])
    with open(os.path.join(d, f"pid.{rank}"), "w") as f:
        f.write(str(os.getpid()))
    if mode == "hang":
        time.sleep(3600)
        return 0
    if mode == "stubborn":
        import signal
        signal.signal(signal.SIGTERM, signal.SIG_IGN)
        with open(os.path.join(d, f"armed.{rank}"), "w") as f:
            f.write("1")
        time.sleep(3600)
        return 0
    if mode == "orphan":
        import signal
        time.sleep(3.0)                       # (the launching process records its launcher's descendants twice a second)
        if rank == 1:
            os.kill(os.getppid(), signal.SIGKILL)
        time.sleep(3600)
        return 0
    from test_sweep_cli_cpu import SEQ_ROOT, _oracle_runner
    from reart_amd import sweep

    def runner(spec):
        if rank == 1 and spec["cano_idx"] >= 1:
            os._exit(7)                       # not an exception the sweep could turn into a NaN record: the process is gone
        return _oracle_runner(spec)

    return sweep.main(["--seq_root", SEQ_ROOT, "--seqs", "seq_tiny", "--cano", "all", "--n_iter", "2", "--energy", "--gpus", "2",
                       "--save_root", sys.argv[3]], runner=runner)


if __name__ == "__main__":
    sys.exit(main())
