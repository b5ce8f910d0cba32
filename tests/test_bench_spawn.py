"""CPU suite: `python bench.py --gpus N` started plainly must launch its own N ranks.

The launcher is a parent that never touches a GPU: it spawns N fresh children with the torch.distributed
environment set, relays rank 0's JSON line and fails if any rank fails.  Checked here with stand-in child
commands (no GPU in this container): the environment and arguments the children receive, the clean failure
when fewer GPUs are visible than ranks were asked for, and that a failing rank takes the run down.
"""
import io
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ECHO_CHILD = r"""
import json, os, sys
keys = ["RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY"]
rec = {k: os.environ.get(k) for k in keys}
rec["argv"] = sys.argv[1:]
open(os.path.join(sys.argv[1], "rank_%s.json" % os.environ["RANK"]), "w").write(json.dumps(rec))
if os.environ["RANK"] == "0":
    print(json.dumps({"metric": "stand-in", "n_gpus": int(os.environ["WORLD_SIZE"])}))
"""


@pytest.mark.parametrize("world", [2, 4, 8])
def test_children_get_rank_environment_and_arguments(tmp_path, world):
    import bench

    out = io.StringIO()
    rc = bench.spawn_ranks(world, ["--gpus", str(world)], device_count=world,
                           child_cmd=[sys.executable, "-c", ECHO_CHILD, str(tmp_path), "--gpus", str(world), "--steps", "3"],
                           timeout=120, out=out)
    assert rc == 0
    line = json.loads(out.getvalue().strip().splitlines()[-1])          # rank 0's line is relayed, nothing else
    assert line == {"metric": "stand-in", "n_gpus": world}
    ports = set()
    for r in range(world):
        rec = json.loads((tmp_path / f"rank_{r}.json").read_text())
        assert rec["RANK"] == str(r) and rec["LOCAL_RANK"] == str(r) and rec["WORLD_SIZE"] == str(world)
        assert rec["MASTER_ADDR"] == "127.0.0.1"                          # the container hostname may not resolve
        assert rec["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"                   # dmabuf IPC, or RCCL cannot share buffers
        assert rec["argv"][1:] == ["--gpus", str(world), "--steps", "3"]
        ports.add(rec["MASTER_PORT"])
    assert len(ports) == 1 and 0 < int(ports.pop()) < 65536


def test_too_few_gpus_fails_cleanly(capsys):
    import bench

    rc = bench.spawn_ranks(2, ["--gpus", "2"], device_count=1, child_cmd=[sys.executable, "-c", "raise SystemExit(0)"])
    assert rc != 0
    assert "needs 2 visible GPUs" in capsys.readouterr().err


def test_plain_invocation_on_this_gpu_less_container_fails_cleanly():
    """The real command line, end to end: no GPU here, so `--gpus 2` must exit non-zero with a message (and must not
    hang, exec, or print a JSON line)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert p.returncode != 0
    assert "visible GPUs" in p.stderr
    assert not p.stdout.strip()


def test_a_failing_rank_fails_the_run_and_stops_the_others(tmp_path):
    import bench

    child = ("import os, sys, time\n"
             "open(os.path.join(sys.argv[1], 'pid_' + os.environ['RANK']), 'w').write(str(os.getpid()))\n"
             "if os.environ['RANK'] == '1': sys.exit(7)\n"
             "time.sleep(120)\n")
    out = io.StringIO()
    rc = bench.spawn_ranks(2, [], device_count=2, child_cmd=[sys.executable, "-c", child, str(tmp_path)], timeout=60, out=out)
    assert rc == 7
    assert out.getvalue() == ""                                           # no line when a rank failed
    pid0 = int((tmp_path / "pid_0").read_text())
    with pytest.raises(ProcessLookupError):                               # rank 0 was stopped (by PID), not left sleeping
        os.kill(pid0, 0)


def test_default_workloads():
    import bench

    def per_gpu(argv, world):
        return bench.keys_per_gpu_log2(bench.parse_args(argv), world)

    assert per_gpu([], 1) == (28, "weak")                                 # configs[2]: 1 GiB on one GPU, unchanged
    assert per_gpu(["--gpus", "8"], 8) == (27, "strong")                  # configs[3]: 2^30 keys over 8 GPUs
    assert per_gpu(["--gpus", "2"], 2) == (29, "strong")
    assert per_gpu(["--gpus", "4", "--total-log2-keys", "28"], 4) == (26, "strong")   # the metric's "1 GiB at 1/2/4/8"
    assert per_gpu(["--gpus", "4", "--log2-keys", "28"], 4) == (28, "weak")
    with pytest.raises(SystemExit):
        bench.parse_args(["--gpus", "3"])
