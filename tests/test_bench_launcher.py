"""`python bench.py --gpus N` (N > 1) outside a launcher starts the N ranks itself, as a child process, and refuses a line that is not an
N-GPU line (the flag used to be parsed and ignored).  CPU-only: the child here is a stand-in that prints a line."""
import json
import os
import subprocess
import sys
import types

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402  (top level imports the standard library only: no GPU, no torch)


def test_launcher_command_is_the_drivers_form():
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "7", "--warmup", "2"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    tail = cmd[cmd.index(os.path.join(REPO, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "7", "--warmup", "2"]


@pytest.mark.parametrize("line, gpus, ok", [
    ({"n_gpus": 2, "rccl_ranks": 2, "value": 1.0}, 2, True),
    ({"n_gpus": 1, "rccl_ranks": None, "value": 1.0}, 2, False),   # the r3 behaviour: silently one GPU
    ({"n_gpus": 8, "rccl_ranks": 4, "value": 1.0}, 8, False),      # RCCL saw fewer ranks
    (None, 2, False),
])
def test_child_line_is_checked(line, gpus, ok):
    text = "RCCL banner\n" + (json.dumps(line) + "\n" if line is not None else "")
    got, err = bench.check_child_line(text, gpus)
    assert (err is None) == ok
    if line is not None:
        assert json.loads(got) == line


@pytest.mark.parametrize("n_gpus, rc_expected", [(2, 0), (1, 3)])
def test_launch_ranks_relays_or_fails(n_gpus, rc_expected, capfd):
    args = types.SimpleNamespace(gpus=2, master_port=0)
    fake = [sys.executable, "-c", "import json; print(json.dumps({'n_gpus': %d, 'rccl_ranks': %d, 'value': 5.0}))" % (n_gpus, n_gpus)]
    rc = bench.launch_ranks(args, [], cmd=fake)
    out = capfd.readouterr().out
    assert rc == rc_expected
    assert json.loads(out.strip().splitlines()[-1])["n_gpus"] == n_gpus  # the line is relayed either way; the exit code says whether to trust it


def test_world_size_must_match_gpus():
    """under a launcher that started another number of ranks than --gpus, bench.py exits before touching the GPU"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=60)
    assert r.returncode == 3 and "--gpus 2" in r.stderr and not r.stdout.strip()


def test_suitesparse_dimensions_are_asserted():
    """--mtx-dir: a file named like a BASELINE matrix must have its SuiteSparse dimensions -- a stand-in cannot pass"""
    ok = {"num_rows": 130228, "num_cols": 130228, "nnz": 2032536}
    bench.assert_suitesparse_dims("/data/cage12.mtx", ok)
    with pytest.raises(AssertionError):
        bench.assert_suitesparse_dims("/data/cage12.mtx", dict(ok, nnz=2032535))
    with pytest.raises(AssertionError):
        bench.assert_suitesparse_dims("/data/webbase-1M.mtx", ok)
