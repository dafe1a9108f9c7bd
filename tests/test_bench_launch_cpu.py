"""`python bench.py --gpus N` starts its own ranks (the reference launches its ranks itself: train_net.py:132-139) -- checked here
without a GPU: the environment / argv the launcher builds, that the launcher path is taken before torch.cuda is touched, and the
gradient reducer's one-backward-per-step guard."""
import importlib.util
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_child_env_and_argv():
    b = _bench()
    env = b.child_env(3, 8, 29600, base={"PATH": "/bin", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert env["RANK"] == "3" and env["LOCAL_RANK"] == "3" and env["WORLD_SIZE"] == "8" and env["LOCAL_WORLD_SIZE"] == "8"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29600" and env["PATH"] == "/bin"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert b.child_env(0, 2, 1, base={})["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    argv = b.child_argv(["--gpus", "8", "--steps", "5", "--warmup", "2"])
    assert argv[0] == sys.executable and argv[1] == os.path.join(ROOT, "bench.py") and argv[2:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]


def test_self_launch_spawns_one_process_per_rank_with_rank_env(tmp_path, monkeypatch):
    """The launcher starts N children, each with its own RANK, and returns their worst exit code; the children here are a stub
    (no GPU in this container), spawned through the real `self_launch`."""
    b = _bench()
    stub = tmp_path / "stub.py"
    stub.write_text("import os, sys\nopen(os.path.join(os.path.dirname(__file__), 'r' + os.environ['RANK']), 'w').write("
                    "os.environ['WORLD_SIZE'] + ' ' + os.environ['MASTER_ADDR'] + ' ' + os.environ['MASTER_PORT'] + ' ' + ' '.join(sys.argv[1:]))\n"
                    "sys.exit(3 if os.environ['RANK'] == '1' else 0)\n")
    monkeypatch.setattr(b, "child_argv", lambda argv: [sys.executable, str(stub)] + list(argv))

    class A:
        gpus = 2

    rc = b.self_launch(A(), ["--gpus", "2", "--steps", "1"])
    assert rc == 3
    got = [(tmp_path / f"r{r}").read_text().split() for r in range(2)]
    assert got[0][0] == got[1][0] == "2" and got[0][1] == "127.0.0.1" and got[0][2] == got[1][2] and got[0][3:] == ["--gpus", "2", "--steps", "1"]


def test_gpus_2_without_launcher_takes_the_launcher_path_before_touching_the_gpu():
    """No outer launcher (RANK unset): the parent must only spawn ranks.  The ranks then fail on the missing GPU -- not on an
    assertion about WORLD_SIZE (round-2 behaviour)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode != 0
    assert "WORLD_SIZE=" not in out.stderr and "AssertionError" not in out.stderr.split("launch with")[0][-400:]
    assert "launch with torch.distributed.run" not in out.stderr


def test_grad_reducer_rejects_a_second_gradient_for_one_parameter_in_a_step():
    from coin_amd.parallel import GradReducer

    w = torch.nn.Parameter(torch.ones(4))
    v = torch.nn.Parameter(torch.ones(4))
    red = GradReducer([w, v])
    (w.sum() + v.sum()).backward()
    assert red.finalize() == 1.0
    w.grad = v.grad = None
    (w * 2).sum().backward()
    with pytest.raises(RuntimeError, match="second gradient"):
        (w * 3).sum().backward()
    red.remove()
