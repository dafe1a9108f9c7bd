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


def test_lazy_test_set_is_sharded_and_mapped_on_demand():
    """Round-2 ADVICE: the evaluation set is mapped image by image when it is reached (not kept on the device), sharded over the ranks."""
    from coin_amd.data import LazyTestSet
    from coin_amd.engine.base import BASE_Trainer

    dicts = [{"image_id": f"{i:03d}"} for i in range(7)]
    mapped = []

    def mapper(d):
        mapped.append(d["image_id"])
        return {"image_id": d["image_id"], "image": torch.zeros(3, 4, 4)}

    shards = [LazyTestSet(None, dicts, mapper=mapper, rank=r, world_size=2) for r in range(2)]
    assert [len(s) for s in shards] == [4, 3] and mapped == []          # nothing mapped at construction
    assert [d["image_id"] for d in shards[0]] + [d["image_id"] for d in shards[1]] == [d["image_id"] for d in dicts]

    class Model(torch.nn.Module):
        def forward(self, batch, branch="test"):
            return [{"n": len(batch)} for _ in batch]

    class Ev:
        def reset(self):
            self.seen = []

        def process(self, inputs, outputs):
            self.seen.append(([i["image_id"] for i in inputs], len(mapped)))

        def evaluate(self):
            return self.seen

    mapped.clear()
    seen = BASE_Trainer.test(Model(), shards[0], Ev(), batch_size=3)
    # batches of 3 + the remainder; when the first batch is processed only its 3 images have been mapped
    assert [ids for ids, _ in seen] == [["000", "001", "002"], ["003"]] and seen[0][1] == 3
    mapped.clear()
    assert len(BASE_Trainer.test(Model(), shards[0], Ev(), batch_size=1)) == 4   # re-iterable: a second evaluation maps again
